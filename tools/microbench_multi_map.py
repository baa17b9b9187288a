"""config 5 (32x32, 8 agents, 8 lasers) with ONE map vs 64 distinct generated maps (1024 envs each), 65 536 envs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, mapgen

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

n = 65536
for label, maps in (("one map", mapgen.config5(0)), ("64 maps x 1024 envs", [mapgen.generate(seed=s) for s in range(64)])):
    bw = BatchedWorld(maps, n)
    t = [0]
    def full():
        bw.step(sample=True, auto_reset=True, seed=1, t=t[0]); t[0] += 1
    us = timeit(full)
    print(f"config 5, {label}: {us:.1f} us per step ({20617*n/us/1e3:.0f} GB/s)  kernel {bw.kernel_info()}", flush=True)
    del bw
