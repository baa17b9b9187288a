import sys, os
sys.path.insert(0, "/root/repo")
import torch
from lle_amd import BatchedWorld, Map
def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n = 65536
for pes in (False, True):
    bw = BatchedWorld(Map(level=6), n)
    if pes:
        g = torch.Generator().manual_seed(0)
        bw.set_sources(torch.randint(0, 4, (n, 3), generator=g, dtype=torch.uint8))
    t = [0]
    def full():
        bw.step(sample=True, auto_reset=True, seed=1, t=t[0]); t[0] += 1
    us = timeit(full)
    print(f"level 6 n={n} per_env_sources={pes}: {us:.2f} us per step ({1937*n/us/1e3:.0f} GB/s)  stats={bw.stats()}", flush=True)
