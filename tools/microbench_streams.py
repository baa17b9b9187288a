"""Does splitting the batch over k HIP streams (k independent half/quarter batches stepped concurrently) overlap the
state-machine phase of one part with the observation stream of another?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map

n = 65536
for k in (1, 2, 4):
    parts = [BatchedWorld(Map(level=6), n // k) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run(steps, t0):
        for t in range(t0, t0 + steps):
            for i, (p, s) in enumerate(zip(parts, streams)):
                with torch.cuda.stream(s):
                    p.step(sample=True, auto_reset=True, seed=1, t=t, env_offset=i * (n // k))
    run(20, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e6
    print(f"streams={k}: {dt:.2f} us per step of {n} envs ({1937*n/dt/1e3:.0f} GB/s)", flush=True)
