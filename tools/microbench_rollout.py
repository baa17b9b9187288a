"""Fused rollout (lle_batch_rollout): us per step for several steps-per-launch and ring sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map

def run(n, T, R, launches=20):
    bw = BatchedWorld(Map(level=6), n)
    ring = bw.make_ring(R) if R else None
    for _ in range(3):
        bw.rollout(T, seed=1, ring=ring, ring_pos=bw.t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        bw.rollout(T, seed=1, ring=ring, ring_pos=bw.t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (launches * T)
    print(f"n={n} steps/launch={T} ring={R}: {us:.2f} us/step  {1937*n/us/1e3:.0f} GB/s algorithmic", flush=True)

for n in (65536, 262144):
    for T, R in ((1, 0), (4, 0), (16, 0), (16, 4), (64, 4), (64, 8)):
        if n == 262144 and R == 8:
            continue
        run(n, T, R)
