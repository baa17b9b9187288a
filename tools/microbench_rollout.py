"""Fused rollout (lle_batch_rollout): us per step for several steps-per-launch and ring sizes (true HBM writes when the
ring is larger than the 256 MiB Infinity Cache)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map

def run(n, T, R, launches=24, auto_reset=True):
    bw = BatchedWorld(Map(level=6), n)
    ring = bw.make_ring(R) if R else None
    for _ in range(3):
        bw.rollout(T, auto_reset=auto_reset, seed=1, ring=ring, ring_pos=bw.t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        bw.rollout(T, auto_reset=auto_reset, seed=1, ring=ring, ring_pos=bw.t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (launches * T)
    print(f"n={n} steps/launch={T} ring={R} ({R*n*1872/1e6:.0f} MB): {us:.2f} us/step  {(1891+48/T)*n/us/1e3:.0f} GB/s", flush=True)

for rep in range(2):
    for T, R in ((8, 8), (16, 8), (16, 16), (32, 8), (64, 8), (16, 4)):
        run(65536, T, R)
