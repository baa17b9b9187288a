"""Throughput of the other BASELINE.json configurations (not bench lines: reported in DESIGN.md)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map, mapgen
from tools.microbench import timeit

def bytes_per_step(m):
    A, G, L = m.n_agents, m.n_gems, m.n_sources
    S = 2 * A + 3 * ((A + 7) // 8) + (G + 7) // 8 + 4 * L
    return m.obs_bytes + 2 * S + A + A + (1 + 2 * A)

for name, mp, n in (("config2 level1 n=4096", Map(level=1), 4096), ("level1 n=65536", Map(level=1), 65536),
                    ("config3 level6 n=65536", Map(level=6), 65536), ("config5 32x32x8 n=65536", Map(mapgen.config5(0)), 65536),
                    ("config5 32x32x8 n=16384", Map(mapgen.config5(0)), 16384)):
    bw = BatchedWorld(mp, n)
    t = [0]
    def full():
        bw.step(sample=True, auto_reset=True, seed=1, t=t[0]); t[0] += 1
    us = timeit(full, iters=50)
    B = bytes_per_step(mp)
    print(f"{name}: {us:.1f} us/step, {n/us:.1f} M env-steps/s, {B} B/env-step -> {B*n/us/1e3:.0f} GB/s ({B*n/us/1e3/80:.1f} % of 8 TB/s)  {bw.kernel_info()}", flush=True)
    del bw
