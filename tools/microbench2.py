"""Logic-phase variants (no observation write) to see where phase 1 spends its time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map
from tools.microbench import timeit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for epw in (32, 64):
    bw = BatchedWorld(Map(level=6), n, envs_per_wave=epw)
    stay = torch.full((n, 4), 4, dtype=torch.uint8, device="cuda")
    t = [0]
    def v1():
        bw.step(sample=True, auto_reset=True, seed=1, t=t[0], write_obs=False); t[0] += 1
    def v2():
        bw.step(sample=True, auto_reset=False, seed=1, t=t[0], write_obs=False); t[0] += 1
    def v3():
        bw.step(stay, write_obs=False)
    bw.reset(); r1 = timeit(v1)
    bw.reset(); r3 = timeit(v3)
    bw.reset(); r2 = timeit(v2)   # all agents die eventually -> mostly STAY-only lanes
    inv = torch.full((n, 4), 7, dtype=torch.uint8, device="cuda")
    def v4():
        bw.step(inv, write_obs=False)     # invalid action: checks only, no step
    bw.reset(); r4 = timeit(v4)
    e0 = torch.empty(1, device="cuda")
    def v5():
        e0.add_(1)   # a trivial torch kernel: the launch-to-launch floor
    r5 = timeit(v5)
    print(f"n={n} epw={epw}: trivial-kernel floor {r5:.2f} us | sample+autoreset {r1:.2f} us | explicit STAY (fresh envs, no deaths) {r3:.2f} us | sample no-reset (mostly dead) {r2:.2f} us | invalid actions (no step) {r4:.2f} us", flush=True)
