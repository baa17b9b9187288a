"""One map per environment (and 2 / 4 / 8 per map) at scale: creation time, table memory, us per step -- a small shape (12 x 13, 4 agents) and
config 5's.  python tools/one_map_per_env.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, mapgen

def run(label, gen, n, per):
    n_maps = n // per
    t0 = time.perf_counter()
    maps = [Map(gen(s)) for s in range(n_maps)]
    t1 = time.perf_counter()
    bw = BatchedWorld(maps, n, autotune_ms=0) if n_maps > 1 else BatchedWorld(maps[0], n, autotune_ms=0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fn = bw.sampled_stepper(auto_reset=True, seed=1)
    us = min(timeit(fn, iters=40, warm=8) for _ in range(3))
    st = bw.stats()
    print(f"{label:22s} {n:6d} envs, {per:5d} per map: {us:9.1f} us per step = {n / us:7.1f} M env-steps/s; maps compiled in {t1 - t0:5.1f} s, batch created in {t2 - t1:5.1f} s, "
          f"tables {n_maps * maps[0].table_bytes / 1e6:7.1f} MB  {bw.kernel_info()}  invalid {st['invalid']}", flush=True)
    del bw, fn, maps
    torch.cuda.empty_cache()

small = lambda s: mapgen.generate(12, 13, 4, 4, 4, seed=s, n_voids=2)
for n, per in ((65536, 65536), (65536, 8), (65536, 4), (65536, 2), (65536, 1), (8192, 8192), (8192, 1)):
    run("12x13, 4 agents", small, n, per)
for n, per in ((16384, 16384), (16384, 8), (16384, 4), (16384, 1)):
    run("config 5 (32x32, 8)", lambda s: mapgen.config5(s), n, per)
