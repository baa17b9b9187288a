"""Where the multi-map tax of small blocks comes from (config 5's shape, 65 536 envs): one map at four and at two wavefronts per workgroup, against
1 024 x 64, 4 096 x 16 distinct maps.  us per step (HIP events).
(Its "copies of ONE map" rows are distinct memory like distinct maps, so they do NOT separate the table reads from the launch shape -- the comparison that does
is in profiles/r05_multi_map.md section 1: a build in which every workgroup reads map 0's tables.)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
def run(label, maps, **env):
    for k, v in env.items():
        os.environ[k] = v
    _capi.refresh_tuning()
    bw = BatchedWorld(maps, n, autotune_ms=0)
    fn = bw.sampled_stepper(auto_reset=True, seed=1)
    us = min(timeit(fn, iters=60, warm=10) for _ in range(3))
    print(f"{label:44s} {us:8.2f} us  {bw.kernel_info()}", flush=True)
    for k in env:
        os.environ.pop(k)
    _capi.refresh_tuning()
    del bw
    torch.cuda.empty_cache()

one = mapgen.config5(0)
run("one map (4 wavefronts per workgroup)", one)
run("one map, LLE_STEP_WPW=2", one, LLE_STEP_WPW="2")
run("one map, LLE_STEP_WPW=1", one, LLE_STEP_WPW="1")
for n_maps in (1024, 4096):
    maps = [Map(mapgen.config5(s)) for s in range(n_maps)]
    run(f"{n_maps} maps x {n // n_maps}", maps)
    same = [Map(one) for _ in range(n_maps)]
    run(f"{n_maps} copies of ONE map x {n // n_maps}", same)
    del maps, same
maps = [Map(mapgen.config5(s)) for s in range(8192)]
run("8192 maps x 8", maps)
