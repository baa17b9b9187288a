"""Small maps (12 x 13, 4 agents, 4 lasers), 65 536 envs: one map, 4 096 x 16, 8 192 x 8; (the comparison of profiles/r05_multi_map.md section 1 ran this script a second time on a
DIAGNOSTIC build in which every workgroup reads map 0's tables -- one line in capi.cpp's launch(): K.table_stride = 0 under LLE_DEBUG_SAME_TABLES; results wrong,
timing only; the shipped library has no such switch)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
gen = lambda s: mapgen.generate(12, 13, 4, 4, 4, seed=s, n_voids=2)
def run(label, maps):
    res = []
    for rep in range(2):
        bw = BatchedWorld(maps, n, autotune_ms=0)
        fn = bw.sampled_stepper(auto_reset=True, seed=1)
        us = min(timeit(fn, iters=60, warm=10) for _ in range(3))
        fill = min(timeit(bw.row_fill_prober(), iters=30, warm=5) for _ in range(2))
        res.append(f"{us:7.2f} (fill {fill:6.2f}: {fill / us:.3f})")
        del bw, fn
        torch.cuda.empty_cache()
    print(f"{os.environ.get('LABEL', 'default'):14s} {label:14s} " + " / ".join(res), flush=True)

run("one map", gen(0))
for n_maps in (4096, 8192):
    run(f"{n_maps} x {n // n_maps}", [Map(gen(s)) for s in range(n_maps)])
