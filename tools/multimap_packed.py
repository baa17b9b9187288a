"""Packed table image (tables.h off_packed; LLE_PACKED_TABLES=0 / 1) on config 5's shape, 65 536 envs: us per step and the share of the arena's own
row-fill time, both settings on the SAME arena."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
one = BatchedWorld(mapgen.config5(0), n, autotune_ms=0)
fill = min(timeit(one.row_fill_prober(), iters=30, warm=5) for _ in range(2))
us = min(timeit(one.sampled_stepper(auto_reset=True, seed=1), iters=60, warm=10) for _ in range(3))
print(f"one map: {us:.1f} us (fill {fill:.1f}: {fill / us:.3f})", flush=True)
del one
torch.cuda.empty_cache()
for n_maps in (1024, 4096, 8192):
    bw = BatchedWorld([Map(mapgen.config5(s)) for s in range(n_maps)], n, autotune_ms=0)
    fill = min(timeit(bw.row_fill_prober(), iters=30, warm=5) for _ in range(2))
    out = []
    for setting in ("0", "1", "0", "1"):
        os.environ["LLE_PACKED_TABLES"] = setting
        _capi.refresh_tuning()
        fn = bw.sampled_stepper(auto_reset=True, seed=1)
        us = min(timeit(fn, iters=60, warm=10) for _ in range(3))
        out.append(f"packed={setting}: {us:.1f} ({fill / us:.3f})")
    os.environ.pop("LLE_PACKED_TABLES")
    _capi.refresh_tuning()
    print(f"{n_maps} x {n // n_maps} (fill {fill:.1f}): " + "  ".join(out), flush=True)
    del bw, fn
    torch.cuda.empty_cache()
