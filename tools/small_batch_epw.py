"""Small batches (BASELINE configs[1]: level 1 x 4 096): us per single step by environments per wavefront (LLE_STEP_EPW), HIP events."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi

for level, n in ((1, 4096), (1, 1024), (6, 4096), (1, 16384)):
    row = []
    for epw in (1, 2, 4, 8, 16):
        os.environ["LLE_STEP_EPW"] = str(epw)
        _capi.refresh_tuning()
        bw = BatchedWorld(Map(level=level), n, autotune_ms=0)
        fn = bw.sampled_stepper(auto_reset=True, seed=1)
        got = bw.kernel_info()["envs_per_wave"]
        us = min(timeit(fn, iters=400, warm=40) for _ in range(3))
        row.append(f"epw {epw}{'' if got == epw else f'->{got}'}: {us:5.2f}")
        del bw
    os.environ.pop("LLE_STEP_EPW")
    _capi.refresh_tuning()
    bw = BatchedWorld(Map(level=level), n)  # the constructor's default: autotuned
    fn = bw.sampled_stepper(auto_reset=True, seed=1)
    us = min(timeit(fn, iters=400, warm=40) for _ in range(3))
    print(f"level {level} x {n}: " + "  ".join(row) + f"  | constructor default (epw {bw.kernel_info()['envs_per_wave']}): {us:5.2f} us", flush=True)
for level, n in ((1, 4096), (1, 1024)):
    for ms in (0, 10, 10, 30):
        bw = BatchedWorld(Map(level=level), n, autotune_ms=ms)
        fn = bw.sampled_stepper(auto_reset=True, seed=1)
        us = min(timeit(fn, iters=400, warm=40) for _ in range(3))
        t = bw.tuning()
        print(f"level {level} x {n} autotune_ms={ms}: {us:5.2f} us  epw {t['envs_per_wave']} heads {t['row_heads']} wt {t['write_through']} rot {t['rotate_rows']} | {t['log']}", flush=True)
