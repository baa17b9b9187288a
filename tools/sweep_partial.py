"""E x batches x store policy sweep of the lane-per-(env, observer) partial kernel (LLE_PARTIAL_E / LLE_PARTIAL_BATCHES / LLE_PARTIAL_WT): us per launch at 65 536 envs.
GPU box: python tools/sweep_partial.py"""
import os, sys, itertools
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen
n = 65536
for label, m, Es in (("level6", Map(level=6), (2, 4, 8, 16)), ("cfg5", Map(mapgen.config5(0)), (1, 2, 4, 8))):
    bw = BatchedWorld(m, n)
    for t in range(8):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
    for k in (3, 5, 7):
        d = bw.obs_desc(_capi.LLE_OBS_PARTIAL, k)
        buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device="cuda")
        buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
        for wt in ("0", "1"):
            row = []
            for E, B in itertools.product(Es, (1, 2, 4, 8)):
                os.environ.update(LLE_PARTIAL_E=str(E), LLE_PARTIAL_BATCHES=str(B), LLE_PARTIAL_WT=wt)
                __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
                us = timeit(lambda: bw.observe_as(_capi.LLE_OBS_PARTIAL, k, out=buf), iters=30, warm=3)
                row.append(f"E{E}B{B}:{us:6.1f}")
            print(label, f"{k}x{k} wt={wt}", " ".join(row), flush=True)
