"""LLE_STEP_INCREMENTAL_OBS A/B: the step kernel writing whole rows vs only the lines dynamic state can change.  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, mapgen  # noqa: E402

for label, mk, sizes in (("level 6", lambda: Map(level=6), (4096, 16384, 65536, 131072, 262144)), ("level 1", lambda: Map(level=1), (4096, 65536)),
                         ("level 5", lambda: Map(level=5), (65536,)), ("config 5", lambda: Map(mapgen.config5(0)), (16384, 65536))):
    for n in sizes:
        m = mk()
        bw = BatchedWorld(m, n)
        full, incr = bw.sampled_stepper(seed=1), bw.sampled_stepper(seed=1, incremental_obs=True)
        row = []
        for rep in range(2):
            row.append(f"full {timeit(full, iters=100, warm=10):6.2f} incr {timeit(incr, iters=100, warm=10):6.2f}")
        h = m.row_head
        print(f"{label} n={n} rows {m.obs_stride} B: " + " | ".join(row), flush=True)
        del bw
        torch.cuda.empty_cache()
