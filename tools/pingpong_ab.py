"""Single-step launches past the Infinity Cache with the walk of the environments alternating (LLE_PINGPONG=1) or not (=0).  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import algo_bytes, stepper, timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, mapgen  # noqa: E402

for label, mk, sizes in (("level 6", lambda: Map(level=6), (65536, 131072, 196608, 262144, 393216, 524288)),
                         ("config 5", lambda: Map(mapgen.config5(0)), (8192, 16384, 32768, 65536))):
    for n in sizes:
        m = mk()
        bw = BatchedWorld(m, n)
        step, probe = stepper(bw), bw.row_fill_prober()
        row = []
        for pp in ("0", "1", "0", "1"):
            os.environ["LLE_PINGPONG"] = pp
            __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
            row.append(f"{pp}: step {timeit(step, iters=60, warm=6):6.1f} fill {timeit(probe, iters=60, warm=6):6.1f}")
            bw.observe()
        os.environ.pop("LLE_PINGPONG", None)
        __import__("lle_amd")._capi.refresh_tuning()
        auto = timeit(step, iters=60, warm=6)
        print(f"{label} n={n} ({m.obs_stride * n / 1e6:.0f} MB of rows): " + " | ".join(row)
              + f" | auto: step {auto:6.1f} us = {algo_bytes(m) * n / auto / 1e3:.0f} GB/s", flush=True)
        del bw, step, probe
        torch.cuda.empty_cache()
