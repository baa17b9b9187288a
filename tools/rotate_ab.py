"""Row rotation (obs_stream.hpp row_rotation, LLE_ROW_ROTATE=0 / 1) A/B on one box: the step kernel and its row-fill probe, with the
alternating walk off (every row to DRAM) and on, on three arenas per size (the write rate depends on the allocation).  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import algo_bytes, stepper, timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, _capi, mapgen  # noqa: E402


def setenv(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _capi.refresh_tuning()


for label, mk, sizes in (("level 6", lambda: Map(level=6), (65536, 131072, 262144, 524288)),
                         ("level 1", lambda: Map(level=1), (65536, 524288)),
                         ("generated 16x16 a4", lambda: Map(mapgen.generate(16, 16, 4, 4, seed=3)), (65536,)),
                         ):
    for n in sizes:
        m = mk()
        if m is None:
            continue
        keep = []
        for arena in range(3):
            bw = BatchedWorld(m, n)
            keep.append(bw)  # (alive: the next arena is other memory)
            step, probe = stepper(bw), bw.row_fill_prober()
            cells = []
            for pp in ("0", "1"):
                for rot in ("0", "1"):
                    setenv(LLE_PINGPONG=pp, LLE_ROW_ROTATE=rot)
                    cells.append(f"walk{pp} rot{rot}: step {timeit(step, iters=60, warm=6):6.1f} fill {timeit(probe, iters=60, warm=6):6.1f}")
                    bw.observe()
            setenv(LLE_PINGPONG=None, LLE_ROW_ROTATE=None)
            print(f"{label} n={n} ({m.obs_stride * n / 1e6:.0f} MB, {algo_bytes(m) * n / 1e6:.0f} MB algorithmic) arena {arena}: " + " | ".join(cells), flush=True)
        del keep, bw, step, probe
        torch.cuda.empty_cache()
