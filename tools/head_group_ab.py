"""Row heads stored by every wavefront (1), by every second one for itself and its neighbour (2), by the first of a workgroup for all
of it (4): us per step on the current step kernel, and the observation buffers compared.  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, _capi, mapgen  # noqa: E402

for label, mk, sizes in (("level 6", lambda: Map(level=6), (32768, 65536, 131072, 262144)), ("level 5", lambda: Map(level=5), (65536,)),
                         ("generated 16x16 a4 l4", lambda: Map(mapgen.generate(16, 16, 4, 4, seed=3)), (65536,))):
    for n in sizes:
        cells, ref = [], None
        for rep in range(2):
            for g in (1, 2, 4):
                os.environ["LLE_HEAD_GROUP"] = str(g)
                _capi.refresh_tuning()
                bw = BatchedWorld(mk(), n)
                step = bw.sampled_stepper(seed=1)
                cells.append(f"{g}: {min(timeit(step, iters=200, warm=20) for _ in range(2)):6.2f}")
                if rep == 0:
                    bw2 = BatchedWorld(mk(), n)
                    for t in range(6):
                        bw2.step(sample=True, auto_reset=True, seed=5, t=t)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = bw2.obs.clone()
                    else:
                        assert torch.equal(ref, bw2.obs), (label, n, g)
                    del bw2
                del bw, step
                torch.cuda.empty_cache()
        print(f"{label} n={n}: " + "  ".join(cells), flush=True)
