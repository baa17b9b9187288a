import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map
def mk(n, pes):
    bw = BatchedWorld(Map(level=6), n)
    if pes:
        g = torch.Generator().manual_seed(0)
        bw.set_sources(torch.randint(0, 4, (n, 3), generator=g, dtype=torch.uint8))
    return bw
for n in (8192, 16384, 32768, 65536, 131072, 196608, 262144, 524288):
    for pes in (False, True):
        bw = mk(n, pes)
        A, G = bw.map.n_agents, bw.map.n_gems
        st, rw, av = (torch.empty((n, 3 * A + G), device="cuda"), torch.empty((n, 1), device="cuda"), torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"))
        eo = bw.make_env_outputs(state=st, reward=rw, available=av)
        row = []
        for label, kw in (("step", dict()), ("fused", dict(env_out=eo))):
            r = []
            for heads in ("0", "1", "0", "1"):
                os.environ["LLE_ROW_HEADS"] = heads
                __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
                r.append(timeit(lambda: bw.step(sample=True, auto_reset=True, seed=1, **kw), iters=60 if n > 65536 else 200, warm=10))
            os.environ.pop("LLE_ROW_HEADS")
            __import__("lle_amd")._capi.refresh_tuning()
            row.append(f"{label}: heads0 {min(r[0], r[2]):6.2f} heads1 {min(r[1], r[3]):6.2f}")
        print(f"n={n:6d} pes={int(pes)}: " + " | ".join(row), flush=True)
        del bw, eo, st, rw, av
        torch.cuda.empty_cache()
