"""Where the per-env-sources step kernel (MODE 5) loses against the default one: sampled steps, with / without the observation,
with / without row heads, with / without the recolouring of resets and the fused outputs."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def mk(pes):
    bw = BatchedWorld(Map(level=6), n)
    if pes:
        g = torch.Generator().manual_seed(0)
        bw.set_sources(torch.randint(0, 4, (n, 3), generator=g, dtype=torch.uint8))
    return bw
for heads in ("1", "0"):
    os.environ["LLE_ROW_HEADS"] = heads
    __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
    for pes in (False, True):
        bw = mk(pes)
        st, rw, av = (torch.empty((n, 16), device="cuda"), torch.empty((n, 1), device="cuda"), torch.empty((n, 4, 5), dtype=torch.uint8, device="cuda"))
        eo = bw.make_env_outputs(state=st, reward=rw, available=av)
        row = []
        for label, kw in (("step", dict()), ("noobs", dict(write_obs=False)), ("fused", dict(env_out=eo)),
                          ("recolour", dict(recolour_resets=True)), ("recolour+fused", dict(recolour_resets=True, env_out=eo))):
            if "recolour" in label and not pes:
                continue
            us = timeit(lambda: bw.step(sample=True, auto_reset=True, seed=1, **kw), iters=200, warm=20)
            row.append(f"{label} {us:6.2f}")
        print(f"heads={heads} pes={pes}: " + " | ".join(row), flush=True)
        del bw
