"""Which of the fused LLE.step outputs costs what: us per step (HIP events) of level 6 x 65 536 with subsets of the outputs.  GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, _capi  # noqa: E402
from lle_amd.batched import _current_stream_handle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def run(names, write_obs=True, pes=False):
    bw = BatchedWorld(Map(level=6), n)
    A, G = bw.map.n_agents, bw.map.n_gems
    if pes:
        g = torch.Generator(device="cuda").manual_seed(1)
        bw.set_sources(colours=torch.randint(0, A, (n, bw.map.n_sources), generator=g, device="cuda", dtype=torch.uint8))
    every = dict(state=torch.empty((n, 3 * A + G), device="cuda"), reward=torch.empty(n, device="cuda"), done=torch.empty(n, dtype=torch.uint8, device="cuda"),
                 available=torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"), alive=torch.empty((n, A), dtype=torch.uint8, device="cuda"))
    keep = {k: every[k] for k in names}
    eo = bw.make_env_outputs(**keep)
    L, h, dev = _capi.lib(), C.c_void_p(bw.h), bw.device
    flags = _capi.LLE_STEP_SAMPLE_ACTIONS | _capi.LLE_STEP_AUTO_RESET | (0 if write_obs else _capi.LLE_STEP_NO_OBS)
    tt = [0]

    def step():
        rc = L.lle_batch_step_outputs(h, None, flags, 1, tt[0], 0, C.byref(eo), _current_stream_handle(dev))
        assert rc == 0, rc
        tt[0] += 1
    us = min(timeit(step, iters=200, warm=30) for _ in range(3))
    del bw
    torch.cuda.empty_cache()
    return us


for pes in (False, True):
    print("per-env sources" if pes else "the map's sources")
    for names in ((), ("done",), ("reward", "done"), ("available",), ("state",), ("state", "reward", "done", "available"), ("state", "reward", "done", "available", "alive")):
        print(f"  {'+'.join(names) or 'none (general kernel)':45s} rows {run(names, True, pes):6.2f}   no rows {run(names, False, pes):6.2f}", flush=True)

# the general kernel WITHOUT an output descriptor (two copies of the map), against the default kernel
from lle_prof import stepper  # noqa: E402
for label, mk in (("default kernel, one map", lambda: BatchedWorld(Map(level=6), n)), ("general kernel, two copies of the map, no descriptor", lambda: BatchedWorld([Map(level=6), Map(level=6)], n))):
    bw = mk()
    fn = bw.sampled_stepper(seed=1)
    print(f"  {label:55s} rows {min(timeit(fn, iters=200, warm=30) for _ in range(3)):6.2f}", flush=True)
    del bw, fn
    torch.cuda.empty_cache()
