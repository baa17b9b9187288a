"""Where the per-environment-sources tax of a step goes: us per launch (HIP events, 65 536 envs of level 6 unless given) for
plain / + fused outputs / per-env sources / + outputs / + recolouring resets, each with and without observation rows.  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import timeit  # noqa: E402

from lle_amd import BatchedWorld, Map  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def run(pes, outputs, recolour, write_obs, incr=False):
    bw = BatchedWorld(Map(level=level), n)
    A, G = bw.map.n_agents, bw.map.n_gems
    if pes:
        g = torch.Generator(device="cuda").manual_seed(1)
        bw.set_sources(colours=torch.randint(0, A, (n, bw.map.n_sources), generator=g, device="cuda", dtype=torch.uint8))
    eo = None
    if outputs:
        keep = dict(state=torch.empty((n, 3 * A + G), device="cuda"), reward=torch.empty((n, 4), device="cuda") if False else torch.empty(n, device="cuda"),
                    done=torch.empty(n, dtype=torch.uint8, device="cuda"), available=torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"))
        eo = bw.make_env_outputs(**keep)
        bw._keep = keep

    import ctypes as C

    from lle_amd import _capi
    from lle_amd.batched import _current_stream_handle
    L, h, dev = _capi.lib(), C.c_void_p(bw.h), bw.device
    flags = (_capi.LLE_STEP_SAMPLE_ACTIONS | _capi.LLE_STEP_AUTO_RESET | (_capi.LLE_STEP_RECOLOUR_RESETS if recolour else 0) |
             (0 if write_obs else _capi.LLE_STEP_NO_OBS) | (_capi.LLE_STEP_INCREMENTAL_OBS if incr else 0))
    tt = [0]

    def step():
        if eo is not None:
            rc = L.lle_batch_step_outputs(h, None, flags, 1, tt[0], 0, C.byref(eo), _current_stream_handle(dev))
        else:
            rc = L.lle_batch_step(h, None, flags, 1, tt[0], 0, _current_stream_handle(dev))
        assert rc == 0, rc
        tt[0] += 1
    us = min(timeit(step, iters=200, warm=30) for _ in range(3))
    name = bw.kernel_info()
    del bw
    torch.cuda.empty_cache()
    return us, name


print(f"level {level} x {n} envs: us per step (launch-to-launch, HIP events)")
for label, pes, outputs, recolour in (("plain", 0, 0, 0), ("plain + outputs", 0, 1, 0), ("per-env sources", 1, 0, 0), ("per-env sources + outputs", 1, 1, 0),
                                      ("per-env sources + outputs + recolour", 1, 1, 1)):
    a, info = run(pes, outputs, recolour, True)
    b, _ = run(pes, outputs, recolour, False)
    c, _ = run(pes, outputs, recolour, True, incr=True)
    print(f"{label:40s} rows {a:6.2f}   no rows {b:6.2f}   incremental rows {c:6.2f}   {info}", flush=True)
