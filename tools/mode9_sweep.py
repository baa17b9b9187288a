"""Step kernel MODE 9 (the partial 7 x 7 observation written by the step launch) at level 6 x 65 536: the launcher's rule against the window sets
forced on (LLE_PARTIAL_SETS=1) and other batch sizes (LLE_PARTIAL_E): us per step of BatchedLLE(obs_type="partial7x7").step(fused=True)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedLLE, Map, _capi

n = 65536
for k in (7, 5):
    env = BatchedLLE(Map(level=6), n, seed=1, obs_type=f"partial{k}x{k}")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(30):
        a = torch.multinomial(env.available_actions().reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
        env.step(a, auto_reset=True, fused=True)
    acts = torch.full((n, 4), 4, dtype=torch.uint8, device="cuda")
    step = lambda: env.step(acts, auto_reset=True, fused=True)
    for sets in ("", "1"):
        for E in ("", "1", "2", "4"):
            for name, v in (("LLE_PARTIAL_SETS", sets), ("LLE_PARTIAL_E", E)):
                if v:
                    os.environ[name] = v
                else:
                    os.environ.pop(name, None)
            _capi.refresh_tuning()
            try:
                us = min(timeit(step, iters=150, warm=20) for _ in range(3))
                print(f"partial{k}x{k} MODE 9: sets={sets or 'rule'} E={E or 'rule'}: {us:6.2f} us", flush=True)
            except Exception as e:
                print(f"partial{k}x{k} sets={sets} E={E}: {e!r}"[:160], flush=True)
    os.environ.pop("LLE_PARTIAL_SETS", None); os.environ.pop("LLE_PARTIAL_E", None)
    _capi.refresh_tuning()
    del env
