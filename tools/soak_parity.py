"""Soak differential (GPU box): HIP path vs the CPU oracle on long random rollouts, every buffer of every env compared
bit-for-bit after every step (state, ordered events, availability, error codes, the full int8 observation).
Beyond the test suite's sizes; prints env-steps compared per map.  Usage: python tools/soak_parity.py [seconds_per_map]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as om  # noqa: E402  (test infrastructure: this tool is a checker, not product)
from oracle.levels import LEVELS  # noqa: E402
from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, unpack_engine  # noqa: E402
from lle_amd import BatchedWorld, mapgen  # noqa: E402

om.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
maps = {f"level{k}": (v, 32768) for k, v in LEVELS.items()}
maps.update({k: (v, 8192) for k, v in EXTRA_MAPS.items()})
maps["config5"] = (mapgen.config5(0), 4096)
total = 0
for name, (text, n) in maps.items():
    ob, bw = om.OracleBatch(text, n), BatchedWorld(text, n)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    t, t0 = 0, time.time()
    while time.time() - t0 < budget:
        auto = (t // 64) % 2 == 0  # alternate: auto-reset regime / episodes running into all-dead, all-STAY states (Q1, Q2)
        bw.step(sample=True, auto_reset=auto, seed=2026, t=t, env_offset=11)
        ostep = ob.step(None, auto_reset=auto, seed=2026, t=t, env_offset=11)
        eng = unpack_engine(bw.host_buffers(), *dims)
        assert_step_equal(eng, ostep, f"{name} t={t}")
        assert_state_equal(eng, ob.dump(), f"{name} t={t}")
        t += 1
    s = bw.stats()
    total += n * t
    print(f"{name:28s} {n:6d} envs x {t:5d} steps = {n * t:11d} env-steps bit-exact "
          f"(deaths {s['deaths']}, gems {s['gems']}, exits {s['exits']}, auto-resets {s['auto_resets']})", flush=True)
    del bw, ob
print(f"total {total} env-steps, every buffer equal after every step")
