"""Fused rollouts with per-environment sources (step_kernel MODE 3, beam masks in the LDS record) against single steps (MODE 5 / 8),
every buffer after every launch, on maps with 8-20 sources; sources re-drawn every 64 steps.  GPU box: python tools/soak_mode3.py [seconds per map]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lle_amd import BatchedWorld, mapgen
from tests.parity_util import EXTRA_MAPS, legal_colours
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
maps = {"gen_12x13_4agents_8lasers": mapgen.generate(12, 13, 4, 8, 4, seed=2), "many_agents": EXTRA_MAPS["many_agents"],
        "gen_20_lasers": EXTRA_MAPS["gen_20_lasers"], "config5_32x32": EXTRA_MAPS["config5_32x32"], "gen_16x16_12agents": EXTRA_MAPS["gen_16x16_12agents"]}
rng = np.random.default_rng(1)
for name, text in maps.items():
    n = 2048
    a, b = BatchedWorld(text, n), BatchedWorld(text, n)
    A, L = a.map.n_agents, a.map.n_sources
    t, t0, T = 0, time.time(), 8
    ring = a.make_ring(3)
    while time.time() - t0 < budget:
        if t % 64 == 0:
            colours = torch.from_numpy(legal_colours(a.map, rng.integers(0, A, size=(n, L), dtype=np.uint8)))
            enabled = torch.from_numpy((rng.integers(0, 1 << min(L, 30), size=n) | (rng.integers(0, 2, size=n) * ((1 << L) - 1))).astype(np.int64).astype(np.int32))
            mask = torch.from_numpy((rng.random(n) < 0.5).astype(np.uint8))
            for w in (a, b):
                w.set_sources(colours, enabled, mask)
        a.rollout(T, auto_reset=(t // 64) % 2 == 0, seed=5, t=t, ring=ring if (t // 8) % 2 else None, ring_pos=t)
        for k in range(T):
            b.step(sample=True, auto_reset=(t // 64) % 2 == 0, seed=5, t=t + k)
        for key in ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done"):
            assert torch.equal(getattr(a, key), getattr(b, key)), (name, t, key)
        if (t // 8) % 2:
            assert torch.equal(ring["obs"][(t + T - 1) % 3], b.obs), (name, t, "ring obs")
        else:
            assert torch.equal(a.obs, b.obs), (name, t, "obs")
        t += T
    print(f"{name:28s} {n} envs x {t} steps: fused rollouts with per-env sources == single steps ({a.kernel_info()})", flush=True)
