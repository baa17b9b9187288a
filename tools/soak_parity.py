"""Soak differential (GPU box): HIP path vs the CPU oracle on long random rollouts, every buffer of every env compared
bit-for-bit after every step (state, ordered events, availability, error codes, the full int8 observation).
Beyond the test suite's sizes; prints env-steps compared per map.
Usage: python tools/soak_parity.py [seconds_per_map] [per-env-sources | full-size | recolour-resets | rollouts | exits | incremental | dtype]
With `dtype` the batch's rows are fp16 / bf16 / fp32 in turn (lle_batch_options.obs_dtype: widened at the store, round 5) -- single steps, and every 16th
step a fused rollout of 4 steps in place; the values must be the oracle's int8 tensor exactly.
With `incremental` every step carries LLE_STEP_INCREMENTAL_OBS (only the lines of a row that dynamic state can change are written):
the observation compared after every step is the buffer's whole content.
With `exits` the exits MOVE every 40 steps (World.exit_pos = [...], world.rs:195-234: lle_map_set_exits + lle_batch_update_map on
the GPU side, set_exit_positions on every oracle world): random legal cells, one of them under a beam where the map has one.
With `rollouts` the GPU side runs lle_batch_rollout (8 steps per launch into a trajectory ring of 8 slots: the fused
kernels, MODE 1) and the oracle the same 8 steps one by one: every slot of the observation / action / reward rings and
the final state of every launch are compared.
With `recolour-resets` every finished env is reset AND re-coloured inside the step kernel (LLE_STEP_AUTO_RESET |
LLE_STEP_RECOLOUR_RESETS: LLE.reset with randomize_lasers); the oracle side resets such an env, sets the colours of the
documented draws (hash of seed ^ RECOLOUR_SALT, env, t, source; uniform over the colours the source may take) and steps.
Maps without sources or with a cell of more than two laser layers are skipped.
With `full-size` only the BASELINE configurations at their full batch (and level 6 at 262 144 envs, the HBM-regime run of
bench.py) are compared, a few steps each: the oracle side then dominates the time.
With `per-env-sources` every env has its own source colours / enabled flags (lle_batch_set_sources; each oracle env is
its own world object and receives the same set_colour / enable / disable calls), re-drawn every 48 steps for a random
half of the envs: the general step-kernel modes.  Maps without sources are skipped."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as om  # noqa: E402  (test infrastructure: this tool is a checker, not product)
from oracle.levels import LEVELS  # noqa: E402
from tests.parity_util import EXTRA_MAPS, LONG_MAPS, assert_state_equal, assert_step_equal, legal_colours, unpack_engine  # noqa: E402
from lle_amd import BatchedWorld, mapgen  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

om.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
per_env = len(sys.argv) > 2 and sys.argv[2] == "per-env-sources"
full_size = len(sys.argv) > 2 and sys.argv[2] == "full-size"
recolour = len(sys.argv) > 2 and sys.argv[2] == "recolour-resets"
rollouts = len(sys.argv) > 2 and sys.argv[2] == "rollouts"
exits_mode = len(sys.argv) > 2 and sys.argv[2] == "exits"
incremental = len(sys.argv) > 2 and sys.argv[2] == "incremental"
dtype_mode = len(sys.argv) > 2 and sys.argv[2] == "dtype"
DTYPES = [torch.float16, torch.bfloat16, torch.float32]
rng = np.random.default_rng(7)


class Mirror:
    """The same per-env source changes on the oracle worlds, with the Python binding's rule that enable / disable only
    act when the flag changes (pylaser_source.rs:55-75)."""

    def __init__(self, ob, n, L):
        self.ob, self.n, self.L = ob, n, L
        self.enabled = np.array([[bool(s[4]) for s in ob.world(0).sources()]] * n)

    def apply(self, colours, enabled, mask):
        for e in np.nonzero(mask)[0]:
            w = self.ob.world(int(e))
            for l in range(self.L):
                w.set_source(l, colour=int(colours[e, l]))
                want = bool((int(enabled[e]) >> l) & 1)
                if want != self.enabled[e, l]:
                    w.set_source(l, enabled=want)
                    self.enabled[e, l] = want


def redraw(bw, mirror, A, L, n):
    colours = legal_colours(bw.map, rng.integers(0, A, (n, L)).astype(np.uint8))  # (colours that cross a start are refused)
    enabled = (rng.integers(0, 1 << min(L, 30), n) | rng.integers(0, 2, n) * ((1 << L) - 1)).astype(np.int64) & ((1 << L) - 1)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    bw.set_sources(colours=torch.from_numpy(colours).cuda(), enabled=torch.from_numpy(enabled.astype(np.int32)).cuda(),
                   env_mask=torch.from_numpy(mask).cuda())
    mirror.apply(colours, enabled, mask)

maps = {f"level{k}": (v, 32768) for k, v in LEVELS.items()}
maps.update({k: (v, 8192) for k, v in EXTRA_MAPS.items()})
maps.update({k: (v, 8192) for k, v in LONG_MAPS.items()})  # beams longer than 32 cells: chains of beam words (round 4)
maps["gen_3x60_long_beams"] = (mapgen.generate(3, 60, 3, 3, 2, wall_fraction=0.0, n_voids=1, seed=4, max_beam=254), 8192)
maps["config5"] = (mapgen.config5(0), 4096)
# 5-8 sources with few agents: beam masks in the LDS record AND row heads (32 768 envs = 2 048 wavefronts: heads on)
maps["gen_12x13_4agents_8lasers"] = (mapgen.generate(12, 13, 4, 8, 4, seed=2), 32768)
maps["gen_12x13_2agents_8lasers"] = (mapgen.generate(12, 13, 2, 8, 4, seed=2), 65536)
if full_size:
    maps = {"cfg2 level1 x 4096": (LEVELS[1], 4096), "cfg3 level6 x 65536": (LEVELS[6], 65536), "level6 x 262144": (LEVELS[6], 262144),
            "cfg5 32x32 x 65536": (mapgen.config5(0), 65536)}
total = 0
for name, (text, n) in maps.items():
    if per_env or recolour:
        n = min(n, 2048)  # (the per-env source calls on the oracle side are Python loops)
    if rollouts:
        n = min(n, 8192)
    if exits_mode:
        n = min(n, 4096)  # (set_exit_positions on the oracle side is one call per world)
    ob, bw = om.OracleBatch(text, n), BatchedWorld(text, n, obs_dtype=DTYPES[len(name) % 3] if dtype_mode else None)
    dims = ob.dims
    L = bw.map.n_sources
    if (per_env or recolour) and L == 0:
        continue
    if recolour and (bw.map.max_cell_layers > 2 or bw.map.n_beam_words != L):
        continue  # (the in-kernel draw is per beam word: maps with a beam of several words use lle_batch_reset_sources)
    mirror = Mirror(ob, n, L) if per_env else None
    t, t0 = 0, time.time()
    if recolour:
        from lle_amd._capi import RECOLOUR_SALT
        allowed = [[c for c in range(ob.A) if bw.map.colour_allowed(s, c)] for s in range(L)]
        start = legal_colours(bw.map, rng.integers(0, ob.A, (n, L)).astype(np.uint8))
        bw.set_sources(colours=torch.from_numpy(start).cuda())
        for e in range(n):
            for s in range(L):
                ob.world(e).set_source(s, colour=int(start[e, s]))
        colours = start.copy()
    if rollouts:
        T = 8
        ring = bw.make_ring(T)
    while rollouts and time.time() - t0 < budget:
        bw.rollout(T, auto_reset=True, seed=2026, t=t, env_offset=11, ring=ring, ring_pos=0)
        robs, ract, rrew = ring["obs"].cpu().numpy(), ring["actions"].cpu().numpy(), ring["reward"].cpu().numpy()
        for j in range(T):
            ostep = ob.step(None, auto_reset=True, seed=2026, t=t + j, env_offset=11)
            assert np.array_equal(robs[j], ostep["obs"]), f"{name} t={t + j}: ring obs"
            assert np.array_equal(ract[j], ostep["actions"]), f"{name} t={t + j}: ring actions"
            cnt = (ostep["ev_count"] & 0x7F).astype(np.int64)
            ev = np.where((np.arange(ostep["events"].shape[1])[None, :] < cnt[:, None]), ostep["events"][:, :, 0], 255)
            want = np.stack([(ev == 1).sum(1), (ev == 0).sum(1), (ev == 2).sum(1)], 1)  # gems, exits, deaths of the step
            assert np.array_equal(rrew[j][:, :3].astype(np.int64), want), f"{name} t={t + j}: ring reward counts"
        eng = unpack_engine(bw.host_buffers(), *dims)
        eng["actions"] = ostep["actions"]  # (with a ring the actions of a step are in its ring slot, compared above)
        assert_step_equal(eng, ostep, f"{name} t={t + T - 1} (last step of the launch)", check_obs=False)  # (obs: the ring)
        assert_state_equal(eng, ob.dump(), f"{name} t={t + T - 1}")
        t += T
    while recolour and time.time() - t0 < budget:
        over = bw.done.cpu().numpy().astype(bool)
        for e in np.nonzero(over)[0]:
            w = ob.world(int(e))
            w.reset()
            for s in range(L):
                if allowed[s]:
                    colours[e, s] = allowed[s][(om.action_hash(2026 ^ RECOLOUR_SALT, 11 + int(e), t, s) * len(allowed[s])) >> 16]
                    w.set_source(s, colour=int(colours[e, s]))
        bw.step(sample=True, auto_reset=True, recolour_resets=True, seed=2026, t=t, env_offset=11)
        ostep = ob.step(None, auto_reset=False, seed=2026, t=t, env_offset=11)
        eng = unpack_engine(bw.host_buffers(), *dims)
        assert np.array_equal(eng["ev_count"] >> 7, over.astype(np.uint8)), f"{name} t={t}: which envs were reset"
        eng["ev_count"] = eng["ev_count"] & 0x7F
        assert_step_equal(eng, ostep, f"{name} t={t}")
        assert_state_equal(eng, ob.dump(), f"{name} t={t}")
        assert np.array_equal(bw.src_colour.cpu().numpy()[:, :L], colours), f"{name} t={t}: colours"  # (words == sources on these maps)
        t += 1
    while not recolour and not rollouts and time.time() - t0 < budget:
        if per_env and t % 48 == 0:
            redraw(bw, mirror, ob.A, L, n)
            eng = unpack_engine(bw.host_buffers(), *dims)
            assert_state_equal(eng, ob.dump(), f"{name} t={t} after set_sources")
        if exits_mode and t % 40 == 39:
            from tests.test_gpu_exits import legal_exits
            new_exits = legal_exits(bw.map, rng, ob.A + int(rng.integers(0, 3)))
            bw.set_exits(new_exits)
            for e in range(n):
                ob.world(e).set_exits(new_exits)
            eng = unpack_engine(bw.host_buffers(), *dims)
            assert_state_equal(eng, ob.dump(), f"{name} t={t} after set_exits")
            assert np.array_equal(eng["obs"][:64], np.stack([ob.world(e).obs() for e in range(64)])), f"{name} t={t}: observation after set_exits"
        auto = (t // 64) % 2 == 0  # alternate: auto-reset regime / episodes running into all-dead, all-STAY states (Q1, Q2)
        if dtype_mode and t % 16 == 15:  # a fused rollout in place: the last step's rows are in `obs`
            bw.rollout(4, auto_reset=True, seed=2026, t=t, env_offset=11)
            for j in range(4):
                ostep = ob.step(None, auto_reset=True, seed=2026, t=t + j, env_offset=11)
            eng = unpack_engine(bw.host_buffers(), *dims)
            assert np.array_equal(eng["obs"], ostep["obs"]), f"{name} t={t}: rows after the fused rollout"
            assert_state_equal(eng, ob.dump(), f"{name} t={t} (rollout)")
            t += 4
            continue
        bw.step(sample=True, auto_reset=auto, seed=2026, t=t, env_offset=11, incremental_obs=incremental or (dtype_mode and t % 5 == 2))
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
