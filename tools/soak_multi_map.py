"""Soak differential for batches of MANY maps (lle_batch_create_multi): every round builds a batch of distinct generated maps with 8 / 16 / 24 / 64 / 1 / 3 / 5 / 2
environments each -- config 5's shape (split rows: the bit-form template, and the packed table image where a map's block fills four wavefronts) and a
small shape (whole rows) --, steps it with sampled actions and auto-reset, and compares every buffer of every env with that map's own oracle batch after
every step (state, ordered events, availability, error codes, the full observation), a fused rollout of 4 steps every 8th step.
Usage: python tools/soak_multi_map.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as om  # noqa: E402  (test infrastructure: this tool is a checker, not product)
from tests.parity_util import assert_state_equal, assert_step_equal, unpack_engine  # noqa: E402
from lle_amd import BatchedWorld, mapgen  # noqa: E402

om.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
shapes = {"config5": lambda s: mapgen.config5(s),
          "gen_12x13_a4_l4": lambda s: mapgen.generate(12, 13, 4, 4, 4, seed=s, n_voids=2),
          "gen_16x16_a8_l6": lambda s: mapgen.generate(16, 16, 8, 6, 5, seed=s, n_voids=3)}
t_end = time.time() + budget
total, rnd, seed0 = 0, 0, 5000
while time.time() < t_end:
    for shape, gen in shapes.items():
        per = (8, 16, 24, 64, 1, 3, 5, 2)[rnd % 8]
        n_maps = 96 if shape == "config5" else 160
        texts = []
        while len(texts) < n_maps:
            seed0 += 1
            try:
                texts.append(gen(seed0))
            except RuntimeError:
                pass  # (the rejection sampler found no placement for this seed)
        n = n_maps * per
        bw = BatchedWorld(texts, n)
        obs = [om.OracleBatch(t, per) for t in texts]
        dims = [(ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W) for ob in obs]

        def check(osteps, where):
            bufs = bw.host_buffers()
            for m, ob in enumerate(obs):
                eng = unpack_engine({k: v[m * per:(m + 1) * per] for k, v in bufs.items()}, *dims[m])
                if osteps is not None:
                    assert_step_equal(eng, osteps[m], f"{where} map {m}")
                assert_state_equal(eng, ob.dump(), f"{where} map {m}")
        t, steps = 0, 0
        while steps < 48 and time.time() < t_end + 20:
            if t % 8 == 7:
                bw.rollout(4, auto_reset=True, seed=77 + rnd, t=t, env_offset=3)
                for tt in range(t, t + 4):
                    osteps = [ob.step(None, auto_reset=True, seed=77 + rnd, t=tt, env_offset=3 + m * per) for m, ob in enumerate(obs)]
                check(osteps, f"{shape} x{per} rollout t={t}")
                t, steps = t + 4, steps + 4
            else:
                auto = t >= 4
                bw.step(sample=True, auto_reset=auto, seed=77 + rnd, t=t, env_offset=3)
                check([ob.step(None, auto_reset=auto, seed=77 + rnd, t=t, env_offset=3 + m * per) for m, ob in enumerate(obs)], f"{shape} x{per} t={t}")
                t, steps = t + 1, steps + 1
        st = bw.stats()
        total += steps * n
        print(f"{shape:18s} {n_maps:4d} maps x {per:2d} envs x {steps:3d} steps = {steps * n:9d} env-steps bit-exact "
              f"(deaths {st['deaths']}, gems {st['gems']}, exits {st['exits']}, auto-resets {st['auto_resets']}) {bw.kernel_info()}", flush=True)
        del bw
    rnd += 1
print(f"total {total} env-steps over {rnd} rounds, every buffer equal after every step")
