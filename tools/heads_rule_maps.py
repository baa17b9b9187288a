import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, mapgen
from tests.parity_util import EXTRA_MAPS, legal_colours
maps = {"level1": Map(level=1), "level3": Map(level=3), "level5": Map(level=5), "gen 12x13 4 agents 8 lasers": Map(mapgen.generate(12, 13, 4, 8, 4, seed=2)),
        "gen 16x16 8 agents 4 lasers": Map(mapgen.generate(16, 16, 8, 4, 4, seed=1)), "nested": Map(EXTRA_MAPS["nested"])}
for name, m in maps.items():
    for n in (1024, 4096, 65536, 262144):
        for pes in (False, True):
            if pes and m.n_sources == 0: continue
            bw = BatchedWorld(m, n)
            if pes:
                rng = np.random.default_rng(0)
                bw.set_sources(torch.from_numpy(legal_colours(bw.map, rng.integers(0, m.n_agents, size=(n, m.n_sources), dtype=np.uint8))))
            A, G = m.n_agents, m.n_gems
            st, rw, av = (torch.empty((n, 3 * A + G), device="cuda"), torch.empty((n, 1), device="cuda"), torch.empty((n, A, 5), dtype=torch.uint8, device="cuda"))
            eo = bw.make_env_outputs(state=st, reward=rw, available=av)
            row = []
            for label, kw in (("step", dict()), ("fused", dict(env_out=eo))):
                if label == "step" and not pes: continue   # (MODE 0 / 6: rule unchanged)
                r = []
                for heads in ("0", "1", "0", "1"):
                    os.environ["LLE_ROW_HEADS"] = heads
                    __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
                    r.append(timeit(lambda: bw.step(sample=True, auto_reset=True, seed=1, **kw), iters=40 if n > 65536 else 150, warm=10))
                os.environ.pop("LLE_ROW_HEADS")
                __import__("lle_amd")._capi.refresh_tuning()
                auto = timeit(lambda: bw.step(sample=True, auto_reset=True, seed=1, **kw), iters=40 if n > 65536 else 150, warm=10)
                row.append(f"{label}: {min(r[0], r[2]):7.2f} / {min(r[1], r[3]):7.2f} (auto {auto:7.2f})")
            print(f"{name:28s} n={n:6d} pes={int(pes)} head {m.row_head} {bw.kernel_info()['kernel']}: " + " | ".join(row), flush=True)
            del bw, eo, st, rw, av
            torch.cuda.empty_cache()
