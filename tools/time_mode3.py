"""Per-env sources at 65 536 envs on many-source maps: single step (MODE 5 / 8) vs fused rollout (MODE 3), us per step.  GPU box."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from lle_prof import timeit
from lle_amd import BatchedWorld, mapgen
from tests.parity_util import EXTRA_MAPS, legal_colours
n = 65536
for name, text in (("gen_20_lasers (4 agents)", EXTRA_MAPS["gen_20_lasers"]), ("4 agents 8 lasers", mapgen.generate(12, 13, 4, 8, 4, seed=2)),
                   ("2 agents 8 lasers", mapgen.generate(12, 13, 2, 8, 4, seed=2)), ("2 agents 12 lasers", mapgen.generate(12, 13, 2, 12, 4, seed=3)),
                   ("1 agent 20 lasers", mapgen.generate(14, 14, 1, 20, 3, seed=4))):
    bw = BatchedWorld(text, n)
    rng = np.random.default_rng(0)
    bw.set_sources(torch.from_numpy(legal_colours(bw.map, rng.integers(0, bw.map.n_agents, size=(n, bw.map.n_sources), dtype=np.uint8))))
    single = min(timeit(lambda: bw.step(sample=True, auto_reset=True, seed=1), iters=60, warm=5) for _ in range(2))
    T = 8
    roll = min(timeit(lambda: bw.rollout(T, auto_reset=True, seed=1), iters=20, warm=3) for _ in range(2)) / T
    print(f"{os.environ.get('LABEL', ''):10s} {name:26s} per-env sources, 65536 envs: single step {single:7.1f} us | fused rollout {roll:7.1f} us per step ({bw.kernel_info()['kernel']})", flush=True)
    del bw
    torch.cuda.empty_cache()
