import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from lle_prof import timeit, stepper
from lle_amd import BatchedWorld, Map, mapgen
from tests.parity_util import legal_colours
n = 262144
def run(label, bw):
    step = stepper(bw)
    row = []
    for pp in ("0", "1", "0", "1"):
        os.environ["LLE_PINGPONG"] = pp
        __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
        row.append(f"{pp}: {timeit(step, iters=40, warm=6):6.1f}")
    os.environ.pop("LLE_PINGPONG")
    __import__("lle_amd")._capi.refresh_tuning()
    print(f"{label}: " + " | ".join(row) + f"  ({bw.kernel_info()})", flush=True)
bw = BatchedWorld(Map(level=6), n)
rng = np.random.default_rng(0)
bw.set_sources(torch.from_numpy(legal_colours(bw.map, rng.integers(0, 4, size=(n, bw.map.n_sources), dtype=np.uint8))))
run("level 6 x 262144, per-env sources (MODE 5)", bw); del bw; torch.cuda.empty_cache()
bw = BatchedWorld([Map(level=6), Map(level=6)], n)
run("level 6 twice, two maps x 131072 (MODE 4)", bw); del bw; torch.cuda.empty_cache()
from lle_amd import BatchedLLE
env = BatchedLLE(Map(level=6), n, randomize_lasers=True)
env.reset()
acts = torch.zeros(n, 4, dtype=torch.uint8, device="cuda") + 4
for pp in ("0", "1", "0", "1"):
    os.environ["LLE_PINGPONG"] = pp
    __import__("lle_amd")._capi.refresh_tuning()  # (the library reads its overrides once per process)
    print(f"BatchedLLE.step fused, randomize_lasers, 262144 envs, LLE_PINGPONG={pp}: {timeit(lambda: env.step(acts, auto_reset=True, fused=True), iters=40, warm=6):6.1f} us", flush=True)
