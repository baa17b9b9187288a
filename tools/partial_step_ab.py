"""BatchedLLE(obs_type="partial7x7").step in one launch (step kernel MODE 9) and the standalone partial observers: us per call."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedLLE, BatchedWorld, Map, _capi

n = 65536
for k in (3, 5, 7):
    env = BatchedLLE(Map(level=6), n, seed=1, obs_type=f"partial{k}x{k}")
    env.reset()
    acts = torch.full((n, 4), 4, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)

    def step():
        env.step(acts, auto_reset=True, fused=True)
    for _ in range(30):  # some movement first
        a = torch.multinomial(env.available_actions().reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
        env.step(a, auto_reset=True, fused=True)
    print(f"partial{k}x{k} one launch: {min(timeit(step, iters=200, warm=20) for _ in range(3)):6.2f} us", flush=True)
    del env
bw = BatchedWorld(Map(level=6), n)
for t in range(30):
    bw.step(sample=True, auto_reset=True, seed=1, t=t)
for k in (3, 5, 7):
    call = bw.bound_observer(_capi.LLE_OBS_PARTIAL, k)
    print(f"partial{k}x{k} observer:   {min(timeit(call, iters=200, warm=20) for _ in range(3)):6.2f} us", flush=True)
