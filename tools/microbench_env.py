"""BatchedLLE.step end to end (us per call, 65 536 envs, level 6) and its pieces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedLLE, Map


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


n = 65536
for kw in (dict(), dict(walkable_lasers=False), dict(multi_objective=True), dict(obs_type="partial7x7"), dict(randomize_lasers=True)):
    env = BatchedLLE(Map(level=6), n, **kw)
    env.reset()
    w = env.world
    acts = torch.full((n, env.n_agents), 4, dtype=torch.uint8, device=w.device)
    print(f"{kw}: step(auto_reset) {timeit(lambda: env.step(acts, auto_reset=True)):.1f} us | world.step {timeit(lambda: w.step(acts, auto_reset=True)):.1f}"
          f" | get_state {timeit(env.get_state):.1f} | reward {timeit(env.reward):.1f} | done {timeit(lambda: env.done):.1f}"
          f" | available_actions {timeit(env.available_actions):.1f} | get_observation {timeit(env.get_observation):.1f}", flush=True)
