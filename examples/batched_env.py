"""The reference's LLE environment, batched: reset / step / available_actions with the same option names
(obs_type, state_type, walkable_lasers, randomize_lasers, multi_objective), device tensors with a leading env axis."""
import torch

from lle_amd import BatchedLLE, Map

env = BatchedLLE(Map(level=6), 4096, obs_type="partial5x5", state_type="normalized-state", walkable_lasers=False,
                 randomize_lasers=True, seed=0)
obs, state = env.reset()
print("obs", tuple(obs.shape), obs.dtype, "| state", tuple(state.shape), state.dtype)
episode_return = torch.zeros(env.n_envs, device=obs.device)
for t in range(200):
    avail = env.available_actions()                              # bool [n, A, 5]; never empty for an alive agent
    logits = torch.rand(avail.shape, device=avail.device).masked_fill(~avail, -1.0)
    actions = logits.argmax(-1)                                  # a random available action per agent
    dead_ends = ~avail.any(-1)                                   # (walkable_lasers=False can leave a corpse without any)
    actions[dead_ends] = 4
    out = env.step(actions, auto_reset=True)                     # finished envs restart first, with fresh laser colours
    episode_return += out["reward"][:, 0]
print("mean reward per env over 200 steps:", float(episode_return.mean()), "| done now:", int(out["done"].sum()))
