"""Random-policy data collection with lle_amd on one MI355X: 65 536 level-6 environments, 64 steps.

Two ways to do the same thing:
  1. one launch per step (`BatchedWorld.step`): the caller sees every step's tensors in place;
  2. one launch for 16 steps (`BatchedWorld.rollout`): observations / actions / reward counts of every step land in a
     trajectory ring, the state machine of one step runs under the observation stream of another.
"""
import time

import torch

from lle_amd import BatchedWorld, Map

n, steps = 65536, 64
bw = BatchedWorld(Map(level=6), n)

torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(steps):
    bw.step(sample=True, auto_reset=True, seed=0, t=t)      # uniform over each agent's available actions
    # bw.obs [n, C, H, W] int8, bw.reward [n, 4] u8 (gems, exits, deaths, all-arrived), bw.done [n] are now step t's
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"step():    {n * steps / dt / 1e9:.2f} G env-steps/s   {bw.stats(reset=True)}")

ring = bw.make_ring(8)                                       # 8 slots of [n, C*H*W] observations (+ actions, reward counts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps // 16):
    bw.rollout(16, auto_reset=True, seed=0, t=steps + 16 * k, ring=ring, ring_pos=16 * k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"rollout(): {n * steps / dt / 1e9:.2f} G env-steps/s   {bw.stats()}")
