"""Target for rocprofv3 --pmc runs: 60 steps of level 6 / 65 536 envs, with or without the observation write."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map
obs = len(sys.argv) < 2 or sys.argv[1] != "noobs"
bw = BatchedWorld(Map(level=6), 65536)
for t in range(60):
    bw.step(sample=True, auto_reset=True, seed=1, t=t, write_obs=obs)
torch.cuda.synchronize()
