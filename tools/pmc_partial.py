"""Target for rocprofv3 --pmc runs: the partial k x k observer alone (level 6, 65 536 envs, k = 7), 40 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map, _capi
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
bw = BatchedWorld(Map(level=6), 65536)
for t in range(20):
    bw.step(sample=True, auto_reset=True, seed=1, t=t)
for _ in range(40):
    bw.observe_as(_capi.LLE_OBS_PARTIAL, k)
torch.cuda.synchronize()
