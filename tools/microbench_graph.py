"""Does replaying the single-step launches from a HIP graph shorten the launch boundary?  20 steps captured, replayed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map

n = 65536
bw = BatchedWorld(Map(level=6), n)
K = 20
def steps(t0):
    for t in range(t0, t0 + K):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
steps(0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(10):
    steps(K * (r + 1))
e1.record(); torch.cuda.synchronize()
print(f"plain launches: {e0.elapsed_time(e1) * 1e3 / (10 * K):.2f} us per step", flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    steps(1000)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        steps(2000)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record(s)
    for r in range(10):
        g.replay()
    e1.record(s); torch.cuda.synchronize()
print(f"graph replay:   {e0.elapsed_time(e1) * 1e3 / (10 * K):.2f} us per step", flush=True)
