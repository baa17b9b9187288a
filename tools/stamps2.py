"""Percentiles of the in-kernel timeline stamps, and how they depend on the wave index."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lle_amd import BatchedWorld, Map, _capi

n = 65536
bw = BatchedWorld(Map(level=6), n)
nb = n // 16
stamps = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
for t in range(30):
    bw.step(sample=True, auto_reset=True, seed=1, t=t)
torch.cuda.synchronize()
_capi.lib().lle_batch_step_stamped(bw.h, 3, 1, 30, stamps.data_ptr(), bw._stream())
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype("float64") * 0.01
t0 = s[:, 0].min()
names = ["entry", "tables", "state", "logic done", "stored", "obs issued", "drained", "rows copied"]
for i in [0, 7, 1, 2, 3, 4, 5, 6]:
    col = s[:, i] - t0
    print(f"{names[i]:12s} p10 {np.percentile(col,10):6.2f} p50 {np.percentile(col,50):6.2f} p90 {np.percentile(col,90):6.2f} p99 {np.percentile(col,99):6.2f} max {col.max():6.2f}")
d = s[:, 6] - t0
for q in range(8):
    sl = d[q * nb // 8:(q + 1) * nb // 8]
    print(f"waves [{q*nb//8},{(q+1)*nb//8}): drained p50 {np.percentile(sl,50):6.2f} max {sl.max():6.2f}; logic done p50 {np.percentile(s[q*nb//8:(q+1)*nb//8,3]-t0,50):6.2f}; entry p50 {np.percentile(s[q*nb//8:(q+1)*nb//8,0]-t0,50):6.2f}")
slow = np.argsort(-(s[:, 3] - t0))[:10]
print("slowest logic waves:", slow.tolist(), (s[slow, 3] - t0).round(2).tolist(), "their entry", (s[slow, 0] - t0).round(2).tolist(), "state", (s[slow, 2] - t0).round(2).tolist())
