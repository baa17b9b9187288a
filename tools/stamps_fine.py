"""DIAGNOSTIC: fine timeline of the DEFAULT step kernel (MODE 0), per quarter of the grid.  Needs a one-off build in which
step_kernel stamps in every mode with a stride of 16 slots per wavefront and has the extra stamp points 8 (sampled),
9 (conflicts solved), 10 (passes done), 11 (loop top), and in which the launcher does not route stamped launches to
MODE 1 (DESIGN.md section 4 quotes the numbers of that build); the committed kernels stamp in MODE 1 only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lle_amd import BatchedWorld, Map, _capi
n = 65536
bw = BatchedWorld(Map(level=6), n)
nb = n // 16
stamps = torch.zeros(nb, 16, dtype=torch.int64, device="cuda")
for t in range(30):
    bw.step(sample=True, auto_reset=True, seed=1, t=t)
torch.cuda.synchronize()
_capi.lib().lle_batch_step_stamped(bw.h, 3, 1, 30, stamps.data_ptr(), bw._stream())
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype("float64") * 0.01
t0 = s[:, 0].min()
order = [(0, "entry"), (7, "rows requested+written"), (1, "barrier+template copy"), (2, "state in registers"), (11, "loop top"),
         (8, "sampled"), (9, "checked+conflicts"), (10, "passes done"), (3, "state machine done"), (4, "records in LDS"),
         (5, "obs issued"), (6, "drained")]
for q in range(4):
    sl = slice(q * nb // 4, (q + 1) * nb // 4)
    print(f"quarter {q}: " + "  ".join(f"{name} {np.percentile(s[sl, i] - t0, 50):.2f}" for i, name in order))
