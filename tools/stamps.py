"""In-kernel timeline of the step kernel: per-wave s_memrealtime stamps (10 ns ticks), summarised over waves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from lle_amd import BatchedWorld, Map, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for epw in (0, 32):
    bw = BatchedWorld(Map(level=6), n, envs_per_wave=epw or None)
    nb = (n + (epw or 16) - 1) // (epw or 16)
    stamps = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
    for t in range(30):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
    torch.cuda.synchronize()
    rc = _capi.lib().lle_batch_step_stamped(bw.h, 3, 1, 30, stamps.data_ptr(), bw._stream())
    assert rc == 0
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype("float64") * 0.01  # us
    t0 = s[:, 0].min()
    names = ["entry", "tables in LDS", "state requested", "logic done", "state stored", "obs stores issued", "drained", "own table rows copied"]
    print(f"n={n} epw={epw} waves={nb}: (us since the first wave's entry; min / median / max over waves)")
    for i, nm in enumerate(names):
        col = s[:, i] - t0
        print(f"   {nm:20s} {col.min():7.2f} {float(sorted(col)[len(col)//2]):7.2f} {col.max():7.2f}")
