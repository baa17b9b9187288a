"""The other observation builders in the batch's element type (level 6 x 65 536): us per launch, int8 / fp16 / fp32, bound calls; and the cast pass a caller
would otherwise add (torch .to() of the int8 output)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi

n = 65536
rows = {}
for dt in (torch.int8, torch.float16, torch.float32):
    bw = BatchedWorld(Map(level=6), n, obs_dtype=dt)
    for t in range(12):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
    for label, kind, param in (("layered-padded-2", _capi.LLE_OBS_LAYERED_PADDED, 2), ("perspective", _capi.LLE_OBS_PERSPECTIVE, 0),
                               ("partial 3x3", _capi.LLE_OBS_PARTIAL, 3), ("partial 5x5", _capi.LLE_OBS_PARTIAL, 5), ("partial 7x7", _capi.LLE_OBS_PARTIAL, 7)):
        call = bw.bound_observer(kind, param)
        call(); torch.cuda.synchronize()
        us = min(timeit(call, iters=60, warm=10) for _ in range(3))
        mb = call.out.numel() * call.out.element_size() / 1e6
        rows.setdefault(label, []).append(f"{str(dt).split('.')[-1]} {us:6.1f} us ({mb / us / 1e-6 / 1e12 * 1e-6:4.2f} TB/s)" if False else f"{str(dt).split('.')[-1]} {us:6.1f} us ({mb * 1e6 / (us * 1e-6) / 1e12:4.2f} TB/s)")
        if dt == torch.int8:
            x = call.out
            for cdt in (torch.float16, torch.float32):
                fn = lambda: x.to(cdt)
                cus = min(timeit(fn, iters=30, warm=5) for _ in range(2))
                rows[label].append(f"[+ cast to {str(cdt).split('.')[-1]} {cus:6.1f} us]")
    del bw
    torch.cuda.empty_cache()
for k, v in rows.items():
    print(f"{k:18s} " + "  ".join(v))
