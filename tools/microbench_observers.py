"""Launch-to-launch time of the observation builders of observers.hip at 65 536 envs (bytes written / time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map, _capi, mapgen

def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

n = 65536
for label, m in (("level 6", Map(level=6)), ("config5 32x32", Map(mapgen.config5(0)))):
    bw = BatchedWorld(m, n)
    for t in range(8):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
    A = bw.map.n_agents
    for name, kind, param in (("layered (view kernel)", _capi.LLE_OBS_LAYERED, 0), ("layered-padded-2", _capi.LLE_OBS_LAYERED_PADDED, 2),
                              ("perspective", _capi.LLE_OBS_PERSPECTIVE, 0), ("partial3x3", _capi.LLE_OBS_PARTIAL, 3),
                              ("partial7x7", _capi.LLE_OBS_PARTIAL, 7), ("state", _capi.LLE_OBS_STATE, 0),
                              ("normalized-state", _capi.LLE_OBS_NORMALIZED_STATE, 0)):
        d = bw.obs_desc(kind, param)
        buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device="cuda")
        buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
        us = timeit(lambda: bw.observe_as(kind, param, out=buf))
        print(f"{label:14s} {name:22s} {d.bytes/1e6:8.1f} MB  {us:8.2f} us  {d.bytes/us/1e3:6.0f} GB/s", flush=True)
    out = torch.empty((n, A, 5), dtype=torch.uint8, device="cuda")
    for wl in (True, False):
        us = timeit(lambda: bw.available_actions(wl, out=out))
        print(f"{label:14s} available_actions(walkable_lasers={wl})  {us:8.2f} us", flush=True)
    us = timeit(lambda: bw.observe())
    print(f"{label:14s} observe() (layered, world_kernel)  {us:8.2f} us  {bw.map.obs_bytes*n/us/1e3:6.0f} GB/s", flush=True)
    del bw
