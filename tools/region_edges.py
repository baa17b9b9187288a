"""The driver's 20-step region, its edges taken apart: wall us per step (median of 40 regions) by how the region is opened and closed."""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch
from lle_amd import BatchedWorld, Map

n, K = 65536, 20
bw = BatchedWorld(Map(level=6), n)
fn = bw.sampled_stepper(auto_reset=True, seed=1234)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for _ in range(256):
        fn()
    torch.cuda.synchronize()


def region(events, spin, precreate):
    ev0 = ev1 = None
    if events and precreate:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); ev1.record()
    torch.cuda.synchronize()
    if events and not precreate:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events:
        ev0.record()
    for _ in range(K):
        fn()
    t1 = time.perf_counter()
    if events:
        ev1.record()
        if spin:
            while not ev1.query():
                pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6, (ev0.elapsed_time(ev1) / K * 1e3 if events else 0.0)


for label, kw in (("events + spin (bench.py)", dict(events=True, spin=True, precreate=False)), ("events made before + spin", dict(events=True, spin=True, precreate=True)),
                  ("events, blocking synchronize", dict(events=True, spin=False, precreate=False)), ("no events, blocking synchronize", dict(events=False, spin=False, precreate=False))):
    for _ in range(5):
        region(**kw)
    r = [region(**kw) for _ in range(40)]
    print(f"{label:36s} wall {statistics.median(x[0] for x in r):6.2f} (min {min(x[0] for x in r):6.2f})  issue {statistics.median(x[1] for x in r):5.2f}  kernel {statistics.median(x[2] for x in r):6.2f}", flush=True)
