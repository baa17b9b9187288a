"""Timing prototype: state-machine kernel (step without observation) on one HIP stream, observation kernel on a
second, chained by events, so that the state machine of step t+1 overlaps the observation stream of step t.
(Timing only: the observer reads the live state here; the product version hands it a double-buffered record.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lle_amd import BatchedWorld, Map

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
bw = BatchedWorld(Map(level=6), n)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(4)]
evo = [torch.cuda.Event() for _ in range(4)]

def run(steps, t0, lag):
    for t in range(t0, t0 + steps):
        with torch.cuda.stream(s1):
            if lag and t - t0 >= lag:
                s1.wait_event(evo[(t - lag) % 4])   # record buffer reuse: observer of step t-lag is done
            bw.step(sample=True, auto_reset=True, seed=1, t=t, write_obs=False)
            evs[t % 4].record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(evs[t % 4])
            bw.observe()
            evo[t % 4].record(s2)

def fused(steps, t0):
    for t in range(t0, t0 + steps):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)

for name, f in (("fused single kernel", lambda k, t0: fused(k, t0)), ("2 streams, lag 2", lambda k, t0: run(k, t0, 2)),
                ("2 streams, lag 1", lambda k, t0: run(k, t0, 1)), ("2 streams, no reuse wait", lambda k, t0: run(k, t0, 0))):
    f(20, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(400, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400 * 1e6
    print(f"{name}: {dt:.2f} us per step of {n} envs ({1937*n/dt/1e3:.0f} GB/s)", flush=True)
