"""What a multi-map batch pays for (config 5's shape, 65 536 envs): the general instantiation (MODE 4) on ONE map (fused outputs), two maps (tables hot in
every L2), and many maps.  us per step (HIP events), each batch built twice (the arena's placement is part of a batch's speed)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
def run(label, maps, outputs=False, **env):
    for k, v in env.items():
        os.environ[k] = v
    _capi.refresh_tuning()
    res = []
    for rep in range(2):
        bw = BatchedWorld(maps, n, autotune_ms=0)
        if outputs:
            done = torch.empty(n, dtype=torch.uint8, device='cuda'); out = bw.make_env_outputs(done=done)
            fn = lambda bw=bw, out=out: bw.step(sample=True, auto_reset=True, seed=1, env_out=out)
        else:
            fn = bw.sampled_stepper(auto_reset=True, seed=1)
        res.append(min(timeit(fn, iters=60, warm=10) for _ in range(3)))
        del bw, fn
        torch.cuda.empty_cache()
    print(f"{label:44s} " + " / ".join(f"{u:7.2f}" for u in res) + " us", flush=True)
    for k in env:
        os.environ.pop(k)
    _capi.refresh_tuning()

one = mapgen.config5(0)
run("one map", one)
try:
    run("one map, fused outputs (MODE 4)", one, outputs=True)
except Exception as e:
    print("outputs:", repr(e))
for n_maps in (2, 16, 128, 1024):
    maps = [Map(mapgen.config5(s)) for s in range(n_maps)]
    run(f"{n_maps} maps x {n // n_maps}", maps)
    del maps
