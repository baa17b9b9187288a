"""Small blocks of a multi-map batch (65 536 envs): fewer environments per wavefront so that a map's block still fills a four-wavefront
workgroup (one table copy; with split rows one shared row and four wavefronts storing it) -- against full wavefronts in smaller workgroups.
us per step (HIP events), each batch built twice (the arena's placement is part of a batch's speed)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
def run(label, maps, **env):
    for k, v in env.items():
        os.environ[k] = v
    _capi.refresh_tuning()
    res = []
    for rep in range(2):
        bw = BatchedWorld(maps, n, autotune_ms=0)
        fn = bw.sampled_stepper(auto_reset=True, seed=1)
        res.append(min(timeit(fn, iters=60, warm=10) for _ in range(3)))
        del bw, fn
        torch.cuda.empty_cache()
    print(f"{label:52s} " + " / ".join(f"{u:7.2f}" for u in res) + " us", flush=True)
    for k in env:
        os.environ.pop(k)
    _capi.refresh_tuning()

shapes = {"config 5 (32x32, 8 agents)": (lambda s: mapgen.config5(s), ("8", "4")),
          "12x13, 4 agents, 4 lasers": (lambda s: mapgen.generate(12, 13, 4, 4, 4, seed=s, n_voids=2), ("16", "8", "4")),
          "12x13, 2 agents, 2 lasers": (lambda s: mapgen.generate(12, 13, 2, 2, 4, seed=s, n_voids=2), ("32", "16", "8", "4"))}
for name, (gen, epws) in shapes.items():
    print(name, flush=True)
    run("  one map", gen(0))
    for n_maps in (4096, 8192):
        maps = [Map(gen(s)) for s in range(n_maps)]
        for e in epws:
            run(f"  {n_maps} maps x {n // n_maps}, LLE_STEP_EPW={e}", maps, LLE_STEP_EPW=e)
        del maps
