"""In-kernel timeline (s_memrealtime stamps of lle_batch_step_stamped, the rollout build) of config 5's shape on one map and on blocks of maps:
how long a wavefront waits for its tables, runs its state machine, streams its rows.  us, medians over the wavefronts (and p90)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from lle_amd import BatchedWorld, Map, _capi, mapgen

n = 65536
def run(label, maps):
    bw = BatchedWorld(maps, n, autotune_ms=0)
    per = bw.kernel_info()["envs_per_wave"]
    nb = (n + per - 1) // per
    stamps = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
    for t in range(20):
        bw.step(sample=True, auto_reset=True, seed=1, t=t)
    torch.cuda.synchronize()
    assert _capi.lib().lle_batch_step_stamped(bw.h, 3, 1, 30, stamps.data_ptr(), bw._stream()) == 0
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype("float64") * 0.01
    ok = s[:, 0] > 0
    s = s[ok]
    t0 = s[:, 0].min()
    seg = {"entry->own table rows in LDS": s[:, 7] - s[:, 0], "...->barrier, state + slice loaded": s[:, 1] - s[:, 7], "entry->tables": s[:, 1] - s[:, 0], "tables->logic done": s[:, 3] - s[:, 1], "logic->records": s[:, 4] - s[:, 3],
           "records->obs issued": s[:, 5] - s[:, 4], "issued->drained": s[:, 6] - s[:, 5], "whole wave": s[:, 6] - s[:, 0]}
    print(f"{label}: epw {per}, {len(s)} stamped waves, launch {s[:, 6].max() - t0:.1f} us")
    for k, v in seg.items():
        print(f"    {k:36s} p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
    del bw
    torch.cuda.empty_cache()

if len(sys.argv) > 1 and sys.argv[1] == "small":  # 12 x 13 maps with 4 agents, down to a map per environment
    gen = lambda s: mapgen.generate(12, 13, 4, 4, 4, seed=s, n_voids=2)
    n = 16384
    run("one map", gen(0))
    for per in (8, 1):
        run(f"{n // per} maps x {per}", [Map(gen(s)) for s in range(n // per)])
else:
    run("one map", mapgen.config5(0))
    for n_maps in (1024, 4096):
        run(f"{n_maps} maps x {n // n_maps}", [Map(mapgen.config5(s)) for s in range(n_maps)])
