"""Does the real step kernel's HBM-regime rate depend on which allocation its arena landed in?  GPU box."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from lle_prof import timeit, stepper, algo_bytes
from lle_amd import BatchedWorld, Map, mapgen
for label, mk, n, k in (("level 6", lambda: Map(level=6, row_align=128), 262144, 10), ("cfg5", lambda: Map(mapgen.config5(0), row_align=128), 65536, 7)):
    keep = []
    for i in range(k):
        bw = BatchedWorld(mk(), n)
        keep.append(bw)
    for rep in range(2):
        for i, bw in enumerate(keep):
            step = stepper(bw)
            us = timeit(step, iters=40, warm=5)
            probe = bw.row_fill_prober()
            fill = timeit(probe, iters=40, warm=5)
            ms = timeit(lambda: bw.obs_rows.fill_(1), iters=40, warm=5)
            print(f"{label} n={n} arena {i} at {bw.arena.data_ptr():#x}: step {us:7.1f} us ({algo_bytes(bw.map) * n / us / 1e3:.0f} GB/s)  row fill {fill:7.1f}  torch fill {ms:7.1f}", flush=True)
            bw.observe()
    del keep
    torch.cuda.empty_cache()
