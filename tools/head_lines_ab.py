"""Row heads of 0..4 lines (lle_map_set_head_lines) on the current step kernel: us per step.  GPU box."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from lle_prof import timeit  # noqa: E402

from lle_amd import BatchedWorld, Map, mapgen  # noqa: E402

for label, mk, sizes in (("level 6", lambda: Map(level=6), (32768, 65536, 131072)), ("level 5", lambda: Map(level=5), (65536,)), ("level 3", lambda: Map(level=3), (65536,)),
                         ("generated 16x16 a4 l4", lambda: Map(mapgen.generate(16, 16, 4, 4, seed=3)), (65536,))):
    for n in sizes:
        cells = []
        for lines in (0, 1, 2, 3, 4, 5, 6, 8):
            m = mk()
            m.set_head_lines(lines)
            if lines and m.row_head[1] != lines * 128:
                continue  # (the map has no run of that many static lines)
            bw = BatchedWorld(m, n)
            step = bw.sampled_stepper(seed=1)
            cells.append(f"{lines}: {min(timeit(step, iters=200, warm=20) for _ in range(2)):6.2f}")
            del bw, step
            torch.cuda.empty_cache()
        print(f"{label} n={n}: " + "  ".join(cells), flush=True)
