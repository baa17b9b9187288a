"""Pure fill of the partial 7x7 output's shape (level 6 x 65 536: 141 MB): blocks of E envs x 2 160 B per wavefront batch."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from lle_amd import placement
n, pitch = 65536, 2160
buf = torch.empty(n * pitch + 256, dtype=torch.uint8, device="cuda")
buf = buf[(-buf.data_ptr()) % 256:][: n * pitch]
for E, rpw in ((4, 4), (2, 8), (1, 16), (8, 2), (16, 1)):
    us = min(placement.time_row_fill(buf, E * pitch, rows_per_wave=rpw, launches=50) for _ in range(3))
    print(f"blocks of {E} envs ({E * pitch} B), {rpw} per wavefront: {us:6.2f} us = {n * pitch / us / 1e6:6.2f} TB/s", flush=True)
rows = torch.empty(65536 * 1920, dtype=torch.uint8, device="cuda")
us = min(placement.time_row_fill(rows, 1920, rows_per_wave=16, launches=50) for _ in range(3))
print(f"layered rows 1920 B x 16 per wavefront: {us:6.2f} us = {65536 * 1920 / us / 1e6:6.2f} TB/s")
