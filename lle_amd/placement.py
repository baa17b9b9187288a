"""Placement of big output buffers: past the 256 MB Infinity Cache the write rate of a buffer depends on where the allocation landed.

profiles/r04_alloc_probe.md: the same row stream runs at 5.75 TB/s on some hipMalloc'ed buffers and at 6.6-6.9 on others of the same process,
reproducibly per buffer; physically contiguous memory (hipDeviceMallocContiguous) is reproducibly of the SLOW kind, so the rate is a
property of the physical pages behind the buffer, which an unprivileged process can neither see nor choose.  What it can do is sample:
allocate a few candidates side by side (all alive until every one is timed: distinct physical memory), time the step kernel's store
pattern on each (lle_probe_fill_rows), keep the fastest.  `BatchedWorld(placement_candidates=k)` does this for the batch's arena;
`pick_fastest` is the same for trajectory rings and observer outputs (BatchedWorld.make_ring / bound_observer take the same argument).
Worth it only for buffers larger than the cache; an option, never a default: it holds k buffers transiently."""
import ctypes as C

import torch

from . import _capi

INFINITY_CACHE_BYTES = 256 << 20


def time_row_fill(t, row_bytes, rows_per_wave=16, launches=10):
    """us per launch of the row-fill probe over the uint8 / int8 tensor `t` (contiguous, a whole number of rows)."""
    nbytes = t.numel() * t.element_size()  # (rings of a batch with a wider observation type: row_bytes is the pitch in BYTES)
    assert t.is_contiguous() and nbytes % row_bytes == 0 and t.data_ptr() % 16 == 0
    L, st = _capi.lib(), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    n_rows = nbytes // row_bytes
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(3 + launches):
        if k == 3:
            e0.record()
        rc = L.lle_probe_fill_rows(t.data_ptr(), n_rows, row_bytes, rows_per_wave, st)
        if rc != 0:
            raise RuntimeError(f"lle_probe_fill_rows failed ({rc}): {L.lle_last_error().decode()}")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


def pick_fastest(alloc, row_bytes, k, rows_per_wave=16):
    """k candidates from `alloc()` (each a contiguous 1-byte-element CUDA tensor of the same size), the row-fill probe on each, the
    fastest kept and zeroed, the others released.  Returns (tensor, record) with record = {"candidates", "row_fill_us", "chosen"}."""
    cands = []
    for _ in range(max(1, int(k))):
        try:
            cands.append(alloc())
        except torch.cuda.OutOfMemoryError:
            if not cands:
                raise
            break
    if len(cands) == 1:
        return cands[0], {"candidates": 1, "row_fill_us": [], "chosen": 0}
    times = [time_row_fill(t, row_bytes, rows_per_wave) for t in cands]
    best = min(range(len(cands)), key=times.__getitem__)
    keep = cands[best]
    del cands
    torch.cuda.empty_cache()
    keep.zero_()
    return keep, {"candidates": len(times), "row_fill_us": [round(v, 2) for v in times], "chosen": best}
