"""BatchedWorld: n lock-stepped worlds of one map resident on one MI355X, exposed as torch tensors.

The device work (reset / step / set_state / layered observation) is done by the HIP kernels behind the C ABI
of include/lle_hip.h; torch only provides the device arena, the stream and (for multi-GPU) torch.distributed.
Every tensor below is a zero-copy view into the batch's arena and is overwritten by the next call.
"""
import ctypes as C
import os

import torch

from . import _capi
from ._capi import (LLE_BUF_COUNT, BUFFER_NAMES, LLE_STEP_AUTO_RESET, LLE_STEP_INCREMENTAL_OBS, LLE_STEP_NO_OBS, LLE_STEP_RECOLOUR_RESETS,
                    LLE_STEP_SAMPLE_ACTIONS, BufferDesc, Map)

_TORCH_DTYPES = {
    "pos": torch.uint8, "bits": torch.int64, "gems": torch.int32, "beams": torch.int32, "avail": torch.uint8,
    "actions": torch.uint8, "err": torch.uint8, "evcount": torch.uint8, "events": torch.uint8, "done": torch.uint8,
    "obs": torch.int8, "stats": torch.int64, "req_pos": torch.uint8, "req_gems": torch.int32, "req_alive": torch.int16, "reward": torch.uint8,
    "src_colour": torch.uint8, "src_enabled": torch.int32,
}


# The raw handle of torch's CURRENT stream on a device: the bound calls below read it on every call (an integer from the C
# extension, ~0.1 us) instead of freezing the stream that was current when they were bound -- a caller that enters another
# `torch.cuda.stream(...)` later gets its launches there, ordered with its own tensors.
# obs_dtype of BatchedWorld -> (LLE_DTYPE_*, torch dtype); strings for hosts that do not hold a torch dtype
_OBS_DTYPES = {None: (_capi.LLE_DTYPE_I8, torch.int8), torch.int8: (_capi.LLE_DTYPE_I8, torch.int8), torch.float16: (_capi.LLE_DTYPE_F16, torch.float16),
               torch.bfloat16: (_capi.LLE_DTYPE_BF16, torch.bfloat16), torch.float32: (_capi.LLE_DTYPE_F32, torch.float32)}
_OBS_DTYPES.update({"int8": _OBS_DTYPES[torch.int8], "float16": _OBS_DTYPES[torch.float16], "bfloat16": _OBS_DTYPES[torch.bfloat16],
                    "float32": _OBS_DTYPES[torch.float32]})

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_stream_handle(device):
    if _raw_stream is not None:
        return _raw_stream(device.index or 0)
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("lle_amd: no HIP device is visible. The batched World runs only on the GPU "
                           "(there is no CPU fallback); build and run on an MI355X box.")


class BatchedWorld:
    """n_envs independent `World`s of one map stepped by one kernel launch.

    Attributes (torch tensors on `device`, views into one arena):
      pos [n,A,2] u8 (i,j) - bits [n] i64 (alive 0-15 | arrived 16-31 | occupant 32-47 | 48-63 set_state bookkeeping) - gems [n] i32 (bit g collected)
      beams [n,L] i32 (bit k = on at offset k) - avail [n,A] u8 (bit a = Action a) - actions [n,A] u8
      err [n] u8 - evcount [n] u8 - events [n,2A] u8 (type<<4|agent) - done [n] u8 - obs [n,C,H,W] i8
    """

    # Batches from this size on run lle_batch_autotune at construction (AUTOTUNE_MS of GPU time): below it a launch is a few
    # microseconds of latency whatever the rules say.  LLE_AUTOTUNE_MS in the environment overrides the budget (0: never).
    AUTOTUNE_MIN_ENVS = 2048
    AUTOTUNE_MS = 10.0

    def __init__(self, map_or_text, n_envs, device=None, envs_per_wave=None, row_align=None, placement_candidates=None, obs_dtype=None,
                 autotune_ms=None, incremental_obs=False):
        """`map_or_text`: a Map / map text, or a LIST of them for a batch of several maps -- map m then owns the envs
        [m * n_envs / len(maps), (m + 1) * n_envs / len(maps)); the maps must agree on height, width and the numbers of
        agents, sources and gems; any number of envs per map, down to one map per env (round 5; a wavefront serves one map, so blocks of 64 and
        more keep the kernels' full shape, smaller ones cost throughput: INTEGRATION.md section 6).
        `row_align`: pitch of the observation rows in bytes (Map.set_row_align: applied to the maps given).
        `placement_candidates`: k > 1 allocates k arenas, times the step kernel's store pattern on each
        (lle_batch_probe_row_fill) and keeps the fastest; the others are released (torch.cuda.empty_cache()).  Worth it only
        when the rows of a step exceed the 256 MB Infinity Cache: there the write rate depends on where the allocation landed
        (profiles/r03_hbm_fronts.md: 5.8 ... 6.8 TB/s over twelve buffers of one process).  `self.placement` records the timings.
        `obs_dtype`: element type of `obs` and of the observation rings -- torch.int8 (default), torch.float16, torch.bfloat16 or
        torch.float32 (the reference's, python/lle/observations.py:223).  The kernels widen at the store (lle_batch_options.obs_dtype):
        the tensor a learner reads comes out of the step launch in its own type, with the values of the int8 tensor.
        `autotune_ms`: GPU time lle_batch_autotune may spend, right here, timing the step launcher's alternatives on this batch's own
        arena (environments per wavefront, row heads, store policy, split rows, walk, rotation) -- None: AUTOTUNE_MS for batches of
        AUTOTUNE_MIN_ENVS environments and more (the default since round 5: the product a caller gets is the product bench.py
        measures), 0: the library's default rules.  Results never depend on it; `tuning()` says what was chosen.
        `incremental_obs`: every in-place single step of this batch writes only the lines of a row that dynamic state can change
        (LLE_STEP_INCREMENTAL_OBS; INTEGRATION.md section 5d says why this is an opt-in: the caller promises not to write into `obs`).
        `check_obs()` verifies the buffer against the state at any time; LLE_DEBUG_CHECK_OBS=1 does so after every step."""
        _require_gpu()
        many = isinstance(map_or_text, (list, tuple))
        items = list(map_or_text) if many else [map_or_text]
        if sum(not isinstance(m, Map) for m in items) >= 512:
            # thousands of map texts (a map per environment, python/lle/generator/world_builder.py:84-89): lle_map_parse is host-only C++ and ctypes
            # releases the GIL around it, so the maps compile side by side (65 536 maps of 12 x 13: 5.2 s -> about 1 s on 8 cores)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
                self.maps = list(ex.map(lambda m: m if isinstance(m, Map) else Map(m), items, chunksize=256))
        else:
            self.maps = [m if isinstance(m, Map) else Map(m) for m in items]
        if row_align is not None:  # on copies: the caller's Map objects keep their pitch (a live batch elsewhere may hold them)
            self.maps = [m.clone() for m in self.maps]
            for m in self.maps:
                m.set_row_align(row_align)
        self.map = self.maps[0]  # common dimensions
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.n_envs = int(n_envs)
        self.obs_dtype = _OBS_DTYPES[obs_dtype][1]
        opt = _capi.BatchOptions(_OBS_DTYPES[obs_dtype][0])
        L = _capi.lib()
        if many and self.n_envs % len(self.maps) != 0:
            raise ValueError("n_envs must be a multiple of the number of maps")
        self.envs_per_map = self.n_envs // len(self.maps)
        handles = (C.c_void_p * len(self.maps))(*[m.h for m in self.maps])
        nbytes = L.lle_batch_arena_bytes_opt(handles, len(self.maps), self.envs_per_map, C.byref(opt))
        if nbytes <= 0:
            raise RuntimeError(L.lle_last_error().decode())
        def create():
            arena = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = arena[(-arena.data_ptr()) % 256:][:nbytes]
            h = L.lle_batch_create_opt(handles, len(self.maps), self.envs_per_map, self.device.index or 0, base.data_ptr(), nbytes, C.byref(opt),
                                       self._stream())
            if not h:
                raise RuntimeError(f"lle_batch_create failed: {L.lle_last_error().decode()}")
            return arena, base, h

        self.placement = None
        with torch.cuda.device(self.device):
            k = max(1, int(placement_candidates or 1))
            if k == 1:
                self.arena, self._base, self.h = create()
            else:
                self.arena, self._base, self.h = self._place(create, k)
        if envs_per_wave is not None:
            self.set_envs_per_wave(envs_per_wave)
        self._bind()
        self.t = 0
        self.incremental_obs = bool(incremental_obs)
        self._debug_check_obs = os.environ.get("LLE_DEBUG_CHECK_OBS", "0") == "1"
        if autotune_ms is None:
            env_ms = os.environ.get("LLE_AUTOTUNE_MS")
            autotune_ms = float(env_ms) if env_ms not in (None, "") else (self.AUTOTUNE_MS if self.n_envs >= self.AUTOTUNE_MIN_ENVS else 0.0)
        if autotune_ms and autotune_ms > 0 and envs_per_wave is None:  # (envs_per_wave selects the diagnostic lane-per-env kernel: nothing to tune)
            self.autotune(autotune_ms)

    def _place(self, create, k):
        """k candidate arenas side by side (all alive until every one is timed: distinct physical memory), the row-fill
        probe on each, the fastest kept."""
        L = _capi.lib()
        cands, times = [], []
        try:
            for _ in range(k):
                try:
                    cands.append(create())
                except torch.cuda.OutOfMemoryError:  # fewer candidates than asked for: place among those that fit
                    if not cands:
                        raise
                    break
            k = len(cands)
            st = self._stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _, _, h in cands:
                for _ in range(3):
                    self._check(L.lle_batch_probe_row_fill(h, 0, st))
                e0.record()
                for _ in range(10):
                    self._check(L.lle_batch_probe_row_fill(h, 0, st))
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) * 100.0)  # us per launch
            best = min(range(k), key=times.__getitem__)
        except Exception:
            for _, _, h in cands:
                L.lle_batch_free(h)
            raise
        for i, (_, _, h) in enumerate(cands):
            if i != best:
                L.lle_batch_free(h)
        arena, base, h = cands[best]
        del cands
        torch.cuda.empty_cache()
        self._check(L.lle_batch_reset(h, None, self._stream()))  # the probe overwrote the rows: back to the start state's observation
        self.placement = {"candidates": k, "row_fill_us": [round(t, 2) for t in times], "chosen": best}
        return arena, base, h

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"lle_amd C ABI call failed ({rc}): {_capi.lib().lle_last_error().decode()}")

    def _bind(self):
        L = _capi.lib()
        m = self.map
        self._desc = {}
        for which in range(LLE_BUF_COUNT):
            d = BufferDesc()
            self._check(L.lle_batch_get_buffer(self.h, which, C.byref(d)))
            name = BUFFER_NAMES[which]
            dt = _TORCH_DTYPES[name]
            self._desc[name] = (int(d.arena_offset), int(d.bytes), [int(d.shape[k]) for k in range(d.ndim)], int(d.elem_bytes),
                                [int(d.stride[k]) for k in range(d.ndim)])
            raw = self._base[d.arena_offset:d.arena_offset + d.bytes]
            if name == "obs":
                assert int(d.elem_bytes) == self.obs_dtype.itemsize
                t = raw.view(self.obs_dtype)[: self.n_envs * m.obs_stride].view(self.n_envs, m.obs_stride)
                self.obs_rows = t
                t = t[:, : m.obs_bytes].unflatten(1, (m.n_layers, m.height, m.width))
            else:
                # per-agent buffers are strided by the kernel's agent bound (lle_buffer_desc.stride, in elements)
                shape = [int(d.shape[k]) for k in range(d.ndim)]
                stride = [int(d.stride[k]) for k in range(d.ndim)]
                t = torch.as_strided(raw.view(dt), shape, stride)
            setattr(self, "stats_blocks" if name == "stats" else name, t)

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                _capi.lib().lle_batch_free(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown: module globals may already be gone)
                pass
            self.h = None

    def autotune(self, budget_ms=20.0):
        """Time the step launcher's alternatives on this batch's own arena and keep the fastest (lle_batch_autotune): environments
        per wavefront, row heads, store policy, split rows, alternating walk.  The trials are real steps: the batch ends reset, with
        its counters at zero -- call it right after construction.  Returns tuning()."""
        self._check(_capi.lib().lle_batch_autotune(self.h, float(budget_ms), self._stream()))
        self.t = 0
        return self.tuning()

    def tuning(self):
        """The rules a plain single step of this batch is launched with (lle_batch_tuning) and the autotune log."""
        info, log = _capi.TuningInfo(), C.create_string_buffer(2048)
        self._check(_capi.lib().lle_batch_tuning(self.h, C.byref(info), log, 2048))
        out = {n: int(getattr(info, n)) for n, _ in _capi.TuningInfo._fields_}
        out["log"] = log.value.decode()
        return out

    def set_envs_per_wave(self, epw):
        self._check(_capi.lib().lle_batch_set_envs_per_wave(self.h, int(epw)))

    def kernel_info(self):
        name = C.create_string_buffer(64)
        lds, epw = C.c_int32(0), C.c_int32(0)
        _capi.lib().lle_batch_kernel_info(self.h, name, 64, C.byref(lds), C.byref(epw))
        return {"kernel": name.value.decode(), "lds_bytes": lds.value, "envs_per_wave": epw.value}

    # ------------------------------------------------------------------ World API, batched
    @property
    def n_agents(self):
        return self.map.n_agents

    def reset(self, env_mask=None):
        """World.reset for every env (or those with env_mask != 0)."""
        mp = None
        if env_mask is not None:
            env_mask = env_mask.to(self.device, torch.uint8).contiguous()
            mp = env_mask.data_ptr()
        self._check(_capi.lib().lle_batch_reset(self.h, mp, self._stream()))
        self.t = 0

    def step(self, actions=None, sample=False, auto_reset=False, seed=0, t=None, env_offset=0, write_obs=True, env_out=None,
             recolour_resets=False, incremental_obs=False):
        """World.step + Layered.observe for every env.  env_out: an `_capi.EnvOutputs` (make_env_outputs) whose tensors the
        step kernel fills in the same launch (lle_batch_step_outputs: LLE.step's state / reward / done / available / ...).

        actions: uint8 tensor [n, A] on the device (Action values), or None with sample=True to draw uniformly from the
        available actions with the counter-based sampler (seed, env_offset + env, t, agent)."""
        flags = 0
        ap = None
        if sample:
            flags |= LLE_STEP_SAMPLE_ACTIONS
        else:
            if actions is None:
                raise ValueError("actions is required unless sample=True")
            if actions.dtype != torch.uint8 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(self.device, torch.uint8).contiguous()
            if tuple(actions.shape) != (self.n_envs, self.map.n_agents):
                raise ValueError(f"Invalid number of actions: given {tuple(actions.shape)}, expected {(self.n_envs, self.map.n_agents)}")
            ap = actions.data_ptr()
        if auto_reset:
            flags |= LLE_STEP_AUTO_RESET
        if recolour_resets:  # (LLE.reset with randomize_lasers inside the step: per-env sources, see LLE_STEP_RECOLOUR_RESETS)
            flags |= LLE_STEP_RECOLOUR_RESETS
        if not write_obs:
            flags |= LLE_STEP_NO_OBS
        if incremental_obs or self.incremental_obs:  # only the lines of a row that dynamic state can change (LLE_STEP_INCREMENTAL_OBS): same content of `obs`
            flags |= LLE_STEP_INCREMENTAL_OBS
        if t is None:
            t = self.t
        if env_out is not None:
            self._check(_capi.lib().lle_batch_step_outputs(self.h, ap, flags, int(seed), int(t), int(env_offset), C.byref(env_out), self._stream()))
        else:
            self._check(_capi.lib().lle_batch_step(self.h, ap, flags, int(seed), int(t), int(env_offset), self._stream()))
        self.t = t + 1
        if self._debug_check_obs and write_obs and not (env_out is not None and env_out.partial):
            bad = self.check_obs()
            if bad:
                raise AssertionError(f"LLE_DEBUG_CHECK_OBS: the rows of {bad} environments differ from their state after step t={t} "
                                     "(an incremental step over a buffer somebody else wrote into?)")

    def sampled_stepper(self, auto_reset=True, seed=0, env_offset=0, write_obs=True, incremental_obs=False):
        """A zero-argument callable for hot loops: one `step(sample=True, ...)` per call with the arguments and the
        flags bound once, the time index advancing by one per call -- the C-ABI call
        and nothing else per step (`step()` itself spends a few microseconds in Python per launch)."""
        fn, h, dev = _capi.lib().lle_batch_step, self.h, self.device
        flags = (LLE_STEP_SAMPLE_ACTIONS | (LLE_STEP_AUTO_RESET if auto_reset else 0) | (0 if write_obs else LLE_STEP_NO_OBS) |
                 (LLE_STEP_INCREMENTAL_OBS if (incremental_obs or self.incremental_obs) else 0))
        seed, env_offset = int(seed), int(env_offset)

        def one_step():
            t = self.t
            rc = fn(h, None, flags, seed, t, env_offset, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
            self.t = t + 1
        return one_step

    def make_ring(self, slots, placement_candidates=None):
        """Trajectory rings for `rollout`: obs [R,n,C,H,W] int8, actions [R,n,A] uint8, reward [R,n,4] uint8.
        placement_candidates: k > 1 samples k allocations of the observation ring and keeps the one the step kernel's store pattern
        writes fastest (lle_amd.placement; `ring["placement"]` records the timings) -- for rings larger than the 256 MB Infinity Cache."""
        from . import placement
        m = self.map
        pitch = self._desc["actions"][4][0]
        shape = (int(slots), self.n_envs, m.obs_stride)
        k = int(placement_candidates or 1)
        if k > 1 and shape[0] * shape[1] * shape[2] * self.obs_dtype.itemsize > placement.INFINITY_CACHE_BYTES:
            rows, placed = placement.pick_fastest(lambda: torch.zeros(shape, dtype=self.obs_dtype, device=self.device), m.obs_stride * self.obs_dtype.itemsize, k,
                                                  rows_per_wave=self.kernel_info()["envs_per_wave"])
        else:
            rows, placed = torch.zeros(shape, dtype=self.obs_dtype, device=self.device), None
        ring = {
            "slots": int(slots),
            "placement": placed,
            "obs_rows": rows,
            "actions_rows": torch.zeros(slots, self.n_envs, pitch, dtype=torch.uint8, device=self.device),
            "reward": torch.zeros(slots, self.n_envs, 4, dtype=torch.uint8, device=self.device),
        }
        ring["obs"] = ring["obs_rows"][:, :, : m.obs_bytes].unflatten(2, (m.n_layers, m.height, m.width))
        ring["actions"] = ring["actions_rows"][:, :, : m.n_agents]
        return ring

    def rollout(self, n_steps, auto_reset=True, seed=0, t=None, env_offset=0, ring=None, ring_pos=0, write_obs=True, sample=True):
        """n_steps fused steps in one launch (lle_batch_rollout); identical results to n_steps calls of step().  With a ring
        (make_ring) step j lands in slot (ring_pos + j) % slots.  sample=True: actions drawn on the device and written to
        the action ring; sample=False: step j READS its joint actions from ring["actions"][(ring_pos + j) % slots] (fill
        them first; without a ring every step takes `self.actions`)."""
        flags = (LLE_STEP_SAMPLE_ACTIONS if sample else 0) | (LLE_STEP_AUTO_RESET if auto_reset else 0) | (0 if write_obs else LLE_STEP_NO_OBS)
        if t is None:
            t = self.t
        rp = None
        if ring is not None:
            r = _capi.RolloutRing(ring["slots"], 0, int(ring_pos), ring["obs_rows"].data_ptr(), ring["actions_rows"].data_ptr(),
                                  ring["reward"].data_ptr())
            rp = C.byref(r)
        self._check(_capi.lib().lle_batch_rollout(self.h, int(n_steps), flags, int(seed), int(t), int(env_offset), rp, self._stream()))
        self.t = t + n_steps

    def set_state(self, positions, gems_collected, agents_alive):
        """World.set_state for every env.  positions u8 [n,A,2]; gems_collected bool [n,G]; agents_alive bool [n,A].
        Per-env result in `err` (0, or LLE_ENV_*), events in `events`/`evcount`."""
        G, A = self.map.n_gems, self.map.n_agents
        self.req_pos.copy_(positions.to(self.device, torch.uint8).view(self.n_envs, A, 2))
        gw = (1 << torch.arange(G, device=self.device, dtype=torch.int64))
        self.req_gems.copy_((gems_collected.to(self.device, torch.int64).view(self.n_envs, G) * gw).sum(1).to(torch.int32))
        aw = (1 << torch.arange(A, device=self.device, dtype=torch.int64))
        self.req_alive.copy_((agents_alive.to(self.device, torch.int64).view(self.n_envs, A) * aw).sum(1).to(torch.int16))
        self._check(_capi.lib().lle_batch_set_state(self.h, self._stream()))

    def update_sources(self):
        """Push self.map's current source colours / enabled flags to the device (LaserSource.enable/disable/set_colour)."""
        self._check(_capi.lib().lle_batch_update_sources(self.h, self.map.h, self._stream()))

    def update_map(self, map_index=0):
        """Push self.maps[map_index] to the device after Map.set_exits (World.exit_pos = ...) or Map.set_source: tables,
        reset states and the observation follow, the dynamic state of live envs stays (lle_batch_update_map)."""
        self._check(_capi.lib().lle_batch_update_map(self.h, int(map_index), self.maps[map_index].h, self._stream()))

    def set_exits(self, exits, map_index=0):
        """World.exit_pos = exits for every env of map `map_index` (src/core/world.rs:195-234)."""
        self.maps[map_index].set_exits(exits)
        self.update_map(map_index)

    def set_sources(self, colours=None, enabled=None, env_mask=None, reset_first=False, write_obs=True):
        """Per-environment laser sources (LLE.reset with randomize_lasers, python/lle/env/env.py:198-200; LaserSource
        enable / disable): colours u8 [n, L] and/or enabled masks i32 [n] (bit l = source l on), optionally only for the
        envs with env_mask != 0.  An env given a colour >= n_agents is left unchanged with err = LLE_ENV_INVALID_COLOUR, one
        given a colour that puts another agent's start on the beam with err = LLE_ENV_COLOUR_CROSSES_START.
        From the first call on `src_colour` / `src_enabled` hold each env's sources.
        reset_first: World.reset of the selected envs in the same launch, before the update (lle_batch_reset_sources:
        what `reset(env_mask)` followed by this call leaves; env_mask may then be `self.done` itself; write_obs=False when
        a step follows before anybody reads the observation)."""
        cp = ep = mp = None
        if colours is not None:
            colours = colours.to(self.device, torch.uint8).contiguous()
            assert colours.shape == (self.n_envs, self.map.n_sources)
            cp = colours.data_ptr()
        if enabled is not None:
            enabled = enabled.to(self.device, torch.int32).contiguous()
            assert enabled.shape == (self.n_envs,)
            ep = enabled.data_ptr()
        if env_mask is not None:
            env_mask = env_mask.to(self.device, torch.uint8).contiguous()
            mp = env_mask.data_ptr()
        if reset_first:
            self._check(_capi.lib().lle_batch_reset_sources(self.h, cp, ep, mp, 0 if write_obs else LLE_STEP_NO_OBS, self._stream()))
        else:
            assert write_obs, "lle_batch_set_sources always rewrites the observation"
            self._check(_capi.lib().lle_batch_set_sources(self.h, cp, ep, mp, self._stream()))

    def observe(self):
        self._check(_capi.lib().lle_batch_observe(self.h, self._stream()))
        return self.obs

    def check_obs(self):
        """Debug aid: the number of environments whose row of `obs` (the padding must be zero) differs from the layered observation of their
        CURRENT state, rebuilt in full by another kernel into a scratch buffer (lle_batch_observe_as(LLE_OBS_LAYERED)).  0 after any step /
        reset / observe / source update -- unless somebody wrote into `obs` between two incremental steps (LLE_STEP_INCREMENTAL_OBS
        leaves the static lines alone) or an observation-less step (write_obs=False) came last.  Costs a pass over the rows: not for hot loops."""
        d = self.obs_desc(_capi.LLE_OBS_LAYERED, 0)
        if not d.supported:
            return 0
        want = self.observe_as(_capi.LLE_OBS_LAYERED, 0)  # int8 [n, C, H, W] over rows of the same pitch
        nb = self.map.obs_bytes
        rows = torch.as_strided(want, (self.n_envs, nb), (int(d.stride[0]), 1))
        pad = self.obs_rows[:, nb:]
        return int(((self.obs_rows[:, :nb] != rows.to(self.obs_dtype)).any(dim=1) | (pad != 0).any(dim=1)).sum())

    # ---- the other observation builders (python/lle/observations.py)
    def obs_desc(self, kind, param=0):
        d = _capi.ObsDesc()
        self._check(_capi.lib().lle_batch_obs_desc(self.h, int(kind), int(param), C.byref(d)))
        return d

    def _desc_dtype(self, kind, d):
        """torch dtype of an observation's elements: float32 for the state kinds; the batch's element type (obs_dtype; int8 by default) for the
        layered-style ones -- every kernel that writes those widens at the store."""
        if int(kind) in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE):
            return torch.float32
        assert int(d.elem_bytes) == self.obs_dtype.itemsize
        return self.obs_dtype

    def observe_as(self, kind, param=0, out=None):
        """Observation `kind` (lle_amd._capi.LLE_OBS_*) of every env, written by the kernels of observers.hip into `out`
        (a uint8 device tensor of obs_desc(kind, param).bytes bytes; allocated when None).  Returns a strided view of
        that buffer with the reference's per-env shape behind the env axis: the batch's element type (int8 unless obs_dtype was given)
        for the layered kinds, float32 for the state kinds.  Raises IndexError where the reference does (a laser colour without a layer)."""
        d = self.obs_desc(kind, param)
        if not d.supported:
            raise IndexError("a laser colour has no layer in this observation (the reference raises IndexError too)")
        if out is None:
            out = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device=self.device)
            out = out[(-out.data_ptr()) % 256:][: int(d.bytes)]
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= d.bytes and out.data_ptr() % 16 == 0
        self._check(_capi.lib().lle_batch_observe_as(self.h, int(kind), int(param), out.data_ptr(), out.numel(), self._stream()))
        dt = self._desc_dtype(kind, d)
        flat = out[: int(d.bytes)].view(dt)
        return torch.as_strided(flat, [int(d.shape[k]) for k in range(d.ndim)], [int(d.stride[k]) for k in range(d.ndim)])

    # ---- bound calls: arguments and output buffer fixed ONCE; per call the C-ABI call on torch's current stream and nothing else.  observe_as() / available_actions() / env_outputs() spend 10-15 us per call in Python (descriptor query,
    # allocation, view construction, argument conversion) around kernels of 4-6 us: a host that steps in a loop uses these.
    def bound_observer(self, kind, param=0, out=None, placement_candidates=None):
        """A zero-argument callable that writes observation `kind` into ONE persistent buffer (`call.out`: the strided view
        observe_as() returns; overwritten by every call).  placement_candidates: k > 1 samples k allocations of an output larger than
        the Infinity Cache and keeps the one written fastest (lle_amd.placement; `call.placement`)."""
        from . import placement
        d = self.obs_desc(kind, param)
        if not d.supported:
            raise IndexError("a laser colour has no layer in this observation (the reference raises IndexError too)")
        placed = None
        if out is None:
            nbytes, k = int(d.bytes), int(placement_candidates or 1)
            pitch = int(d.stride[0]) * int(d.elem_bytes)  # bytes of one env's record
            if k > 1 and nbytes > placement.INFINITY_CACHE_BYTES and pitch % 16 == 0 and nbytes % pitch == 0:
                out, placed = placement.pick_fastest(lambda: torch.empty(nbytes, dtype=torch.uint8, device=self.device), min(pitch, 1 << 24), k)
            else:
                out = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
                out = out[(-out.data_ptr()) % 256:][:nbytes]
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= d.bytes and out.data_ptr() % 16 == 0
        dt = self._desc_dtype(kind, d)
        view = torch.as_strided(out[: int(d.bytes)].view(dt), [int(d.shape[k]) for k in range(d.ndim)], [int(d.stride[k]) for k in range(d.ndim)])
        fn = _capi.lib().lle_batch_observe_as
        args, dev = (C.c_void_p(self.h), C.c_int(int(kind)), C.c_int(int(param)), C.c_void_p(out.data_ptr()), C.c_int64(out.numel())), self.device

        def call():
            rc = fn(*args, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
            return view
        call.out, call.buffer, call.placement = view, out, placed
        return call

    def bound_available_actions(self, walkable_lasers=True, out=None):
        """Zero-argument callable for available_actions(): bool [n, A, 5] in `call.out`, overwritten by every call."""
        if out is None:
            out = torch.empty((self.n_envs, self.map.n_agents, 5), dtype=torch.uint8, device=self.device)
        fn = _capi.lib().lle_batch_available_actions
        args, dev = (C.c_void_p(self.h), C.c_int(int(bool(walkable_lasers))), C.c_void_p(out.data_ptr())), self.device
        view = out.view(torch.bool)

        def call():
            rc = fn(*args, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
            return view
        call.out, call.buffer = view, out
        return call

    def bound_env_outputs(self, **tensors):
        """Zero-argument callable for env_outputs() over fixed tensors (same keyword arguments)."""
        o = self.make_env_outputs(**tensors)
        fn = _capi.lib().lle_batch_env_outputs
        args, dev = (C.c_void_p(self.h), C.byref(o)), self.device

        def call():
            rc = fn(*args, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
        call.struct, call.tensors = o, tensors  # (kept alive with the callable)
        return call

    def bound_step(self, auto_reset=False, recolour_resets=False, write_obs=True, seed=0, env_offset=0, env_out=None, incremental_obs=False):
        """`fn(actions)` = step(actions, ...) with the flags, the stream and the output struct fixed; `actions` must already be a
        contiguous uint8 [n, A] tensor on this device (no conversion, no checks).  The time index advances by one per call."""
        flags = ((LLE_STEP_AUTO_RESET if auto_reset else 0) | (LLE_STEP_RECOLOUR_RESETS if recolour_resets else 0) |
                 (0 if write_obs else LLE_STEP_NO_OBS) | (LLE_STEP_INCREMENTAL_OBS if (incremental_obs or self.incremental_obs) else 0))
        L, h, dev, seed, env_offset = _capi.lib(), C.c_void_p(self.h), self.device, int(seed), int(env_offset)
        fn = L.lle_batch_step_outputs if env_out is not None else L.lle_batch_step
        tail = (C.byref(env_out),) if env_out is not None else ()

        def call(actions):
            t = self.t
            rc = fn(h, actions.data_ptr(), flags, seed, t, env_offset, *tail, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
            self.t = t + 1
        return call

    def available_actions(self, walkable_lasers=True, out=None):
        """LLE.available_actions (python/lle/env/env.py:146-163) for every env: bool [n, A, 5] in Action value order."""
        if out is None:
            out = torch.empty((self.n_envs, self.map.n_agents, 5), dtype=torch.uint8, device=self.device)
        self._check(_capi.lib().lle_batch_available_actions(self.h, int(bool(walkable_lasers)), out.data_ptr(), self._stream()))
        return out.view(torch.bool)

    def make_env_outputs(self, state=None, normalize_state=False, reward=None, multi_objective=False, done=None, available=None,
                         walkable_lasers=True, alive=None, arrived=None, partial=None, partial_k=0):
        """The lle_env_outputs struct over the given device tensors (None = not wanted); see env_outputs for the shapes.
        partial / partial_k (step(env_out=...) only): a uint8 buffer of obs_desc(LLE_OBS_PARTIAL, k).bytes bytes that the STEP launch fills
        with the partial k x k observation instead of writing the layered one (lle_batch_step_outputs; `partial_view(buf, k)` shapes it)."""
        o = _capi.EnvOutputs()
        if partial is not None:
            assert partial.dtype == torch.uint8 and partial.is_contiguous() and partial.device == self.device and partial.data_ptr() % 16 == 0
            assert partial.numel() >= int(self.obs_desc(_capi.LLE_OBS_PARTIAL, int(partial_k)).bytes)
            o.partial, o.partial_k = partial.data_ptr(), int(partial_k)
        for name, t in (("state", state), ("reward", reward), ("done", done), ("available", available), ("alive", alive), ("arrived", arrived)):
            if t is not None:
                assert t.is_contiguous() and t.device == self.device, name
                setattr(o, name, t.data_ptr())
        o.normalize_state, o.reward_kind, o.walkable_lasers = int(bool(normalize_state)), int(bool(multi_objective)), int(bool(walkable_lasers))
        return o

    def partial_buffer(self, k):
        """(buffer, view): a uint8 buffer for make_env_outputs(partial=...) and its view [n, A, 2A + 3, k, k] in the batch's element type."""
        d = self.obs_desc(_capi.LLE_OBS_PARTIAL, int(k))
        if not d.supported:
            raise IndexError("a laser colour has no layer in this observation (the reference raises IndexError too)")
        buf = torch.empty(int(d.bytes) + 256, dtype=torch.uint8, device=self.device)
        buf = buf[(-buf.data_ptr()) % 256:][: int(d.bytes)]
        view = torch.as_strided(buf.view(self.obs_dtype), [int(d.shape[q]) for q in range(d.ndim)], [int(d.stride[q]) for q in range(d.ndim)])
        return buf, view

    def env_outputs(self, state=None, normalize_state=False, reward=None, multi_objective=False, done=None, available=None,
                    walkable_lasers=True, alive=None, arrived=None):
        """Everything `LLE.step` returns besides the observation, in one launch (lle_batch_env_outputs): pass the
        tensors to fill -- state f32 [n, 3A+G], reward f32 [n, 1] ([n, 4] with multi_objective), done / available
        [n, A, 5] / alive / arrived [n, A] as uint8 or bool -- and leave the others None."""
        o = _capi.EnvOutputs()
        for name, t in (("state", state), ("reward", reward), ("done", done), ("available", available), ("alive", alive), ("arrived", arrived)):
            if t is not None:
                assert t.is_contiguous() and t.device == self.device, name
                setattr(o, name, t.data_ptr())
        o.normalize_state, o.reward_kind, o.walkable_lasers = int(bool(normalize_state)), int(bool(multi_objective)), int(bool(walkable_lasers))
        self._check(_capi.lib().lle_batch_env_outputs(self.h, C.byref(o), self._stream()))

    def row_fill_prober(self, value=0):
        """A zero-argument callable: one launch that overwrites `obs` with the step kernel's store pattern and nothing else
        (lle_batch_probe_row_fill): the write ceiling of this box for this batch shape.  Call observe() afterwards."""
        fn, h, dev, v = _capi.lib().lle_batch_probe_row_fill, self.h, self.device, int(value)

        def probe():
            rc = fn(h, v, _current_stream_handle(dev))
            if rc != 0:
                self._check(rc)
        return probe

    def _noop_launch(self):
        """Profiling aid: a launch that loads the tables and the state and does nothing else (step with every
        action invalid and no observation write is the closest public equivalent)."""
        self._check(_capi.lib().lle_batch_step(self.h, None, LLE_STEP_NO_OBS | 0x100, 0, 0, 0, self._stream()))

    # ---- reward / done epilogue: `reward` holds the per-step counts the reference's
    # strategies are functions of; the two strategies themselves are two lines of torch
    def reward_single_objective(self):
        """SingleObjective.compute_reward (python/lle/env/reward_strategy.py:58-75): +1 per gem, +1 per exit, -1 per
        death, +1 when every agent has arrived (the death override of that class never fires: `death_reward` is never
        assigned).  float32 [n]."""
        r = self.reward.to(torch.float32)
        return r[:, 0] + r[:, 1] - r[:, 2] + r[:, 3]

    def reward_multi_objective(self):
        """MultiObjective.compute_reward (reward_strategy.py:90-109): [gem, exit, death, done]; a death zeroes the others.
        float32 [n, 4]."""
        r = self.reward.to(torch.float32)
        dead = r[:, 2] > 0
        out = torch.stack([r[:, 0], r[:, 1], -r[:, 2], r[:, 3]], dim=1)
        out[dead, 0] = 0
        out[dead, 1] = 0
        out[dead, 3] = 0
        return out

    def snapshot(self):
        """Exact checkpoint of the dynamic state (nothing re-derived, unlike get_state/set_state): a uint8 tensor."""
        n = _capi.lib().lle_batch_snapshot_bytes(self.h)
        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(_capi.lib().lle_batch_snapshot(self.h, buf.data_ptr(), self._stream()))
        return buf

    def restore(self, snapshot):
        self._check(_capi.lib().lle_batch_restore(self.h, snapshot.data_ptr(), self._stream()))

    def stats(self, reset=False):
        out = (C.c_int64 * 8)()
        self._check(_capi.lib().lle_batch_stats(self.h, out, int(reset), self._stream()))
        keys = ["env_steps", "agent_steps", "gems", "exits", "deaths", "invalid", "auto_resets", "reward_sum"]
        return {k: int(out[i]) for i, k in enumerate(keys)}

    # decoded views (small torch ops, for convenience)
    def agents_alive(self):
        return ((self.bits.unsqueeze(1) >> torch.arange(self.map.n_agents, device=self.device)) & 1).bool()

    def agents_arrived(self):
        return ((self.bits.unsqueeze(1) >> (16 + torch.arange(self.map.n_agents, device=self.device))) & 1).bool()

    def gems_collected(self):
        return ((self.gems.to(torch.int64).unsqueeze(1) >> torch.arange(self.map.n_gems, device=self.device)) & 1).bool()

    def observation(self):
        """(n, A, C, H, W) view of the layered observation: the reference tiles one (C,H,W) slice A times
        (python/lle/observations.py:266); here it is a broadcast view, not a copy."""
        return self.obs.unsqueeze(1).expand(-1, self.map.n_agents, -1, -1, -1)

    def host_small_state(self):
        """One device->host copy of everything except the observation, as numpy arrays in the LLE_BUF_* layout."""
        import numpy as np
        lo, hi = self._desc["pos"][0], self._desc["obs"][0]
        host = self._base[lo:hi].cpu().numpy()
        np_dt = {"pos": np.uint8, "bits": np.uint64, "gems": np.uint32, "beams": np.uint32, "avail": np.uint8,
                 "actions": np.uint8, "err": np.uint8, "evcount": np.uint8, "events": np.uint8, "done": np.uint8}
        out = {}
        for name, dt in np_dt.items():
            off, nbytes, shape, elem, stride = self._desc[name]
            flat = host[off - lo: off - lo + nbytes].view(dt)
            out[name] = np.lib.stride_tricks.as_strided(flat, shape, [v * elem for v in stride])
        return out

    def host_buffers(self, names=("pos", "bits", "gems", "beams", "avail", "actions", "err", "evcount", "events", "done", "obs")):
        """numpy copies in the LLE_BUF_* layout (synchronises)."""
        torch.cuda.synchronize(self.device)
        out = {}
        for nme in names:
            t = self.obs_rows if nme == "obs" else getattr(self, nme)
            if t.dtype == torch.bfloat16:  # (numpy has no bfloat16: the values are small integers, exact in float32)
                t = t.float()
            out[nme] = t.cpu().numpy()
        import numpy as np
        if "bits" in out:
            out["bits"] = out["bits"].view(np.uint64)
        if "gems" in out:
            out["gems"] = out["gems"].view(np.uint32)
        if "beams" in out:
            out["beams"] = out["beams"].view(np.uint32)
        return out
