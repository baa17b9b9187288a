"""ctypes binding of liblle_hip.so (the C ABI of include/lle_hip.h).

This is the binding a reference maintainer would write in place of the PyO3 glue of
src/bindings/world/pyworld.rs: plain pointers and sizes.  The library is built in-tree by
`lle_amd.build.build_native()` (hipcc, gfx950).  There is no fallback: if the library is missing the import
of anything that needs it raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLE_HIP_LIB") or os.path.join(_HERE, "liblle_hip.so")  # override: A/B runs of two builds

# enum lle_hip.h
(LLE_BUF_POS, LLE_BUF_BITS, LLE_BUF_GEMS, LLE_BUF_BEAMS, LLE_BUF_AVAIL, LLE_BUF_ACTIONS, LLE_BUF_ERR, LLE_BUF_EVCOUNT,
 LLE_BUF_EVENTS, LLE_BUF_DONE, LLE_BUF_OBS, LLE_BUF_STATS, LLE_BUF_REQ_POS, LLE_BUF_REQ_GEMS, LLE_BUF_REQ_ALIVE,
 LLE_BUF_REWARD, LLE_BUF_SRC_COLOUR, LLE_BUF_SRC_ENABLED, LLE_BUF_COUNT) = range(19)
BUFFER_NAMES = ["pos", "bits", "gems", "beams", "avail", "actions", "err", "evcount", "events", "done", "obs", "stats",
                "req_pos", "req_gems", "req_alive", "reward", "src_colour", "src_enabled"]
LLE_POS_START, LLE_POS_EXIT, LLE_POS_WALL, LLE_POS_VOID, LLE_POS_GEM = range(5)
LLE_STEP_SAMPLE_ACTIONS, LLE_STEP_AUTO_RESET, LLE_STEP_NO_OBS, LLE_STEP_RECOLOUR_RESETS, LLE_STEP_INCREMENTAL_OBS = 1, 2, 4, 8, 16
RECOLOUR_SALT = 0xC01055EED  # the colour draws of LLE_STEP_RECOLOUR_RESETS hash (seed ^ RECOLOUR_SALT, env, t, laser_id)
LLE_ENV_INVALID_WORLD_STATE, LLE_ENV_OUT_OF_WORLD_POSITION, LLE_ENV_INVALID_AGENT_POSITION = 0x40, 0x41, 0x42
LLE_ENV_INVALID_COLOUR = 0x43
LLE_ENV_COLOUR_CROSSES_START = 0x44
LLE_ERR_NO_DEVICE = -5
LLE_ERR_UNSUPPORTED = -4
(LLE_OBS_LAYERED, LLE_OBS_LAYERED_PADDED, LLE_OBS_PERSPECTIVE, LLE_OBS_PARTIAL, LLE_OBS_STATE,
 LLE_OBS_NORMALIZED_STATE) = range(6)

PARSE_ERROR_NAMES = {
    1: "EmptyWorld", 2: "NoAgents", 3: "InvalidTile", 4: "NotEnoughExitTiles", 5: "DuplicateStartTile",
    6: "InconsistentDimensions", 7: "InvalidAgentId", 8: "InvalidDirection", 9: "AgentWithoutStart",
    10: "NotEnoughStartTiles", 11: "TomlUnsupported", 12: "InvalidLevel", 13: "Limit",
}

EXPORTS = [
    "lle_abi_version", "lle_last_status", "lle_last_error", "lle_action_hash",
    "lle_map_parse", "lle_map_level", "lle_map_free", "lle_map_get_info", "lle_map_positions", "lle_map_sources",
    "lle_map_set_source", "lle_map_set_exits", "lle_map_clone", "lle_map_colour_allowed", "lle_map_reset_beam", "lle_map_set_row_align", "lle_map_set_head_lines", "lle_map_row_head", "lle_map_row_head_env_sources", "lle_map_row_head_env_sources_second", "lle_map_row_dynamic_lines", "lle_map_laser_tiles", "lle_map_world_string",
    "lle_batch_arena_bytes", "lle_batch_create", "lle_batch_arena_bytes_multi", "lle_batch_create_multi", "lle_batch_n_maps", "lle_batch_free", "lle_batch_get_buffer", "lle_batch_n_envs",
    "lle_batch_reset", "lle_batch_step", "lle_batch_rollout", "lle_batch_set_state", "lle_batch_update_sources", "lle_batch_update_map", "lle_batch_observe",
    "lle_batch_snapshot_bytes", "lle_batch_snapshot", "lle_batch_restore",
    "lle_batch_set_sources", "lle_batch_reset_sources", "lle_batch_obs_desc", "lle_batch_observe_as", "lle_batch_available_actions", "lle_batch_env_outputs", "lle_batch_step_outputs",
    "lle_comm_unique_id", "lle_comm_create", "lle_comm_create_all", "lle_comm_free", "lle_comm_rank", "lle_batch_stats_allreduce",
    "lle_batch_stats_allreduce_group", "lle_comm_allreduce_i64", "lle_comm_allreduce_i64_group",
    "lle_batch_stats", "lle_batch_kernel_info", "lle_batch_set_envs_per_wave", "lle_batch_step_stamped", "lle_batch_probe_row_fill",
    "lle_batch_autotune", "lle_batch_tuning", "lle_tuning_refresh", "lle_probe_read_rows", "lle_probe_fill_rows",
    "lle_debug_launched", "lle_debug_reachable", "lle_debug_reset_launched",
    "lle_batch_arena_bytes_opt", "lle_batch_create_opt", "lle_batch_obs_dtype",
]


LLE_DTYPE_I8, LLE_DTYPE_F16, LLE_DTYPE_BF16, LLE_DTYPE_F32 = 0, 1, 2, 3


class BatchOptions(C.Structure):
    """lle_batch_options (include/lle_hip.h)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("obs_dtype", C.c_int32), ("reserved", C.c_int32 * 6)]

    def __init__(self, obs_dtype=0):
        super().__init__()
        self.struct_bytes = C.sizeof(BatchOptions)
        self.obs_dtype = int(obs_dtype)


class MapInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "height", "width", "n_agents", "n_gems", "n_sources", "n_layers", "n_exits", "n_walls", "n_voids",
        "n_laser_tiles", "obs_bytes", "obs_stride", "max_beam_len", "max_cell_layers", "obs_supported", "table_bytes", "n_beam_words", "dyn_row_bytes")]


class SourceInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("i", "j", "direction", "agent_id", "enabled", "length", "laser_id")]


class LaserTile(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("i", "j", "laser_id", "offset", "layer", "word", "bit")]


class EnvOutputs(C.Structure):
    """lle_env_outputs (include/lle_hip.h)."""
    _fields_ = [("state", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("available", C.c_void_p),
                ("alive", C.c_void_p), ("arrived", C.c_void_p), ("normalize_state", C.c_int32), ("reward_kind", C.c_int32),
                ("walkable_lasers", C.c_int32), ("partial_k", C.c_int32), ("partial", C.c_void_p)]


class TuningInfo(C.Structure):
    """lle_tuning_info (include/lle_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("envs_per_wave", "row_heads", "write_through", "split_rows", "alternating_walk", "rotate_rows", "autotuned", "head_group")]


class RolloutRing(C.Structure):
    _fields_ = [("ring_slots", C.c_int32), ("pad", C.c_int32), ("ring_pos", C.c_uint64), ("obs", C.c_void_p),
                ("actions", C.c_void_p), ("reward", C.c_void_p)]


class BufferDesc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("arena_offset", C.c_int64), ("bytes", C.c_int64), ("elem_bytes", C.c_int32),
                ("ndim", C.c_int32), ("shape", C.c_int64 * 3), ("stride", C.c_int64 * 3)]


class ObsDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("param", C.c_int32), ("elem_bytes", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 6), ("stride", C.c_int64 * 6), ("bytes", C.c_int64), ("supported", C.c_int32),
                ("pad", C.c_int32)]


_lib = None


def lib():
    """Load liblle_hip.so.  torch is imported first when available so that the HIP runtime torch bundles
    (same SONAME, libamdhip64.so.7) is the single runtime instance of the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  lle_amd has no fallback implementation.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:  # pragma: no cover - torch is plumbing, the map functions work without it
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32
    L.lle_abi_version.restype = i32
    L.lle_last_status.restype = i32
    L.lle_last_error.restype = C.c_char_p
    L.lle_action_hash.restype = u64
    L.lle_action_hash.argtypes = [u64, u64, u64, u64]
    L.lle_map_parse.restype = vp
    L.lle_map_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.lle_map_level.restype = vp
    L.lle_map_level.argtypes = [i32, C.POINTER(C.c_int)]
    L.lle_map_free.argtypes = [vp]
    L.lle_map_row_dynamic_lines.restype = i32
    L.lle_map_row_dynamic_lines.argtypes = [vp, C.POINTER(C.c_uint8), i32]
    L.lle_map_get_info.restype = i32
    L.lle_map_get_info.argtypes = [vp, C.POINTER(MapInfo)]
    L.lle_map_positions.restype = i32
    L.lle_map_positions.argtypes = [vp, i32, C.POINTER(C.c_int32), i32]
    L.lle_map_sources.restype = i32
    L.lle_map_sources.argtypes = [vp, C.POINTER(SourceInfo), i32]
    L.lle_map_set_source.restype = i32
    L.lle_map_set_source.argtypes = [vp, i32, i32, i32]
    L.lle_map_set_exits.restype = i32
    L.lle_map_set_exits.argtypes = [vp, C.POINTER(C.c_int32), i32, C.POINTER(C.c_int)]
    L.lle_map_clone.restype = vp
    L.lle_map_clone.argtypes = [vp]
    L.lle_map_colour_allowed.restype = i32
    L.lle_map_colour_allowed.argtypes = [vp, i32, i32]
    L.lle_map_set_row_align.restype = i32
    L.lle_map_set_row_align.argtypes = [vp, i32]
    L.lle_map_reset_beam.restype = C.c_int64
    L.lle_map_reset_beam.argtypes = [vp, i32, i32]
    L.lle_map_set_head_lines.restype = i32
    L.lle_map_set_head_lines.argtypes = [vp, i32]
    L.lle_map_row_head.restype = i32
    L.lle_map_row_head.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.lle_map_row_head_env_sources.restype = i32
    L.lle_map_row_head_env_sources.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.lle_map_row_head_env_sources_second.restype = i32
    L.lle_map_row_head_env_sources_second.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.lle_map_laser_tiles.restype = i32
    L.lle_map_laser_tiles.argtypes = [vp, C.POINTER(LaserTile), i32]
    L.lle_map_world_string.restype = C.c_size_t
    L.lle_map_world_string.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.lle_batch_arena_bytes.restype = i64
    L.lle_batch_arena_bytes.argtypes = [vp, i64]
    L.lle_batch_create.restype = vp
    L.lle_batch_create.argtypes = [vp, i64, i32, vp, i64, vp]
    L.lle_batch_arena_bytes_multi.restype = i64
    L.lle_batch_arena_bytes_multi.argtypes = [C.POINTER(vp), i32, i64]
    L.lle_batch_create_multi.restype = vp
    L.lle_batch_create_multi.argtypes = [C.POINTER(vp), i32, i64, i32, vp, i64, vp]
    L.lle_batch_n_maps.restype = i32
    L.lle_batch_n_maps.argtypes = [vp]
    L.lle_batch_arena_bytes_opt.restype = i64
    L.lle_batch_arena_bytes_opt.argtypes = [C.POINTER(vp), i32, i64, C.POINTER(BatchOptions)]
    L.lle_batch_create_opt.restype = vp
    L.lle_batch_create_opt.argtypes = [C.POINTER(vp), i32, i64, i32, vp, i64, C.POINTER(BatchOptions), vp]
    L.lle_batch_obs_dtype.restype = i32
    L.lle_batch_obs_dtype.argtypes = [vp]
    L.lle_batch_free.argtypes = [vp]
    L.lle_batch_get_buffer.restype = i32
    L.lle_batch_get_buffer.argtypes = [vp, i32, C.POINTER(BufferDesc)]
    L.lle_batch_n_envs.restype = i64
    L.lle_batch_n_envs.argtypes = [vp]
    L.lle_batch_snapshot_bytes.restype = i64
    L.lle_batch_snapshot_bytes.argtypes = [vp]
    L.lle_batch_snapshot.restype = i32
    L.lle_batch_snapshot.argtypes = [vp, vp, vp]
    L.lle_batch_restore.restype = i32
    L.lle_batch_restore.argtypes = [vp, vp, vp]
    L.lle_batch_reset.restype = i32
    L.lle_batch_reset.argtypes = [vp, vp, vp]
    L.lle_batch_step.restype = i32
    L.lle_batch_step.argtypes = [vp, vp, u32, u64, u64, i64, vp]
    L.lle_batch_rollout.restype = i32
    L.lle_batch_rollout.argtypes = [vp, u32, u32, u64, u64, i64, C.POINTER(RolloutRing), vp]
    L.lle_batch_set_state.restype = i32
    L.lle_batch_set_state.argtypes = [vp, vp]
    L.lle_batch_update_sources.restype = i32
    L.lle_batch_update_sources.argtypes = [vp, vp, vp]
    L.lle_batch_update_map.restype = i32
    L.lle_batch_update_map.argtypes = [vp, i32, vp, vp]
    L.lle_batch_observe.restype = i32
    L.lle_batch_observe.argtypes = [vp, vp]
    L.lle_batch_set_sources.restype = i32
    L.lle_batch_set_sources.argtypes = [vp, vp, vp, vp, vp]
    L.lle_batch_reset_sources.restype = i32
    L.lle_batch_reset_sources.argtypes = [vp, vp, vp, vp, u32, vp]
    L.lle_batch_obs_desc.restype = i32
    L.lle_batch_obs_desc.argtypes = [vp, i32, i32, C.POINTER(ObsDesc)]
    L.lle_batch_observe_as.restype = i32
    L.lle_batch_observe_as.argtypes = [vp, i32, i32, vp, i64, vp]
    L.lle_batch_available_actions.restype = i32
    L.lle_batch_available_actions.argtypes = [vp, i32, vp, vp]
    L.lle_batch_env_outputs.restype = i32
    L.lle_batch_env_outputs.argtypes = [vp, C.POINTER(EnvOutputs), vp]
    L.lle_batch_step_outputs.restype = i32
    L.lle_batch_step_outputs.argtypes = [vp, vp, u32, u64, u64, i64, C.POINTER(EnvOutputs), vp]
    L.lle_batch_stats.restype = i32
    L.lle_batch_stats.argtypes = [vp, C.POINTER(C.c_int64), i32, vp]
    L.lle_comm_unique_id.restype = i32
    L.lle_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.lle_comm_create.restype = vp
    L.lle_comm_create.argtypes = [C.POINTER(C.c_uint8), i32, i32, i32]
    L.lle_comm_create_all.restype = i32
    L.lle_comm_create_all.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int)]
    L.lle_comm_free.argtypes = [vp]
    L.lle_comm_rank.restype = i32
    L.lle_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lle_batch_stats_allreduce.restype = i32
    L.lle_batch_stats_allreduce.argtypes = [vp, vp, C.POINTER(C.c_int64), i32, vp]
    L.lle_batch_stats_allreduce_group.restype = i32
    L.lle_batch_stats_allreduce_group.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(C.c_int64), i32]
    L.lle_comm_allreduce_i64_group.restype = i32
    L.lle_comm_allreduce_i64_group.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, i32]
    L.lle_comm_allreduce_i64.restype = i32
    L.lle_comm_allreduce_i64.argtypes = [vp, vp, i32, i32, vp]
    L.lle_batch_kernel_info.restype = i32
    L.lle_batch_kernel_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.lle_batch_step_stamped.restype = i32
    L.lle_batch_step_stamped.argtypes = [vp, u32, u64, u64, vp, vp]
    L.lle_batch_probe_row_fill.restype = i32
    L.lle_batch_probe_row_fill.argtypes = [vp, u32, vp]
    L.lle_batch_set_envs_per_wave.restype = i32
    L.lle_batch_set_envs_per_wave.argtypes = [vp, i32]
    L.lle_batch_autotune.restype = i32
    L.lle_batch_autotune.argtypes = [vp, C.c_double, vp]
    L.lle_batch_tuning.restype = i32
    L.lle_batch_tuning.argtypes = [vp, C.POINTER(TuningInfo), C.c_char_p, C.c_size_t]
    L.lle_probe_fill_rows.restype = i32
    L.lle_probe_fill_rows.argtypes = [vp, i64, i64, i32, vp]
    L.lle_probe_read_rows.restype = i32
    L.lle_probe_read_rows.argtypes = [vp, vp, i64, vp]
    L.lle_tuning_refresh.restype = None
    L.lle_tuning_refresh.argtypes = []
    for fn in (L.lle_debug_launched, L.lle_debug_reachable):
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_char_p, C.c_size_t]
    L.lle_debug_reset_launched.restype = None
    L.lle_debug_reset_launched.argtypes = []
    _lib = L
    return L


def _debug_names(fn):
    need = fn(None, 0)
    buf = C.create_string_buffer(need)
    fn(buf, need)
    return [n for n in buf.value.decode().split("\n") if n]


def launched_kernels():
    """Names of the kernel instantiations this process has launched so far (lle_debug_launched)."""
    return _debug_names(lib().lle_debug_launched)


def reachable_kernels():
    """Names of every kernel instantiation the launchers' dispatch can reach (lle_debug_reachable)."""
    return _debug_names(lib().lle_debug_reachable)


def refresh_tuning():
    """Read the LLE_* tuning overrides from the environment again (lle_tuning_refresh): the library reads them once per process."""
    lib().lle_tuning_refresh()


class MapParseError(ValueError):
    """A map could not be parsed/compiled; `.kind` is the reference's ParseError variant name."""

    def __init__(self, code):
        self.code = code
        self.kind = PARSE_ERROR_NAMES.get(code, f"ParseError{code}")
        super().__init__(self.kind)


class Map:
    """Host-side compiled map (lle_map*).  Needs no GPU."""

    def __init__(self, text=None, level=None, row_align=None, _handle=None):
        """`row_align`: pitch of an observation row in bytes (lle_map_set_row_align; default 0 = automatic: whole 128-byte
        lines when that pads the row by at most 1/32 -- level 6: 1 872 -> 1 920 B --, 16 otherwise)."""
        L = lib()
        err = C.c_int(0)
        if _handle is not None:
            self.h = _handle
        elif level is not None:
            self.h = L.lle_map_level(int(level), C.byref(err))
        else:
            data = text.encode()
            self.h = L.lle_map_parse(data, len(data), C.byref(err))
        if not self.h:
            raise MapParseError(err.value)
        if row_align is not None and L.lle_map_set_row_align(self.h, int(row_align)) != 0:
            raise ValueError(L.lle_last_error().decode())
        self.refresh()

    def refresh(self):
        info = MapInfo()
        lib().lle_map_get_info(self.h, C.byref(info))
        self.info = info
        for name, _ in MapInfo._fields_:
            setattr(self, name, int(getattr(info, name)))

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().lle_map_free(self.h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass
            self.h = None

    def clone(self):
        """An independent copy (lle_map_clone): sources, exits, row alignment and head lines included."""
        h = lib().lle_map_clone(self.h)
        if not h:
            raise MemoryError("lle_map_clone failed")
        return Map(_handle=h)

    def set_exits(self, exits):
        """World::set_exit_positions on the host object (lle_map_set_exits).  Raises MapParseError("NotEnoughExitTiles") or
        ValueError (a cell where the reference panics); the map is untouched then."""
        flat = [int(v) for p in exits for v in p]
        if len(flat) != 2 * len(exits):
            raise ValueError("exit positions are (i, j) pairs")
        buf = (C.c_int32 * max(len(flat), 1))(*flat)
        err = C.c_int(0)
        rc = lib().lle_map_set_exits(self.h, buf, len(exits), C.byref(err))
        if rc != 0:
            if err.value:
                raise MapParseError(err.value)
            raise ValueError(lib().lle_last_error().decode())
        self.refresh()

    def positions(self, which):
        n = lib().lle_map_positions(self.h, which, None, 0)
        buf = (C.c_int32 * max(2 * n, 1))()
        lib().lle_map_positions(self.h, which, buf, n)
        return [(int(buf[2 * k]), int(buf[2 * k + 1])) for k in range(n)]

    def sources(self):
        n = lib().lle_map_sources(self.h, None, 0)
        arr = (SourceInfo * max(n, 1))()
        lib().lle_map_sources(self.h, arr, n)
        return [arr[k] for k in range(n)]

    def source_first_words(self):
        """First beam word of every source (lle_map_info.n_beam_words): a beam of `length` cells takes ceil(length / 32) consecutive
        32-bit words of an env's LLE_BUF_BEAMS / LLE_BUF_SRC_COLOUR record, at least one; == laser_id when no beam exceeds 32 cells."""
        out, w = [], 0
        for s in self.sources():
            out.append(w)
            w += max(1, -(-int(s.length) // 32))
        return out

    def laser_tiles(self):
        n = lib().lle_map_laser_tiles(self.h, None, 0)
        arr = (LaserTile * max(n, 1))()
        lib().lle_map_laser_tiles(self.h, arr, n)
        return [arr[k] for k in range(n)]

    def colour_allowed(self, laser_id, agent_id):
        """May source `laser_id` take colour `agent_id`?  (no start of another agent on its beam, pylaser_source.rs:121-139)"""
        rc = lib().lle_map_colour_allowed(self.h, int(laser_id), int(agent_id))
        if rc < 0:
            raise ValueError(lib().lle_last_error().decode())
        return bool(rc)

    def reset_beam(self, laser_id, agent_id):
        """Beam mask of source `laser_id` right after World.reset when it is enabled and has colour `agent_id` (lle_map_reset_beam)."""
        v = lib().lle_map_reset_beam(self.h, int(laser_id), int(agent_id))
        if v < 0:
            raise ValueError(lib().lle_last_error().decode())
        return int(v)

    def set_row_align(self, align):
        if lib().lle_map_set_row_align(self.h, int(align)) != 0:
            raise ValueError(lib().lle_last_error().decode())
        self.refresh()

    def set_head_lines(self, lines):
        """Most 128-byte lines of a row that the step kernel stores ahead of its state machine (lle_map_set_head_lines)."""
        if lib().lle_map_set_head_lines(self.h, int(lines)) != 0:
            raise ValueError(lib().lle_last_error().decode())
        self.refresh()

    @property
    def row_head(self):
        """(first byte, number of bytes) of the row's head: lines no agent, beam or gem can change (lle_map_row_head)."""
        a, n = C.c_int32(0), C.c_int32(0)
        lib().lle_map_row_head(self.h, C.byref(a), C.byref(n))
        return a.value, n.value

    @property
    def row_dynamic_lines(self):
        """One bool per 128-byte line of a row: can dynamic state change it?  (lle_map_row_dynamic_lines; what incremental steps write)"""
        n = lib().lle_map_row_dynamic_lines(self.h, None, 0)
        buf = (C.c_uint8 * max(n, 1))()
        lib().lle_map_row_dynamic_lines(self.h, buf, n)
        return [bool(buf[k]) for k in range(n)]

    @property
    def row_head_env_sources(self):
        """(first byte, number of bytes) of the row's head under per-environment source colours (lle_map_row_head_env_sources)."""
        a, n = C.c_int32(0), C.c_int32(0)
        lib().lle_map_row_head_env_sources(self.h, C.byref(a), C.byref(n))
        return a.value, n.value

    @property
    def row_head_env_sources_second(self):
        """(first byte, number of bytes) of the second run of head lines under per-environment source colours."""
        a, n = C.c_int32(0), C.c_int32(0)
        lib().lle_map_row_head_env_sources_second(self.h, C.byref(a), C.byref(n))
        return a.value, n.value

    def set_source(self, laser_id, enabled=None, agent_id=None):
        rc = lib().lle_map_set_source(self.h, laser_id, -1 if enabled is None else int(bool(enabled)),
                                      -1 if agent_id is None else int(agent_id))
        if rc != 0:
            raise ValueError(lib().lle_last_error().decode())
        self.refresh()

    def world_string(self):
        n = lib().lle_map_world_string(self.h, None, 0)
        buf = C.create_string_buffer(n)
        lib().lle_map_world_string(self.h, buf, n)
        return buf.value.decode()
