"""TEST-ONLY host build of the device state machine (see hostsim.cpp).  Never imported by lle_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

from lle_amd import _capi, _decode

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libhostsim.so")
_SRCS = [os.path.join(_HERE, "hostsim.cpp"), os.path.join(_ROOT, "lle_amd", "csrc", "map_compile.cpp")]
_DEPS = _SRCS + [os.path.join(_ROOT, "lle_amd", "csrc", f) for f in ("step_logic.hpp", "step_lanes.hpp", "observers_logic.hpp", "tables.h", "map_compile.hpp")]
ENGINES = {"env": 0, "lanes": 1, "lanes_no_shortcut": 2}
DEFAULT_ENGINE = ["env"]  # tests/test_hostsim_lanes.py reruns the CPU suites with "lanes": the step_kernel restatement


def build():
    if not os.path.exists(_LIB) or any(os.path.getmtime(_LIB) < os.path.getmtime(d) for d in _DEPS):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", _LIB] + _SRCS)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.hs_create.restype = C.c_void_p
        L.hs_create.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
        L.hs_free.argtypes = [C.c_void_p]
        L.hs_set_engine.argtypes = [C.c_void_p, C.c_int]
        L.hs_lane_passes.restype = C.c_int64
        L.hs_lane_passes.argtypes = [C.c_void_p]
        L.hs_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.hs_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int64]
        L.hs_set_state.argtypes = [C.c_void_p]
        L.hs_observe.argtypes = [C.c_void_p]
        L.hs_set_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.hs_set_exits.restype = C.c_int
        L.hs_set_exits.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.hs_set_sources.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_observe_as.restype = C.c_int
        L.hs_observe_as.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.hs_available_actions.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hs_template_from_bits.restype = C.c_int64
        L.hs_template_from_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_tables_from_packed.restype = C.c_int64
        L.hs_tables_from_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int)]
        L.hs_buffer.restype = C.c_void_p
        L.hs_buffer.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


class SimError(Exception):
    def __init__(self, kind, agent=None):
        super().__init__(kind)
        self.kind, self.agent = kind, agent


_DT = {"pos": np.uint8, "bits": np.uint64, "gems": np.uint32, "beams": np.uint32, "avail": np.uint8, "actions": np.uint8,
       "err": np.uint8, "evcount": np.uint8, "events": np.uint8, "done": np.uint8, "obs": np.int8, "stats": np.int64,
       "req_pos": np.uint8, "req_gems": np.uint32, "req_alive": np.uint16}


class SimBatch:
    """n envs of one map in the host simulator; buffers are numpy views with the device layout."""

    def __init__(self, text, n=1):
        self.L = lib()
        err = C.c_int(0)
        self.h = self.L.hs_create(text.encode(), n, C.byref(err))
        if not self.h:
            raise SimError(_capi.PARSE_ERROR_NAMES.get(err.value, str(err.value)))
        self.n = n
        self.set_engine(DEFAULT_ENGINE[0])
        self.map = _capi.Map(text)  # static description through the product's host-only map functions
        m = self.map
        A, Ls = m.n_agents, max(m.n_beam_words, 1)  # (LLE_BUF_BEAMS holds beam WORDS: == n_sources unless a beam is longer than 32 cells)
        self.shapes = {"pos": (n, A, 2), "bits": (n,), "gems": (n,), "beams": (n, Ls), "avail": (n, A), "actions": (n, A),
                       "err": (n,), "evcount": (n,), "events": (n, 2 * A), "done": (n,), "obs": (n, m.obs_stride),
                       "stats": (8,), "req_pos": (n, A, 2), "req_gems": (n,), "req_alive": (n,)}

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_free(self.h)
            self.h = None

    def set_engine(self, engine):
        """"env": step_logic.hpp (one lane per environment, world_kernel); "lanes": step_lanes.hpp (one lane per agent, the
        step_kernel restatement); "lanes_no_shortcut": the same with every move_agents pass executed."""
        self.L.hs_set_engine(self.h, ENGINES[engine])

    def lane_passes(self):
        """move_agents passes the lane engines have executed so far (all envs, all steps)."""
        return int(self.L.hs_lane_passes(self.h))

    def buf(self, name):
        which = _capi.BUFFER_NAMES.index(name)
        shape = self.shapes[name]
        count = int(np.prod(shape))
        ptr = self.L.hs_buffer(self.h, which)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(_DT[name]))), shape=(count,))
        return arr.reshape(shape)

    def template_from_bits(self):
        """(rebuilt from the blob's bit form with the kernel's arithmetic, the template itself), or (None, template) when the map has no bit form."""
        a, b = np.zeros(self.map.obs_stride, np.int8), np.zeros(self.map.obs_stride, np.int8)
        r = int(self.L.hs_template_from_bits(self.h, a.ctypes.data, b.ctypes.data))
        assert r >= 0, "malformed bit section"
        return (a if r else None), b

    def tables_from_packed(self):
        """(the table section rebuilt from the blob's packed image the way the kernel fills LDS -- None when the map has no image --, the section itself)."""
        cap = 1 << 22
        a, b, has = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), C.c_int(0)
        n = int(self.L.hs_tables_from_packed(self.h, a.ctypes.data, b.ctypes.data, cap, C.byref(has)))
        assert n >= 0, "malformed packed image"
        return (a[:n] if has.value else None), b[:n]

    def reset(self):
        self.L.hs_reset(self.h, None)

    def step(self, actions=None, flags=0, seed=0, t=0, env_offset=0):
        ap = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
            ap = actions.ctypes.data
        self.L.hs_step(self.h, ap, flags, seed, t, env_offset)

    def set_state(self):
        self.L.hs_set_state(self.h)

    def set_sources(self, colours=None, enabled=None, env_mask=None):
        """Per-environment source colours u8 [n, L] / enabled masks u32 [n] (lle_batch_set_sources)."""
        keep = []

        def ptr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        self.L.hs_set_sources(self.h, ptr(colours, np.uint8), ptr(enabled, np.uint32), ptr(env_mask, np.uint8))

    def observe_as(self, kind, param=0):
        """Logical (unpadded) array of observation `kind` for every env, or None when the reference would raise
        IndexError (a laser colour without a layer)."""
        m, n = self.map, self.n
        A, H, W, G = m.n_agents, m.height, m.width, m.n_gems
        shape, dtype = {
            _capi.LLE_OBS_LAYERED: ((n, 2 * A + 4, H, W), np.int8),
            _capi.LLE_OBS_LAYERED_PADDED: ((n, 2 * (A + param) + 4, H, W), np.int8),
            _capi.LLE_OBS_PERSPECTIVE: ((n, A, 2 * A + 4, H, W), np.int8),
            _capi.LLE_OBS_PARTIAL: ((n, A, 2 * A + 3, param, param), np.int8),
            _capi.LLE_OBS_STATE: ((n, 3 * A + G), np.float32),
            _capi.LLE_OBS_NORMALIZED_STATE: ((n, 3 * A + G), np.float32),
        }[kind]
        out = np.zeros(shape, dtype)
        rc = self.L.hs_observe_as(self.h, kind, param, out.ctypes.data)
        return out if rc == 0 else None

    def available_actions(self, walkable_lasers=True):
        out = np.zeros((self.n, self.map.n_agents, 5), np.uint8)
        self.L.hs_available_actions(self.h, int(walkable_lasers), out.ctypes.data)
        return out.astype(bool)

    def set_source(self, laser_id, enabled=None, colour=None):
        self.L.hs_set_source(self.h, laser_id, -1 if enabled is None else int(enabled), -1 if colour is None else int(colour))
        self.map.set_source(laser_id, enabled=enabled, agent_id=colour)


    def set_exits(self, exits):
        flat = [int(v) for p in exits for v in p]
        rc = self.L.hs_set_exits(self.h, (C.c_int32 * max(len(flat), 1))(*flat), len(exits))
        if rc > 0:
            raise SimError(_capi.PARSE_ERROR_NAMES[rc])
        if rc < 0:
            raise SimError("Panic")
        self.map.set_exits(exits)


class SimWorld:
    """KAT surface over a 1-env SimBatch (mirrors oracle.OracleWorld)."""

    def __init__(self, map_str=None, level=None):
        if level is not None:
            from oracle.levels import LEVELS
            map_str = LEVELS[level]
        self.b = SimBatch(map_str, 1)
        m = self.b.map
        self.height, self.width, self.n_agents, self.n_gems, self.n_sources = m.height, m.width, m.n_agents, m.n_gems, m.n_sources
        self.start_pos = m.positions(_capi.LLE_POS_START)
        self.exit_pos = m.positions(_capi.LLE_POS_EXIT)
        self.wall_pos = m.positions(_capi.LLE_POS_WALL)
        self.void_pos = m.positions(_capi.LLE_POS_VOID)
        self.gem_pos = m.positions(_capi.LLE_POS_GEM)

    def sources(self):
        return [(s.i, s.j, s.direction, s.agent_id, s.enabled, s.length) for s in self.b.map.sources()]

    def reset(self):
        self.b.reset()

    def step(self, actions):
        if len(actions) != self.n_agents:
            raise SimError("InvalidNumberOfActions")
        self.b.step(np.asarray(actions, np.uint8).reshape(1, -1))
        err = int(self.b.buf("err")[0])
        if err:
            raise SimError("InvalidAction", agent=err - 1)
        return _decode.events_list(self.b.buf("evcount")[0], self.b.buf("events")[0])

    def set_state(self, positions, gems, alive):
        if len(gems) != self.n_gems:
            raise SimError("InvalidNumberOfGems")
        if len(positions) != self.n_agents:
            raise SimError("InvalidNumberOfAgents")
        if any(max(p) > 255 for p in positions):
            if len(set(map(tuple, positions))) != len(positions):
                raise SimError("InvalidWorldState")
            raise SimError("OutOfWorldPosition")
        self.b.buf("req_pos")[0] = _decode.pack_positions(positions)
        self.b.buf("req_gems")[0] = _decode.pack_bits(gems)
        self.b.buf("req_alive")[0] = _decode.pack_bits(alive)
        self.b.set_state()
        err = int(self.b.buf("err")[0])
        if err:
            raise SimError({0x40: "InvalidWorldState", 0x41: "OutOfWorldPosition", 0x42: "InvalidAgentPosition"}[err])
        return _decode.events_list(self.b.buf("evcount")[0], self.b.buf("events")[0])

    def positions(self):
        return _decode.positions(self.b.buf("pos")[0])

    def _bits(self):
        return _decode.agent_bits(self.b.buf("bits")[0], self.n_agents)

    def alive(self):
        return self._bits()[0]

    def arrived(self):
        return self._bits()[1]

    def gems_collected(self):
        return _decode.gem_bits(self.b.buf("gems")[0], self.n_gems)

    def n_gems_collected(self):
        direct = {(t.i, t.j) for t in self.b.map.laser_tiles()}
        return sum(1 for c, p in zip(self.gems_collected(), self.gem_pos) if c and p not in direct)

    def available_actions(self):
        return _decode.avail_lists(self.b.buf("avail")[0])

    def lasers(self):
        return _decode.lasers_listing(self.b.map.laser_tiles(), self.b.map.sources(), self.b.buf("beams")[0])

    def beam_bits(self, laser_id):
        return _decode.beam_bits(self.b.buf("beams")[0], self.b.map.source_first_words()[laser_id], self.b.map.sources()[laser_id].length)

    def set_source(self, laser_id, enabled=None, colour=None):
        self.b.set_source(laser_id, enabled, colour)

    def set_exits(self, exits):
        self.b.set_exits(exits)
        self.exit_pos = self.b.map.positions(_capi.LLE_POS_EXIT)

    def collect_gem(self, i, j):
        """Gem.collect() (pygem.rs:52-66) the way the facade does it: the gem's bit in the env's word, observation rebuilt."""
        gems = self.b.map.positions(_capi.LLE_POS_GEM)
        under_beam = {(t.i, t.j) for t in self.b.map.laser_tiles()}
        if (i, j) not in gems or (i, j) in under_beam:
            raise SimError("ValueError")
        self.b.buf("gems")[0] |= np.uint32(1 << gems.index((i, j)))
        self.b.L.hs_observe(self.b.h)

    def tile_agent(self, i, j):
        """Tile::agent (tile.rs:86-95): the occupant of cell (i, j), -1 = none."""
        occ = self._bits()[2]
        for a, p in enumerate(self.positions()):
            if occ[a] and tuple(p) == (i, j):
                return a
        return -1

    def obs(self):
        m = self.b.map
        if not m.obs_supported:
            raise SimError("IndexError")
        return self.b.buf("obs")[0][: m.obs_bytes].reshape(m.n_layers, m.height, m.width).copy()
