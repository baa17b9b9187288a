"""ctypes front-end of the CPU oracle (oracle/lle_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package `lle_amd` never does.  See the header of lle_oracle.c for what the oracle is and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .levels import LEVELS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblle_oracle.so")

PARSE_ERRORS = {
    1: "EmptyWorld", 2: "NoAgents", 3: "InvalidTile", 4: "NotEnoughExitTiles", 5: "DuplicateStartTile",
    6: "InconsistentDimensions", 7: "InvalidAgentId", 8: "InvalidDirection", 9: "AgentWithoutStart",
    10: "NotEnoughStartTiles", 11: "TomlUnsupported",
}
RUNTIME_ERRORS = {
    -1: "InvalidNumberOfActions", -2: "InvalidNumberOfGems", -3: "InvalidNumberOfAgents",
    -4: "InvalidWorldState", -5: "OutOfWorldPosition", -6: "InvalidAgentPosition",
}


class OracleError(Exception):
    def __init__(self, kind, agent=None):
        super().__init__(f"{kind}" + (f" (agent {agent})" if agent is not None else ""))
        self.kind = kind
        self.agent = agent


def build(force=False):
    """Compile liblle_oracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "lle_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblle_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
        p8, pi32, pi64 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.ow_parse.restype = vp
        L.ow_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.ow_free.argtypes = [vp]
        L.ow_reset.argtypes = [vp]
        L.ow_step.restype = i32
        L.ow_step.argtypes = [vp, p8, i32, p8, i32, C.POINTER(C.c_int)]
        L.ow_get_state.argtypes = [vp, pi32, p8, p8]
        L.ow_set_state.restype = i32
        L.ow_set_state.argtypes = [vp, pi32, i32, p8, i32, p8, p8, i32, C.POINTER(C.c_int)]
        for name in ("height", "width", "n_agents", "n_gems", "n_sources", "n_exits", "n_walls", "n_voids", "panics",
                     "n_gems_collected"):
            getattr(L, "ow_" + name).restype = i32
            getattr(L, "ow_" + name).argtypes = [vp]
        L.ow_panic_msg.restype = C.c_char_p
        L.ow_panic_msg.argtypes = [vp]
        for name in ("agents_positions", "start_positions", "exit_positions", "wall_positions", "void_positions",
                     "gem_positions", "sources"):
            getattr(L, "ow_" + name).argtypes = [vp, pi32]
        L.ow_agents_flags.argtypes = [vp, p8, p8]
        L.ow_available_mask.argtypes = [vp, p8]
        L.ow_available_list.restype = i32
        L.ow_available_list.argtypes = [vp, i32, p8]
        L.ow_beam_bits.argtypes = [vp, i32, p8]
        L.ow_lasers.restype = i32
        L.ow_lasers.argtypes = [vp, pi32, i32]
        L.ow_tile_agent.restype = i32
        L.ow_tile_agent.argtypes = [vp, i32, i32]
        L.ow_gem_collect.restype = i32
        L.ow_gem_collect.argtypes = [vp, i32, i32]
        L.ow_set_exits.restype = i32
        L.ow_set_exits.argtypes = [vp, pi32, i32]
        L.ow_source_set_enabled.argtypes = [vp, i32, i32]
        L.ow_source_set_agent_id.argtypes = [vp, i32, i32]
        L.ow_layered_obs.restype = i32
        L.ow_layered_obs.argtypes = [vp, C.POINTER(C.c_int8)]
        L.ow_action_hash.restype = u64
        L.ow_action_hash.argtypes = [u64, u64, u64, u64]
        L.ow_sample_action.restype = i32
        L.ow_sample_action.argtypes = [C.c_uint8, u64, u64, u64, u64]
        L.ow_batch_create.restype = vp
        L.ow_batch_create.argtypes = [C.c_char_p, i64, C.POINTER(C.c_int)]
        L.ow_batch_free.argtypes = [vp]
        L.ow_batch_world.restype = vp
        L.ow_batch_world.argtypes = [vp, i64]
        L.ow_batch_reset.argtypes = [vp]
        L.ow_batch_step_range.argtypes = [vp, i64, i64, vp, i32, u64, u64, i64, vp, vp, vp, vp, vp, vp]
        L.ow_batch_dump.argtypes = [vp, i64, i64, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ow_batch_rollout.argtypes = [vp, i32, u64, i32, vp, pi64]
        L.ow_set_thread_pinning.argtypes = [i32]
        _lib = L
    return _lib


def _u8(n):
    return (C.c_uint8 * max(n, 1))()


def _i32(n):
    return (C.c_int32 * max(n, 1))()


def _pairs(buf, n):
    return [(int(buf[2 * k]), int(buf[2 * k + 1])) for k in range(n)]


class OracleWorld:
    """One world of the oracle, with the small uniform surface the KAT runner needs."""

    def __init__(self, map_str, _handle=None, _owner=None):
        self.L = lib()
        self._owner = _owner
        if _handle is not None:
            self.h = _handle
        else:
            err = C.c_int(0)
            self.h = self.L.ow_parse(map_str.encode(), C.byref(err))
            if not self.h:
                raise OracleError(PARSE_ERRORS.get(err.value, f"ParseError{err.value}"))
        self.map_str = map_str
        L, h = self.L, self.h
        self.height, self.width = L.ow_height(h), L.ow_width(h)
        self.n_agents, self.n_gems, self.n_sources = L.ow_n_agents(h), L.ow_n_gems(h), L.ow_n_sources(h)

    @staticmethod
    def level(n):
        if n not in LEVELS:
            raise OracleError("InvalidLevel")
        return OracleWorld(LEVELS[n])

    def __del__(self):
        if getattr(self, "h", None) and self._owner is None:
            self.L.ow_free(self.h)
            self.h = None

    # -- static layout
    def _poslist(self, fn, n):
        buf = _i32(2 * n)
        fn(self.h, buf)
        return _pairs(buf, n)

    @property
    def start_pos(self):
        return self._poslist(self.L.ow_start_positions, self.n_agents)

    @property
    def exit_pos(self):
        return self._poslist(self.L.ow_exit_positions, self.L.ow_n_exits(self.h))

    @property
    def wall_pos(self):
        return self._poslist(self.L.ow_wall_positions, self.L.ow_n_walls(self.h))

    @property
    def void_pos(self):
        return self._poslist(self.L.ow_void_positions, self.L.ow_n_voids(self.h))

    @property
    def gem_pos(self):
        return self._poslist(self.L.ow_gem_positions, self.n_gems)

    def sources(self):
        """[(i, j, direction, agent_id, enabled, len)] in laser_id order."""
        buf = _i32(6 * self.n_sources)
        self.L.ow_sources(self.h, buf)
        return [tuple(int(buf[6 * s + q]) for q in range(6)) for s in range(self.n_sources)]

    # -- dynamics
    def reset(self):
        self.L.ow_reset(self.h)

    def step(self, actions):
        acts = (C.c_uint8 * max(len(actions), 1))(*actions)
        cap = 4 * self.n_agents
        ev = _u8(2 * cap)
        n = C.c_int(0)
        rc = self.L.ow_step(self.h, acts, len(actions), ev, cap, C.byref(n))
        if rc > 0:
            raise OracleError("InvalidAction", agent=rc - 1)
        if rc < 0:
            raise OracleError(RUNTIME_ERRORS[rc])
        return _pairs(ev, n.value)

    def set_state(self, positions, gems, alive):
        flat = [v for p in positions for v in p]
        pos = (C.c_int32 * max(len(flat), 1))(*flat)
        g = (C.c_uint8 * max(len(gems), 1))(*[int(x) for x in gems])
        al = (C.c_uint8 * max(len(alive), 1))(*[int(x) for x in alive])
        cap = 2 * self.n_agents
        ev = _u8(2 * cap)
        n = C.c_int(0)
        rc = self.L.ow_set_state(self.h, pos, len(positions), g, len(gems), al, ev, cap, C.byref(n))
        if rc != 0:
            raise OracleError(RUNTIME_ERRORS[rc])
        return _pairs(ev, n.value)

    def get_state(self):
        pos, g, al = _i32(2 * self.n_agents), _u8(self.n_gems), _u8(self.n_agents)
        self.L.ow_get_state(self.h, pos, g, al)
        return (_pairs(pos, self.n_agents), [bool(g[k]) for k in range(self.n_gems)],
                [bool(al[k]) for k in range(self.n_agents)])

    def positions(self):
        return self._poslist(self.L.ow_agents_positions, self.n_agents)

    def alive(self):
        return self.get_state()[2]

    def arrived(self):
        al, ar = _u8(self.n_agents), _u8(self.n_agents)
        self.L.ow_agents_flags(self.h, al, ar)
        return [bool(ar[k]) for k in range(self.n_agents)]

    def gems_collected(self):
        return self.get_state()[1]

    def n_gems_collected(self):
        return self.L.ow_n_gems_collected(self.h)

    def available_actions(self):
        out = []
        for a in range(self.n_agents):
            buf = _u8(5)
            n = self.L.ow_available_list(self.h, a, buf)
            out.append([int(buf[k]) for k in range(n)])
        return out

    def available_mask(self):
        buf = _u8(self.n_agents)
        self.L.ow_available_mask(self.h, buf)
        return [int(buf[k]) for k in range(self.n_agents)]

    def lasers(self):
        """[(i, j, laser_id, agent_id, is_on, is_enabled)], outer layer first at each position (world.rs:159-172)."""
        n = self.L.ow_lasers(self.h, None, 0)
        buf = _i32(6 * n)
        self.L.ow_lasers(self.h, buf, n)
        return [tuple(int(buf[6 * k + q]) for q in range(6)) for k in range(n)]

    def beam_bits(self, laser_id):
        ln = self.sources()[laser_id][5]
        buf = _u8(ln)
        self.L.ow_beam_bits(self.h, laser_id, buf)
        return [bool(buf[k]) for k in range(ln)]

    def set_source(self, laser_id, enabled=None, colour=None):
        if enabled is not None:
            self.L.ow_source_set_enabled(self.h, laser_id, int(enabled))
        if colour is not None:
            self.L.ow_source_set_agent_id(self.h, laser_id, int(colour))

    def set_exits(self, exits):
        """World::set_exit_positions (world.rs:195-234)."""
        flat = [int(v) for p in exits for v in p]
        rc = self.L.ow_set_exits(self.h, (C.c_int32 * max(len(flat), 1))(*flat), len(exits))
        if rc > 0:
            raise OracleError(PARSE_ERRORS[rc])
        if rc < 0:
            raise OracleError("Panic")

    def tile_agent(self, i, j):
        return self.L.ow_tile_agent(self.h, i, j)

    def collect_gem(self, i, j):
        """Gem.collect() (pygem.rs:52-66)."""
        if self.L.ow_gem_collect(self.h, int(i), int(j)) != 0:
            raise OracleError("ValueError")

    def obs(self):
        c = 2 * self.n_agents + 4
        arr = np.zeros((c, self.height, self.width), dtype=np.int8)
        rc = self.L.ow_layered_obs(self.h, arr.ctypes.data_as(C.POINTER(C.c_int8)))
        if rc != 0:
            raise OracleError("IndexError")
        return arr

    def panics(self):
        return self.L.ow_panics(self.h), self.L.ow_panic_msg(self.h).decode()


class OracleBatch:
    """N independent oracle worlds of one map, stepped together, results in the canonical comparison layout."""

    def __init__(self, map_str, n_envs):
        self.L = lib()
        err = C.c_int(0)
        self.h = self.L.ow_batch_create(map_str.encode(), n_envs, C.byref(err))
        if not self.h:
            raise OracleError(PARSE_ERRORS.get(err.value, f"ParseError{err.value}"))
        self.n = n_envs
        w = OracleWorld(map_str, _handle=self.L.ow_batch_world(self.h, 0), _owner=self)
        self.world0 = w
        self.A, self.G, self.Ls, self.H, self.W = w.n_agents, w.n_gems, w.n_sources, w.height, w.width
        self.C = 2 * self.A + 4
        self.beam_stride = max([s[5] for s in w.sources()] + [1])
        # where the engines under test keep a beam (include/lle_hip.h lle_map_info.n_beam_words): ceil(len / 32) consecutive 32-bit
        # words per source, at least one; offset k of source s = bit k % 32 of word first_words[s] + k // 32
        self.first_words, nw = [], 0
        for src in w.sources():
            self.first_words.append(nw)
            nw += max(1, -(-int(src[5]) // 32))

    @property
    def dims(self):
        """The arguments of tests.parity_util.unpack_engine behind `bufs`."""
        return (self.A, self.G, self.Ls, self.beam_stride, self.C, self.H, self.W, self.first_words)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ow_batch_free(self.h)
            self.h = None

    def world(self, e):
        return OracleWorld(self.world0.map_str, _handle=self.L.ow_batch_world(self.h, e), _owner=self)

    def reset(self):
        self.L.ow_batch_reset(self.h)

    def step(self, actions=None, auto_reset=False, seed=0, t=0, env_offset=0, want_obs=True, stats=None):
        """actions: uint8 [n, A] or None (sample).  Returns dict of numpy arrays."""
        n, A = self.n, self.A
        out = {
            "actions": np.zeros((n, A), np.uint8), "err": np.zeros(n, np.int32), "ev_count": np.zeros(n, np.uint8),
            "events": np.zeros((n, 2 * A, 2), np.uint8),
        }
        if want_obs:
            out["obs"] = np.zeros((n, self.C, self.H, self.W), np.int8)
        ap = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
            assert actions.shape == (n, A)
            ap = actions.ctypes.data
        sp = stats.ctypes.data if stats is not None else None
        self.L.ow_batch_step_range(self.h, 0, n, ap, int(auto_reset), seed, t, env_offset,
                                   out["actions"].ctypes.data, out["err"].ctypes.data, out["ev_count"].ctypes.data,
                                   out["events"].ctypes.data, out["obs"].ctypes.data if want_obs else None, sp)
        return out

    def dump(self):
        n, A, G, Ls = self.n, self.A, self.G, self.Ls
        d = {
            "pos": np.zeros((n, A, 2), np.uint8), "alive": np.zeros((n, A), np.uint8),
            "arrived": np.zeros((n, A), np.uint8), "occupant": np.zeros((n, A), np.uint8),
            "gems": np.zeros((n, max(G, 1)), np.uint8)[:, :G], "avail": np.zeros((n, A), np.uint8),
            "beams": np.zeros((n, max(Ls, 1), self.beam_stride), np.uint8)[:, :Ls],
        }
        gems = np.zeros((n, G), np.uint8)
        beams = np.zeros((n, Ls, self.beam_stride), np.uint8)
        self.L.ow_batch_dump(self.h, 0, n, self.beam_stride, d["pos"].ctypes.data, d["alive"].ctypes.data,
                             d["arrived"].ctypes.data, d["occupant"].ctypes.data,
                             gems.ctypes.data if G else None, beams.ctypes.data if Ls else None, d["avail"].ctypes.data)
        d["gems"], d["beams"] = gems, beams
        return d

    def rollout(self, steps, seed, n_threads, obs=None):
        stats = np.zeros(8, np.int64)
        self.L.ow_batch_rollout(self.h, steps, seed, n_threads, obs.ctypes.data if obs is not None else None,
                                stats.ctypes.data_as(C.POINTER(C.c_int64)))
        return stats


def set_thread_pinning(on):
    """Pin thread k of OracleBatch.rollout to the k-th CPU this process may run on (bench.py cpu_baseline)."""
    lib().ow_set_thread_pinning(int(bool(on)))


def action_hash(seed, env, t, agent):
    return int(lib().ow_action_hash(seed, env, t, agent))


def sample_action(mask, seed, env, t, agent):
    return int(lib().ow_sample_action(mask, seed, env, t, agent))
