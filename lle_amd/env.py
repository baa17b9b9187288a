"""BatchedLLE: n lock-stepped `LLE` environments on one MI355X.

The reference's `LLE` (python/lle/env/env.py:39-255) is a thin host class around `World.step`: events -> reward
strategy, `compute_done`, observation / state generators, `available_actions` (optionally without moves into foreign
lasers), `randomize_lasers` on reset.  Every one of those pieces already runs on the GPU behind the C ABI
(include/lle_hip.h); this class strings them together with the reference's argument names and meanings and returns
device tensors with a leading env axis.  No marlenv dependency, no extras generators, no PBRS shaping and no rendering
 -- the hot path and its immediate callers only.

    env = BatchedLLE(Map(level=6), 65536, obs_type="layered", randomize_lasers=True)
    obs, state = env.reset()
    step = env.step(actions)            # dict: obs, state, reward, done, available_actions
"""
import os
from enum import IntEnum

import torch

from . import _capi
from .batched import BatchedWorld


class DeathStrategy(IntEnum):
    """python/lle/env/env.py:23-37."""
    END = 0      # the episode ends when an agent dies
    RESPAWN = 1  # (not implemented by the reference either: env.py:106-107)

    @staticmethod
    def from_str(value):
        if value == "end":
            return DeathStrategy.END
        if value == "respawn":
            return DeathStrategy.RESPAWN
        raise ValueError(f"Unknown death strategy: {value}")

_OBS_KINDS = {
    "layered": (_capi.LLE_OBS_LAYERED, 0), "flattened": (_capi.LLE_OBS_LAYERED, 0),
    "partial3x3": (_capi.LLE_OBS_PARTIAL, 3), "partial5x5": (_capi.LLE_OBS_PARTIAL, 5), "partial7x7": (_capi.LLE_OBS_PARTIAL, 7),
    "state": (_capi.LLE_OBS_STATE, 0), "normalized-state": (_capi.LLE_OBS_NORMALIZED_STATE, 0),
    "perspective": (_capi.LLE_OBS_PERSPECTIVE, 0),
    "layered-padded-1": (_capi.LLE_OBS_LAYERED_PADDED, 1), "layered-padded-2": (_capi.LLE_OBS_LAYERED_PADDED, 2),
    "layered-padded-3": (_capi.LLE_OBS_LAYERED_PADDED, 3),
}


class BatchedLLE:
    """Arguments follow `LLE.__init__` / `Builder` (python/lle/env/env.py:72-114, builder.py:30-116):
    obs_type / state_type: ObservationType values ("layered", "flattened", "partial3x3", ..., "state", "normalized-state",
    "perspective", "layered-padded[-k]"; padding_size for plain "layered-padded"); obs_dtype (the layered-style observations in float32 -- the reference's --,
    float16 or bfloat16 instead of int8, straight from the kernels); walkable_lasers; randomize_lasers;
    multi_objective (MultiObjective instead of SingleObjective); death_strategy "end" only, like the reference."""

    def __init__(self, maps, n_envs, obs_type="layered", state_type="state", walkable_lasers=True, randomize_lasers=False,
                 multi_objective=False, death_strategy="end", padding_size=0, device=None, seed=0, name=None, incremental_obs=False, obs_dtype=None):
        if death_strategy == "respawn":
            raise NotImplementedError("Respawn strategy is not implemented yet")  # env.py:106-107
        if death_strategy != "end":
            raise ValueError(f"Unknown death strategy: {death_strategy}")
        self.death_strategy = DeathStrategy.END
        self._name = name
        obs_type, state_type = getattr(obs_type, "value", obs_type), getattr(state_type, "value", state_type)  # ObservationType or its string
        self.obs_type, self.state_type = str(obs_type), str(state_type)
        self._obs_kind = self._kind(self.obs_type, padding_size)
        self._state_kind = self._kind(self.state_type, padding_size)
        # obs_dtype: element type of the layered-style observations (layered, flattened, padded, perspective, partial: values -1 / 0 / 1) --
        # torch.float32 is the reference's (python/lle/observations.py:223), float16 / bfloat16 what a learner's first layer usually reads; every
        # kernel that writes them widens at the store (BatchedWorld(obs_dtype=...)).  The state vector is float32 whatever the type.
        if obs_dtype is not None and all(k[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE) for k in (self._obs_kind, self._state_kind)):
            raise ValueError("obs_dtype is the element type of the layered-style observations: obs_type and state_type are both state vectors")
        self.world = BatchedWorld(maps, n_envs, device=device, obs_dtype=obs_dtype)
        self.n_envs, self.n_agents, self.n_actions = self.world.n_envs, self.world.map.n_agents, 5
        # the step kernel's own (layered) observation is only written when somebody reads it
        self._needs_layered = _capi.LLE_OBS_LAYERED in (self._obs_kind[0], self._state_kind[0])
        self.walkable_lasers = bool(walkable_lasers)
        self.randomize_lasers = bool(randomize_lasers)
        if self.randomize_lasers:
            # LLE.reset draws `source.set_colour(random.randint(0, n_agents - 1))` (env.py:198-200) and set_colour raises
            # ValueError for a colour that puts another agent's start on the beam (pylaser_source.rs:121-139): on such a
            # map the reference fails at the first reset that draws the pair; a batch draws every pair at once.
            for m in self.world.maps:
                for s in m.sources():
                    for c in range(m.n_agents):
                        if not m.colour_allowed(s.laser_id, c):
                            raise ValueError(f"randomize_lasers: laser source {s.laser_id} at {(s.i, s.j)} cannot be changed to agent ID "
                                             f"{c} since it would cross the start position of another agent")
        self.multi_objective = bool(multi_objective)
        # incremental_obs: steps write only the lines of the layered rows that dynamic state can change (LLE_STEP_INCREMENTAL_OBS: the
        # others keep their bytes from the last full write) -- same observation, a third fewer bytes on level 6; the caller must not write
        # into `world.obs` itself
        self._incr = bool(incremental_obs)
        self._gen = torch.Generator(device=self.world.device)
        self._gen.manual_seed(int(seed))
        self._seed_value = int(seed)
        self._t = 0
        # auto-resets under randomize_lasers re-colour inside the step kernel (LLE_STEP_RECOLOUR_RESETS) unless a cell
        # of the map carries more than two laser layers (then: a launch of lle_batch_reset_sources per step)
        # or a beam is longer than 32 cells (the in-kernel draw is per beam word)
        self._recolour_in_step = self.randomize_lasers and all(m.max_cell_layers <= 2 and m.n_beam_words == m.n_sources for m in self.world.maps)
        self._fused = None  # output tensors + lle_env_outputs of the one-launch step (step(..., fused=True))
        # ... which also writes a partial k x k OBSERVATION itself when nobody reads the layered one, the sources are the map's own and
        # the map has at most 8 beam words (lle_batch_step_outputs: `partial`)
        self._fused_partial = (self._obs_kind[0] == _capi.LLE_OBS_PARTIAL and not self._needs_layered and not self.randomize_lasers
                               and all(m.n_beam_words <= 8 for m in self.world.maps)
                               and self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE, _capi.LLE_OBS_PARTIAL))
        self._bound = {}    # bound calls over persistent buffers (step(..., persistent=True)): BatchedWorld.bound_*

    @staticmethod
    def _kind(name, padding_size):
        if name == "layered-padded":
            return (_capi.LLE_OBS_LAYERED_PADDED, int(padding_size))
        if name == "rgb-image":
            raise NotImplementedError("rendering is outside the scope of lle_amd (SURVEY.md section 2, row 11)")
        try:
            return _OBS_KINDS[name]
        except KeyError:
            raise ValueError(f"Unknown observation type: {name}") from None

    # ------------------------------------------------------------------ construction the reference's way (env.py:222-243, builder.py)
    @staticmethod
    def from_str(world_string):
        """`LLE.from_str(...)`: a Builder; `.build(n_envs)` makes the batch."""
        return Builder(world_string)

    @staticmethod
    def from_file(path):
        from .world import _LEVEL_NAMES
        name = str(path).lower()
        if name in _LEVEL_NAMES:  # (World.from_file takes the standard levels' names: src/core/levels.rs:10-19)
            return Builder(_capi.Map(level=_LEVEL_NAMES[name])).name(f"LLE-{os.path.basename(str(path))}")
        if not os.path.exists(path):
            raise FileNotFoundError(str(path))
        with open(path) as f:
            return Builder(f.read()).name(f"LLE-{os.path.basename(str(path))}")

    @staticmethod
    def level(level):
        """Load a predefined level between 1 and 6 (env.py:238-243)."""
        return Builder(_capi.Map(level=int(level))).name(f"LLE-lvl{level}")

    @property
    def name(self):
        return self._name if self._name is not None else "LLE"  # (marlenv's default is the class name, env.py:116-120)

    # ------------------------------------------------------------------ static description (env.py:72-143)
    @property
    def width(self):
        return self.world.map.width

    @property
    def height(self):
        return self.world.map.height

    @property
    def observation_shape(self):
        """Per-env shape of `get_observation()` (the reference's `observation_shape`, without its tiled agent axis for the
        kinds whose agents all see the same tensor)."""
        return tuple(self._shape_of(self._obs_kind, self.obs_type, state=False))

    @property
    def state_shape(self):
        """LLE.state_shape = get_state().shape (env.py:96): the state generator's observation of agent 0."""
        return tuple(self._shape_of(self._state_kind, self.state_type, state=True))

    def _shape_of(self, kind, name, state):
        k, p = kind
        m, A = self.world.map, self.n_agents
        if k in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE):
            return [3 * A + m.n_gems]
        d = self.world.obs_desc(k, p)
        shape = [int(d.shape[i]) for i in range(1, d.ndim)]
        if state and k in (_capi.LLE_OBS_PARTIAL, _capi.LLE_OBS_PERSPECTIVE):
            shape = shape[1:]  # agent 0's slice
        if name == "flattened":
            n = 1
            for v in shape:
                n *= v
            return [n]
        return shape

    @property
    def agent_state_size(self):
        """StateGenerator.unit_size = 2 (i, j per agent; observations.py:174); other state types have none (env.py:135-140)."""
        if self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE):
            return 2
        raise NotImplementedError(f"State type {self.state_type} does not support `agent_state_size`.")

    # ------------------------------------------------------------------ LLE API, batched
    def seed(self, seed_value):
        """LLE.seed (env.py:245-247): seeds the colour randomisation (v1 maps have a single start per agent)."""
        self._gen.manual_seed(int(seed_value))
        self._seed_value = int(seed_value)
        # the persistent step's bound calls carry the seed of the in-kernel colour draws: bind them again with the new one
        for key in [k for k in self._bound if isinstance(k, tuple) and k[0] == "step"]:
            del self._bound[key]

    @property
    def done(self):
        """bool [n]: LLE.compute_done (env.py:253-254) -- every agent arrived, or somebody died."""
        return self.world.done.view(torch.bool)  # the kernel writes 0 / 1: a view, no launch

    def compute_done(self):
        return self.done

    @property
    def n_arrived(self):
        """int64 [n]: RewardStrategy.n_arrived (reward_strategy.py:22-25,33-39) = agents that have reached an exit this episode."""
        arrived = (self.world.bits >> 16) & 0xFFFF
        return sum((arrived >> a) & 1 for a in range(self.n_agents))

    def reset(self, env_mask=None, seed=None, colours=None):
        """LLE.reset (env.py:189-203) for every env, or those with env_mask != 0: world.reset(), then -- with
        randomize_lasers -- a fresh colour in [0, n_agents) for every source (`colours` u8 [n, L] overrides the draw)."""
        if seed is not None:
            self.seed(seed)
        self._reset_world(env_mask, colours)
        return self.get_observation(), self.get_state()

    def _reset_world(self, env_mask, colours=None, write_obs=True):
        """world.reset() and, with randomize_lasers, the recolouring of the same envs: one launch either way
        (lle_batch_reset or lle_batch_reset_sources)."""
        w = self.world
        if self.randomize_lasers or colours is not None:
            if colours is None:
                colours = torch.randint(0, self.n_agents, (self.n_envs, w.map.n_sources), generator=self._gen,
                                        device=w.device, dtype=torch.uint8)
            w.set_sources(colours=colours, env_mask=env_mask, reset_first=True, write_obs=write_obs)
        else:
            w.reset(env_mask)

    def set_state(self, positions, gems_collected, agents_alive=None):
        """LLE.set_state (env.py:208-217) for every env: World.set_state with the reference's semantics (lossy
        re-derivation of the beams, InvalidWorldState rules); `done` is recomputed from the new state.  positions u8
        [n, A, 2], gems_collected bool [n, G], agents_alive bool [n, A] (default: all alive).  Returns the per-env error
        codes (0, or LLE_ENV_*; such an env keeps / gets the state World.set_state leaves it in)."""
        w = self.world
        if agents_alive is None:
            agents_alive = torch.ones((self.n_envs, self.n_agents), dtype=torch.bool, device=w.device)
        w.set_state(positions, gems_collected, agents_alive)
        return w.err

    def agents_alive(self):
        """bool [n, A]: the `is-alive-k` entries of Step.info (env.py:174-176)."""
        return ((self.world.bits.unsqueeze(1) >> torch.arange(self.n_agents, device=self.world.device)) & 1).bool()

    def agents_arrived(self):
        """bool [n, A]: the `has-arrived-k` entries of Step.info."""
        return ((self.world.bits.unsqueeze(1) >> (16 + torch.arange(self.n_agents, device=self.world.device))) & 1).bool()

    def _observe(self, kind):
        k, p = kind
        if k == _capi.LLE_OBS_LAYERED:
            return self.world.obs  # written by the step / reset / set_sources kernel itself
        return self.world.observe_as(k, p)

    def get_observation(self):
        """The observation of every env with the reference's per-env shape behind the env axis.  Kinds whose agents all
        see the same tensor carry ONE copy (the reference tiles it n_agents times, observations.py:151,266):
        broadcast with `.unsqueeze(1).expand(-1, n_agents, ...)` if the learner wants the tiled layout."""
        obs = self._observe(self._obs_kind)
        return obs.flatten(1) if self.obs_type == "flattened" else obs

    def get_state(self):
        """LLE.get_state (env.py:205-206): the state generator's observation of agent 0."""
        if self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE):
            w = self.world
            st = torch.empty((self.n_envs, 3 * self.n_agents + w.map.n_gems), dtype=torch.float32, device=w.device)
            w.env_outputs(state=st, normalize_state=self._state_kind[0] == _capi.LLE_OBS_NORMALIZED_STATE)
            return st
        st = self._observe(self._state_kind)
        if self._state_kind[0] in (_capi.LLE_OBS_PARTIAL, _capi.LLE_OBS_PERSPECTIVE):
            st = st[:, 0]
        return st.flatten(1) if self.state_type == "flattened" else st

    def available_actions(self):
        """LLE.available_actions (env.py:146-163): bool [n, n_agents, 5]."""
        return self.world.available_actions(self.walkable_lasers)

    def reward(self):
        """Reward of the last step: SingleObjective float32 [n, 1] or MultiObjective float32 [n, 4]
        (reward_strategy.py:58-75, 90-109)."""
        reward = torch.empty((self.n_envs, 4 if self.multi_objective else 1), dtype=torch.float32, device=self.world.device)
        self.world.env_outputs(reward=reward, multi_objective=self.multi_objective)
        return reward

    def _fresh_outputs(self):
        """(tensors, struct) for ONE step of the one-launch path: like _fused_outputs, allocated anew (the struct travels in the
        launch's kernel arguments: 21.04 us per step against 21.02 with persistent tensors)."""
        keep, self._fused = self._fused, None
        try:
            return self._fused_outputs()
        finally:
            self._fused = keep

    def _fused_outputs(self):
        """Persistent output tensors of the one-launch step and the struct over them (lle_batch_step_outputs passes the struct in the kernel arguments)."""
        if self._fused is None:
            w, n, dev = self.world, self.n_envs, self.world.device
            fused_state = self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE)
            t = {"state": torch.empty((n, 3 * self.n_agents + w.map.n_gems), dtype=torch.float32, device=dev) if fused_state else None,
                 "reward": torch.empty((n, 4 if self.multi_objective else 1), dtype=torch.float32, device=dev),
                 "available": torch.empty((n, self.n_agents, 5), dtype=torch.uint8, device=dev), "partial": None, "partial_buf": None}
            kw = {}
            if self._fused_partial:
                # the partial k x k observation written by the step launch itself (lle_batch_step_outputs, step kernel MODE 9)
                t["partial_buf"], t["partial"] = w.partial_buffer(self._obs_kind[1])
                kw = dict(partial=t["partial_buf"], partial_k=self._obs_kind[1])
            o = w.make_env_outputs(state=t["state"], normalize_state=self._state_kind[0] == _capi.LLE_OBS_NORMALIZED_STATE, reward=t["reward"],
                                   multi_objective=self.multi_objective, available=t["available"], walkable_lasers=True, **kw)
            self._fused = (t, o)
        return self._fused

    def step(self, actions, auto_reset=False, fused=None, persistent=False):
        """LLE.step (env.py:165-187) for every env.  actions: integer tensor [n, n_agents] (Action values).
        By default (fused=None) state / reward / available_actions are written by the step kernel itself (lle_batch_step_outputs:
        ONE launch per step, 21.1 us at 65 536 level-6 envs against 25.1 in two) into tensors allocated for this step -- whenever
        that kernel can serve the env (walkable_lasers; the partial observation too where step kernel MODE 9 covers the map);
        otherwise in two launches.  fused=False forces the two launches (step, then lle_batch_env_outputs).
        fused=True (needs walkable_lasers): the one launch into PERSISTENT tensors that the next step overwrites (no allocation
        per step).
        persistent=True: the two-launch step (any walkable_lasers, any observation / state type) through calls bound once
        (BatchedWorld.bound_*) into persistent tensors that the next step overwrites: the host side of a step is two C-ABI
        calls (three with an observation type other than layered) instead of allocations, descriptor queries and views.
        The reference refuses to step a finished environment (`Cannot step in a done environment`); here such an env
        is the caller's to reset -- or pass auto_reset=True: an env that is done when the step starts is reset first
        (with fresh colours under randomize_lasers), the usual vector-env convention.
        The step kernel's layered observation (`world.obs`) is only refreshed when obs_type or state_type is layered.
        Returns a dict of device tensors: obs, state, reward, done, available_actions, err (per-env error code of
        World.step: 0 or 1 + the agent whose action was not available, the env then being left untouched)."""
        w = self.world
        if actions.dtype is not torch.uint8 or actions.device != w.device or not actions.is_contiguous():
            actions = actions.to(w.device, torch.uint8).contiguous()
        if fused and not self.walkable_lasers:
            raise ValueError("the one-launch step writes available_actions with walkable_lasers only")
        if persistent and not fused:
            return self._step_persistent(actions, auto_reset)
        fresh = None
        if fused is None:  # the default: one launch where the step kernel can write the outputs, into this step's own tensors
            fused = False
            if self.walkable_lasers:
                fresh = self._fresh_outputs()
        env_out = fresh[1] if fresh is not None else (self._fused_outputs()[1] if fused else None)
        if auto_reset:
            if self._recolour_in_step:
                # world.reset() + a fresh colour per source for the envs that are over, inside the step kernel; the draws
                # are keyed by (seed, env, step counter, source), not by the torch generator that reset() uses
                w.step(actions, auto_reset=True, recolour_resets=True, seed=self._seed_value, t=self._t, write_obs=self._needs_layered,
                       env_out=env_out, incremental_obs=self._incr)
            elif self.randomize_lasers:
                # (the kernel reads an env's mask byte before it rewrites its `done`; the step rewrites the observation)
                self._reset_world(w.done, write_obs=False)
                w.step(actions, write_obs=self._needs_layered, env_out=env_out, incremental_obs=self._incr)
            else:
                w.step(actions, auto_reset=True, write_obs=self._needs_layered, env_out=env_out, incremental_obs=self._incr)
        else:
            w.step(actions, write_obs=self._needs_layered, env_out=env_out, incremental_obs=self._incr)
        self._t += 1
        if fused or fresh is not None:
            t = fresh[0] if fresh is not None else self._fused[0]
            if t["partial"] is not None:  # (written by the step launch: no observer launch behind it)
                obs = t["partial"]
                state = t["state"] if t["state"] is not None else (obs[:, 0] if self._state_kind == self._obs_kind else self.get_state())
                if t["state"] is None and self.state_type == "flattened":
                    state = state.flatten(1)
            else:
                obs, state = self.get_observation(), (t["state"] if t["state"] is not None else self.get_state())
            return {"obs": obs, "state": state, "reward": t["reward"],
                    "done": self.done, "available_actions": t["available"].view(torch.bool), "err": w.err}
        return self._outputs()

    def _step_persistent(self, actions, auto_reset):
        """step(persistent=True): see step()."""
        w, b = self.world, self._bound
        key = ("step", bool(auto_reset))
        if key not in b:
            n, dev = self.n_envs, w.device
            plain_state = self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE)
            if "outs" not in b:
                t = {"state": torch.empty((n, 3 * self.n_agents + w.map.n_gems), dtype=torch.float32, device=dev) if plain_state else None,
                     "reward": torch.empty((n, 4 if self.multi_objective else 1), dtype=torch.float32, device=dev),
                     "available": torch.empty((n, self.n_agents, 5), dtype=torch.uint8, device=dev)}
                b["outs"] = (t, w.bound_env_outputs(state=t["state"], normalize_state=self._state_kind[0] == _capi.LLE_OBS_NORMALIZED_STATE,
                                                    reward=t["reward"], multi_objective=self.multi_objective, available=t["available"],
                                                    walkable_lasers=self.walkable_lasers))
                b["obs"] = None if self._obs_kind[0] == _capi.LLE_OBS_LAYERED else w.bound_observer(*self._obs_kind)
                b["state"] = None if plain_state or self._state_kind[0] == _capi.LLE_OBS_LAYERED else (
                    b["obs"] if self._state_kind == self._obs_kind and b["obs"] is not None else w.bound_observer(*self._state_kind))
            recolour = auto_reset and self._recolour_in_step
            in_kernel_reset = auto_reset and (recolour or not self.randomize_lasers)
            b[key] = w.bound_step(auto_reset=in_kernel_reset, recolour_resets=recolour, write_obs=self._needs_layered, seed=self._seed_value,
                                  incremental_obs=self._incr)
        if auto_reset and self.randomize_lasers and not self._recolour_in_step:
            self._reset_world(w.done, write_obs=False)
        w.t = self._t
        b[key](actions)
        self._t += 1
        t, outs = b["outs"]
        outs()
        obs = w.obs if b["obs"] is None else b["obs"]()
        if self.obs_type == "flattened":
            obs = obs.flatten(1)
        if t["state"] is not None:
            state = t["state"]
        elif b["state"] is None:
            state = w.obs
        else:
            state = b["state"].out if b["state"] is b["obs"] else b["state"]()
            if self._state_kind[0] in (_capi.LLE_OBS_PARTIAL, _capi.LLE_OBS_PERSPECTIVE):
                state = state[:, 0]
        if t["state"] is None and self.state_type == "flattened":
            state = state.flatten(1)
        return {"obs": obs, "state": state, "reward": t["reward"], "done": self.done, "available_actions": t["available"].view(torch.bool),
                "err": w.err}

    def _outputs(self):
        """obs / state / reward / done / available_actions / err after a step: one launch of lle_batch_env_outputs for
        the small tensors (the layered observation was written by the step kernel itself); only a state type other than
        "state" / "normalized-state" or an observation type other than layered costs a further observer launch.
        The tensors are freshly allocated except `obs` (layered), `done` and `err`, which view the world's buffers."""
        w, n, dev = self.world, self.n_envs, self.world.device
        fused_state = self._state_kind[0] in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE)
        state = torch.empty((n, 3 * self.n_agents + w.map.n_gems), dtype=torch.float32, device=dev) if fused_state else None
        reward = torch.empty((n, 4 if self.multi_objective else 1), dtype=torch.float32, device=dev)
        avail = torch.empty((n, self.n_agents, 5), dtype=torch.uint8, device=dev)
        w.env_outputs(state=state, normalize_state=self._state_kind[0] == _capi.LLE_OBS_NORMALIZED_STATE, reward=reward,
                      multi_objective=self.multi_objective, available=avail, walkable_lasers=self.walkable_lasers)
        return {"obs": self.get_observation(), "state": state if fused_state else self.get_state(), "reward": reward,
                "done": self.done, "available_actions": avail.view(torch.bool), "err": w.err}


class Builder:
    """`lle.level(6).obs_type("layered").randomize_lasers().build()` of the reference (python/lle/env/builder.py:12-166), for the
    batch: the same chain, and `build(n_envs)` returns a BatchedLLE.  `pbrs` and `add_extras` belong to the parts of `LLE` that are
    outside this package (reward shaping and extras generators) and say so."""

    def __init__(self, map_or_text):
        self._map = map_or_text
        self._obs_type, self._state_type = "layered", "state"
        self._death_strategy, self._walkable_lasers = "end", True
        self._env_name, self._multi_objective, self._randomize_lasers = "LLE", False, False
        self._padding_size = 0

    def obs_type(self, obs_type):
        from .observations import ObservationType
        self._obs_type = ObservationType.from_str(obs_type).value if isinstance(obs_type, str) else obs_type.value
        return self

    def state_type(self, state_type):
        from .observations import ObservationType
        self._state_type = ObservationType.from_str(state_type).value if isinstance(state_type, str) else state_type.value
        return self

    def walkable_lasers(self, walkable_lasers):
        self._walkable_lasers = bool(walkable_lasers)
        return self

    def death_strategy(self, death_strategy):
        self._death_strategy = death_strategy
        return self

    def name(self, name):
        self._env_name = name
        return self

    def multi_objective(self):
        if not self._multi_objective:
            self._multi_objective = True
            self._env_name = f"{self._env_name}-MO"  # builder.py:74-76
        return self

    def randomize_lasers(self):
        self._randomize_lasers = True
        return self

    def pbrs(self, *args, **kwargs):
        raise NotImplementedError("potential-based reward shaping is outside the scope of lle_amd (SURVEY.md section 2, row 9)")

    def add_extras(self, *extras):
        if not extras:
            return self
        raise NotImplementedError("extras generators are outside the scope of lle_amd (SURVEY.md section 2, row 9)")

    def build(self, n_envs=1, device=None, seed=0, obs_dtype=None):
        """(n_envs, device, seed, obs_dtype: what a batch needs beyond the reference's builder -- BatchedLLE's arguments of the same names)"""
        return BatchedLLE(self._map, n_envs, obs_type=self._obs_type, state_type=self._state_type, walkable_lasers=self._walkable_lasers,
                          randomize_lasers=self._randomize_lasers, multi_objective=self._multi_objective, death_strategy=self._death_strategy,
                          padding_size=self._padding_size, device=device, seed=seed, name=self._env_name, obs_dtype=obs_dtype)


def level(level):
    """`lle.level(n)` (python/lle/__init__.py)."""
    return BatchedLLE.level(level)


def from_str(world_string):
    return BatchedLLE.from_str(world_string)


def from_file(path):
    return BatchedLLE.from_file(path)
