"""Observation generators with the reference's interface (python/lle/observations.py), computed on the GPU.

Every tensor is produced by a HIP kernel from the world's device state (layered: the step kernel / observe kernel;
the others: lle_amd/csrc/observers.hip behind `lle_batch_observe_as`); the classes here only adapt it to the
reference's `ObservationGenerator` protocol -- same class names, attributes (`A0`, `LASER_0`, `WALL`, ...), `shape`,
`observe()` returning float32 with the reference's (n_agents, ...) leading axis, `to_world_state`.
"rgb-image" is out of scope (rendering).
"""
from enum import Enum

import numpy as np

from . import _capi
from .world import WorldState


class ObservationType(str, Enum):
    """python/lle/observations.py:38-60"""

    NORMALIZED_STATE = "normalized-state"
    STATE = "state"
    RGB_IMAGE = "rgb-image"
    LAYERED = "layered"
    FLATTENED = "flattened"
    PARTIAL_3x3 = "partial3x3"
    PARTIAL_5x5 = "partial5x5"
    PARTIAL_7x7 = "partial7x7"
    LAYERED_PADDED = "layered-padded"
    LAYERED_PADDED_1AGENT = "layered-padded-1"
    LAYERED_PADDED_2AGENTS = "layered-padded-2"
    LAYERED_PADDED_3AGENTS = "layered-padded-3"
    AGENT0_PERSPECTIVE_LAYERED = "perspective"

    @staticmethod
    def from_str(s):
        return ObservationType(s)

    def get_observation_generator(self, world, padding_size=0):
        """python/lle/observations.py:66-97"""
        T = ObservationType
        if self is T.NORMALIZED_STATE:
            return StateGenerator(world, normalize=True)
        if self is T.STATE:
            return StateGenerator(world, normalize=False)
        if self is T.RGB_IMAGE:
            raise NotImplementedError("rendering is outside the scope of lle_amd (SURVEY.md section 2, row 11)")
        if self is T.LAYERED:
            return Layered(world)
        if self is T.FLATTENED:
            return FlattenedLayered(world)
        if self in (T.PARTIAL_3x3, T.PARTIAL_5x5, T.PARTIAL_7x7):
            return PartialGenerator(world, {T.PARTIAL_3x3: 3, T.PARTIAL_5x5: 5, T.PARTIAL_7x7: 7}[self])
        if self is T.LAYERED_PADDED:
            return LayeredPadded(world, padding_size)
        if self in (T.LAYERED_PADDED_1AGENT, T.LAYERED_PADDED_2AGENTS, T.LAYERED_PADDED_3AGENTS):
            return LayeredPadded(world, {T.LAYERED_PADDED_1AGENT: 1, T.LAYERED_PADDED_2AGENTS: 2, T.LAYERED_PADDED_3AGENTS: 3}[self])
        if self is T.AGENT0_PERSPECTIVE_LAYERED:
            return AgentZeroPerspective(world)
        raise ValueError(f"Unknown observation type: {self}")


class ObservationGenerator:
    """python/lle/observations.py:100-143"""

    def __init__(self, world):
        self._world = world

    def observe(self):
        raise NotImplementedError

    def get_state(self):
        return self.observe()[0]

    def to_world_state(self, data):
        raise NotImplementedError(f"This method is not implemented for {self.__class__.__name__}")

    def set_world(self, new_world):
        self._world = new_world

    def reset(self):
        """Static layers live in the device tables and follow the world's sources by themselves; nothing is cached here."""


class StateGenerator(ObservationGenerator):
    """python/lle/observations.py:137-175"""

    def __init__(self, world, normalize):
        super().__init__(world)
        self.n_gems = world.n_gems
        self.n_agents = world.n_agents
        self.normalize = normalize
        if normalize:
            self.dimensions = np.array([world.height, world.width] * world.n_agents)
        else:
            self.dimensions = np.array([1.0, 1.0] * world.n_agents)

    def observe(self):
        kind = _capi.LLE_OBS_NORMALIZED_STATE if self.normalize else _capi.LLE_OBS_STATE
        state = self._world.observation(kind)
        return np.tile(state, reps=(self._world.n_agents, 1))

    def to_world_state(self, data):
        data[: self._world.n_agents * 2] = data[: self._world.n_agents * 2] * self.dimensions
        return WorldState.from_array(data.tolist(), self.n_agents, self.n_gems)

    @property
    def obs_type(self):
        return ObservationType.STATE

    @property
    def shape(self):
        return (self._world.n_agents * 3 + self.n_gems,)

    @property
    def unit_size(self):
        return 2


class LayeredPadded(ObservationGenerator):
    """python/lle/observations.py:196-271"""

    def __init__(self, world, padding_size=0):
        super().__init__(world)
        self.padding_size = padding_size
        self.width, self.height = world.width, world.height
        self.n_agents = world.n_agents + padding_size
        self.A0 = 0
        self.LASER_0 = self.A0 + self.n_agents
        self.WALL = self.LASER_0 + self.n_agents
        self.VOID = self.WALL + 1
        self.GEM = self.VOID + 1
        self.EXIT = self.GEM + 1
        self._shape = (self.EXIT + 1, world.height, world.width)
        self.ordered_gem_pos = sorted(gem.pos for gem in world.gems) if hasattr(world, "gems") else []

    @property
    def shape(self):
        return self._shape

    @property
    def obs_type(self):
        return ObservationType.LAYERED

    def _single(self):
        if self.padding_size == 0:
            return self._world.layered_observation()
        return self._world.observation(_capi.LLE_OBS_LAYERED_PADDED, self.padding_size)

    def observe(self):
        obs = self._single().astype(np.float32)
        return np.tile(obs, (self.n_agents, 1, 1, 1))

    def to_world_state(self, data):
        """python/lle/observations.py:243-252 (assumes every agent alive)"""
        _, i, j = np.nonzero(data[self.A0: self.A0 + self.n_agents])
        agents_positions = [(int(i[n]), int(j[n])) for n in range(self.n_agents)]
        gems_collected = [bool(data[self.GEM, gi, gj] == 0.0) for gi, gj in self.ordered_gem_pos]
        return WorldState(agents_positions, gems_collected)


class Layered(LayeredPadded):
    def __init__(self, world):
        super().__init__(world, padding_size=0)


class FlattenedLayered(ObservationGenerator):
    """python/lle/observations.py:279-303"""

    def __init__(self, world):
        super().__init__(world)
        self.layered = Layered(world)
        size = 1
        for s in self.layered.shape:
            size = size * s
        self._shape = (size,)

    def observe(self):
        return self.layered.observe().reshape(self._world.n_agents, -1)

    @property
    def obs_type(self):
        return ObservationType.FLATTENED

    @property
    def shape(self):
        return self._shape

    @property
    def unit_size(self):
        return 0

    def set_world(self, new_world):
        self.layered.set_world(new_world)
        return super().set_world(new_world)


class PartialGenerator(ObservationGenerator):
    """python/lle/observations.py:306-369"""

    def __init__(self, world, square_size):
        super().__init__(world)
        assert square_size % 2 == 1, "Can only use odd numbers for the square size"
        self.size = square_size
        self._shape = (world.n_agents + world.n_agents + 3, self.size, self.size)
        self._center = self.size // 2
        self.WALL = world.n_agents
        self.LASER_0 = self.WALL + 1
        self.GEM = self.LASER_0 + world.n_agents
        self.EXIT = self.GEM + 1

    @property
    def shape(self):
        return self._shape

    @property
    def obs_type(self):
        return ObservationType.PARTIAL_3x3

    def observe(self):
        return self._world.observation(_capi.LLE_OBS_PARTIAL, self.size).astype(np.float32)


class AgentZeroPerspective(Layered):
    """python/lle/observations.py:372-395"""

    @property
    def obs_type(self):
        return ObservationType.AGENT0_PERSPECTIVE_LAYERED

    def observe(self):
        return self._world.observation(_capi.LLE_OBS_PERSPECTIVE).astype(np.float32)
