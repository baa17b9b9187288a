"""The reference's observation generators (python/lle/observations.py) over the GPU observer kernels.

Every tensor comes from a HIP kernel reading the world's device state (layered: the step / observe kernel; the other
kinds: lle_amd/csrc/observers.hip behind `lle_batch_observe_as`).  The classes below only give those tensors the
reference's protocol -- class names, the channel attributes (`A0`, `LASER_0`, `WALL`, ...), `shape`, `obs_type`,
`observe()` as float32 with the (n_agents, ...) leading axis, `to_world_state` -- so that code written against
`ObservationType(...).get_observation_generator(world)` runs unchanged.  "rgb-image" (rendering) is out of scope
(SURVEY.md section 2 row 11).
"""
from enum import Enum
from typing import Literal

import numpy as np

from . import _capi
from .world import WorldState


# the preset names as a typing Literal (python/lle/observations.py:21-35; pinned by python/tests/test_observations.py:8-17)
ObservationTypeLiteral = Literal["layered", "flattened", "partial3x3", "partial5x5", "partial7x7", "state", "rgb-image", "perspective",
                                 "normalized-state", "layered-padded-1", "layered-padded-2", "layered-padded-3", "layered-padded"]


class ObservationType(str, Enum):
    """The presets of python/lle/observations.py:38-60 (same names, same string values)."""

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
        """Same dispatch as python/lle/observations.py:66-97, as a table."""
        if self is ObservationType.RGB_IMAGE:
            raise NotImplementedError("rendering is outside the scope of lle_amd (SURVEY.md section 2, row 11)")
        make = {
            "normalized-state": lambda: StateGenerator(world, normalize=True),
            "state": lambda: StateGenerator(world, normalize=False),
            "layered": lambda: Layered(world),
            "flattened": lambda: FlattenedLayered(world),
            "partial3x3": lambda: PartialGenerator(world, 3),
            "partial5x5": lambda: PartialGenerator(world, 5),
            "partial7x7": lambda: PartialGenerator(world, 7),
            "layered-padded": lambda: LayeredPadded(world, padding_size),
            "layered-padded-1": lambda: LayeredPadded(world, 1),
            "layered-padded-2": lambda: LayeredPadded(world, 2),
            "layered-padded-3": lambda: LayeredPadded(world, 3),
            "perspective": lambda: AgentZeroPerspective(world),
        }
        return make[self.value]()


class ObservationGenerator:
    """Protocol of python/lle/observations.py:100-143.  `_obs_type` / `_kind` / `_param` are set by the subclasses;
    `_tile` says whether the reference repeats the tensor once per agent (np.tile) or already returns one per agent."""

    _obs_type = None
    _kind = None
    _param = 0
    _tile = True

    def __init__(self, world):
        self._world = world

    @property
    def obs_type(self):
        return self._obs_type

    @property
    def shape(self):
        return self._shape

    def _device_tensor(self):
        """One copy of the observation, from the GPU (IndexError where the reference raises it)."""
        return self._world.observation(self._kind, self._param)

    def observe(self):
        # float32 like the reference.  The layered tensor arrives in that type from the kernels (World.layered_observation: no cast
        # here, `copy=False` hands it through); the observers' int8 outputs (padded / perspective / partial) are cast once
        single = self._device_tensor().astype(np.float32, copy=False)
        if not self._tile:
            return single
        reps = getattr(self, "n_agents", self._world.n_agents)
        return np.tile(single, (reps,) + (1,) * single.ndim)

    def get_state(self):
        return self.observe()[0]

    def to_world_state(self, data):
        raise NotImplementedError(f"This method is not implemented for {self.__class__.__name__}")

    def set_world(self, new_world):
        self._world = new_world

    def reset(self):
        """The static layers live in the device tables and follow the world's sources; nothing is cached on the host."""


class StateGenerator(ObservationGenerator):
    """[i0, j0, ..., gems, alive], optionally divided by (height, width): python/lle/observations.py:137-175."""

    _obs_type = ObservationType.STATE

    def __init__(self, world, normalize):
        super().__init__(world)
        self.n_gems, self.n_agents, self.normalize = world.n_gems, world.n_agents, normalize
        self._kind = _capi.LLE_OBS_NORMALIZED_STATE if normalize else _capi.LLE_OBS_STATE
        self.dimensions = np.array(([world.height, world.width] if normalize else [1.0, 1.0]) * world.n_agents)
        self._shape = (world.n_agents * 3 + world.n_gems,)
        self.unit_size = 2

    def to_world_state(self, data):
        k = 2 * self._world.n_agents
        data[:k] = data[:k] * self.dimensions
        return WorldState.from_array(data.tolist(), self.n_agents, self.n_gems)


class LayeredPadded(ObservationGenerator):
    """(2(A+p)+4, H, W) channels: agents, laser colours, WALL, VOID, GEM, EXIT (python/lle/observations.py:196-271)."""

    _obs_type = ObservationType.LAYERED
    _kind = _capi.LLE_OBS_LAYERED_PADDED

    def __init__(self, world, padding_size=0):
        super().__init__(world)
        self._param = self.padding_size = int(padding_size)
        self.width, self.height = world.width, world.height
        n = self.n_agents = world.n_agents + self.padding_size
        self.A0, self.LASER_0, self.WALL, self.VOID, self.GEM, self.EXIT = 0, n, 2 * n, 2 * n + 1, 2 * n + 2, 2 * n + 3
        self._shape = (2 * n + 4, world.height, world.width)
        self.ordered_gem_pos = sorted(g.pos for g in world.gems)

    def _device_tensor(self):
        if self.padding_size == 0:
            return self._world.layered_observation()  # the tensor the step kernel keeps up to date
        return super()._device_tensor()

    def to_world_state(self, data):
        """observations.py:243-252 (assumes that every agent is alive)."""
        _, rows, cols = np.nonzero(data[self.A0: self.A0 + self.n_agents])
        positions = [(int(rows[k]), int(cols[k])) for k in range(self.n_agents)]
        return WorldState(positions, [bool(data[self.GEM, i, j] == 0.0) for i, j in self.ordered_gem_pos])


class Layered(LayeredPadded):
    def __init__(self, world):
        super().__init__(world, padding_size=0)


class FlattenedLayered(ObservationGenerator):
    """The layered tensor as one row per agent (python/lle/observations.py:279-303)."""

    _obs_type = ObservationType.FLATTENED

    def __init__(self, world):
        super().__init__(world)
        self.layered = Layered(world)
        self._shape = (int(np.prod(self.layered.shape)),)
        self.unit_size = 0

    def observe(self):
        return self.layered.observe().reshape(self._world.n_agents, -1)

    def set_world(self, new_world):
        self.layered.set_world(new_world)
        super().set_world(new_world)


class PartialGenerator(ObservationGenerator):
    """(A, 2A+3, k, k): a k x k window around every agent (python/lle/observations.py:306-369)."""

    _obs_type = ObservationType.PARTIAL_3x3
    _kind = _capi.LLE_OBS_PARTIAL
    _tile = False

    def __init__(self, world, square_size):
        super().__init__(world)
        assert square_size % 2 == 1, "Can only use odd numbers for the square size"
        a = world.n_agents
        self._param = self.size = square_size
        self._center = square_size // 2
        self.WALL, self.LASER_0, self.GEM, self.EXIT = a, a + 1, 2 * a + 1, 2 * a + 2
        self._shape = (2 * a + 3, square_size, square_size)


class AgentZeroPerspective(Layered):
    """Observer k sees itself and its laser colour in channels 0 / LASER_0 (python/lle/observations.py:372-395)."""

    _obs_type = ObservationType.AGENT0_PERSPECTIVE_LAYERED
    _kind = _capi.LLE_OBS_PERSPECTIVE
    _tile = False

    def _device_tensor(self):
        return self._world.observation(self._kind, 0)
