"""Layered observation generator with the reference's interface (python/lle/observations.py:196-279).

The tensor itself is produced on the GPU by the step kernel (int8, one (C,H,W) slice per environment); this class
only adapts it to the reference's `ObservationGenerator` protocol: `observe()` returns float32 (A, C, H, W)."""
import numpy as np


class LayeredPadded:
    def __init__(self, world, padding_size=0):
        if padding_size != 0:
            raise NotImplementedError("agent padding is not on the accelerated path (SURVEY.md section 8(f), rank 3)")
        self._world = world
        self.width, self.height = world.width, world.height
        self.n_agents = world.n_agents
        self.A0 = 0
        self.LASER_0 = self.A0 + self.n_agents
        self.WALL = self.LASER_0 + self.n_agents
        self.VOID = self.WALL + 1
        self.GEM = self.VOID + 1
        self.EXIT = self.GEM + 1
        self._shape = (self.EXIT + 1, world.height, world.width)

    @property
    def shape(self):
        return self._shape

    def reset(self):
        """Static layers live in the device tables and are refreshed by World itself; nothing to cache here."""

    def observe(self):
        obs = self._world.layered_observation().astype(np.float32)
        return np.tile(obs, (self.n_agents, 1, 1, 1))

    def get_state(self):
        return self.observe()[0]


class Layered(LayeredPadded):
    def __init__(self, world):
        super().__init__(world, padding_size=0)
