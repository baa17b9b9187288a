"""lle_amd -- batched, MI355X-native `World.step()` for the Laser Learning Environment.

Drop-in for the hot path of yamoling/lle: `World` (single environment, reference API) and `BatchedWorld`
(tens of thousands of lock-stepped environments per kernel launch, torch tensors over device buffers).
The compute path is hand-written HIP for gfx950 behind the C ABI of include/lle_hip.h; there is no CPU fallback.
"""
from . import exceptions, tiles, types, world  # the reference's submodule paths: lle.tiles, lle.exceptions, lle.types, lle.world
from ._capi import Map, MapParseError
from .world import (Action, Agent, Direction, EventType, Gem, InvalidActionError, InvalidLevelError, InvalidWorldStateError,
                    Laser, LaserSource, ParsingError, World, WorldEvent, WorldState)

__version__ = "0.3.0"  # (round 3 of this build; the reference exposes lle.__version__: python/tests/test_imports.py:35-39)


def __getattr__(name):
    # BatchedWorld imports torch; keep `import lle_amd` light for parse-only users
    if name == "BatchedWorld":
        from .batched import BatchedWorld
        return BatchedWorld
    if name in ("BatchedLLE", "Builder", "DeathStrategy", "level", "from_str", "from_file"):  # (lle.level(6).obs_type(...).build())
        from . import env
        return getattr(env, name)
    if name in ("Layered", "LayeredPadded", "ObservationType", "StateGenerator", "FlattenedLayered", "PartialGenerator",
                "AgentZeroPerspective"):
        from . import observations
        return getattr(observations, name)
    raise AttributeError(name)


__all__ = ["Action", "Agent", "AgentZeroPerspective", "BatchedLLE", "BatchedWorld", "Direction", "EventType", "FlattenedLayered", "Gem", "InvalidActionError", "InvalidLevelError",
           "InvalidWorldStateError", "Laser", "LaserSource", "Layered", "LayeredPadded", "Map", "MapParseError",
           "ObservationType", "ParsingError", "PartialGenerator", "StateGenerator", "World", "WorldEvent", "WorldState", "__version__", "exceptions", "tiles", "types", "world"]
