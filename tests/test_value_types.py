"""The reference's value-type tests replayed on the facade's types (no GPU needed): python/tests/test_actions.py (all of
it), python/tests/test_world.py:606-640 (WorldState subclassing / constructor), python/tests/test_serialization.py:9-16.
The types are API surface of the drop-in (`lle_amd.Action`, `lle_amd.WorldState`), not the hot path; the reference
file:line of every assertion is given."""
import copy
import pickle

import pytest

from lle_amd import Action, WorldState


def test_action_equality_count_names_hash():   # test_actions.py:4-38
    assert Action.NORTH == Action.NORTH
    assert Action(0) == Action(0)
    values = [Action.NORTH, Action.SOUTH, Action.EAST, Action.WEST, Action.WEST]
    assert [values.count(a) for a in (Action.NORTH, Action.SOUTH, Action.EAST, Action.WEST)] == [1, 1, 1, 2]
    assert {a.name for a in Action.variants()} == {"NORTH", "SOUTH", "EAST", "WEST", "STAY"}
    hashes, actions = set(), set()
    for a in Action.variants():
        assert hash(a) == hash(a)
        hashes.add(hash(a))
        actions.add(a)
    assert len(hashes) == len(actions) == 5


def test_action_deepcopy_and_pickle_give_new_equal_objects():   # test_actions.py:41-55
    for a in Action.variants():
        assert copy.deepcopy(a) == a
        b = pickle.loads(pickle.dumps(a))
        assert a == b
        assert a is not b


def test_action_from_delta_uses_dx_dy():   # test_actions.py:58-110 (quirk Q8: (dx, dy), not (di, dj))
    assert Action.from_delta(0, 0) == Action.STAY
    assert Action.from_delta(0, -1) == Action.NORTH
    assert Action.from_delta(0, 1) == Action.SOUTH
    assert Action.from_delta(1, 0) == Action.EAST
    assert Action.from_delta(-1, 0) == Action.WEST
    for delta in [(2, 0), (-2, 0), (0, 2), (0, -2), (1, 1), (-1, 1), (1, -1), (-1, -1), (5, 3)]:
        with pytest.raises(ValueError):
            Action.from_delta(*delta)
    assert Action.NORTH.delta == (-1, 0) and Action.EAST.delta == (0, 1)   # Action.delta is (di, dj): src/action.rs:18-26


def test_action_values_and_invalid_values():   # pyaction.rs:13-25,40-52
    assert [a.value for a in Action.variants()] == [0, 1, 2, 3, 4] and Action.cardinality() == 5
    for bad in (5, -1, 23):
        with pytest.raises(ValueError):
            Action(bad)
    assert Action.NORTH.opposite() == Action.SOUTH and Action.STAY.opposite() == Action.STAY


def test_subclass_world_state():   # test_world.py:606-623
    class WS(WorldState):
        def __init__(self, other, agents_positions, gems_collected, agents_alive=None):
            super().__init__(agents_positions, gems_collected=gems_collected, agents_alive=agents_alive)
            self.other = other

    s1, s2 = WS(4, [(0, 0)], [False], [True]), WS(5, [(0, 0)], [False], [True])
    assert s1 == s2


def test_world_state_constructor():   # test_world.py:633-640
    assert all(WorldState([(0, 0)], [False]).agents_alive)
    assert all(WorldState([(0, 0)], [True], [True]).agents_alive)
    assert WorldState([(0, 0), (1, 1)], [True], [False, True]).agents_alive == [False, True]


def test_subclass_world_and_standard_levels_parse_without_a_gpu():   # test_world.py:626-630, :447-451; test_world.rs:438-453
    from lle_amd import World

    class W(World):
        pass

    assert W("S0 . X").width == 3
    for i in range(1, 7):
        a, b, c = World.level(i), World.from_file(f"lvl{i}"), World.from_file(f"level{i}")
        assert a.world_string == b.world_string == c.world_string and a.n_agents == b.n_agents
    with pytest.raises(FileNotFoundError):
        World.from_file("no/such/file.toml")


def test_direction():   # python/tests/test_direction.py (all of it)
    from lle_amd import Direction

    ds = [Direction.NORTH, Direction.SOUTH, Direction.EAST, Direction.WEST]
    for d in ds:
        assert d == d
        for d2 in ds:
            if d is not d2:
                assert d != d2
    assert Direction("N") == Direction.NORTH and Direction("W") == Direction.WEST
    with pytest.raises(ValueError):
        Direction("z")
    assert [d.delta for d in ds] == [(-1, 0), (1, 0), (0, 1), (0, -1)]
    assert [d.opposite() for d in ds] == [Direction.SOUTH, Direction.NORTH, Direction.WEST, Direction.EAST]
    # the rest of the stub (python/lle/tiles/__init__.pyi:150-200, src/bindings/tiles/pydirection.rs:82-165)
    assert [d.name for d in ds] == ["N", "S", "E", "W"] and [repr(d) for d in ds] == ["N", "S", "E", "W"]
    assert [d.is_horizontal for d in ds] == [False, False, True, True] and [d.is_vertical for d in ds] == [True, True, False, False]
    assert [Direction.from_str(c) for c in "NSEW"] == ds
    with pytest.raises(ValueError, match="Invalid direction string."):
        Direction.from_str("north")
    assert [hash(d) for d in (Direction.NORTH, Direction.EAST, Direction.SOUTH, Direction.WEST)] == [0, 1, 2, 3]
    import copy
    import pickle
    assert all(pickle.loads(pickle.dumps(d)) is d and copy.deepcopy(d) is d for d in ds)


def test_import_paths():   # python/tests/test_imports.py:1-27,35-39 (solver / rendering / characterization paths: out of scope)
    import lle_amd
    from lle_amd import __version__, exceptions, tiles, world  # noqa: F401
    from lle_amd.exceptions import InvalidActionError, InvalidLevelError, InvalidWorldStateError, ParsingError
    from lle_amd.tiles import Direction, Gem, Laser, LaserSource
    from lle_amd.types import AgentId, LaserId, Position
    from lle_amd.world import Action, EventType, World, WorldEvent, WorldState  # noqa: F401

    assert isinstance(__version__, str) and lle_amd.__version__ == __version__
    assert (tiles.Gem, tiles.Laser, tiles.LaserSource, tiles.Direction) == (Gem, Laser, LaserSource, Direction)
    assert all(issubclass(e, ValueError) for e in (InvalidActionError, InvalidLevelError, InvalidWorldStateError, ParsingError))
    assert (AgentId, LaserId) == (int, int) and Position == tuple[int, int]
    assert lle_amd.Gem is Gem and lle_amd.ParsingError is ParsingError


def test_typing_observation_type_literal():   # python/tests/test_observations.py:8-17
    from typing import get_args

    from lle_amd.observations import ObservationType, ObservationTypeLiteral

    for name in get_args(ObservationTypeLiteral):
        assert any(name == o for o in ObservationType), f"{name} is not a valid ObservationType"
    for o in ObservationType:
        assert o in get_args(ObservationTypeLiteral)
