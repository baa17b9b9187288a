"""`World`: drop-in for the reference's `lle.world.World` (src/bindings/world/pyworld.rs:42-76, API spec
python/lle/world/__init__.pyi:23-340), backed by a one-environment batch on the GPU.

Same method / property names, argument meaning and exception classes as the reference, so that reference-style
tests read unchanged.  Parsing and every static property work without a GPU (host map compiler); anything that
touches the dynamic state needs an MI355X and raises RuntimeError otherwise -- there is no CPU fallback.
For throughput use `lle_amd.BatchedWorld`; this facade pays one kernel launch + one small device->host copy per call.
"""
import copy
import itertools
import os
from enum import Enum

import numpy as np

from . import _capi, _decode
from ._capi import LLE_POS_EXIT, LLE_POS_GEM, LLE_POS_START, LLE_POS_VOID, LLE_POS_WALL, Map, MapParseError


# ------------------------------------------------------------------------------------------------ exceptions
class InvalidWorldStateError(ValueError):
    """Raised when the state of the world is invalid.  (src/bindings/pyexceptions.rs:8-13)"""


class InvalidActionError(ValueError):
    """Raised when the action taken by an agent is invalid.  (pyexceptions.rs:15-20)"""


class ParsingError(ValueError):
    """Raised when there is a problem while parsing a world string.  (pyexceptions.rs:22-27)"""


class InvalidLevelError(ValueError):
    """Raised when the level asked does not exist.  (pyexceptions.rs:29-34)"""


_PARSE_MESSAGES = {
    "EmptyWorld": "Empty world: no tiles", "NoAgents": "No agents in the world",
    "TomlUnsupported": "TOML (v2) world descriptions are not supported by lle_amd (v1 text maps only)",
    "Limit": "The map exceeds a static limit of lle_amd (agents<=16, sources<=32, gems<=32, side<=255, 32 beam words of 32 cells over all beams)",
}


def _parse_error(e):
    err = ParsingError(_PARSE_MESSAGES.get(e.kind, e.kind))
    err.kind = e.kind
    return err


# ------------------------------------------------------------------------------------------------ value types
class Action:
    """src/bindings/world/pyaction.rs:13-88.  A value class like the PyO3 one, not a Python Enum: `Action(0) == Action.NORTH`,
    equal hashes, but pickling / deep-copying gives a NEW equal object (`python/tests/test_actions.py:49-55` asserts
    `a == b and a is not b`)."""
    __slots__ = ("_value",)
    _NAMES = ("NORTH", "SOUTH", "EAST", "WEST", "STAY")

    def __init__(self, value):
        if isinstance(value, Action):
            value = value._value
        if isinstance(value, bool) or not isinstance(value, (int, np.integer)) or not 0 <= int(value) <= 4:
            raise ValueError(f"Invalid action value: {value}. Valid values for actions are between 0 and 4.")
        object.__setattr__(self, "_value", int(value))

    def __setattr__(self, key, val):
        raise AttributeError("Action is immutable")

    @property
    def value(self):
        return self._value

    @property
    def name(self):
        return Action._NAMES[self._value]

    @staticmethod
    def variants():
        return [Action.NORTH, Action.SOUTH, Action.EAST, Action.WEST, Action.STAY]

    @staticmethod
    def cardinality():
        return 5

    @property
    def delta(self):
        return {0: (-1, 0), 1: (1, 0), 2: (0, 1), 3: (0, -1), 4: (0, 0)}[self._value]  # src/action.rs:18-26

    def opposite(self):
        return {0: Action.SOUTH, 1: Action.NORTH, 2: Action.WEST, 3: Action.EAST, 4: Action.STAY}[self._value]

    @staticmethod
    def from_delta(di, dj):
        """(dx, dy) convention of the binding (pyaction.rs:71-88): (-1,0) is WEST, (0,-1) is NORTH."""
        table = {(0, 0): Action.STAY, (-1, 0): Action.WEST, (1, 0): Action.EAST, (0, -1): Action.NORTH, (0, 1): Action.SOUTH}
        if (di, dj) not in table:
            raise ValueError(f"Invalid delta: ({di}, {dj}). Valid deltas for actions are (-1, 0), (1, 0), (0, -1), or (0, 1).")
        return table[(di, dj)]

    def __eq__(self, other):
        return isinstance(other, Action) and self._value == other._value

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(("lle.Action", self._value))

    def __int__(self):
        return self._value

    def __reduce__(self):
        return (Action, (self._value,))

    def __deepcopy__(self, memo):
        return Action(self._value)

    def __repr__(self):
        return _ACTION_DEBUG[self._value]

    def __str__(self):
        return self.name


_ACTION_DEBUG = ["North", "South", "East", "West", "Stay"]
for _v, _n in enumerate(Action._NAMES):
    setattr(Action, _n, Action(_v))


class EventType(Enum):
    """src/bindings/world/pyevent.rs:9-16"""
    AGENT_EXIT = 0
    GEM_COLLECTED = 1
    AGENT_DIED = 2


class Direction(Enum):
    """src/core/tiles/direction.rs:8-18, src/bindings/tiles/pydirection.rs: built from the map's letters too (`Direction("N")`)."""
    NORTH = 0
    EAST = 1
    SOUTH = 2
    WEST = 3

    @classmethod
    def _missing_(cls, value):
        letters = {"N": cls.NORTH, "E": cls.EAST, "S": cls.SOUTH, "W": cls.WEST}
        if value in letters:
            return letters[value]
        raise ValueError(f"Invalid direction string: {value}")

    @staticmethod
    def from_str(direction):
        """pydirection.rs:82-90: one of "N", "E", "S", "W"; ValueError otherwise."""
        letters = {"N": Direction.NORTH, "E": Direction.EAST, "S": Direction.SOUTH, "W": Direction.WEST}
        if direction not in letters:
            raise ValueError("Invalid direction string.")
        return letters[direction]

    def opposite(self):
        return Direction((self.value + 2) % 4)

    @property
    def delta(self):
        return {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}[self.value]

    @property
    def is_horizontal(self):
        return self in (Direction.EAST, Direction.WEST)  # pydirection.rs:112-117

    @property
    def is_vertical(self):
        return not self.is_horizontal

    @property
    def name(self):
        """The map's letter, as in the reference (pydirection.rs:129-138) -- not the enum member's identifier."""
        return "NESW"[self.value]

    def __repr__(self):
        return self.name

    def __hash__(self):
        return self.value  # pydirection.rs:158-165


class WorldEvent:
    """src/bindings/world/pyevent.rs:37-58"""

    def __init__(self, event_type, agent_id):
        self.event_type = event_type
        self.agent_id = agent_id

    def __repr__(self):
        return f"{self.event_type.name}, agent id: {self.agent_id}"

    def __eq__(self, other):
        return isinstance(other, WorldEvent) and (self.event_type, self.agent_id) == (other.event_type, other.agent_id)

    def __hash__(self):
        return hash((self.event_type, self.agent_id))


class WorldState:
    """src/bindings/world/pyworld_state.rs:36-64: (agents_positions, gems_collected, agents_alive=all True)."""

    def __init__(self, agents_positions, gems_collected, agents_alive=None):
        self.agents_positions = [tuple(int(v) for v in p) for p in agents_positions]
        self.gems_collected = [bool(g) for g in gems_collected]
        self.agents_alive = [True] * len(self.agents_positions) if agents_alive is None else [bool(a) for a in agents_alive]

    def as_array(self):
        """[i0, j0, ..., gems..., alive...] as float32 (pyworld_state.rs:79-101)."""
        flat = [float(v) for p in self.agents_positions for v in p]
        flat += [1.0 if g else 0.0 for g in self.gems_collected] + [1.0 if a else 0.0 for a in self.agents_alive]
        return np.array(flat, dtype=np.float32)

    @staticmethod
    def from_array(array, n_agents, n_gems):
        array = list(array)
        if len(array) != n_agents * 3 + n_gems:
            raise ValueError(f"The array must have a length of {n_agents * 3 + n_gems}.")
        pos = [(int(array[2 * i]), int(array[2 * i + 1])) for i in range(n_agents)]
        gems = [array[2 * n_agents + i] == 1.0 for i in range(n_gems)]
        alive = [array[2 * n_agents + n_gems + i] == 1.0 for i in range(n_agents)]
        return WorldState(pos, gems, alive)

    def _key(self):
        return (tuple(self.agents_positions), tuple(self.gems_collected), tuple(self.agents_alive))

    def __eq__(self, other):
        return isinstance(other, WorldState) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return (f"WorldState {{ agents_positions: {self.agents_positions}, gems_collected: {self.gems_collected}, "
                f"agents_alive: {self.agents_alive} }}")

    def __deepcopy__(self, memo):
        return WorldState(self.agents_positions, self.gems_collected, self.agents_alive)


# ------------------------------------------------------------------------------------------------ tile snapshots
class Laser:
    """Snapshot of one laser tile (src/bindings/tiles/pylaser.rs:18-38)."""

    def __init__(self, laser_id, agent_id, direction, is_on, is_enabled, pos, occupant):
        self.laser_id, self.agent_id, self.direction = laser_id, agent_id, direction
        self.is_on, self.is_enabled, self.pos, self.agent = bool(is_on), bool(is_enabled), pos, occupant

    @property
    def is_off(self):
        return not self.is_on

    @property
    def is_disabled(self):
        return not self.is_enabled

    def __repr__(self):
        return (f"Laser(laser_id={self.laser_id}, is_on={_rust_bool(self.is_on)}, direction={self.direction.name}, "
                f"agent_id={self.agent_id}, agent={self.agent})")  # pylaser.rs:83-96


def _rust_bool(b):
    return "true" if b else "false"


class Gem:
    """One gem (src/bindings/tiles/pygem.rs:13-88): `pos` and `is_collected` are a snapshot taken when the list was built,
    `agent` reads the world, `collect()` marks the gem collected IN the world (no event, no reward: tiles/gem.rs:17-19)."""

    def __init__(self, world, index, pos, is_collected):
        self._world, self._index = world, index
        self.pos, self.is_collected = pos, bool(is_collected)

    @property
    def agent(self):
        if self.pos in self._world._laser_cells:  # the tile there is a Laser, not a Gem (pygem.rs:69-76)
            return None
        return self._world._occupant_at(self._world._state(), self.pos)

    def collect(self):
        if self.pos in self._world._laser_cells:  # `inner` is World::at_mut: a gem under a beam is a Laser tile (pygem.rs:52-62)
            raise ValueError(f"Tile at {self.pos} is not a gem")
        self._world._collect_gem(self._index)
        self.is_collected = True

    def __repr__(self):
        a = self.agent
        return f"Gem(pos={self.pos}, is_collected={_rust_bool(self.is_collected)}, agent={'None' if a is None else f'Some({a})'})"

    __str__ = __repr__


class Agent:
    """src/bindings/pyagent.rs:6-36"""

    def __init__(self, num, alive, arrived):
        self.num, self.is_alive, self.is_dead, self.has_arrived = num, bool(alive), not alive, bool(arrived)


class LaserSource:
    """Handle on one laser source (src/bindings/tiles/pylaser_source.rs:20-142); mutators act on the world."""

    def __init__(self, world, info):
        self._world = world
        self.laser_id = int(info.laser_id)
        self.pos = (int(info.i), int(info.j))
        self.direction = Direction(int(info.direction))
        self._agent_id = int(info.agent_id)
        self._enabled = bool(info.enabled)

    @property
    def is_enabled(self):
        return self._enabled

    @is_enabled.setter
    def is_enabled(self, value):
        self._set_status(bool(value))

    @property
    def is_disabled(self):
        return not self._enabled

    @is_disabled.setter
    def is_disabled(self, value):
        self._set_status(not bool(value))

    def enable(self):
        self._set_status(True)

    def disable(self):
        self._set_status(False)

    def _set_status(self, enabled):
        if enabled == self._enabled:
            return
        self._world._set_source(self.laser_id, enabled=enabled)
        self._enabled = enabled

    @property
    def agent_id(self):
        return self._agent_id

    @agent_id.setter
    def agent_id(self, new_agent_id):
        self.set_colour(new_agent_id)

    def set_agent_id(self, new_agent_id):
        self.set_colour(new_agent_id)

    def set_colour(self, colour):
        """PyLaserSource.set_agent_id (src/bindings/tiles/pylaser_source.rs:107-141), quirk included: the world's colour
        is changed BEFORE the start positions are checked, so a change refused with "would cross the start position"
        has recoloured the world all the same; only this snapshot keeps the old id."""
        colour = int(colour)
        if colour < 0:
            raise OverflowError("can't convert negative int to unsigned")
        w = self._world
        if colour >= w.n_agents:
            raise ValueError("Agent ID is greater than the number of agents")
        w._set_source(self.laser_id, colour=colour)  # :114-119
        # :121-139: the beam must not cross a possible start of another agent
        cells = {(t.i, t.j) for t in w._map.laser_tiles() if t.laser_id == self.laser_id}
        for agent, starts in enumerate(w.random_start_pos):
            hit = cells & set(starts)
            if agent != colour and hit:
                raise ValueError(f"Laser source cannot be changed to agent ID {colour} since it would cross the start "
                                 f"position of agent {agent} at {sorted(hit)}")
        self._agent_id = colour

    def __eq__(self, other):
        return (isinstance(other, LaserSource) and (self.agent_id, self.direction, self.laser_id, self.pos) ==
                (other.agent_id, other.direction, other.laser_id, other.pos))

    def __hash__(self):
        return self.laser_id

    def __repr__(self):
        return (f"LaserSource(laser_id={self.laser_id}, is_enabled={_rust_bool(self.is_enabled)}, direction={self.direction.name}, "
                f"agent_id={self.agent_id})")  # pylaser_source.rs:167-175


# ------------------------------------------------------------------------------------------------ World
_LEVEL_NAMES = {f"lvl{k}": k for k in range(1, 7)}
_LEVEL_NAMES.update({f"level{k}": k for k in range(1, 7)})


class World:
    """The environment in which the agents evolve (reference docstring: pyworld.rs:29-41).

        w1 = World.level(5)
        w2 = World("S0 X")
        w3 = World.from_file("lvl1")
    """

    def __init__(self, map_str, _map=None, device=None):
        if _map is None:
            try:
                _map = Map(str(map_str))
            except MapParseError as e:
                raise _parse_error(e) from None
        self._map = _map
        self._device = device
        self._batch_obj = None
        m = _map
        self.height, self.width, self.n_agents, self.n_gems = m.height, m.width, m.n_agents, m.n_gems
        self._exit_pos = m.positions(LLE_POS_EXIT)
        self.wall_pos = m.positions(LLE_POS_WALL)
        self.void_pos = m.positions(LLE_POS_VOID)
        self.start_pos = m.positions(LLE_POS_START)
        self.random_start_pos = [[p] for p in self.start_pos]
        self._gem_pos = m.positions(LLE_POS_GEM)
        self._laser_tiles = m.laser_tiles()
        self._laser_cells = {(t.i, t.j) for t in self._laser_tiles}

    # ---- constructors (pyworld.rs:147-200)
    @staticmethod
    def level(level):
        try:
            return World(None, _map=Map(level=int(level)))
        except MapParseError as e:
            if e.kind == "InvalidLevel":
                raise InvalidLevelError(f"Invalid level: {level}. Expected a level between 1 and 6.") from None
            raise _parse_error(e) from None

    @staticmethod
    def from_file(filename):
        name = str(filename).lower()
        if name in _LEVEL_NAMES:  # src/core/levels.rs:10-19
            return World.level(_LEVEL_NAMES[name])
        if not os.path.exists(filename):
            raise FileNotFoundError(str(filename))
        with open(filename) as f:
            return World(f.read())

    def save(self, filename):
        """Save the world configuration to the given file (pyworld.rs:183-193: `world_string` written as is)."""
        try:
            with open(filename, "w") as f:
                f.write(self.world_string)
        except OSError as e:
            raise ValueError(f"Could not write to file: {filename}: {e}") from None

    @property
    def exit_pos(self):
        """The (i, j) position of each exit (pyworld.rs:46-56)."""
        return self._exit_pos

    @exit_pos.setter
    def exit_pos(self, exit_pos):
        """World::set_exit_positions (src/core/world.rs:195-234) through the setter of the binding (pyworld.rs:203-209): the
        current exits become floor tiles, the given cells exits -- also under a laser --, whoever stands on a swapped tile stays
        there and nobody's `has_arrived` changes.  Fewer exits than agents: ParsingError (NotEnoughExitTiles), world untouched.
        A cell that is not a floor (wall, laser source, gem, void, the same cell twice) or out of the world: the reference
        PANICS half way through the swap and leaves a poisoned world behind; here it is a ValueError and the world is untouched."""
        pos = [tuple(int(v) for v in p) for p in exit_pos]
        if any(v < 0 for p in pos for v in p):
            raise OverflowError("can't convert negative int to unsigned")
        try:
            self._map.set_exits(pos)
        except MapParseError as e:
            err = ParsingError(f"Not enough exit tiles: {self.n_agents} starts, {len(pos)} exits" if e.kind == "NotEnoughExitTiles"
                               else _PARSE_MESSAGES.get(e.kind, e.kind))
            err.kind = e.kind
            raise err from None
        self._exit_pos = pos
        if self._batch_obj is not None:
            self._batch_obj.update_map()

    # ---- device batch (lazy: parsing needs no GPU, dynamics do)
    @property
    def _batch(self):
        if self._batch_obj is None:
            from .batched import BatchedWorld
            # creation resets (World::new, world.rs:82).  The layered observation of the facade's one environment leaves the kernels as
            # float32, the reference's dtype (python/lle/observations.py:223): lle_batch_options.obs_dtype -- no cast on the host
            import torch
            self._batch_obj = BatchedWorld(self._map, 1, device=self._device, obs_dtype=torch.float32)
            self._map = self._batch_obj.map  # (the batch may hold a copy: mutators go through the object it pushes from)
        return self._batch_obj

    def _state(self):
        return {k: v[0] for k, v in self._batch.host_small_state().items()}

    def _events(self, st):
        return [WorldEvent(EventType(t), a) for t, a in _decode.events_list(st["evcount"], st["events"])]

    # ---- dynamics
    def reset(self):
        """Reset the world to its original state (src/core/world.rs:411-432)."""
        self._batch.reset()

    @staticmethod
    def _extract_actions(action):
        err = TypeError("Action must be of type Action or list[Action]")  # pyworld.rs:123-141
        if isinstance(action, Action):
            return [action]
        try:
            items = list(action)
        except TypeError:
            raise err from None
        if not all(isinstance(a, Action) for a in items):
            raise err
        return items

    def step(self, action):
        """Simultaneously perform an action for each agent (src/core/world.rs:435-475).  Returns the events.

        Raises InvalidActionError if an agent takes an unavailable action (the world is left untouched) and
        ValueError if the number of actions differs from the number of agents."""
        import torch
        actions = self._extract_actions(action)
        if len(actions) != self.n_agents:
            raise ValueError(f"Invalid number of actions: given {len(actions)}, expected {self.n_agents}")
        b = self._batch
        b.step(torch.tensor([[a.value for a in actions]], dtype=torch.uint8, device=b.device))
        st = self._state()
        err = int(st["err"])
        if err:
            agent = err - 1
            avail = [Action(a) for a in _decode.avail_lists(st["avail"])[agent]]
            raise InvalidActionError(f"Invalid action for agent {agent}: available actions: {avail!r}, "
                                     f"taken action: {actions[agent]!r}")
        return self._events(st)

    def available_actions(self):
        """Per agent, in the reference's order [STAY, NORTH, EAST, SOUTH, WEST] filtered (world.rs:343-363)."""
        return [[Action(a) for a in lst] for lst in _decode.avail_lists(self._state()["avail"])]

    def available_joint_actions(self):
        return [list(j) for j in itertools.product(*self.available_actions())]

    def get_state(self):
        st = self._state()
        alive, _, _ = _decode.agent_bits(st["bits"], self.n_agents)
        return WorldState(_decode.positions(st["pos"]), _decode.gem_bits(st["gems"], self.n_gems), alive)

    def set_state(self, state):
        """Force the world to a given state (src/core/world.rs:515-597, same checks, same lossy beam re-derivation)."""
        import torch
        if len(state.gems_collected) != self.n_gems:
            raise InvalidWorldStateError(f"Invalid number of gems: given {len(state.gems_collected)}, expected {self.n_gems}")
        if len(state.agents_positions) != self.n_agents:
            raise InvalidWorldStateError(f"Invalid number of agents: given {len(state.agents_positions)}, expected {self.n_agents}")
        pos = [tuple(p) for p in state.agents_positions]
        if any(v < 0 for p in pos for v in p):
            raise OverflowError("can't convert negative int to unsigned")
        if len(set(pos)) != len(pos):
            raise InvalidWorldStateError(f"Invalid world state: There are two agents at the same position. Wrong state: {state!r}")
        for p in pos:
            if p[0] >= self.height or p[1] >= self.width:
                raise IndexError(f"Position {p} is out of the world's boundaries")
        b = self._batch
        dev = b.device
        b.set_state(torch.tensor([pos], dtype=torch.uint8, device=dev).view(1, self.n_agents, 2),
                    torch.tensor([state.gems_collected], dtype=torch.bool, device=dev).view(1, self.n_gems),
                    torch.tensor([state.agents_alive], dtype=torch.bool, device=dev).view(1, self.n_agents))
        st = self._state()
        err = int(st["err"])
        if err == _capi.LLE_ENV_INVALID_AGENT_POSITION:
            bad = next(p for p in pos if p in self.wall_pos)
            raise InvalidWorldStateError(f"Invalid agent position {bad}: The tile is not walkable")
        if err == _capi.LLE_ENV_OUT_OF_WORLD_POSITION:
            raise IndexError("Position is out of the world's boundaries")
        if err == _capi.LLE_ENV_INVALID_WORLD_STATE:
            raise InvalidWorldStateError("Invalid world state: The given state is invalid (e.g. an agent whose alive status "
                                         f"was set to `true` died).. Wrong state: {state!r}")
        return self._events(st)

    def set_agents_positions(self, agents_positions):
        state = self.get_state()
        state.agents_positions = [tuple(p) for p in agents_positions]
        return self.set_state(state)

    def set_agent_position(self, agent_id, position):
        if agent_id >= self.n_agents:
            raise ValueError(f"Agent id {agent_id} is out of bounds")
        state = self.get_state()
        state.agents_positions[agent_id] = tuple(position)
        return self.set_state(state)

    def seed(self, seed_value):
        """v1 maps give every agent a single start, for which the reference's reset consumes no randomness
        (src/utils/mod.rs:63); nothing to seed."""

    # ---- read-only views
    @property
    def agents_positions(self):
        return _decode.positions(self._state()["pos"])

    @property
    def agents(self):
        alive, arrived, _ = _decode.agent_bits(self._state()["bits"], self.n_agents)
        return [Agent(a, alive[a], arrived[a]) for a in range(self.n_agents)]

    def _occupant_at(self, st, pos):
        _, _, occ = _decode.agent_bits(st["bits"], self.n_agents)
        for a, p in enumerate(_decode.positions(st["pos"])):
            if occ[a] and p == pos:
                return a
        return None

    @property
    def gems(self):
        st = self._state()
        col = _decode.gem_bits(st["gems"], self.n_gems)
        return [Gem(self, g, p, c) for g, (p, c) in enumerate(zip(self._gem_pos, col))]

    def _collect_gem(self, index):
        """Gem.collect (pygem.rs:52-66): bit `index` of the env's gem word, then the observation rebuilt from the state."""
        bw = self._batch
        bit = 1 << int(index)
        bw.gems[0] |= bit - (1 << 32) if bit >= 1 << 31 else bit  # (an int32 word)
        bw.observe()

    @property
    def gems_collected(self):
        """Number of collected gems; like the reference it ignores gems lying under a beam (world.rs:265-275)."""
        col = _decode.gem_bits(self._state()["gems"], self.n_gems)
        return sum(1 for p, c in zip(self._gem_pos, col) if c and p not in self._laser_cells)

    def gem_at(self, position):
        position = tuple(position)
        if position[0] >= self.height or position[1] >= self.width:
            raise IndexError("Position out of bounds")
        for g in self.gems:
            if g.pos == position and position not in self._laser_cells:
                return g
        raise ValueError(f"Tile at position {position} is not a gem")

    @property
    def lasers(self):
        """Every laser tile (outer layer and the one directly below, src/core/world.rs:159-172)."""
        st = self._state()
        srcs = self._map.sources()
        rows = _decode.lasers_listing(self._laser_tiles, srcs, st["beams"])
        return [Laser(lid, col, Direction(int(srcs[lid].direction)), on, en, (i, j), self._occupant_at(st, (i, j)))
                for (i, j, lid, col, on, en) in rows]

    @property
    def laser_sources(self):
        return [LaserSource(self, s) for s in self._map.sources()]

    def source_at(self, position):
        position = tuple(position)
        if position[0] >= self.height or position[1] >= self.width:
            raise IndexError("Position out of bounds")
        for s in self.laser_sources:
            if s.pos == position:
                return s
        raise ValueError(f"Tile at position {position} is not a laser source")

    def _set_source(self, laser_id, enabled=None, colour=None):
        self._map.set_source(laser_id, enabled=enabled, agent_id=colour)
        if self._batch_obj is not None:
            self._batch_obj.update_sources()

    @property
    def n_laser_colours(self):
        return len({int(s.agent_id) for s in self._map.sources()})

    @property
    def world_string(self):
        return self._map.world_string()

    @property
    def image_dimensions(self):
        return (32 * self.width + 1, 32 * self.height + 1)  # src/unit_tests/test_renderer.rs:24

    def get_image(self):
        raise NotImplementedError("rendering is outside the scope of lle_amd (SURVEY.md section 2, row 11)")

    def layered_observation(self):
        """(C, H, W) float32 layered observation of the current state (python/lle/observations.py:254-266), from the GPU: the
        kernels store it in that type (a batch of one environment created with obs_dtype = float32)."""
        b = self._batch
        if not self._map.obs_supported:
            raise IndexError("a laser colour addresses a layer beyond 2*n_agents+4")
        import torch
        torch.cuda.synchronize(b.device)
        return b.obs[0].cpu().numpy()

    def observation(self, kind, param=0):
        """Observation `kind` (lle_amd._capi.LLE_OBS_*) of the current state as a numpy array, from the GPU kernels of
        observers.hip.  Raises IndexError where the reference does."""
        import torch
        b = self._batch
        out = b.observe_as(kind, param)[0]
        torch.cuda.synchronize(b.device)
        return out.cpu().numpy()

    def available_actions_mask(self, walkable_lasers=True):
        """LLE.available_actions (python/lle/env/env.py:146-163): bool (n_agents, 5) in Action value order."""
        import torch
        b = self._batch
        out = b.available_actions(walkable_lasers)[0]
        torch.cuda.synchronize(b.device)
        return out.cpu().numpy()

    # ---- copy / pickle (pyworld.rs:546-604): (world_string, state), restored through set_state
    def __getstate__(self):
        return (self.world_string, self.get_state())

    def __setstate__(self, state):
        world_string, world_state = state
        self.__init__(world_string)
        self.set_state(world_state)

    def __getnewargs__(self):
        return ("S0 X",)

    def __deepcopy__(self, memo):
        clone = World(self.world_string, device=self._device)
        for s, c in zip(self._map.sources(), clone._map.sources()):
            if not s.enabled:
                clone._set_source(int(c.laser_id), enabled=False)
        clone.set_state(self.get_state())
        return clone

    def __repr__(self):
        pos = "".join(f"Agent {i} position: {p}, " for i, p in enumerate(self.agents_positions))
        return (f"World(height={self.height}, width={self.width}, n_gems={self.n_gems}, n_agents={self.n_agents}, "
                f"world_string={self.world_string}){pos}")


def deepcopy_world(world):
    return copy.deepcopy(world)
