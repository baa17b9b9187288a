"""KAT surface (see tests/kat_runner.py) over the product's `lle_amd.World` facade, i.e. through the HIP kernels."""
from lle_amd import Action, World
from lle_amd.world import InvalidActionError, InvalidWorldStateError, ParsingError, WorldState


class KatError(Exception):
    def __init__(self, kind, agent=None):
        super().__init__(kind)
        self.kind, self.agent = kind, agent


class GpuWorld:
    def __init__(self, map_str=None, level=None):
        try:
            self.w = World.level(level) if level is not None else World(map_str)
        except ParsingError as e:
            raise KatError(e.kind) from None
        w = self.w
        self.height, self.width, self.n_agents, self.n_gems = w.height, w.width, w.n_agents, w.n_gems
        self.n_sources = len(w.laser_sources)
        self.start_pos, self.exit_pos, self.wall_pos, self.void_pos = w.start_pos, w.exit_pos, w.wall_pos, w.void_pos
        self.gem_pos = w._gem_pos

    def sources(self):
        return [(s.i, s.j, s.direction, s.agent_id, s.enabled, s.length) for s in self.w._map.sources()]

    def reset(self):
        self.w.reset()

    def step(self, actions):
        try:
            ev = self.w.step([Action(a) for a in actions])
        except InvalidActionError as e:
            agent = int(str(e).split("agent ")[1].split(":")[0])
            raise KatError("InvalidAction", agent) from None
        return [(e.event_type.value, e.agent_id) for e in ev]

    def set_state(self, positions, gems, alive):
        try:
            ev = self.w.set_state(WorldState(positions, gems, alive))
        except IndexError:
            raise KatError("OutOfWorldPosition") from None
        except InvalidWorldStateError as e:
            msg = str(e)
            kind = ("InvalidNumberOfGems" if "number of gems" in msg else "InvalidNumberOfAgents" if "number of agents" in msg
                    else "InvalidAgentPosition" if "agent position" in msg else "InvalidWorldState")
            raise KatError(kind) from None
        return [(e.event_type.value, e.agent_id) for e in ev]

    def positions(self):
        return self.w.agents_positions

    def alive(self):
        return [a.is_alive for a in self.w.agents]

    def arrived(self):
        return [a.has_arrived for a in self.w.agents]

    def gems_collected(self):
        return self.w.get_state().gems_collected

    def n_gems_collected(self):
        return self.w.gems_collected

    def available_actions(self):
        return [[a.value for a in lst] for lst in self.w.available_actions()]

    def lasers(self):
        return [(l.pos[0], l.pos[1], l.laser_id, l.agent_id, int(l.is_on), int(l.is_enabled)) for l in self.w.lasers]

    def beam_bits(self, laser_id):
        from lle_amd import _decode
        return _decode.beam_bits(self.w._state()["beams"], self.w._map.source_first_words()[laser_id], self.w._map.sources()[laser_id].length)

    def set_source(self, laser_id, enabled=None, colour=None):
        src = self.w.laser_sources[laser_id]
        if enabled is not None:
            src.enable() if enabled else src.disable()
        if colour is not None:
            self.w._set_source(laser_id, colour=colour)  # core-level set_agent_id (laser_source.rs:45-47), no start check

    def obs(self):
        return self.w.layered_observation()

    def set_exits(self, exits):
        try:
            self.w.exit_pos = exits  # the property setter of the facade (pyworld.rs:203-209)
        except ParsingError as e:
            raise KatError(e.kind) from None
        except ValueError:
            raise KatError("Panic") from None
        self.exit_pos = self.w.exit_pos

    def collect_gem(self, i, j):
        try:
            self.w.gem_at((i, j)).collect()  # the facade's Gem.collect (pygem.rs:52-66)
        except ValueError:
            raise KatError("ValueError") from None

    def tile_agent(self, i, j):
        a = self.w._occupant_at(self.w._state(), (i, j))
        return -1 if a is None else a

    # ---- binding-level operations: the product facade itself is under test here
    def clone(self):
        import copy
        c = object.__new__(GpuWorld)
        c.w = copy.deepcopy(self.w)  # PyWorld.__deepcopy__ = World::clone (pyworld.rs:557-559, world.rs:645-652)
        assert c.w.agents_positions is not self.w.agents_positions
        for name in ("height", "width", "n_agents", "n_gems", "n_sources", "start_pos", "exit_pos", "wall_pos", "void_pos", "gem_pos"):
            setattr(c, name, getattr(self, name))
        return c

    def set_agent_position(self, agent, pos):
        try:
            ev = self.w.set_agent_position(agent, pos)
        except IndexError:
            raise KatError("OutOfWorldPosition") from None
        except InvalidWorldStateError:
            raise KatError("InvalidWorldState") from None
        except ValueError as e:
            assert "out of bounds" in str(e)
            raise KatError("AgentIdOutOfBounds") from None
        return [(e.event_type.value, e.agent_id) for e in ev]

    def set_colour_checked(self, laser_id, colour):
        src = self.w.laser_sources[laser_id]
        before = src.agent_id
        try:
            src.set_colour(colour)
        except OverflowError:
            raise KatError("OverflowError") from None
        except ValueError:
            assert src.agent_id == before  # the caller's snapshot keeps the old id (pylaser_source.rs:141 is not reached)
            raise KatError("ValueError") from None
