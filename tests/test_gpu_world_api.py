"""API-surface tests of the reference's `World` that need the dynamics (python/tests/test_world.py:65-106)."""
import pytest

pytestmark = pytest.mark.gpu


def test_world_step_something_else_than_action():   # test_world.py:65-77
    from lle_amd import World

    world = World("S0 X . .\n.  . . .\n.  . . .")
    world.reset()
    with pytest.raises(TypeError):
        world.step(23)


def test_world_step_tuple_and_invalid_sequence_action():   # test_world.py:92-106
    from lle_amd import Action, World

    world = World("S0 X . .\n.  . . .\n.  . . .")
    world.reset()
    world.step((Action.SOUTH,))
    assert world.agents_positions == [(1, 0)]
    with pytest.raises(TypeError, match="Action must be of type Action or list\\[Action\\]"):
        world.step((23,))
    assert world.step(Action.SOUTH) == [] and world.agents_positions == [(2, 0)]   # a bare Action for a single agent (:52-62)


def test_gem_collect_and_the_tile_reprs():
    """Gem.collect() / Gem.agent (src/bindings/tiles/pygem.rs:52-88) and the strings the bindings print
    (pygem.rs:78-88, pylaser.rs:83-96, pylaser_source.rs:167-175: Rust's `true` / `false`, `Some(0)` / `None`, the direction's letter)."""
    from lle_amd import Action, World

    world = World("S0 G . X\nL0E G . .")
    world.reset()
    free, under = world.gems
    assert (free.pos, under.pos) == ((0, 1), (1, 1)) and not free.is_collected
    assert repr(free) == str(free) == "Gem(pos=(0, 1), is_collected=false, agent=None)"
    free.collect()
    assert free.is_collected and world.gems[0].is_collected and world.gems_collected == 1
    assert world.get_state().gems_collected == [True, False]
    assert int(world.layered_observation()[4, 0, 1]) == 0 and int(world.layered_observation()[4, 1, 1]) == 1
    assert world.step(Action.EAST) == []   # nothing left to collect there (gem.rs:26-33)
    assert free.agent == 0 and repr(world.gems[0]) == "Gem(pos=(0, 1), is_collected=true, agent=Some(0))"
    with pytest.raises(ValueError, match="is not a gem"):
        under.collect()   # a Laser tile wraps it (World::at_mut)
    assert under.agent is None and not world.gems[1].is_collected
    assert repr(world.laser_sources[0]) == "LaserSource(laser_id=0, is_enabled=true, direction=E, agent_id=0)"
    assert repr(world.lasers[0]).startswith("Laser(laser_id=0, is_on=true, direction=E, agent_id=0, agent=None")
    world.reset()
    assert [g.is_collected for g in world.gems] == [False, False]


def test_perspective_generator_reports_its_observation_type():   # python/tests/test_observations.py:404-407
    from lle_amd import World
    from lle_amd.observations import AgentZeroPerspective, ObservationType

    assert AgentZeroPerspective(World("S0 X")).obs_type is ObservationType.AGENT0_PERSPECTIVE_LAYERED


def test_world_and_action_cross_a_thread_boundary():   # python/tests/test_world.py:262-288 (the binding declares World Send + Sync)
    import threading

    from lle_amd import Action, World

    class Holder(threading.Thread):
        def __init__(self, item):
            super().__init__()
            self.item, self.result = item, None

        def run(self):
            w = self.item
            self.result = w.step(Action.EAST) if isinstance(w, World) else w.delta

    world = World("S0 . X")
    world.reset()
    threads = [Holder(Action.NORTH), Holder(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert threads[0].result == (-1, 0) and threads[1].result == [] and world.agents_positions == [(0, 1)]


def test_standard_level_names_and_setstate_onto_another_world():
    """python/tests/test_world.py:447-451 (`lvl3` / `level3` are level names for from_file, src/core/levels.rs:10-19) and
    src/unit_tests/test_pyworld.rs:4-9 (__setstate__ turns an existing world into the pickled one)."""
    from lle_amd import World

    for i in range(1, 7):
        a, b, c = World.level(i), World.from_file(f"lvl{i}"), World.from_file(f"level{i}")
        assert a.world_string == b.world_string == c.world_string
    world = World.level(1)
    other = World("S0 X")
    other.__setstate__(world.__getstate__())
    assert (other.width, other.height, other.n_agents) == (world.width, world.height, 1) and other.get_state() == world.get_state()
    other.reset()
    world.reset()
    assert other.agents_positions == world.agents_positions


def test_sampled_stepper_equals_step():
    """BatchedWorld.sampled_stepper(): the bound hot-loop callable takes the same steps as step(sample=True, ...)."""
    import pytest
    import torch

    from lle_amd import BatchedWorld
    from oracle.levels import LEVELS

    a, b = BatchedWorld(LEVELS[6], 3000), BatchedWorld(LEVELS[6], 3000)
    one = a.sampled_stepper(auto_reset=True, seed=9, env_offset=4)
    for t in range(25):
        one()
        b.step(sample=True, auto_reset=True, seed=9, env_offset=4)
    assert a.t == b.t == 25
    for k in ("pos", "bits", "gems", "beams", "avail", "actions", "events", "evcount", "done", "obs"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.stats() == b.stats()
