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
