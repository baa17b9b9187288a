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
