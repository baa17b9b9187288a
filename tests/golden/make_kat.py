"""Known-answer tests for the World.step() hot path, hand-transcribed from the reference's own tests.

The reference (yamoling/lle) holds no golden-vector files: every expectation is an inline assertion in
its Rust / Python tests.  Each case below restates ONE such test as data: the map text, the script of
calls, and exactly the assertions the reference test makes (`ref` cites the test's file:line relative to
the reference repository).  Keys prefixed `derived_` are NOT asserted by the reference: they are extra
expectations derived by reading the reference source and are reported separately.

Running this file rewrites tests/golden/kat_world.json (the committed fixture).  Nothing here imports
or executes the reference.

Codes: actions N=0 S=1 E=2 W=3 STAY=4; events EXIT=0 GEM=1 DIED=2 written as [type, agent].
"""
import json
import os

N, S, E, W, STAY = 0, 1, 2, 3, 4
EXIT, GEM, DIED = 0, 1, 2

CASES = []


def case(name, ref, map=None, level=None, script=(), parse_error=None, static=None):
    c = {"name": name, "ref": ref}
    if map is not None:
        c["map"] = map
    if level is not None:
        c["level"] = level
    if parse_error:
        c["parse_error"] = parse_error
    if static:
        c["static"] = static
    c["script"] = list(script)
    CASES.append(c)


def reset():
    return {"op": "reset"}


def step(actions, **kw):
    return {"op": "step", "actions": list(actions), **kw}


def expect(**kw):
    return {"op": "expect", **kw}


def set_state(positions, gems, alive=None, **kw):
    return {"op": "set_state", "positions": positions, "gems": gems,
            "alive": alive if alive is not None else [True] * len(positions), **kw}


def source(laser_id, **kw):
    return {"op": "source", "laser_id": laser_id, **kw}


# --------------------------------------------------------------------------- src/unit_tests/test_world.rs
case("tile_type", "src/unit_tests/test_world.rs:24-61",
     map="\n    S0 . G\n    L0E X @\n    ",
     static={"start_pos": [[0, 0]], "gem_pos_contains": [[0, 2]], "sources": [[1, 0, 0]],
             "exit_pos_contains": [[1, 1]], "n_walls": 2, "wall_pos_contains": [[1, 2], [1, 0]]},
     script=[reset(), expect(laser_colours=[[1, 1, 0]])])

case("duplicate_start_pos", "src/unit_tests/test_world.rs:63-75", map="S0 S0 X X", parse_error="DuplicateStartTile")

case("start_pos_order", "src/unit_tests/test_world.rs:77-86", map="S1 S0 X X",
     static={"start_pos": [[0, 1], [0, 0]]},
     script=[reset(), expect(positions=[[0, 1], [0, 0]])])

case("start_pos_order_lvl6", "src/unit_tests/test_world.rs:88-97", level=6,
     static={"start_pos": [[0, 4], [0, 5], [0, 6], [0, 7]], "n_agents": 4},
     script=[reset(), expect(positions=[[0, 4], [0, 5], [0, 6], [0, 7]])])

case("laser_blocked_by_wall", "src/unit_tests/test_world.rs:99-115",
     map="\n        . L0S .\n        .  .  .\n        X  @  S0\n        .  .  .",
     script=[reset(), expect(no_laser_at=[[2, 1], [3, 1]])])

MAP_BLOCKED_ON_RESET = """
        @ @ L0S @  @
        @ .  .  .  @
        @ X  S0 .  @
        @ .  .  .  @
        @ @  @  @  @"""
case("laser_blocked_on_reset", "src/unit_tests/test_world.rs:117-135", map=MAP_BLOCKED_ON_RESET,
     script=[reset(), expect(alive=[True], lasers_on=[[1, 2, True], [2, 2, False], [3, 2, False]])])

MAP_FACING = """
         @ @ L0S @  @
         @ X  .  S0 @
         @ .  .  .  @
         @ X  .  S1 @
         @ @ L1N  @ @"""
case("facing_lasers", "src/unit_tests/test_world.rs:137-155", map=MAP_FACING,
     script=[reset(), step([W, W]), expect(alive=[True, True], all_lasers="off")])

case("event_exit_when_staying", "src/unit_tests/test_world.rs:157-169", map="S0 X .\n         S1 . X",
     script=[reset(), step([E, STAY], n_events=1), step([STAY, STAY], n_events=0)])

case("facing_lasers_agent_dies", "src/unit_tests/test_world.rs:171-185", map=MAP_FACING,
     script=[reset(), step([W, STAY]), expect(alive_of={"0": False})])

case("empty_world", "src/unit_tests/test_world.rs:187-196", map="", parse_error="EmptyWorld")

MAP_S0_G_X2 = "\n        S0 . G\n        X  . .\n    "
case("force_state_invalid_number_of_agents", "src/unit_tests/test_world.rs:198-225", map=MAP_S0_G_X2,
     script=[reset(), set_state([[1, 2], [0, 0]], [True], error="InvalidNumberOfAgents")])
case("force_state_invalid_number_of_gems", "src/unit_tests/test_world.rs:227-251", map=MAP_S0_G_X2,
     script=[reset(), set_state([[1, 2]], [True, False], error="InvalidNumberOfGems")])

case("complex_laser_blocking", "src/unit_tests/test_world.rs:253-282",
     map="\n    G L0E X . X\n    G G . . L1W\n    @ S0 . . @\n    . @ . . .\n    S1 G . . G",
     script=[reset(), expect(lasers_on=[[0, 3, True]]),
             set_state([[0, 2], [0, 3]], [False] * 5),
             expect(lasers_on=[[0, 3, False]], alive=[True, True]),
             step([STAY, E]),
             expect(alive=[True, True], lasers_on=[[0, 3, False]])])

case("set_state_available_actions", "src/unit_tests/test_world.rs:302-321",
     map="\n        .  . . @ . . . @ . X\n        .  @ . @ . @ . @ . @\n        S0 @ . . . @ . . . @\n    ",
     script=[reset(), set_state([[0, 0]], []), expect(avail_sets=[[S, E, STAY]])])

case("die_in_void", "src/unit_tests/test_world.rs:323-329", map="S0 V X",
     script=[reset(), step([E]), expect(alive=[False])])

case("num_gems_collected", "src/unit_tests/test_world.rs:331-342", map="S0 G X",
     script=[reset(), expect(n_gems_collected=0), step([E]), expect(n_gems_collected=1),
             step([STAY]), expect(n_gems_collected=1), step([E]), expect(n_gems_collected=1)])
case("num_agents_arrived", "src/unit_tests/test_world.rs:344-355", map="S0 G X",
     script=[reset(), expect(n_arrived=0), step([E]), expect(n_arrived=0),
             step([STAY]), expect(n_arrived=0), step([E]), expect(n_arrived=1)])

case("parse_inconsistent_row_lengths", "src/unit_tests/test_world.rs:357-378", map="X S0 .\n         . .",
     parse_error="InconsistentDimensions")
case("parse_inconsistent_start_exit_tiles", "src/unit_tests/test_world.rs:380-392", map="S1 S0 X",
     parse_error="NotEnoughExitTiles")
case("parse_no_agents", "src/unit_tests/test_world.rs:394-403", map=". . G", parse_error="NoAgents")

case("vertex_conflict_rs", "src/unit_tests/test_world.rs:405-419", map="\n    S0 X .\n    .  . .\n    S1 X .",
     script=[reset(), step([S, N]), expect(positions=[[0, 0], [2, 0]])])

case("reset_x10", "src/unit_tests/test_world.rs:421-436", map="S0 G X",
     script=sum(([reset(), expect(positions=[[0, 0]]), step([E], events=[[GEM, 0]]), step([E], events=[[EXIT, 0]])]
                 for _ in range(10)), []))

for lvl in range(1, 7):
    case(f"standard_level_{lvl}", "src/unit_tests/test_world.rs:438-453", level=lvl,
         static={"height": 12, "width": 13}, script=[reset()])

case("force_state", "src/unit_tests/test_world.rs:455-470", map=MAP_S0_G_X2,
     script=[reset(), set_state([[1, 2]], [True]), expect(positions=[[1, 2]], gems=[True])])
case("force_end_state", "src/unit_tests/test_world.rs:472-487", map=MAP_S0_G_X2,
     script=[reset(), set_state([[1, 0]], [True]), expect(positions=[[1, 0]], gems=[True])])

MAP_FORCE_DIES = "\n        S0 S1 G\n        X  X L0W\n    "
case("force_state_agent_dies", "src/unit_tests/test_world.rs:489-507 + tests/world_integration_tests.rs:229-251",
     map=MAP_FORCE_DIES,
     script=[reset(), set_state([[1, 0], [1, 1]], [False], [True, False]),
             expect(alive=[True, False], arrived=[True, False])])

case("wrong_world_state", "src/unit_tests/test_world.rs:526-541", map="\n        S0 L0S X\n        S1  .  X\n    ",
     script=[reset(), set_state([[0, 0], [1, 1]], [], error="InvalidWorldState")])

case("beam_single_source", "src/unit_tests/test_world.rs:661-679", map="\n        L0E . L0S\n        S0  X  @",
     script=[reset(), expect(beam={"0": [[0, 1]], "1": []}, sources=[[0, 0, 0], [0, 2, 0]])])

case("source_laser_id_is_source_index", "src/unit_tests/test_world.rs:681-695",
     map="\nS0  S1  S2  S3  S4  S5  S6  S7  S8  S9  S10\nL0S L1S L2S L3S L4S L5S L6S L7S L8S L9S L10S\n .  .   .   .   .   .   .   .   .   .   .\n X  X   X   X   X   X   X   X   X   X   X\n",
     static={"sources": [[1, k, k] for k in range(11)], "n_agents": 11})

case("beam_long", "src/unit_tests/test_world.rs:697-718", map="\n        L0E .  .  .  .  @\n        S0  .  .  .  X  @",
     script=[reset(), expect(beam={"0": [[0, 1], [0, 2], [0, 3], [0, 4]]})])
case("beam_south_direction", "src/unit_tests/test_world.rs:720-735",
     map="\n        @  L0S @\n        .  .   .\n        .  .   .\n        S0 X   @",
     script=[reset(), expect(beam={"0": [[1, 1], [2, 1], [3, 1]]})])
case("beam_agent_on_beam_tile", "src/unit_tests/test_world.rs:737-754",
     map="\n        L0E .  .  .\n        .   S0 .  X\n        .   .  .  . ",
     script=[reset(), step([N]), expect(positions=[[0, 1]], beam={"0": [[0, 1], [0, 2], [0, 3]]})])
case("beam_two_sources", "src/unit_tests/test_world.rs:756-774",
     map="\n        @ L0E .  .  .\n        .  .  .  .  .\n        @ L1E .  @  @\n        S0 .  .  X  .\n        S1 .  .  X  .",
     script=[expect(beam_len={"0": 3, "1": 1})])

# --------------------------------------------------------------------------- tests/world_integration_tests.rs
case("available_actions_rs", "tests/world_integration_tests.rs:3-17", map="\n    S0 . G\n    L0E X .\n    ",
     script=[reset(), expect(avail_sets=[[E, STAY]])])

MAP_AVAIL2 = "\n    .  S1 .\n    .  S0 G\n    L0E X X\n    "
case("available_actions_two_agents", "tests/world_integration_tests.rs:19-41", map=MAP_AVAIL2,
     script=[reset(), expect(avail_sets=[[S, E, W, STAY], [E, W, STAY]])])
case("available_actions_exit", "tests/world_integration_tests.rs:43-92", map=MAP_AVAIL2,
     script=[reset(), expect(avail_sets=[[S, E, W, STAY], [E, W, STAY]]),
             step([S, E]), expect(avail_sets=[[STAY], [S, W, STAY]]),
             step([STAY, S]), expect(avail_sets=[[STAY], [N, S, W, STAY]]),
             step([STAY, S]), expect(avail_sets=[[STAY], [STAY]])])

case("parse_empty_world", "tests/world_integration_tests.rs:94-103", map="", parse_error="EmptyWorld")
case("take_action_not_available", "tests/world_integration_tests.rs:105-122", map="S0 X",
     script=[reset(), step([N], error="InvalidAction", error_agent=0)])
case("take_action_not_available_swap", "tests/world_integration_tests.rs:124-141", map="S0 X\nS1 X",
     script=[reset(), step([S, N], error="InvalidAction", error_agent=0)])
case("take_action_walk_into_laser_source", "tests/world_integration_tests.rs:143-160", map="L0E X\nS0 .",
     script=[reset(), step([N], error="InvalidAction", error_agent=0)])
case("take_action_walk_outside_map", "tests/world_integration_tests.rs:162-179", map="L0E X\nS0 .",
     script=[reset(), step([W], error="InvalidAction", error_agent=0)])

case("force_state_agents_have_exited", "tests/world_integration_tests.rs:181-203", map=MAP_S0_G_X2,
     script=[reset(), set_state([[1, 0]], [True], events=[[EXIT, 0]]), expect(arrived=[True])])

case("force_wrong_state_check_laser_not_blocked", "tests/world_integration_tests.rs:205-227",
     map="\n        S1  S0 X\n        L0E  G  X\n    ",
     script=[reset(), set_state([[1, 1], [1, 0]], [True], error="InvalidAgentPosition"),
             expect(all_lasers="on", gems=[False])])

case("set_invalid_state_rs", "tests/world_integration_tests.rs:253-273", map="\n        S0 S1 X\n        @  @  X\n    ",
     script=[reset(), set_state([[1, 0], [1, 1]], [], error="InvalidAgentPosition"),
             expect(positions=[[0, 0], [0, 1]])])

MAP_Q1 = """
        S0 .   G  X
        .  .  L2W X
        .  S1  .  X
        . L1N  .  S2"""
case("dead_agent_does_not_block_the_laser", "tests/world_integration_tests.rs:275-309", map=MAP_Q1,
     script=[reset(),
             step([E, N, STAY], event_multiset=[[DIED, 0], [DIED, 1]], derived_events=[[DIED, 1], [DIED, 0]]),
             expect(lasers_on=[[0, 1, True]], derived_beam_bits={"1": [True, False, True]})])

case("world_state_equal", "tests/world_integration_tests.rs:311-330", map="\n        S0 . G\n        .  . X\n    ",
     script=[reset(), expect(positions=[[0, 0]], gems=[False], alive=[True]), step([STAY]),
             expect(positions=[[0, 0]], gems=[False], alive=[True]), step([E]), expect(positions=[[0, 1]])])

MAP_L0W = """
        S0 .   G  X
        .  .  L0W .
        .  S1  .  X
        .  .   .  ."""
case("change_laser_id", "tests/world_integration_tests.rs:332-360", map=MAP_L0W,
     script=[reset(), expect(all_laser_colour=0), source(0, colour=1), expect(all_laser_colour=1, sources=[[1, 2, 1]]),
             step([S, STAY], events=[[DIED, 0]])])
case("disable_laser_source", "tests/world_integration_tests.rs:362-379", map=MAP_L0W,
     script=[reset(), expect(all_lasers="on"), source(0, enabled=False), expect(all_lasers="off"),
             source(0, enabled=True), expect(all_lasers="on")])
case("disable_laser_source_and_block_with_agent", "tests/world_integration_tests.rs:381-407", map="L0E . S0 X",
     script=[reset(), expect(lasers_on=[[0, 1, True]]), source(0, enabled=False), expect(lasers_on=[[0, 1, False]]),
             step([W]), expect(lasers_on=[[0, 2, False]]), step([E]), expect(lasers_on=[[0, 1, False]])])
case("laser_id", "tests/world_integration_tests.rs:409-444",
     map="\n        S0 .   G  X\n        .  .  L0W .\n        .  S1  .  X\n        .  .  L0W  .",
     script=[reset(), expect(laser_ids_by_row={"1": 0, "3": 1}, sources=[[1, 2, 0], [3, 2, 0]])])
case("disable_laser_then_reset_does_not_turn_on", "tests/world_integration_tests.rs:446-456", map="L0E . S0 X",
     script=[reset(), source(0, enabled=False), reset(), expect(lasers_on=[[0, 1, False]], lasers_enabled=[[0, 1, False]])])
case("laser_sources_have_different_laser_ids", "tests/world_integration_tests.rs:458-465", map="L0E . L0E . X S0",
     static={"n_sources": 2}, script=[reset()])
case("laser_on_exit", "tests/world_integration_tests.rs:481-500", map="\n    .   L0S S1\n    S0   .   .\n    L1E  X   X",
     script=[reset(), expect(n_lasers=4, lasers_per_id={"0": 2, "1": 2})])
case("available_joint_actions", "tests/world_integration_tests.rs:502-523", map="S0 . S1 @\n         @   X .  X",
     script=[reset(), expect(avail_sets=[[E, STAY], [S, W, STAY]], n_joint_actions=6)])
case("num_available_joint_actions", "tests/world_integration_tests.rs:525-535", map=" X  S0  .  S1 @\n         S2  @   X  .  X",
     script=[reset(), expect(n_joint_actions=18)])
case("world_state_dead_agents", "tests/world_integration_tests.rs:537-556", map="\n    S0 . G\n    V  . X\n    ",
     script=[reset(), expect(alive=[True]), step([S]), expect(alive=[False]), reset(),
             set_state([[1, 0]], [False], [False]), expect(alive=[False])])
case("blocked_laser_on_spawn", "tests/world_integration_tests.rs:558-577",
     map="\n    . L1S .  X .  .\n    . S1  .  . .  .\n    . S0  .  @ . L0W\n    .  .  .  . .  .\n    @  . L2N . .  X\n    .  .  .  @ .  . ",
     script=[reset(), step([E, E]), reset(), step([E, E])])

# --------------------------------------------------------------------------- tests/tile.rs (tile state machines, restated through maps)
case("tile_gem", "tests/tile.rs:14-50", map="S0 G . X",
     script=[reset(), expect(gems=[False]), step([E], events=[[GEM, 0]]), expect(gems=[True]),
             step([E], n_events=0), expect(gems=[True]), reset(), expect(gems=[False])])
case("tile_laser_agent_survives", "tests/tile.rs:63-82", map="L0E . . .\n. S0 X .",
     script=[reset(), expect(all_lasers="on"), step([N], n_events=0), expect(alive=[True]),
             step([S], n_events=0), expect(alive=[True], all_lasers="on")])
case("tile_laser_agent_dies", "tests/tile.rs:84-94", map="L2E . . .\n. S0 X .",
     script=[reset(), step([N], events=[[DIED, 0]]), expect(alive=[False], all_lasers="on")])
case("tile_void_agent_dies", "tests/tile.rs:96-105", map="S0 V X", script=[reset(), step([E], events=[[DIED, 0]])])

# --------------------------------------------------------------------------- python/tests/test_world.py
case("py_world_tiles", "python/tests/test_world.py:10-14", map="S0 . X",
     static={"start_pos": [[0, 0]], "exit_pos": [[0, 2]]})
MAP_PY_AVAIL = "\n@ @ L0S @  @\n@ .  .  .  @\n@ X  .  S0 @\n@ X  .  S1 @\n@ @  @  @  @\n"
case("py_available_actions", "python/tests/test_world.py:17-33 + python/tests/test_core.py:9-33", map=MAP_PY_AVAIL,
     script=[reset(), expect(avail_sets=[[N, W, STAY], [W, STAY]])])
case("py_parse_wrong_worlds_exits", "python/tests/test_world.py:36-45",
     map="\n            @ @  @ @\n            @ S0 . @\n            @ .  . @\n            @ @  @ @", parse_error="NotEnoughExitTiles")
case("py_parse_wrong_worlds_no_agent", "python/tests/test_world.py:47-49", map="X G", parse_error="NoAgents")
MAP_PY_STEP = "\n        S0 X . .\n        .  . . .\n        .  . . ."
case("py_world_step_one_action", "python/tests/test_world.py:52-62", map=MAP_PY_STEP,
     script=[reset(), step([S], n_events=0), expect(positions=[[1, 0]])])
case("py_world_move", "python/tests/test_world.py:80-89", map="S0 X . .\n.  . . .\n.  . . .",
     script=[reset(), step([S]), step([E]), step([N]), expect(positions=[[0, 1]])])
case("py_world_agents", "python/tests/test_world.py:109-117", map="\n                  S0 S1 S2\n                  X  X  X",
     script=[reset(), expect(alive=[True, True, True])])
case("py_walk_into_wall", "python/tests/test_world.py:120-131",
     map="@ @ @  @ @ @\n@ . .  . . @\n@ . S0 . . @\n@ . .  X . @\n@ @ @  @ @ @",
     script=[reset(), step([S]), step([S], error="InvalidAction", error_agent=0)])
case("py_gem_collected_and_agent_died", "python/tests/test_world.py:134-144", map="\nS0  G  X\nS1 L1N X",
     script=[reset(), step([E, STAY], events=[[DIED, 0]]), expect(n_gems_collected=0, gems=[False])])
case("py_gem_collected_and_agent_has_arrived", "python/tests/test_world.py:147-167", map="\nS0 X . .\n.  . . .\nG  . . .",
     script=[reset(), reset(), step([S]), step([S]), expect(n_gems_collected=1), step([N]), step([N]), step([E]),
             expect(arrived=[True])])
case("py_vertex_conflict", "python/tests/test_world.py:170-182", map="\n        .  X  .  .\n        S0 .  S1  .\n        .  X  .  .",
     script=[reset(), expect(positions=[[1, 0], [1, 2]], gems=[], alive=[True, True]), step([E, W]),
             expect(positions=[[1, 0], [1, 2]], gems=[], alive=[True, True])])
case("py_swapping_conflict", "python/tests/test_world.py:185-198", map="\nS0 X  .  .\n.  .  S1  .\n.  X  .  .",
     script=[reset(), step([S, W]), step([E, W], error="InvalidAction")])
case("py_walk_into_laser_source", "python/tests/test_world.py:201-213",
     map="\n        @ L0S @\n        .  .  .\n        X  .  S0\n        .  .  .",
     script=[reset(), step([W]), step([N]), step([N], error="InvalidAction")])
case("py_walk_outside_map", "python/tests/test_world.py:216-230",
     map="@ @ L0S @  @\n@ .  .  .  @\n@ X  .  S0 @\n@ .  .  .  @\n@ @  .  @  @\n",
     script=[reset(), step([S]), step([W]), step([S]), step([S], error="InvalidAction")])
case("py_world_done", "python/tests/test_world.py:233-250",
     map="\nG  G  . .  S1\nX  .  . @  .\n@  .  G .  .\nG  .  . G  X\n@ L0N . S0 .",
     script=[reset(), step([STAY, W]), step([STAY, W]), step([STAY, W]), step([STAY, W], error="InvalidAction")])
case("py_gems_collected", "python/tests/test_world.py:253-260", map="S0 G X",
     script=[reset(), expect(n_gems_collected=0), step([E]), expect(n_gems_collected=1), step([E]),
             expect(n_gems_collected=1)])
case("py_get_state", "python/tests/test_world.py:318-327", map="S0 G X",
     script=[reset(), expect(positions=[[0, 0]], gems=[False]), step([E]), expect(positions=[[0, 1]], gems=[True])])
case("py_set_state", "python/tests/test_world.py:330-344", map="S0 G X",
     script=[reset(), step([E]), set_state([[0, 0]], [False], events=[]),
             expect(positions=[[0, 0]], n_gems_collected=0),
             set_state([[0, 2]], [True], events=[[EXIT, 0]]), expect(positions=[[0, 2]], n_gems_collected=1)])
MAP_PY_INVALID = "\n        S1  S0 X\n        L0E  G  X"
case("py_set_invalid_state", "python/tests/test_world.py:347-368", map=MAP_PY_INVALID,
     script=[reset(),
             set_state([[0, 0], [0, 1]], [True, True], error="InvalidNumberOfGems"),
             set_state([[0, 0]], [True], error="InvalidNumberOfAgents"),
             set_state([[10, 1], [1, 0]], [True], error="OutOfWorldPosition"),
             set_state([[1, 1], [1, 0]], [True], error="InvalidAgentPosition"),
             set_state([[0, 0], [0, 0]], [True], error="InvalidWorldState")])
case("py_set_invalid_state_dead", "python/tests/test_world.py:371-378", map="\n        S0 L0S X\n        S1  .  X",
     script=[set_state([[0, 0], [0, 1]], [], [True, True], error="InvalidAgentPosition|InvalidWorldState")])
case("py_set_agents_positions_two_agents", "python/tests/test_world.py:426-435",
     map="\n                  S0 . . X\n                  S1 . . X\n                  ",
     script=sum(([set_state([[0, j], [1, j]], []), expect(positions=[[0, j], [1, j]]),
                  set_state([[1, j], [0, j]], []), expect(positions=[[1, j], [0, j]])] for j in range(4)), []))
case("py_set_conflicting_agents_positions", "python/tests/test_world.py:438-444",
     map="\n                  S0 . . X\n                  S1 . . X\n                  ",
     script=[set_state([[0, 0], [0, 0]], [], error="InvalidWorldState")])
case("py_laser_tile_state", "python/tests/test_world.py:454-467", map="L0E S0 . X",
     script=[reset(), expect(n_lasers=3, all_lasers="off"), step([E]),
             expect(lasers_on=[[0, 1, True], [0, 2, False], [0, 3, False]])])
case("py_disable_deadly_laser_source_and_walk_into_it", "python/tests/test_world.py:470-481",
     map="\n        L0S . L0W X\n        S0 S1  .  X\n        ",
     script=[reset(), source(1, enabled=False), step([STAY, N], events=[]), expect(alive=[True, True])])
MAP_COLOUR = "\n        L1E . S1 S0 X\n        L0E .  .  . X\n        "
case("py_change_laser_colour", "python/tests/test_world.py:484-513", map=MAP_COLOUR,
     script=[reset(), expect(n_lasers=8, laser_colour_by_row={"0": 1, "1": 0}), source(1, colour=1), reset(),
             expect(laser_colour_by_row={"1": 1}), step([S, S], events=[]), expect(alive=[True, True])])
case("py_laser_colour_change_remains_after_reset", "python/tests/test_world.py:528-534", map="L0E X X @ S0 S1",
     script=[reset(), source(0, colour=1), reset(), expect(sources=[[0, 0, 1]])])
case("py_change_laser_colour_back", "python/tests/test_world.py:579-603", map=MAP_COLOUR,
     script=[reset(), expect(n_lasers=8), source(1, colour=1), reset(), expect(all_laser_colour=1),
             source(1, colour=0), reset(), expect(laser_colour_by_row={"0": 1, "1": 0})])
case("py_set_state_agent_dead", "python/tests/test_world.py:643-648", map="S0 G X",
     script=[reset(), set_state([[0, 0]], [False], [False]), expect(alive=[False])])
case("py_no_reset", "python/tests/test_world.py:667-669", map="S0 . X", script=[step([E])])
case("py_world_n_agents", "python/tests/test_world.py:672-677", map="S0 S1 X X", static={"n_agents": 2})
case("py_laser_on_start_pos_error", "python/tests/test_world.py:740-748 + src/unit_tests/test_parser_v1.rs:60-78",
     map="\n    S0  S1 X . X\n    L1N .  . . .\n    ", parse_error="AgentWithoutStart")
case("py_laser_sources_in_wall_pos", "python/tests/test_world.py:784-794",
     map="\n        S0 . . X\n       L0E . . X\n        S1 . . X\n       L1E . . X\n",
     static={"wall_pos_contains": [[1, 0], [3, 0]], "sources": [[1, 0, 0], [3, 0, 1]]})
case("py_laser_num_higher_than_n_agents", "python/tests/test_world.py:797-799", map="S0 L1E X",
     static={"sources": [[0, 1, 1]]})
case("py_n_laser_colours", "python/tests/test_world.py:802-829", map="\n        S0 L0E X\n         . L2E X\n        ",
     static={"n_laser_colours": 2})
case("py_n_laser_colours_same", "python/tests/test_world.py:822-829", map="\n        S0 L0E X\n         . L0E X\n        ",
     static={"n_laser_colours": 1})
MAP_RESET_BLOCKED = """
    . .  .  @ .  .  .  . .  .  . . .
    . .  .  . @  X  .  . .  @  . . .
    . .  .  . . L1S .  X .  .  @ . @
    . .  .  . .  .  .  . .  .  . . .
    . S3  . . . S1  .  . .  .  . @ .
    . .  .  @ . S0  .  @ . L0W . . .
    @ .  .  @ .  .  .  . .  .  . . .
    . .  .  . .  .  .  . .  .  . X .
    . .  .  . @  . L2N . .  .  . . .
    . .  S2 . .  .  .  . .  .  . . X
    . .  .  @ .  .  @  . .  @  . . .
    . .  .  . .  .  .  @ .  .  . . ."""
case("py_reset_in_blocked_laser", "python/tests/test_world.py:832-852", map=MAP_RESET_BLOCKED,
     script=[reset(), step([E, E, N, S]), reset(), step([E, E, N, S])])
MAP_MANY = "\n .   .   . . . .\n" + "".join(f"S{k}  L{k}W  . . . X\n" for k in range(14))
case("py_many_agents", "python/tests/test_world.py:855-874", map=MAP_MANY, static={"n_agents": 14, "n_sources": 14})

# --------------------------------------------------------------------------- doc examples (pyworld.rs / __init__.pyi)
case("doc_set_agent_position", "python/lle/world/__init__.pyi:182-190", map="S0 . . X",
     script=[reset(), set_state([[0, 2]], []), step([E], events=[[EXIT, 0]])])
case("doc_gem_at", "python/lle/world/__init__.pyi:198-206", map="S0 G X",
     script=[reset(), expect(gems=[False]), step([E]), expect(gems=[True])])
case("doc_source_at", "python/lle/world/__init__.pyi:215-223", map="S0 L0E X\n.  .   X",
     script=[reset(), expect(sources_enabled=[[0, 1, True]]), source(0, enabled=False), expect(all_lasers="off")])
case("doc_step", "python/lle/world/__init__.pyi:241-254", map="S1 G X S0 X",
     script=[reset(), step([STAY, E], events=[[GEM, 1]]), step([E, E], n_events=2, event_types=[EXIT, EXIT])])
case("doc_available_actions", "python/lle/world/__init__.pyi:268-275", map="S0 @ X",
     script=[reset(), expect(avail_excludes=[[E]], avail_includes=[[STAY]])])
case("doc_available_joint_actions", "python/lle/world/__init__.pyi:284-290", map=". .  .  . .\n. S0 . S1 .\n. X  .  X .\n",
     script=[reset(), expect(n_joint_actions=25)])

# --------------------------------------------------------------------------- src/unit_tests/test_parser_v1.rs
case("parser_v1_invalid_tile", "src/core/parsing/parser_v1.rs:165-171", map="S0 ? X", parse_error="InvalidTile")
case("parser_v1_invalid_agent_id", "src/core/parsing/parser_v1.rs:150-155", map="Sx . X", parse_error="InvalidAgentId")

# --------------------------------------------------------------------------- python/tests/test_observations.py (layered)
case("obs_layered_deactivated_laser", "python/tests/test_observations.py:95-123", map=MAP_PY_AVAIL,
     script=[reset(),
             expect(obs_shape=[8, 5, 5],
                    obs_cells=[[2, 0, 2, -1], [2, 1, 2, 1], [2, 2, 2, 1], [2, 3, 2, 1]], obs_layer_all=[[3, 0]]),
             step([W, STAY]),
             expect(obs_cells=[[2, 0, 2, -1], [2, 2, 2, 0], [2, 3, 2, 0]], obs_layer_all=[[3, 0]])])
case("obs_layered_gems_walls", "python/tests/test_observations.py:126-159",
     map="\n@ @ L0S @  @\n@ .  .  .  @\n@ X  G  S0 @\n@ .  .  .  @\n@ @  @  @  @\n",
     script=[reset(), expect(obs_consistent=True, obs_layer_all=[[3, 0]])])
case("obs_layered_void", "python/tests/test_observations.py:162-178", map="\n    V . . S0\n    . . . .\n    V V G X",
     script=[reset(), expect(obs_layer_exact={"3": [[0, 0], [2, 0], [2, 1]]})])
case("obs_flattened_shape", "python/tests/test_observations.py:181-200",
     map="\n@ @ L0S @  @\n@ .  .  .  @\n@ X  G  S0 @\n@ .  .  .  @\n@ @  @  @  @\n",
     script=[reset(), expect(obs_shape=[6, 5, 5])])
case("obs_laser_colour_above_n_agents", "python/tests/test_observations.py:500-506", map="S0 L1E X",
     script=[expect(obs_cells=[[2, 0, 1, -1], [2, 0, 2, 1]])])
for lvl in range(1, 7):
    case(f"obs_all_shapes_level_{lvl}", "python/tests/test_observations.py:509-518", level=lvl,
         script=[reset(), expect(obs_shape_formula=True, obs_consistent=True)])

# --------------------------------------------------------------------------- round 2: the remaining on-path tests
# (ops added for them: clone_check = World::clone / deepcopy -- a new world from the config + set_state(get_state()),
#  world.rs:645-652, pyworld.rs:557-559; set_agent_position = pyworld.rs:282-299; colour = the BINDING's checked setter
#  LaserSource.set_colour / agent_id, src/bindings/tiles/pylaser_source.rs:107-147, unlike op "source" which is the core
#  LaserSource::set_agent_id)
def clone_check():
    return {"op": "clone_check"}


def set_agent_position(agent, pos, **kw):
    return {"op": "set_agent_position", "agent": agent, "pos": list(pos), **kw}


def colour(laser_id, value, **kw):
    return {"op": "colour", "laser_id": laser_id, "colour": value, **kw}


case("clone_after_step", "src/unit_tests/test_world.rs:283-300", map=MAP_S0_G_X2,
     script=[reset(), step([E]), step([E]), clone_check()])
# tests/tile.rs:52-62 builds one Laser tile (colour 0, beam of 4) around a floor: it is on, and after reset walkable and
# not occupied.  As a world: the agent below the first beam cell may walk NORTH onto it (walkable + not occupied).
case("tile_laser_basic", "tests/tile.rs:52-62", map="L0E . . . . @\n .  S0 . . . X",
     script=[reset(), expect(laser_colours=[[0, 1, 0]], lasers_on=[[0, 1, True], [0, 4, True]], beam_len={"0": 4},
                             avail_includes=[[N]])])
case("py_deepcopy", "python/tests/test_world.py:300-305", map="S0 . X", script=[clone_check()])
case("py_deepcopy_not_initial_state", "python/tests/test_world.py:308-315", map="S0 . X",
     script=[reset(), step([E]), clone_check()])
case("py_set_agent_position", "python/tests/test_world.py:410-414", map="S0 . . X",
     script=[x for j in range(4) for x in (set_agent_position(0, (0, j)), expect(positions=[[0, j]]))])
case("py_set_wrong_agent_position", "python/tests/test_world.py:417-423", map="S0 . . X",
     script=[set_agent_position(25, (0, 0), error="AgentIdOutOfBounds"),      # ValueError in the binding
             set_agent_position(0, (0, 25), error="OutOfWorldPosition")])      # IndexError in the binding
case("py_change_laser_colour_to_negative_colour", "python/tests/test_world.py:516-525", map="L0E S0 . X",
     script=[reset(), colour(0, -1, error="OverflowError"), expect(derived_sources=[[0, 0, 0]])])
# the binding sets the CORE colour before it checks the start positions (pylaser_source.rs:108-119 vs :121-139): the
# refused change has recoloured the world, only the caller's snapshot keeps the old id (derived from the source)
case("py_laser_colour_change_kills_agent_on_start", "python/tests/test_world.py:537-545", map="L0E X X S0 S1",
     script=[reset(), colour(0, 1, error="ValueError"), expect(derived_sources=[[0, 0, 1]], derived_all_laser_colour=1)])
case("py_change_laser_colour_to_invalid_colour", "python/tests/test_world.py:548-575", map="L0E S0 . X",
     script=[reset(), colour(0, 2, error="ValueError"), colour(0, 1, error="ValueError"),
             colour(0, 2, error="ValueError"), colour(0, 1, error="ValueError"),   # (again through the `agent_id` setter)
             expect(derived_sources=[[0, 0, 0]])])                                  # refused before anything is set (:109-113)
case("py_pickled_world_keeps_same_laser_ids", "python/tests/test_serialization.py:41-50", map="L0E L1S S0 S1 X X",
     static={"sources": [[0, 0, 0], [0, 1, 1]]}, script=[clone_check()])
# NOT transcribable: python/tests/test_world.py:751-762 (test_laser_on_start_pos_removed) and python/tests/test_env.py:144-180
# (test_set_state) build their worlds from TOML with several / random start positions: v2 maps are out of scope
# (SURVEY.md section 2 row 5).  Their v1 counterparts are py_laser_on_start_pos_error above and the get_state -> set_state
# round trip of tests/test_gpu_env.py::test_set_state_round_trip_along_a_rollout.

MAP_MULTI_DIGIT = " .   .   . . . .\n" + "".join(f"S{k} L{k}W . . . X\n" for k in range(14))
case("parser_v1_multi_digit_agents_and_sources", "src/unit_tests/test_parser_v1.rs:22-57", map=MAP_MULTI_DIGIT,
     static={"n_agents": 14, "n_sources": 14, "sources": [[k + 1, 1, k] for k in range(14)]},
     script=[reset(), expect(sources=[[k + 1, 1, k] for k in range(14)])])
# `S0` lies on the beam of colour 1 but BEHIND agent 1's only start, which blocks it: the start is kept and the world builds
case("parser_v1_laser_blocked_on_spawn", "src/unit_tests/test_parser_v1.rs:80-98", map="\n    L1E . S1 S0 X\n    L0E .  .  . X\n    ",
     static={"n_agents": 2, "start_pos": [[0, 3], [0, 2]]}, script=[reset(), expect(alive=[True, True])])
# NOT transcribable as they stand: test_parser_v1.rs:5-20 parse `S10 X` / `L10E S0 ... S10 X` to a CONFIG only (they would not
# survive into_world: ten agents without start / one exit for eleven agents); the map above covers multi-digit ids end to end.

case("readme_world_example", "readme.md:56-67", map="S0 G X",
     script=[reset(), expect(avail_sets=[[STAY, E]]), step([E], event_types=[GEM]), step([E], event_types=[EXIT])])
case("doc_deepcopy", "python/lle/world/__init__.pyi:313-324", map="S0 X", script=[reset(), clone_check()])


# --------------------------------------------------------------------------- World.exit_pos = [...] (world.rs:195-234)
def set_exits(exits, **kw):
    return {"op": "set_exits", "exits": [list(p) for p in exits], **kw}


MAP_SET_EXITS = "\n        S0 . G\n        X  . .\n    "
case("set_exits", "src/unit_tests/test_world.rs:583-602", map=MAP_SET_EXITS,
     script=[reset(), expect(n_exits=1, exit_pos_contains=[[1, 0]]), set_exits([(0, 1), (1, 1)]),
             expect(n_exits=2, exit_pos_contains=[[0, 1], [1, 1]])])
case("set_exits_events", "src/unit_tests/test_world.rs:604-641", map=MAP_SET_EXITS,
     script=[reset(), set_exits([(0, 1), (1, 1)]), step([E], n_events=1, event_multiset=[[EXIT, 0]])])
case("set_exits_old_exit_inactive", "src/unit_tests/test_world.rs:643-660", map=MAP_SET_EXITS,
     script=[reset(), set_exits([(0, 1), (1, 1)]), step([S], n_events=0)])
case("py_set_exit_positions", "python/tests/test_world.py:764-781", map="S0 . X",
     script=[reset(), expect(exit_pos=[[0, 2]]), set_exits([(0, 1)]), reset(), expect(exit_pos=[[0, 1]]),
             step([E], event_types=[EXIT]),
             set_exits([(0, 2)]), reset(), expect(exit_pos=[[0, 2]]), step([E], n_events=0), step([E], event_types=[EXIT])])
# (A = 1: EXIT is layer 2A+3 = 5.  The reference's generator caches its static layers until `observer.reset()`; the test calls it.)
case("py_observe_layered_change_exits", "python/tests/test_observations.py:76-92", map="S0 X . .",
     static={"exit_pos": [[0, 1]]},
     script=[set_exits([(0, 2), (0, 3)]), reset(), expect(obs_cells=[[5, 0, 2, 1], [5, 0, 3, 1]], derived_obs_layer_exact={"5": [[0, 2], [0, 3]]})])
# ---- derived from the source (world.rs:195-234, laser.rs:109-115); not asserted by any reference test
case("derived_set_exits_not_enough", "src/core/world.rs:196-201", map="S0 S1 X X",
     script=[reset(), set_exits([(0, 2)], error="NotEnoughExitTiles"), expect(exit_pos=[[0, 2], [0, 3]]),
             set_exits([], error="NotEnoughExitTiles"), expect(exit_pos=[[0, 2], [0, 3]])])
# `other => panic!("Tile is not a floor")` (:230) / Vec index out of bounds: refused up front, world untouched (lle_hip.h)
case("derived_set_exits_where_the_reference_panics", "src/core/world.rs:219-232", map="S0 . @ G V\n . . . . .\nL0E . X . .",
     script=[reset()] + [x for bad in ([(0, 2)], [(0, 3)], [(0, 4)], [(2, 0)], [(7, 0)], [(0, 9)], [(1, 3), (1, 3)])
                         for x in (set_exits(bad, error="Panic"), expect(exit_pos=[[2, 2]]))] +
            [step([S], n_events=0), step([E], n_events=0), step([E], n_events=0), step([S], event_types=[EXIT])])
# ... but under a beam the same cell twice is fine: the second Laser::set_tile finds a Laser again, not an Exit (:221-227)
case("derived_set_exits_twice_the_same_cell_under_a_laser", "src/core/world.rs:221-227", map="S0 . @ G V\n . . . . .\nL0E . X . .",
     script=[reset(), set_exits([(2, 3), (2, 3)]), expect(exit_pos=[[2, 3], [2, 3]], obs_cells=[[5, 2, 2, 0], [5, 2, 3, 1]])])
# the exit under a beam: Laser::set_tile swaps the innermost tile (laser.rs:109-115); the old exit is a plain floor afterwards
case("derived_set_exits_under_a_laser", "src/core/world.rs:204-228", map="L0E . . .\n S0 . X .",
     script=[reset(), set_exits([(0, 2)]), expect(exit_pos=[[0, 2]], obs_cells=[[5, 0, 2, 1], [5, 1, 2, 0]], n_lasers=3),
             step([E], n_events=0), step([E], n_events=0), expect(positions=[[1, 2]], arrived=[False]),
             step([N], events=[[EXIT, 0]]), expect(positions=[[0, 2]], arrived=[True], beam_bits={"0": [True, False, False]}),
             set_exits([(1, 2)]), expect(arrived=[True], tile_agent=[[0, 2, 0]], obs_cells=[[5, 0, 2, 0], [5, 1, 2, 1]]),
             reset(), expect(arrived=[False], all_lasers="on"), step([E], n_events=0), step([E], events=[[EXIT, 0]])])
# whoever stands on a swapped tile stays its occupant, and `has_arrived` is the agent's, not the tile's (:207,224)
case("derived_set_exits_keeps_occupant_and_arrival", "src/core/world.rs:204-228", map="S0 X . S1\n . . . X",
     script=[reset(), step([E, STAY], events=[[EXIT, 0]]), set_exits([(0, 2), (1, 0)]),
             expect(positions=[[0, 1], [0, 3]], arrived=[True, False], tile_agent=[[0, 1, 0], [0, 3, 1], [0, 2, -1]],
                    avail_sets=[[STAY], [STAY, S, W]]),
             step([STAY, W], events=[[EXIT, 1]]), expect(arrived=[True, True]),
             reset(), expect(arrived=[False, False]), step([E, STAY], n_events=0), step([E, STAY], events=[[EXIT, 0]])])
# an agent whose START becomes an exit arrives at the next reset (world.rs:411-432 enters every agent; events dropped)
case("derived_set_exits_on_a_start", "src/core/world.rs:411-432", map="S0 . X",
     script=[reset(), set_exits([(0, 0)]), expect(arrived=[False], avail_sets=[[STAY, E]]),
             reset(), expect(arrived=[True], avail_sets=[[STAY]]), step([E], error="InvalidAction", error_agent=0)])
# a void under a beam turned exit: Laser::set_tile replaces the innermost tile whatever it is (laser.rs:109-115) -- the agent
# survives where it would have died; `void_pos` still lists the cell (world.rs:218 only replaces `exits`): A = 1, VOID = 3, EXIT = 5
case("derived_set_exits_replaces_a_void_under_a_laser", "src/core/tiles/laser.rs:109-115", map="L0S . X\n V S0 .\n . . .",
     script=[reset(), set_exits([(1, 0)]), expect(obs_cells=[[3, 1, 0, 1], [5, 1, 0, 1], [5, 0, 2, 0]]),
             step([W], events=[[EXIT, 0]]), expect(alive=[True], arrived=[True], beam_bits={"0": [False, False]}),
             set_exits([(0, 2)]), reset(), step([W], n_events=0), expect(alive=[True], positions=[[1, 0]], obs_cells=[[3, 1, 0, 1], [5, 1, 0, 0]])])
case("derived_set_exits_then_clone", "src/core/world.rs:645-652", map=MAP_SET_EXITS,
     script=[reset(), set_exits([(0, 1), (1, 1)]), step([E], events=[[EXIT, 0]]), clone_check()])



# ---- Gem.collect() of the bindings (src/bindings/tiles/pygem.rs:52-66 -> tiles/gem.rs:17-19): the reference has no test for it
# (python/tests/test_tiles.py:6 lists it as to do); derived from the two functions.  Layers for A = 1: gem = 4.
def collect_gem(pos, **kw):
    return {"op": "collect_gem", "pos": list(pos), **kw}


# collected without an event; entering it later collects nothing (gem.rs:26-33); reset puts it back (gem.rs:21-24)
case("derived_gem_collect", "src/bindings/tiles/pygem.rs:52-66", map="S0 G . X",
     script=[reset(), expect(gems=[False], obs_cells=[[4, 0, 1, 1]]), collect_gem((0, 1)),
             expect(gems=[True], n_gems_collected=1, obs_cells=[[4, 0, 1, 0]], positions=[[0, 0]]),
             step([E], n_events=0), expect(tile_agent=[[0, 1, 0]]), step([E], n_events=0), step([E], events=[[EXIT, 0]]),
             reset(), expect(gems=[False], obs_cells=[[4, 0, 1, 1]]), step([E], events=[[GEM, 0]])])
# `inner` is World::at_mut: a gem under a beam is a Laser tile there, and so is any other cell -> ValueError, nothing changes
case("derived_gem_collect_refused", "src/bindings/tiles/pygem.rs:52-62", map="S0 . X\nL0E G .",
     script=[reset(), collect_gem((1, 1), error="ValueError"), collect_gem((0, 1), error="ValueError"),
             expect(gems=[False], n_gems_collected=0, obs_cells=[[4, 1, 1, 1]])])
# with the collector standing on ANOTHER gem: only the addressed bit moves
case("derived_gem_collect_one_of_two", "src/bindings/tiles/pygem.rs:52-66", map="S0 G G X",
     script=[reset(), step([E], events=[[GEM, 0]]), collect_gem((0, 2)), expect(gems=[True, True], n_gems_collected=2),
             step([E], n_events=0), step([E], events=[[EXIT, 0]])])

# --------------------------------------------------------------------------- beams longer than 32 cells (derived)
# `LaserBeam.beam` is a Vec<bool> (src/core/tiles/laser.rs:15-21): no bound on its length; turn_on / turn_off run from an offset to the
# END of the Vec (:50-59).  No reference test holds a beam of more than 12 cells, so everything here is derived from those lines and
# from move_agents (src/core/world.rs:477-505); the kernels store such a beam as a chain of 32-cell words (tables.h), and these cases sit
# on the word boundary (offsets 31 / 32) and behind it.
def _grid(h, w, cells):
    rows = [["."] * w for _ in range(h)]
    for (i, j), t in cells.items():
        rows[i][j] = t
    return "\n".join(" ".join(r) for r in rows)


# 3 x 40: the beam of L0E at (0, 0) covers (0, 1) ... (0, 38): 38 cells, offset = column - 1; agents walk up into it from row 1
LONG = _grid(3, 40, {(0, 0): "L0E", (0, 39): "@", (1, 36): "S0", (1, 38): "S1", (2, 0): "X", (2, 1): "X"})
LONG_FAR = _grid(3, 40, {(0, 0): "L0E", (0, 39): "@", (1, 11): "S0", (1, 38): "S1", (2, 0): "X", (2, 1): "X"})
ON38 = [True] * 38

case("derived_long_beam_traced_and_lit", "src/core/parsing/world_config.rs:203-250", map=LONG,
     static={"n_sources": 1, "sources": [[0, 0, 0]]},
     script=[reset(), expect(beam_len={"0": 38}, beam_bits={"0": ON38}, n_lasers=38, all_lasers="on",
                             lasers_on=[[0, 1, True], [0, 32, True], [0, 33, True], [0, 38, True]],
                             obs_cells=[[2, 0, 0, -1], [2, 0, 1, 1], [2, 0, 32, 1], [2, 0, 33, 1], [2, 0, 38, 1], [2, 0, 39, 0]])])
# the owner walks in at offset 35 (second word): bits 35 .. 37 go off, nothing before them (laser.rs:173-182, :57-59)
case("derived_long_beam_owner_cuts_at_offset_35", "src/core/tiles/laser.rs:173-182", map=LONG,
     script=[reset(), step([N, STAY], n_events=0),
             expect(positions=[[0, 36], [1, 38]], alive=[True, True], beam_bits={"0": [True] * 35 + [False] * 3},
                    lasers_on=[[0, 35, True], [0, 36, False], [0, 37, False], [0, 38, False]],
                    obs_cells=[[2, 0, 35, 1], [2, 0, 36, 0], [2, 0, 38, 0]]),
             step([S, STAY], n_events=0), expect(beam_bits={"0": ON38})])  # it leaves: re-lit from offset 35 on (laser.rs:157-162)
# another agent walks into the lit tile at offset 37: dies there (laser.rs:184-197) -- and lives when the owner cuts upstream in the same step
case("derived_long_beam_death_at_offset_37", "src/core/tiles/laser.rs:184-197", map=LONG,
     script=[reset(), step([STAY, N], events=[[DIED, 1]]), expect(alive=[True, False], beam_bits={"0": ON38}),
             reset(), step([N, N], n_events=0), expect(alive=[True, True], positions=[[0, 36], [0, 38]], beam_bits={"0": [True] * 35 + [False] * 3})])
# the owner cuts in the FIRST word (offset 10): every bit behind it goes off, the whole second word included; leaving re-lights them all
case("derived_long_beam_cut_and_relight_across_the_word_boundary", "src/core/tiles/laser.rs:50-59", map=LONG_FAR,
     script=[reset(), step([N, STAY], n_events=0), expect(beam_bits={"0": [True] * 10 + [False] * 28}, lasers_on=[[0, 32, False], [0, 33, False], [0, 38, False]]),
             step([STAY, N], n_events=0), expect(alive=[True, True], positions=[[0, 11], [0, 38]]),   # offset 37 is dark: safe
             step([E, STAY], n_events=0), expect(beam_bits={"0": [True] * 11 + [False] * 27}),         # re-lit from 10, cut again from 11
             step([S, STAY], events=[[DIED, 1]]),                                                      # the owner leaves: all on again, agent 1 stands in it
             expect(alive=[True, False], beam_bits={"0": ON38})])
# the owner starts ON its beam behind the boundary: World::reset cuts from there (world.rs:411-432 -> laser.rs:173-182)
case("derived_long_beam_owner_starts_at_offset_33", "src/core/world.rs:411-432",
     map=_grid(2, 40, {(0, 0): "L0E", (0, 34): "S0", (0, 39): "@", (1, 0): "X"}),
     script=[reset(), expect(beam_bits={"0": [True] * 33 + [False] * 5}, lasers_on=[[0, 33, True], [0, 34, False], [0, 38, False]]),
             step([S], n_events=0), expect(beam_bits={"0": ON38}), reset(), expect(beam_bits={"0": [True] * 33 + [False] * 5})])
# quirk Q1 (stale re-light) ACROSS the boundary.  L1S at (0, 12) lights (1, 12) only (a wall below); L0E at (1, 0) lights (1, 1) ... (1, 38).
# Step 1: the owner of the long beam (agent 0) walks in at offset 10.  Step 2: it moves on to the crossing at offset 11 -- leave re-lights
# from 10, pre_enter cuts from 11, enter: the inner beam (colour 1) is lit -> it dies; agent 1 walks into offset 33 (dark: cut) and lives.
# Pass 2 (somebody died): agent 1 leaves its dark tile -> re-lights 33 .. 37; the dead owner no longer cuts; agent 1 re-enters a lit tile of
# another colour -> dies (tests/world_integration_tests.rs:279-309 is the same mechanism on 3 cells).  Final beam: on x 11, off x 22, on x 5.
Q1_LONG = _grid(3, 40, {(0, 12): "L1S", (1, 0): "L0E", (1, 39): "@", (2, 0): "X", (2, 1): "X", (2, 11): "S0", (2, 12): "@", (2, 34): "S1"})
case("derived_long_beam_stale_relight_across_the_word_boundary", "src/core/world.rs:464-505", map=Q1_LONG,
     static={"n_sources": 2, "sources": [[0, 12, 1], [1, 0, 0]]},
     script=[reset(), expect(beam_len={"0": 1, "1": 38}),
             step([N, STAY], n_events=0), expect(beam_bits={"1": [True] * 10 + [False] * 28}),
             step([E, N], events=[[DIED, 0], [DIED, 1]]),
             expect(alive=[False, False], positions=[[1, 12], [1, 34]], beam_bits={"0": [True], "1": [True] * 11 + [False] * 22 + [True] * 5},
                    lasers_on=[[1, 11, True], [1, 12, False], [1, 32, False], [1, 33, False], [1, 34, True], [1, 38, True]])])
# LaserSource.disable / enable on a long beam: every cell off / on (laser.rs:69-77); a colour change recolours every cell (laser.rs:79-86)
case("derived_long_beam_disable_enable_recolour", "src/core/tiles/laser.rs:69-86", map=LONG,
     script=[reset(), source(0, enabled=False), expect(all_lasers="off", beam_bits={"0": [False] * 38}),
             step([STAY, N], n_events=0), expect(alive=[True, True]),
             source(0, enabled=True), expect(all_lasers="on", beam_bits={"0": ON38}),
             source(0, colour=1), expect(all_laser_colour=1, obs_cells=[[3, 0, 0, -1], [3, 0, 33, 1], [2, 0, 33, 0]])])
# get_state / set_state with a long beam: the beams are re-derived from positions and alive flags (world.rs:515-597)
case("derived_long_beam_set_state", "src/core/world.rs:515-597", map=LONG,
     script=[reset(), set_state([[0, 34], [1, 38]], [], n_events=0), expect(beam_bits={"0": [True] * 33 + [False] * 5}),
             set_state([[1, 36], [0, 38]], [], alive=[True, False], events=[[DIED, 1]]), expect(alive=[True, False], beam_bits={"0": ON38}),
             set_state([[1, 36], [0, 38]], [], error="InvalidWorldState")])  # (asked alive, dies on entering: world.rs:588-594)

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_world.json")
    with open(out, "w") as f:
        json.dump({"codes": {"actions": ["N", "S", "E", "W", "STAY"], "events": ["EXIT", "GEM", "DIED"]}, "cases": CASES},
                  f, indent=1)
    print(f"wrote {len(CASES)} cases to {out}")
