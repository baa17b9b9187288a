"""Known-answer tests for the LLE host class (reward strategies, done, set_state, per-agent metrics), hand-transcribed
from the reference's python/tests/test_env.py.  Same rules as make_kat.py: each case restates ONE reference test as data
(map, script, the assertions that test makes; `ref` = file:line); nothing here imports or executes the reference.

Running this file rewrites tests/golden/kat_env.json.

Script ops: {"op": "expect", "done": bool | null, "available": [[actions of agent 0], ...] | null,
             "derived_available": ...}  (LLE.done / LLE.available_actions as sets of Action values; `derived_` = computed by
             the reference's code at that point but not asserted by its test)
            {"op": "reset"} | {"op": "step", "actions": [...], "reward": r | [gem, exit, death, done] | null,
             "done": bool | null, "metrics": {"has-arrived": [...], "is-alive": [...]} | null}
            {"op": "set_state", "positions": [[i, j]...], "gems": [...], "alive": [...] | null}
            (set_state = LLE.set_state, python/lle/env/env.py:208-217: reward counters restart, World.set_state, the
             events of set_state count towards arrivals / deaths, done is recomputed)
Codes: actions N=0 S=1 E=2 W=3 STAY=4.  Reward constants of python/lle/env/reward_strategy.py:20-23:
GEM = EXIT = DONE = 1, DEATH = -1.
"""
import json
import os

N, S, E, W, STAY = 0, 1, 2, 3, 4
GEM = EXIT = DONE = 1.0
DEATH = -1.0
CASES = []


def case(name, ref, map, script, multi_objective=False):
    CASES.append({"name": name, "ref": ref, "map": map, "multi_objective": multi_objective, "script": script})


reset = {"op": "reset"}


def step(actions, reward=None, done=None, metrics=None):
    return {"op": "step", "actions": list(actions), "reward": reward, "done": done, "metrics": metrics}


case("void_reward", "python/tests/test_env.py:10-15", "S0 V X", [reset, step([E], DEATH, True)])
case("collect_reward", "python/tests/test_env.py:18-27", "S0 X . .\n.  . . .\nG  . . .", [reset, step([S]), step([S], GEM)])
case("time_reward", "python/tests/test_env.py:30-40", "\n    . .  . X\n    . S0 . .\n    . .  . .",
     [reset] + [step([a], 0.0) for a in (N, S, E, W, STAY)])
case("finish_reward", "python/tests/test_env.py:43-54",
     "@ @ @  @ @ @\n@ . .  . . @\n@ . S0 . . @\n@ . .  X . @\n@ @ @  @ @ @", [reset, step([E]), step([S], DONE + EXIT)])
case("arrive_reward_only_once", "python/tests/test_env.py:57-77", "\n    S0 . G\n    S1 X X\n",
     [reset, step([E, STAY], 0), step([STAY, E], 1), step([STAY, STAY], 0), step([STAY, STAY], 0), step([E, STAY], 1),
      step([STAY, STAY], 0), step([S, STAY], 2)])
_play = [reset, step([S]), step([S], GEM, False), step([N], 0), step([N], 0), step([E], DONE + EXIT, True)]
case("reward_after_reset", "python/tests/test_env.py:80-105", "\n    S0 X . .\n    .  . . .\n    G  . . .\n    ", _play * 3)
case("reward_after_set_state", "python/tests/test_env.py:108-120", "\n    S0 . G\n    S1 X X",
     [reset, {"op": "set_state", "positions": [[0, 1], [1, 1]], "gems": [False], "alive": None}, step([E, STAY], GEM)])
case("reward_set_state_all_arrived", "python/tests/test_env.py:123-141", "\n    S0 . G\n    S1 X X",
     # the test sets the world state, reads it back as an env state, resets, and hands that state to LLE.set_state
     [reset, {"op": "set_state", "positions": [[0, 2], [1, 1]], "gems": [True], "alive": None}, step([S, STAY], DONE + EXIT)])
case("reward", "python/tests/test_env.py:183-193", "\n    S0 G .\n    .  . X\n    ",
     [reset, step([E], GEM), step([E], 0.0), step([S], EXIT + DONE)])
case("reward_death", "python/tests/test_env.py:196-206", "\n    S0 L0S X\n    S1  .  X\n    ", [reset, step([STAY, E], DEATH, True)])
case("step_info_arrival_metrics", "python/tests/test_env.py:209-224", "\n    S0 X\n    S1 X\n    ",
     [reset, step([E, STAY], metrics={"has-arrived": [True, False], "is-alive": [True, True]})])
case("step_info_death_metrics", "python/tests/test_env.py:227-242", "\n    S0 L0S X\n    S1  .  X\n    ",
     [reset, step([STAY, E], metrics={"has-arrived": [False, False], "is-alive": [True, False]})])
case("reward_collect_and_death", "python/tests/test_env.py:245-255", "\n    S0 L0S X\n    S1  G  X\n    ",
     [reset, step([STAY, E], DEATH, True)])
case("multi_objective_rewards", "python/tests/test_env.py:258-284", "\n    S0 G .\n    .  . X\n    ",
     [reset, step([E], [1.0, 0, 0, 0]), step([E], [0, 0, 0, 0]), step([S], [0, EXIT, 0, DONE], True)], multi_objective=True)
case("multi_objective_death", "python/tests/test_env.py:287-302", "\n    S0 L0S X\n    S1  G  X\n    ",
     [reset, step([STAY, E], [0, 0, DEATH, 0])], multi_objective=True)


def expect(**kw):
    return {"op": "expect", **kw}


# ---- python/tests/test_core.py (round 2)
MAP_CORE_AVAIL2 = """
@   @ @  @  @ @ @
@   . S0 S1 . . @
@   . .  .  . . @
L0E . .  .  . . @
@   . .  .  . G @
@   . X  X  . . @
@   @ @  @  @ @ @"""
# the reference asserts the mask after reset; inside its loop the check's result is dropped (test_core.py:63-70), so the
# later masks are `derived_`: what LLE.available_actions returns there
_mid = [[N, S, W, STAY], [N, S, E, STAY]]
case("core_available_actions2", "python/tests/test_core.py:36-70", MAP_CORE_AVAIL2,
     [reset, expect(available=[[S, W, STAY], [S, E, STAY]]),
      step([S, S]), expect(derived_available=_mid), step([S, S]), expect(derived_available=_mid),
      step([S, S]), expect(derived_available=_mid), step([S, S]), expect(derived_available=[[STAY], [STAY]])])
case("core_move_end_game", "python/tests/test_core.py:144-161", "\n    S0 X .\n    .  . .\n    .  . .",
     [reset, step([S], done=False), step([S], done=False), step([E], done=False), step([N], done=False), step([N], done=True)])
case("core_force_end_state", "python/tests/test_core.py:164-175", "\n        S0 . G\n        X  . .\n    ",
     [reset, {"op": "set_state", "positions": [[1, 0]], "gems": [True], "alive": None}, expect(done=True)])
# the reference then expects ValueError from env.step: "Cannot step in a done environment" (env.py:166-167) -- agent 1
# is revived by Agent::reset inside set_state, enters the lit beam of colour 0 and dies with an event (world.rs:571-579)
case("core_force_state_agent_dies", "python/tests/test_core.py:178-192", "\n        S0 S1 G\n        X  . L0W\n        .  X  .\n    ",
     [reset, {"op": "set_state", "positions": [[1, 0], [1, 1]], "gems": [False], "alive": [True, False]}, expect(done=True)])

# LLE.done counts EVENTS since the last reset / set_state (env.py:208-217,253-254; reward_strategy.py:58-75): an agent that a forced
# state flags dead on a tile that does not kill it dies WITHOUT an AgentDied (world.rs:571-579), so the episode goes on -- derived
# from the source, no reference test forces such a state; test_core.py:178-192 (above) is the case where the event does fire.
case("derived_forced_dead_agent_without_event", "python/lle/env/env.py:208-217 + src/core/world.rs:571-579 (derived)",
     "\n        S0 S1 G\n        X  . .\n        .  X  .\n    ",
     [reset, {"op": "set_state", "positions": [[1, 1], [0, 1]], "gems": [False], "alive": [True, False]}, expect(done=False),
      step([W, STAY], done=False, metrics={"has-arrived": [True, False], "is-alive": [True, False]}), expect(done=False),
      step([STAY, STAY], done=False)])

# ---- python/tests/test_death_strategy.py, python/tests/test_reward_strategy.py (round 2)
case("death_strategy_end", "python/tests/test_death_strategy.py:4-18", "\nS0  G  X\nS1 L1N X\n", [reset, step([E, STAY], done=True)])
# test_reward_strategy.py feeds hand-built event lists to the strategies; the same lists from real steps.  `free_running`: the
# strategy is exercised past the end of the episode (the reference's LLE would refuse the step of a done env, its strategy
# objects do not care), so adapters must not enforce "Cannot step in a done environment" on these cases.
MAP_RS_ARRIVE = "S0 G X\nS1 . X"           # step 1: [GEM 0]; step 2: [EXIT 0, EXIT 1]
MAP_RS_DEATHS = "S0 V X\nS1 . V\nX  . ."    # step 1: [DIED 0]; step 2: [DIED 1]
case("reward_strategy_single_objective", "python/tests/test_reward_strategy.py:6-32", MAP_RS_ARRIVE,
     [reset, step([E, E], GEM), step([E, E], 2 * EXIT + DONE, True, {"has-arrived": [True, True], "is-alive": [True, True]}),
      reset, step([E, E]), step([E, E], 2 * EXIT + DONE)])
case("reward_strategy_single_objective_deaths", "python/tests/test_reward_strategy.py:24-32", MAP_RS_DEATHS,
     [reset, step([E, E], DEATH, True), reset, step([E, E], DEATH, True), step([STAY, E], DEATH, True, {"has-arrived": [False, False], "is-alive": [False, False]})])
CASES[-1]["free_running"] = True
case("reward_strategy_multi_objective", "python/tests/test_reward_strategy.py:35-62", MAP_RS_ARRIVE,
     [reset, step([E, E], [GEM, 0, 0, 0]), step([E, E], [0, 2 * EXIT, 0, DONE], True)], multi_objective=True)
case("reward_strategy_multi_objective_deaths", "python/tests/test_reward_strategy.py:54-62", MAP_RS_DEATHS,
     [reset, step([E, E], [0, 0, DEATH, 0], True), reset, step([E, E], [0, 0, DEATH, 0]), step([STAY, E], [0, 0, DEATH, 0])], multi_objective=True)
CASES[-1]["free_running"] = True

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_env.json")
    with open(out, "w") as f:
        json.dump({"format": 1, "cases": CASES}, f, indent=1)
    print(f"wrote {len(CASES)} cases to {out}")
