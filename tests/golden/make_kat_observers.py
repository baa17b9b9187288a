"""Known-answer tests for the observation builders other than the layered tensor,
hand-transcribed from the reference's own tests: python/tests/test_observations.py and
python/tests/test_walkable_lasers.py.  Same rules as make_kat.py: each case restates ONE reference test as data (map,
script of calls, the assertions that test makes, `ref` = file:line); nothing here imports or executes the reference.

Running this file rewrites tests/golden/kat_observers.json.

Script ops:   {"op": "reset"} | {"op": "step", "actions": [...]} |
              {"op": "check", "obs": <kind>, "param": p, <assertions>}      kinds: state, normalized-state, layered,
                    layered-padded, flattened, partial, perspective
              {"op": "check_avail", "walkable_lasers": bool, "true": [[agent, action]...], "false": [[agent, action]...]}
Assertions:   "shape": per-agent shape  | "at": [[index..., value], ...] on the (n_agents, ...) tensor |
              "all": [[index-prefix..., value], ...]  (every element under the prefix equals value; a prefix entry may be
              a [start, stop] slice, stop = null for "to the end") |
              "equal": full tensor | "allclose": full tensor |
              "roundtrip_state": true   (to_world_state(observe()[0]) == get_state()) |
              "perspective_of_layered": true  (observer k = layered with layers A0<->A0+k, LASER_0<->LASER_0+k swapped,
                                               and [k, A0, pos_k] == 1)
Layer names usable inside indices: "A0", "LASER_0", "WALL", "VOID", "GEM", "EXIT" (+ integer offsets as ["LASER_0", 1]).
Codes: actions N=0 S=1 E=2 W=3 STAY=4.
"""
import json
import os

N, S, E, W, STAY = 0, 1, 2, 3, 4
CASES = []


def case(name, ref, script, map=None, level=None):
    c = {"name": name, "ref": ref}
    if map is not None:
        c["map"] = map
    if level is not None:
        c["level"] = level
    c["script"] = script
    CASES.append(c)


reset = {"op": "reset"}


def step(*a):
    return {"op": "step", "actions": list(a)}


def check(obs, param=0, **kw):
    return {"op": "check", "obs": obs, "param": param, **kw}


def check_avail(walkable, true=(), false=()):
    return {"op": "check_avail", "walkable_lasers": walkable, "true": [list(x) for x in true], "false": [list(x) for x in false]}


# ------------------------------------------------------------------ python/tests/test_observations.py
case("state_gem_collected", "python/tests/test_observations.py:20-40",
     map="\nS0 X . .\n.  . . .\nG  . . .",
     script=[check("state", shape=[4]), reset, step(S), check("state", all=[[[0, None], 2, 0.0]]),
             step(S), check("state", all=[[[0, None], 2, 1.0]]), step(N), check("state", all=[[[0, None], 2, 1.0]])])

case("normalized_state_roundtrip_level1", "python/tests/test_observations.py:43-50", level=1,
     script=[reset, check("normalized-state", roundtrip_state=True)])
case("state_roundtrip_level1", "python/tests/test_observations.py:53-60", level=1,
     script=[reset, check("state", roundtrip_state=True)])

case("initial_normalized_state_1", "python/tests/test_observations.py:204-213",
     map="S0 X .\n.  . .\n.  . .",
     script=[reset, check("normalized-state", equal=[[0.0, 0.0, 1.0]])])
case("initial_normalized_state_2", "python/tests/test_observations.py:215-225",
     map="\n    S0 X  .\n    .  .  S1\n    .  .  X",
     script=[reset, check("normalized-state", allclose=[[0.0, 0.0, 1 / 3, 2 / 3, 1.0, 1.0]] * 2)])
case("initial_normalized_state_3", "python/tests/test_observations.py:227-237",
     map="\nS0 X  .  .\n.  .  S1  .\n.  X  .  .",
     script=[reset, check("normalized-state", allclose=[[0.0, 0.0, 1 / 3, 1 / 2, 1.0, 1.0]] * 2)])
case("initial_normalized_state_4", "python/tests/test_observations.py:239-249",
     map="\nS0 X  .  G\n.  .  S1  .\n.  X  .  .",
     script=[reset, check("normalized-state", allclose=[[0.0, 0.0, 1 / 3, 1 / 2, 0.0, 1.0, 1.0]] * 2)])

case("partial_3x3", "python/tests/test_observations.py:252-277",
     map="\n    S0 X  @\n    G  S1 @\n    .  .  X",
     script=[reset, check("partial", 3, at=[
         [0, 0, 1, 1, 1], [0, 1, 2, 2, 1],
         [1, 0, 0, 0, 1], [1, 1, 1, 1, 1],
         [0, "GEM", 2, 1, 1], [1, "GEM", 1, 0, 1],
         [1, "EXIT", 2, 2, 1],
         [1, "WALL", 1, 2, 1], [1, "WALL", 0, 2, 1]],
         all=[[0, "WALL", 0]])])

# four agents side by side: in agent k's window, agent m sits at (centre, centre - k + m)
_p7_at = [[k, m, 3, 3 - k + m, 1] for k in range(4) for m in range(4)]
_p7_at += [[1, "EXIT", 3, 6, 1]]
case("partial_7x7", "python/tests/test_observations.py:280-309",
     map="\nS0 S1 S2 S3 X X X X\n",
     script=[reset, check("partial", 7, shape=[11, 7, 7], at=_p7_at,
                          count_nonzero=[[k, m, 1] for k in range(4) for m in range(4)],
                          all=[[0, "EXIT", 0], [2, "EXIT", 3, [5, None], 1], [3, "EXIT", 3, [4, None], 1],
                               [[0, None], "WALL", 0], [[0, None], "GEM", 0],
                               [[0, None], ["LASER_0", 0], 0], [[0, None], ["LASER_0", 1], 0],
                               [[0, None], ["LASER_0", 2], 0], [[0, None], ["LASER_0", 3], 0]])])

case("partial_3x3_lasers", "python/tests/test_observations.py:312-330",
     map="\n    .   L0S S1\n    S0   .   .\n    L1E  X   X\n",
     script=[reset, check("partial", 3, at=[
         [0, "LASER_0", 0, 2, -1], [0, "LASER_0", 1, 2, 1], [0, "LASER_0", 2, 2, 1],
         [0, ["LASER_0", 1], 2, 1, -1], [0, ["LASER_0", 1], 2, 2, 1]])])

case("padded_layered_shapes", "python/tests/test_observations.py:333-344", map="S0 X",
     script=[check("layered", shape=[6, 1, 2]), check("layered-padded", 1, shape=[8, 1, 2]),
             check("layered-padded", 2, shape=[10, 1, 2]), check("layered-padded", 3, shape=[12, 1, 2])])

case("perspective", "python/tests/test_observations.py:347-372",
     map="\n                  S0  S1 S2 X\n                  L0E .  X  .\n                   .  .  X L1W\n                  ",
     script=[reset, check("perspective", shape=[10, 3, 4], at=[
         [0, "A0", 0, 0, 1], [1, "A0", 0, 1, 1], [2, "A0", 0, 2, 1],
         [0, "LASER_0", 1, 0, -1], [1, "LASER_0", 2, 3, -1]],
         all=[[0, "LASER_0", 1, [1, None], 1], [1, "LASER_0", 2, [0, 3], 1]])])

case("perspective2", "python/tests/test_observations.py:375-401",
     map="\n                  S0  S1 S2\n                   .   .  .\n                  L0E  X  .\n                  L1E  X  .\n                  L2E  X  .\n                  ",
     script=[reset, check("perspective", perspective_of_layered=True), step(S, S, S),
             check("perspective", perspective_of_layered=True)])

case("layered_colour_above_n_agents", "python/tests/test_observations.py:500-506", map="S0 L1E X",
     script=[check("layered", at=[[0, ["LASER_0", 1], 0, 1, -1], [0, ["LASER_0", 1], 0, 2, 1]])])

for lvl in range(1, 7):
    case(f"all_shapes_level_{lvl}", "python/tests/test_observations.py:509-518", level=lvl,
         script=[check(k, p, shape_consistent=True) for k, p in
                 (("normalized-state", 0), ("state", 0), ("layered", 0), ("flattened", 0), ("partial", 3), ("partial", 5),
                  ("partial", 7), ("layered-padded", 0), ("layered-padded", 1), ("layered-padded", 2), ("layered-padded", 3),
                  ("perspective", 0))])

# ------------------------------------------------------------------ python/tests/test_walkable_lasers.py
_WL1 = "\n@ @ L0S @  @\n@ .  .  .  @\n@ X  .  S0 @\n@ X  .  S1 @\n@ @  @  @  @\n            "
_WL2 = "\n@ @ L0S @  @\n@ .  .  .  @\n@ X  S0 .  @\n@ X  .  S1 @\n@ @  @  @  @\n            "
_WL3 = "\n@ @ L1S @  @\n@ .  .  .  @\n@ X  .  S0 @\n@ X  .  S1 @\n@ @  @  @  @\n            "
_WL4 = "\n@ @ L1S @  @\n@ .  .  .  @\n@ X  S1 .  @\n@ X  .  S0 @\n@ @  @  @  @\n            "
case("walkable_laser_enabled", "python/tests/test_walkable_lasers.py:4-21", map=_WL1,
     script=[reset, check_avail(True, true=[(0, W), (1, W)])])
case("walkable_laser_disabled_laser_enabled", "python/tests/test_walkable_lasers.py:24-45", map=_WL1,
     script=[reset, check_avail(False, true=[(0, W)], false=[(1, W)])])
case("walkable_laser_disabled_laser_disabled", "python/tests/test_walkable_lasers.py:48-67", map=_WL2,
     script=[reset, check_avail(False, true=[(1, W)])])
case("walkable_laser_disabled_laser_enabled2", "python/tests/test_walkable_lasers.py:71-92", map=_WL3,
     script=[reset, check_avail(False, true=[(1, W)], false=[(0, W)])])
case("walkable_laser_disabled_laser_disabled2", "python/tests/test_walkable_lasers.py:95-113", map=_WL4,
     script=[reset, check_avail(False, true=[(0, W)])])

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_observers.json")
    with open(out, "w") as f:
        json.dump({"format": 1, "cases": CASES}, f, indent=1)
    print(f"wrote {len(CASES)} cases to {out}")
