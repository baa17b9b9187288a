"""Differential fuzz of the lane-per-agent state machine (lle_amd/csrc/step_lanes.hpp, the source the step kernel runs)
against the oracle, in one native process built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the GPU
pool has no sanitizer runs).  tests/hostsim/fuzz_lanes.cpp compares every buffer after every step.

Maps: the quirk maps of the parity suite (stale beams Q1, corpses on gems / exits Q2, three and four beams through one
cell Q4, colour aliasing Q5) plus 2 x 1 000 generated small maps (1-6 agents, 0-7 sources on 3x3 ... 8x9 grids: beams cross,
gems and exits lie under beams, voids).  Both variants of the no-op-pass shortcut."""
import os
import random
import subprocess

import pytest

from tests.parity_util import EXTRA_MAPS

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "hostsim", "fuzz_lanes")
SRCS = [os.path.join(HERE, "hostsim", "fuzz_lanes.cpp"), os.path.join(HERE, "hostsim", "hostsim.cpp"),
        os.path.join(ROOT, "lle_amd", "csrc", "map_compile.cpp")]
ORACLE = os.path.join(ROOT, "oracle", "lle_oracle.c")
DEPS = SRCS + [ORACLE] + [os.path.join(ROOT, "lle_amd", "csrc", f) for f in ("step_lanes.hpp", "step_logic.hpp", "tables.h", "map_compile.hpp",
                                                                              "observers_logic.hpp")]
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def build():
    """-O0 for the template-heavy simulator (the sanitizers' instrumentation of ~160 step_lanes instantiations takes five
    minutes to optimise at -O1, 20 s at -O0), -O1 for the rest; the three compilations run side by side."""
    if os.path.exists(BIN) and all(os.path.getmtime(BIN) >= os.path.getmtime(d) for d in DEPS):
        return BIN
    jobs, objs = [], []
    for src, cc, std, opt in ((ORACLE, "gcc", "-std=gnu11", "-O1"), (SRCS[0], "g++", "-std=c++17", "-O1"), (SRCS[1], "g++", "-std=c++17", "-O0"),
                              (SRCS[2], "g++", "-std=c++17", "-O1")):
        obj = BIN + "_" + os.path.basename(src) + ".o"
        objs.append(obj)
        jobs.append(subprocess.Popen([cc, std, "-Wno-unknown-pragmas", opt, "-c", src, "-o", obj] + SAN))
    assert all(j.wait() == 0 for j in jobs), "sanitized build failed"
    subprocess.check_call(["g++", "-o", BIN] + objs + ["-lpthread"] + SAN)
    for o in objs:
        os.remove(o)
    return BIN


def generated_maps(count, seed=0):
    from lle_amd import mapgen
    rng = random.Random(seed)
    maps = []
    while len(maps) < count:
        h, w = rng.randint(3, 8), rng.randint(3, 9)
        agents = rng.randint(1, min(6, h * w // 4))
        kw = dict(height=h, width=w, n_agents=agents, n_lasers=rng.randint(0, 7), n_gems=rng.randint(0, 4), n_exits=agents + rng.randint(0, 2),
                  wall_fraction=rng.choice([0.0, 0.05, 0.15]), n_voids=rng.randint(0, 3), seed=rng.randint(0, 1 << 30))
        try:
            maps.append(mapgen.generate(**kw))
        except (RuntimeError, ValueError, IndexError):
            continue  # the shape does not admit a legal map (e.g. no room for beams of two cells)
    return maps


@pytest.fixture(scope="module")
def fuzzer():
    return build()


def run(fuzzer, maps, path, envs, steps, engine):
    with open(path, "w") as f:
        f.write("\n===\n".join(m.strip("\n") for m in maps) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([fuzzer, str(path), str(envs), str(steps), str(engine)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=env, timeout=900)
    assert res.returncode == 0, f"rc={res.returncode}\n{res.stdout[-3000:]}\n{res.stderr[-6000:]}"
    assert res.stdout.startswith("OK maps="), res.stdout
    return dict(kv.split("=") for kv in res.stdout.split()[1:])


@pytest.mark.parametrize("engine", [1, 2], ids=["shortcut", "every_pass"])
def test_quirk_maps_long_rollouts(fuzzer, tmp_path, engine):
    maps = [EXTRA_MAPS[k] for k in ("q1", "nested", "voids_gems", "exit_under_beam", "colour_alias", "three_beams", "corridor", "four_layers",
                                    "many_agents", "gen_16x16_12agents", "gen_20_lasers")]
    out = run(fuzzer, maps, tmp_path / "quirks.txt", envs=48, steps=160, engine=engine)
    assert int(out["deaths"]) > 1000 and int(out["env_steps"]) == len(maps) * 48 * 160


@pytest.mark.parametrize("engine", [1, 2], ids=["shortcut", "every_pass"])
def test_generated_small_maps(fuzzer, tmp_path, engine):
    maps = generated_maps(1000, seed=engine)
    out = run(fuzzer, maps, tmp_path / "generated.txt", envs=12, steps=40, engine=engine)
    assert int(out["maps"]) == 1000 and int(out["deaths"]) > 10000
    # World.exit_pos = [...] with random cells every 16 steps (world.rs:195-234): lists taken AND lists refused, on both sides alike
    assert int(out["exits_taken"]) > 50 and int(out["exits_refused"]) > 50, out


def generated_wide_maps(count, seed=0):
    """2-5 rows x 34-80 columns with few walls: horizontal beams of more than 32 cells (chains of beam words, tables.h)."""
    from lle_amd import mapgen
    rng = random.Random(seed)
    maps = []
    while len(maps) < count:
        h, w = rng.randint(2, 5), rng.randint(34, 80)
        agents = rng.randint(1, 5)
        kw = dict(height=h, width=w, n_agents=agents, n_lasers=rng.randint(1, 6), n_gems=rng.randint(0, 4), n_exits=agents + rng.randint(0, 2),
                  wall_fraction=rng.choice([0.0, 0.0, 0.01, 0.03]), n_voids=rng.randint(0, 2), seed=rng.randint(0, 1 << 30), max_beam=254)
        try:
            maps.append(mapgen.generate(**kw))
        except (RuntimeError, ValueError, IndexError):
            continue
    return maps


@pytest.mark.parametrize("engine", [1, 2], ids=["shortcut", "every_pass"])
def test_long_beam_maps(fuzzer, tmp_path, engine):
    """Beams longer than 32 cells under ASan / UBSan: the hand-made maps of tests/parity_util.py LONG_MAPS (long rollouts) and 300
    generated wide maps, with exits moving every 16 steps; most of the generated maps must actually hold a long beam."""
    from tests.parity_util import LONG_MAPS
    out = run(fuzzer, list(LONG_MAPS.values()), tmp_path / "long.txt", envs=48, steps=160, engine=engine)
    assert int(out["long_beam_maps"]) == len(LONG_MAPS) and int(out["deaths"]) > 1000
    out = run(fuzzer, generated_wide_maps(300, seed=10 + engine), tmp_path / "wide.txt", envs=12, steps=48, engine=engine)
    assert int(out["maps"]) == 300 and int(out["long_beam_maps"]) > 120 and int(out["deaths"]) > 3000, out


def test_shortcut_saves_passes_and_nothing_else(fuzzer, tmp_path):
    maps = generated_maps(200, seed=7)
    a = run(fuzzer, maps, tmp_path / "a.txt", envs=12, steps=40, engine=1)
    b = run(fuzzer, maps, tmp_path / "b.txt", envs=12, steps=40, engine=2)
    assert a["deaths"] == b["deaths"] and a["env_steps"] == b["env_steps"]
    assert int(a["lane_passes"]) < int(b["lane_passes"])  # every death costs one more pass without the shortcut
