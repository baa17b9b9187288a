"""Row heads (lle_map_set_head_lines / lle_map_row_head): the default step kernel stores the lines of every row that no
agent, beam or gem can change BEFORE its state machine, and streams the rest afterwards.  The launcher only does it for
launches of one to two rounds of workgroups; LLE_ROW_HEADS=1 forces it so that the small batches here take that path.
Every buffer must equal the oracle's whatever the head size, and equal the run without heads."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu

MAPS = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)


def _generated():
    """Maps with 5-8 sources: their beam masks live in the LDS record AND they have row heads (the lane's share of the masks is
    read up front, ahead of the head stores)."""
    from lle_amd import mapgen
    return {"gen_12x13_4agents_8lasers": mapgen.generate(12, 13, 4, 8, 4, seed=2), "gen_12x13_2agents_8lasers": mapgen.generate(12, 13, 2, 8, 4, seed=2),
            "gen_12x13_1agent_6lasers": mapgen.generate(12, 13, 1, 6, 2, seed=5)}


MAPS.update(_generated())


@pytest.fixture
def heads_forced(monkeypatch):
    monkeypatch.setenv("LLE_ROW_HEADS", "1")


@pytest.mark.parametrize("name", ["level1", "level2", "level3", "level5", "level6", "nested", "colour_alias", "gen_16x16_12agents",
                                  "gen_12x13_4agents_8lasers", "gen_12x13_2agents_8lasers", "gen_12x13_1agent_6lasers"])
@pytest.mark.parametrize("lines", [-1, 1, 4, 8])
def test_heads_match_oracle(oracle_mod, heads_forced, name, lines):
    from lle_amd import BatchedWorld, Map

    m = Map(MAPS[name], row_align=128)
    m.set_head_lines(lines)
    first, nbytes = m.row_head
    assert first % 128 == 0 and nbytes % 128 == 0 and nbytes <= (8 if lines < 0 else lines) * 128
    assert first >= m.n_agents * m.height * m.width or nbytes == 0  # behind the agent layers
    n = 1000
    ob = oracle_mod.OracleBatch(MAPS[name], n)
    bw = BatchedWorld(m, n)
    bw.obs_rows.fill_(55)  # the launch must write every byte of every row itself, head included
    for t in range(24):
        auto = t % 8 != 7
        bw.step(sample=True, auto_reset=auto, seed=99, t=t, env_offset=5)
        check(bw, ob, ob.step(None, auto_reset=auto, seed=99, t=t, env_offset=5), f"{name} lines={lines} t={t}")
        assert int(bw.obs_rows[:, m.obs_bytes:].abs().max()) == 0 if m.obs_stride > m.obs_bytes else True


def test_heads_with_given_actions(oracle_mod, heads_forced):
    """The caller's actions (lle_batch_step `actions` and the batch's own LLE_BUF_ACTIONS) are read up front on this path."""
    import torch

    from lle_amd import BatchedWorld, Map
    from lle_amd._capi import lib

    n = 640
    m = Map(LEVELS[6])
    assert m.row_head[1] > 0
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(m, n)
    rng = np.random.default_rng(3)
    for t in range(30):
        actions = rng.integers(0, 6, size=(n, ob.A), dtype=np.uint8)  # unavailable and out-of-range (5) included
        if t % 2:
            bw.step(torch.from_numpy(actions).cuda())
        else:  # actions already in LLE_BUF_ACTIONS: NULL action pointer, no sampling
            bw.actions.copy_(torch.from_numpy(actions).cuda())
            assert lib().lle_batch_step(bw.h, None, 0, 0, t, 0, bw._stream()) == 0
        check(bw, ob, ob.step(actions), f"t={t}")


@pytest.mark.parametrize("name", ["level6", "level3"])
def test_heads_on_and_off_agree_at_full_batch(name, monkeypatch):
    """65 536 envs (the launcher's own choice is heads ON there): same buffers with LLE_ROW_HEADS=0."""
    import torch

    from lle_amd import BatchedWorld, Map

    n = 65536
    a, b = BatchedWorld(Map(MAPS[name]), n), BatchedWorld(Map(MAPS[name]), n)
    assert a.map.row_head[1] > 0
    for t in range(12):
        monkeypatch.delenv("LLE_ROW_HEADS", raising=False)
        a.step(sample=True, auto_reset=True, seed=4, t=t)
        monkeypatch.setenv("LLE_ROW_HEADS", "0")
        b.step(sample=True, auto_reset=True, seed=4, t=t)
    for k in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "done", "obs"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


@pytest.mark.parametrize("variant", ["per_env_sources", "fused_outputs"])
def test_general_modes_with_heads_in_a_many_round_launch(variant, monkeypatch):
    """262 144 level-6 envs = 16 384 wavefronts, four rounds of workgroups: since round 3 the launcher takes the head kernels there
    too in the general modes (MODE 8 for per-env sources, MODE 7 for the fused LLE.step outputs; kernels.hip row_heads_pay).  Same
    buffers, same outputs as the kernels without heads (MODE 5 / 4), which the small-batch suites hold against the oracle."""
    import torch

    from lle_amd import BatchedWorld, Map
    from tests.parity_util import legal_colours

    n = 262144
    worlds = [BatchedWorld(Map(LEVELS[6]), n) for _ in (0, 1)]
    outs = []
    for bw in worlds:
        A, G = bw.map.n_agents, bw.map.n_gems
        st, rw, av = (torch.zeros((n, 3 * A + G), device="cuda"), torch.zeros((n, 1), device="cuda"),
                      torch.zeros((n, A, 5), dtype=torch.uint8, device="cuda"))
        outs.append((st, rw, av, bw.make_env_outputs(state=st, reward=rw, available=av)))
        if variant == "per_env_sources":
            rng = np.random.default_rng(3)
            bw.set_sources(torch.from_numpy(legal_colours(bw.map, rng.integers(0, A, size=(n, bw.map.n_sources), dtype=np.uint8))))
    for t in range(6):
        for bw, (st, rw, av, eo), heads in zip(worlds, outs, ("1", "0")):
            monkeypatch.setenv("LLE_ROW_HEADS", heads)
            kw = dict(env_out=eo) if variant == "fused_outputs" else dict(recolour_resets=(t % 2 == 1))
            bw.step(sample=True, auto_reset=True, seed=8, t=t, **kw)
    a, b = worlds
    for k in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "done", "obs"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    if variant == "fused_outputs":
        assert all(torch.equal(x, y) for x, y in zip(outs[0][:3], outs[1][:3]))


def test_source_update_moves_the_head(oracle_mod, heads_forced):
    """lle_batch_update_sources recompiles the tables: a colour change moves beam bytes to another layer, and the head with them."""
    from lle_amd import BatchedWorld, Map

    n = 500
    m = Map(LEVELS[6])
    m.set_head_lines(8)
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(m, n)
    heads = {m.row_head}
    for colour in (3, 0, 2, 1):
        if not m.colour_allowed(0, colour):
            continue
        m.set_source(0, agent_id=colour)
        bw.update_sources()
        for e in range(n):
            ob.world(e).set_source(0, colour=colour)
        heads.add(m.row_head)
        for t in range(6):
            bw.step(sample=True, auto_reset=True, seed=8, t=t)
            check(bw, ob, ob.step(None, auto_reset=True, seed=8, t=t), f"colour {colour} t={t}")
    assert len(heads) >= 2, heads


@pytest.mark.parametrize("group", ["2", "4"])
@pytest.mark.parametrize("name", ["level6", "level3", "gen_12x13_4agents_8lasers"])
def test_head_groups_match_oracle(oracle_mod, heads_forced, monkeypatch, name, group):
    """LLE_HEAD_GROUP (step_kernel.hpp HEAD; lle_batch_autotune's `head_group`): one wavefront of every 2 / 4 of a workgroup stores the
    row heads of the whole group, the others go straight to their state machines.  A ragged last workgroup included."""
    from lle_amd import BatchedWorld, Map

    monkeypatch.setenv("LLE_HEAD_GROUP", group)
    m = Map(MAPS[name], row_align=128)
    n = 1000 + 8 * int(group)
    ob = oracle_mod.OracleBatch(MAPS[name], n)
    bw = BatchedWorld(m, n)
    assert bw.tuning()["head_group"] == int(group)
    bw.obs_rows.fill_(55)
    for t in range(16):
        auto = t % 8 != 7
        bw.step(sample=True, auto_reset=auto, seed=7, t=t, env_offset=5)
        check(bw, ob, ob.step(None, auto_reset=auto, seed=7, t=t, env_offset=5), f"{name} group={group} t={t}")


def test_head_groups_in_the_general_kernels(heads_forced, monkeypatch):
    """The same on several maps per batch (MODE 7) and with per-environment sources (MODE 8): equal buffers whatever the group."""
    import torch

    from lle_amd import BatchedWorld, mapgen
    from tests.parity_util import legal_colours

    texts = [mapgen.generate(seed=100 + s, height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2) for s in range(5)]
    per, want = 208, None
    for group in ("1", "2", "4"):
        monkeypatch.setenv("LLE_HEAD_GROUP", group)
        got = []
        for pes in (False, True):
            bw = BatchedWorld(texts, per * len(texts), row_align=128)
            if pes:
                g = torch.Generator(device="cuda").manual_seed(1)
                bw.set_sources(colours=legal_colours(bw.maps, torch.randint(0, 3, (bw.n_envs, bw.map.n_sources), generator=g, device="cuda", dtype=torch.uint8)))
            bw.obs_rows.fill_(55)
            for t in range(12):
                bw.step(sample=True, auto_reset=True, seed=3, t=t)
            torch.cuda.synchronize()
            got.append((bw.obs_rows.clone(), bw.pos.clone(), bw.beams.clone()))
        if want is None:
            want = got
        else:
            for (a, b) in zip(want, got):
                for x, y in zip(a, b):
                    assert torch.equal(x, y), group


@pytest.mark.parametrize("name", ["level6", "level5", "level3", "nested", "gen_12x13_4agents_8lasers"])
@pytest.mark.parametrize("lines", [-1, 1, 2, 8])
def test_heads_under_per_env_sources_write_every_byte(oracle_mod, heads_forced, name, lines):
    """MODE 8 with one or two runs of head lines (lle_map_row_head_env_sources / _second): rows filled with garbage before every step --
    the launch must write every byte of every row itself, whichever lines went out ahead of the state machine."""
    import torch

    from lle_amd import BatchedWorld, Map
    from tests.parity_util import legal_colours

    m = Map(MAPS[name], row_align=128)
    m.set_head_lines(lines)
    n = 1000
    A, L = m.n_agents, m.n_sources
    ob = oracle_mod.OracleBatch(MAPS[name], n)
    bw = BatchedWorld(m, n)
    rng = np.random.default_rng(6)
    colours = legal_colours(m, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    bw.set_sources(torch.from_numpy(colours))
    for e in range(n):
        for l in range(L):
            ob.world(e).set_source(l, colour=int(colours[e, l]))
    for t in range(12):
        bw.obs_rows.fill_(55)
        auto = t % 4 != 3
        bw.step(sample=True, auto_reset=auto, seed=5, t=t, env_offset=2)
        check(bw, ob, ob.step(None, auto_reset=auto, seed=5, t=t, env_offset=2), f"{name} lines={lines} t={t}")
        if m.obs_stride > m.obs_bytes:
            assert int(bw.obs_rows[:, m.obs_bytes:].abs().max()) == 0
