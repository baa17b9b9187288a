"""Bit-exact comparison of an engine's packed buffers (device layout, as numpy) with the oracle's canonical dump."""
import numpy as np


def unpack_engine(bufs, A, G, L, beam_stride, C, H, W, first_words=None):
    """bufs: dict name -> numpy array in the LLE_BUF_* layout.  Returns the oracle's canonical layout.
    first_words: first beam WORD of every source (Map.source_first_words(); None: word == laser_id, i.e. no beam of the map is
    longer than 32 cells): offset k of source s is bit k % 32 of word first_words[s] + k // 32."""
    n = bufs["bits"].shape[0]
    bits = bufs["bits"].astype(np.uint64)
    ar = np.arange(A, dtype=np.uint64)
    out = {
        "pos": bufs["pos"].reshape(n, A, 2).astype(np.uint8),
        "alive": ((bits[:, None] >> ar) & 1).astype(np.uint8),
        "arrived": ((bits[:, None] >> (ar + 16)) & 1).astype(np.uint8),
        "occupant": ((bits[:, None] >> (ar + 32)) & 1).astype(np.uint8),
        "gems": ((bufs["gems"].astype(np.uint64)[:, None] >> np.arange(G, dtype=np.uint64)) & 1).astype(np.uint8),
        "avail": bufs["avail"].reshape(n, A).astype(np.uint8),
    }
    if L and first_words is not None and beam_stride > 32:
        bm = bufs["beams"].reshape(n, -1).astype(np.uint64)
        k = np.arange(beam_stride)
        words = np.minimum(np.asarray(first_words)[:, None] + k[None, :] // 32, bm.shape[1] - 1)  # [L, beam_stride]
        out["beams"] = ((bm[:, words] >> (k % 32).astype(np.uint64)[None, None, :]) & 1).astype(np.uint8)
        # (offsets beyond a shorter beam's words would read another source's word: the oracle reports 0 there)
        n_words = np.diff(list(first_words) + [bm.shape[1]])
        out["beams"] *= (k[None, :] < 32 * n_words[:, None]).astype(np.uint8)[None]
    elif L:
        bm = bufs["beams"].reshape(n, -1)[:, :L].astype(np.uint64)
        out["beams"] = ((bm[:, :, None] >> np.arange(beam_stride, dtype=np.uint64)) & 1).astype(np.uint8)
    else:
        out["beams"] = np.zeros((n, 0, beam_stride), np.uint8)
    if "obs" in bufs and bufs["obs"] is not None:
        out["obs"] = bufs["obs"].reshape(n, -1)[:, : C * H * W].reshape(n, C, H, W)
    ev = bufs["events"].reshape(n, 2 * A)
    out["events"] = np.stack([ev >> 4, ev & 15], axis=-1).astype(np.uint8)
    out["ev_count"] = bufs["evcount"].astype(np.uint8)
    out["err"] = bufs["err"].astype(np.int32)
    out["actions"] = bufs["actions"].reshape(n, A).astype(np.uint8)
    return out


def assert_state_equal(eng, dump, where=""):
    for key in ("pos", "alive", "arrived", "occupant", "gems", "beams", "avail"):
        a, b = eng[key], dump[key]
        if not np.array_equal(a, b):
            bad = np.argwhere(a.reshape(a.shape[0], -1) != b.reshape(b.shape[0], -1))
            e = int(bad[0][0])
            raise AssertionError(f"{where}: '{key}' differs in {len(set(bad[:, 0]))} envs; first env {e}: engine={a[e].tolist()} oracle={b[e].tolist()}")


def assert_step_equal(eng, ostep, where="", check_obs=True):
    """ostep: dict returned by OracleBatch.step()."""
    n, cap = eng["events"].shape[0], eng["events"].shape[1]
    for key in ("actions", "err", "ev_count"):
        if not np.array_equal(eng[key], ostep[key]):
            bad = np.argwhere(eng[key].reshape(n, -1) != ostep[key].reshape(n, -1))
            e = int(bad[0][0])
            raise AssertionError(f"{where}: '{key}' differs; first env {e}: engine={eng[key][e].tolist()} oracle={ostep[key][e].tolist()}")
    cnt = (ostep["ev_count"] & 0x7F).astype(np.int64)
    valid = np.arange(cap)[None, :] < cnt[:, None]
    ea = np.where(valid[:, :, None], eng["events"], 0)
    eb = np.where(valid[:, :, None], ostep["events"], 0)
    if not np.array_equal(ea, eb):
        e = int(np.argwhere((ea != eb).reshape(n, -1))[0][0])
        raise AssertionError(f"{where}: events differ; first env {e}: engine={ea[e].tolist()} oracle={eb[e].tolist()}")
    if check_obs and "obs" in eng and "obs" in ostep:
        if not np.array_equal(eng["obs"], ostep["obs"]):
            bad = np.argwhere((eng["obs"] != ostep["obs"]).reshape(n, -1))
            e = int(bad[0][0])
            idx = np.argwhere(eng["obs"][e] != ostep["obs"][e])
            raise AssertionError(f"{where}: obs differs in {len(set(bad[:, 0]))} envs; first env {e} at (c,i,j)={idx[:5].tolist()} "
                                 f"engine={[int(eng['obs'][e][tuple(q)]) for q in idx[:5]]} oracle={[int(ostep['obs'][e][tuple(q)]) for q in idx[:5]]}")


# maps used by the differential tests beyond the six levels: crossings (nested lasers), voids, gems and exits under
# beams, same-colour nesting, colours without agent, disabled-looking corners
EXTRA_MAPS = {
    "q1": "S0 . G X\n. . L2W X\n. S1 . X\n. L1N . S2",
    "nested": (
        "S0 . G . L1S . X\n"
        "L0E . . G . . .\n"
        ". S1 . . . V .\n"
        "L2E . G . . . X\n"
        ". . S2 . . . X\n"
        ". . . . L0N . ."
    ),
    "voids_gems": (
        "S0 G V . X\n"
        ". V G . X\n"
        "S1 . . G V\n"
        "L0E . . . .\n"
        ". G . V ."
    ),
    "exit_under_beam": "L1E . X X\nS0 . . .\nS1 . . G\nL0E G . .",
    # laser colours >= n_agents are legal (Q5): with A=1 their layers alias WALL (1), VOID (2) and GEM (3)
    "colour_alias": "S0 . . .\n. G X .\nL3E . . .\n. . . L2W\nL1E . G .",
    "three_beams": (
        ". . L0S . . X\n"
        "L1E . . . . X\n"
        ". . . . L2W X\n"
        "S0 S1 G S2 . ."
    ),
    "corridor": "S0 S1 S2 S3 . . G X X X X",
    # cell (2,2) lies under four beams (one per direction): the deepest stack possible; World.lasers() and the
    # observation only expose the two outer layers (quirk Q4) while all four act on agents
    "four_layers": ". . L0S . .\n. . . . .\nL1E . G . L2W\n. . . . .\nS0 S1 L3N X X",
    # python/tests/test_world.py:855-874: 14 agents, 14 sources (the 16x16 kernel instantiation)
    "many_agents": " .   .   . . . .\n" + "".join(f"S{k}  L{k}W  . . . X\n" for k in range(14)),
}


def _add_generated():
    from lle_amd import mapgen
    EXTRA_MAPS["config5_32x32"] = mapgen.config5(0)          # BASELINE.json configs[4]: 8 agents, 8 lasers, crossings
    EXTRA_MAPS["gen_16x16_12agents"] = mapgen.generate(16, 16, 12, 10, 6, seed=3, n_voids=3)
    EXTRA_MAPS["gen_20_lasers"] = mapgen.generate(18, 18, 3, 20, 5, seed=11, wall_fraction=0.05, n_voids=2)  # LM = 32


_add_generated()


def _grid(h, w, cells):
    rows = [["."] * w for _ in range(h)]
    for (i, j), t in cells.items():
        rows[i][j] = t
    return "\n".join(" ".join(r) for r in rows)


# beams longer than 32 cells (a chain of beam words, lle_amd/csrc/tables.h): corridors, a crossing behind the word boundary, several
# long beams, gems / exits / voids under them, three words
LONG_MAPS = {
    "long_corridor": _grid(3, 40, {(0, 0): "L0E", (0, 39): "@", (1, 36): "S0", (1, 38): "S1", (2, 0): "X", (2, 1): "X", (0, 20): "G", (0, 35): "G"}),
    "long_q1": _grid(3, 40, {(0, 12): "L1S", (1, 0): "L0E", (1, 39): "@", (2, 0): "X", (2, 1): "X", (2, 11): "S0", (2, 12): "@", (2, 34): "S1"}),
    "long_crossing": _grid(6, 44, {(2, 0): "L0E", (3, 43): "L1W", (0, 36): "L2S", (5, 8): "L1N", (1, 10): "S0", (4, 20): "S1", (1, 30): "S2",
                                   (5, 0): "X", (5, 1): "X", (5, 2): "X", (2, 40): "G", (3, 3): "G", (2, 36): "V", (3, 35): "X", (4, 36): "G"}),
    "long_three_words": _grid(2, 72, {(0, 0): "L0E", (1, 5): "S0", (1, 40): "S1", (1, 70): "S2", (1, 0): "X", (1, 1): "X", (1, 2): "X", (0, 66): "G"}),
}


def legal_colours(maps, colours):
    """Random per-env source colours made acceptable to lle_batch_set_sources: a colour that would put another agent's start
    on the source's beam (refused with LLE_ENV_COLOUR_CROSSES_START, like the binding's LaserSource.set_colour,
    pylaser_source.rs:121-139) is replaced by the colour the map gives that source.  `maps`: a Map / map text, or the list
    of a multi-map batch (map m owns the m-th block of envs).  `colours`: numpy or torch [n, L]; same type comes back."""
    import torch

    from lle_amd import _capi
    maps = maps if isinstance(maps, (list, tuple)) else [maps]
    maps = [m if isinstance(m, _capi.Map) else _capi.Map(m) for m in maps]
    is_torch = isinstance(colours, torch.Tensor)
    arr = colours.cpu().numpy().copy() if is_torch else np.array(colours, copy=True)
    n, L = arr.shape
    per = n // len(maps)
    for k, m in enumerate(maps):
        block = arr[k * per:(k + 1) * per]
        for s in m.sources():
            for c in range(m.n_agents):
                if not m.colour_allowed(s.laser_id, c):
                    block[block[:, s.laser_id] == c, s.laser_id] = s.agent_id
    return torch.from_numpy(arr).to(colours.device) if is_torch else arr
