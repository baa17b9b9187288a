"""Maps for the instantiation-coverage test (tests/test_gpu_instantiations.py): one v1 text map for every
(lanes per environment, beam registers, crossing beams or not, exact source count) the step kernel is instantiated for.

step_kernel<G, LM, MODE, ML1, LX> (lle_amd/csrc/step_kernel.hpp): G = 1, 2, 4, 8, 16 lanes per environment (agents 1, 2, 3-4, 5-8,
9-16), LM = 4, 8, 16, 32 beam words, ML1 = no cell under two beams, LX = the exact source count of single-layer maps with at most
four.  `build(n_agents, n_sources, crossing, seed)` lays a map out so that those properties hold by construction:

    row 0            the vertical sources of a `crossing` map (L?S at columns 2 and 6, shooting down through every beam row), else floor
    beam rows        L?E . . . @ . . . L?W      two horizontal sources per row, their beams stopped by the wall in the middle
    free rows        between the beam rows and at the bottom: starts (never in a vertical beam's column), exits, gems, voids

Every free row touches a beam row, so a random walk meets beams of other colours within a few steps (deaths, corpses under beams,
gems under beams); at least 29 rows of 9 cells make every plane longer than two 128-byte lines, so that rows aligned to 128 bytes
have static head lines (the MODE 6 / 7 / 8 kernels) whatever the number of agents.  A vertical beam is at most 32 cells (one beam word).
The reference's rules hold (src/core/parsing/world_config.rs:107-250): one start per agent, at least as many exits, no start on a beam
of another colour."""
import random

W = 9
MID = 4
VCOLS = (2, 6)


def build(n_agents, n_sources, crossing=False, seed=0, n_gems=4, variant=0):
    """`variant`: another placement of starts / exits / gems / voids over the SAME walls and sources (the maps of a multi-map batch must
    agree on the dimensions and on the beam words)."""
    rng = random.Random(1000 * seed + 7)
    n_vert = 0
    if crossing:
        assert n_sources >= 2
        n_vert = 2 if n_sources >= 6 else 1
    n_horiz = n_sources - n_vert
    n_beam_rows = (n_horiz + 1) // 2
    free_needed = max(3, -(-(2 * n_agents + n_gems + 6) // 6))
    # rows: top row, then (beam row, free row) pairs, then extra free rows; at least 29 rows, at most 33 (a vertical beam is one word)
    kinds = ["top"]
    for _ in range(n_beam_rows):
        kinds += ["beam", "free"]
    while kinds.count("free") < free_needed or len(kinds) < 29:
        kinds.append("free")
    assert len(kinds) <= 33, (n_agents, n_sources, len(kinds))
    H = len(kinds)
    grid = [["."] * W for _ in range(H)]
    colours = [k % n_agents for k in range(n_sources)]
    rng.shuffle(colours)
    src = 0
    beam_colours = {}  # cell -> set of colours of the beams over it

    def cover(cells, colour):
        for c in cells:
            beam_colours.setdefault(c, set()).add(colour)

    # parse order is row-major: the vertical sources (row 0) come first
    for v in range(n_vert):
        col = VCOLS[v]
        grid[0][col] = f"L{colours[src]}S"
        cover([(i, col) for i in range(1, H)], colours[src])
        src += 1
    placed_h = 0
    for i, kind in enumerate(kinds):
        if kind != "beam":
            continue
        grid[i][MID] = "@"
        if placed_h < n_horiz:
            grid[i][0] = f"L{colours[src]}E"
            cover([(i, j) for j in range(1, MID)], colours[src])
            src += 1
            placed_h += 1
        if placed_h < n_horiz:
            grid[i][W - 1] = f"L{colours[src]}W"
            cover([(i, j) for j in range(MID + 1, W - 1)], colours[src])
            src += 1
            placed_h += 1
    assert src == n_sources
    # features: another shuffle per variant
    prng = random.Random(1000 * seed + 31 * variant + 1)
    free_cells = [(i, j) for i, k in enumerate(kinds) if k in ("free", "top") for j in range(W) if grid[i][j] == "."]
    prng.shuffle(free_cells)
    # starts close to the beams: free rows between beam rows first
    inner = [c for c in free_cells if kinds[c[0]] == "free" and c[0] < 2 * n_beam_rows + 2 and c[1] not in VCOLS]
    outer = [c for c in free_cells if c not in inner and c[1] not in VCOLS and kinds[c[0]] == "free"]
    starts = (inner + outer)[:n_agents]
    assert len(starts) == n_agents
    for a, (i, j) in enumerate(starts):
        grid[i][j] = f"S{a}"
    # exits, gems and voids near the starts (a random walk of twenty-odd steps should meet them), ties in shuffled order
    rest = sorted((c for c in free_cells if c not in starts), key=lambda c: min(abs(c[0] - s[0]) + abs(c[1] - s[1]) for s in starts) // 2)
    n_exits = n_agents + 1
    for (i, j) in rest[:n_exits]:
        grid[i][j] = "X"
    rest = rest[n_exits:]
    gems = []
    # a gem and an exit UNDER a beam where there is one (the tile below a Laser layer), the others plain
    under = [c for c in sorted(beam_colours) if grid[c[0]][c[1]] == "." and kinds[c[0]] == "beam"]
    prng.shuffle(under)
    if under:
        gems.append(under[0])
        if len(under) > 1:
            grid[under[1][0]][under[1][1]] = "X"
    for c in rest:
        if len(gems) >= n_gems:
            break
        gems.append(c)
    for (i, j) in gems:
        grid[i][j] = "G"
    rest = [c for c in rest if c not in gems and c not in beam_colours]
    for (i, j) in rest[:2]:
        grid[i][j] = "V"
    return "\n".join(" ".join(row) for row in grid) + "\n"


# (agents, G): one representative per lane-group size -- not powers of two where the class allows it (idle lanes in every group)
AGENT_CLASSES = [(1, 1), (2, 2), (3, 4), (7, 8), (13, 16)]
# (sources, crossing): LX = 0..4 single-layer, LM = 4 with a crossing, then LM = 8 / 16 / 32 single-layer and crossing
SOURCE_CLASSES = [(0, False), (1, False), (2, False), (3, False), (4, False), (3, True), (6, False), (7, True), (12, False), (12, True),
                  (20, False), (32, True)]


def lm_of(n_sources):
    return 4 if n_sources <= 4 else (8 if n_sources <= 8 else (16 if n_sources <= 16 else 32))


def kernel_name(n_agents, n_sources, crossing, mode):
    g = next(G for a, G in AGENT_CLASSES if a == n_agents)
    lm = lm_of(n_sources)
    ml1 = not crossing
    lx = n_sources if (lm == 4 and ml1) else -1
    return f"step_kernel<{g},{lm},{mode},{'true' if ml1 else 'false'},{lx}>"
