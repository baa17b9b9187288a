"""Seeded generator of v1 text maps for stress configurations (BASELINE.json configs[4]: 32x32, 8 agents, 8 lasers).

Not the reference's SAT-filtered generator (python/lle/generator/, out of scope): a plain rejection sampler that
emits maps obeying the same validity rules the reference enforces at load time (python/lle/generator/placements.py:271-303,
338-368; src/core/parsing/world_config.rs:225-243): one start per agent, at least as many exits as agents, every beam at
least two cells long, no beam over the start of an agent of another colour, distinct source colours.
"""
import random

DELTA = {"N": (-1, 0), "E": (0, 1), "S": (1, 0), "W": (0, -1)}


def generate(height=32, width=32, n_agents=8, n_lasers=8, n_gems=8, n_exits=None, wall_fraction=0.10, n_voids=4, seed=0, max_beam=31):
    """max_beam: longest beam a source may get (31 keeps the maps of earlier rounds, config5 among them, bit for bit; up to 254 for
    maps with beams of several 32-cell words)."""
    rng = random.Random(seed)
    n_exits = n_agents + 2 if n_exits is None else n_exits
    for _attempt in range(1000):
        grid = [["." for _ in range(width)] for _ in range(height)]
        cells = [(i, j) for i in range(height) for j in range(width)]
        for (i, j) in rng.sample(cells, int(wall_fraction * height * width)):
            grid[i][j] = "@"
        beam_cells = {}  # cell -> set of colours crossing it
        ok = True
        colours = list(range(n_lasers))
        rng.shuffle(colours)
        for colour in colours:
            for _try in range(200):
                i, j = rng.choice(cells)
                d = rng.choice("NESW")
                if grid[i][j] != "." or (i, j) in beam_cells:
                    continue
                di, dj = DELTA[d]
                beam = []
                y, x = i + di, j + dj
                while 0 <= y < height and 0 <= x < width and grid[y][x] == ".":
                    beam.append((y, x))
                    y, x = y + di, x + dj
                if not (2 <= len(beam) <= max_beam):
                    continue
                grid[i][j] = f"L{colour % n_agents}{d}"
                for c in beam:
                    beam_cells.setdefault(c, set()).add(colour % n_agents)
                break
            else:
                ok = False
                break
        if not ok:
            continue
        free = [c for c in cells if grid[c[0]][c[1]] == "."]
        rng.shuffle(free)
        # starts: never under a beam of another colour
        starts = []
        for a in range(n_agents):
            for c in free:
                if c not in starts and beam_cells.get(c, set()) <= {a}:
                    starts.append(c)
                    break
        if len(starts) < n_agents:
            continue
        for a, (i, j) in enumerate(starts):
            grid[i][j] = f"S{a}"
        rest = [c for c in free if c not in starts]
        if len(rest) < n_exits + n_gems + n_voids:
            continue
        for (i, j) in rest[:n_exits]:
            grid[i][j] = "X"
        for (i, j) in rest[n_exits:n_exits + n_gems]:
            grid[i][j] = "G"
        placed = 0
        for (i, j) in rest[n_exits + n_gems:]:
            if placed == n_voids:
                break
            if (i, j) not in beam_cells:   # a sprinkle of void tiles, kept off the beams so that rows stay readable
                grid[i][j] = "V"
                placed += 1
        return "\n".join(" ".join(row) for row in grid) + "\n"
    raise RuntimeError("could not generate a map")


def config5(seed=0):
    """BASELINE.json configs[4]: 32x32, 8 agents, 8 lasers (distinct colours 0-7), 8 gems, ~10 % walls."""
    return generate(32, 32, 8, 8, 8, seed=seed)
