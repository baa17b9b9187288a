/*
 * lle_oracle.c -- CPU restatement of yamoling/lle's `World` hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for lle_amd.  It is a literal, tile-object restatement of the
 * reference's sequential algorithm: every cell is a `tile` object, laser cells wrap another tile
 * and share a per-source `beam` (array of bool), exactly like the reference's
 * `Tile` / `Laser` / `LaserBeam`.  It deliberately does NOT share any formulation with the HIP
 * kernels (which use per-beam bitmasks and per-agent occupancy bits), so that the two are
 * independent statements of the same rules.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (lle_amd/) never imports, links or executes anything in oracle/.
 *
 * Parity pinning: the reference is Rust and cannot be built or imported in this environment
 * (no cargo/rustc; `lle.lle` extension absent), so this oracle is pinned by the reference's own
 * known-answer tests, hand-transcribed into tests/golden/kat_*.json (see tests/golden/README.md),
 * and by nothing else.
 *
 * Every function cites the reference file:line (relative to the reference repository root) that
 * it restates.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET (cpu_baseline thread pinning) */
#endif
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

/* ---- value codes (src/bindings/world/pyaction.rs:13-25, pyevent.rs:9-16) ---- */
enum { ACT_NORTH = 0, ACT_SOUTH = 1, ACT_EAST = 2, ACT_WEST = 3, ACT_STAY = 4 };
enum { EV_AGENT_EXIT = 0, EV_GEM_COLLECTED = 1, EV_AGENT_DIED = 2 };
/* src/core/tiles/direction.rs:8-27 */
enum { DIR_NORTH = 0, DIR_EAST = 1, DIR_SOUTH = 2, DIR_WEST = 3 };

/* ParseError variants (src/core/parsing/errors.rs:5-74) that the v1 path can raise. */
enum {
    OW_OK = 0,
    OW_ERR_EMPTY_WORLD = 1,
    OW_ERR_NO_AGENTS = 2,
    OW_ERR_INVALID_TILE = 3,
    OW_ERR_NOT_ENOUGH_EXITS = 4,
    OW_ERR_DUPLICATE_START = 5,
    OW_ERR_INCONSISTENT_DIMENSIONS = 6,
    OW_ERR_INVALID_AGENT_ID = 7,
    OW_ERR_INVALID_DIRECTION = 8, /* the reference panics here (laser_config.rs:22 `.unwrap()`) */
    OW_ERR_AGENT_WITHOUT_START = 9,
    OW_ERR_NOT_ENOUGH_STARTS = 10,
    OW_ERR_TOML_UNSUPPORTED = 11, /* TOML (v2) maps are out of scope */
};

/* RuntimeWorldError variants (src/core/errors.rs:6-45) used by step/set_state. */
enum {
    OW_RT_OK = 0,
    OW_RT_INVALID_NUMBER_OF_ACTIONS = -1,
    OW_RT_INVALID_NUMBER_OF_GEMS = -2,
    OW_RT_INVALID_NUMBER_OF_AGENTS = -3,
    OW_RT_INVALID_WORLD_STATE = -4,
    OW_RT_OUT_OF_WORLD_POSITION = -5,
    OW_RT_INVALID_AGENT_POSITION = -6,
    /* InvalidAction{agent_id} is returned as 1 + agent_id (> 0). */
};

typedef struct { int i, j; } pos_t;

/* src/core/tiles/laser.rs:15-21 `LaserBeam` */
typedef struct beam {
    bool* on;      /* beam: RefCell<Vec<bool>> */
    int len;
    bool enabled;  /* is_enabled */
    int agent_id;  /* colour (mutable) */
    int direction;
    int laser_id;
    pos_t source;
} beam_t;

typedef enum { T_FLOOR, T_WALL, T_VOID, T_EXIT, T_GEM, T_SOURCE, T_LASER } tile_kind;

/* src/core/tiles/tile.rs:10-18 `Tile`, with laser.rs:88-92 `Laser{beam, wrapped, offset}` */
typedef struct tile {
    tile_kind kind;
    int agent;      /* Option<AgentId>; -1 = None (Floor/Exit/Void/Gem slot) */
    bool collected; /* Gem */
    beam_t* beam;   /* Laser / LaserSource */
    int offset;     /* Laser */
    struct tile* wrapped; /* Laser */
} tile_t;

/* src/agent.rs:6-10 */
/* `ghost`: harness bookkeeping, not reference state -- set_state flagged the agent dead without an AgentDied event, so
 * LLE.compute_done (python/lle/env/env.py:208-217,253-254: it counts death EVENTS) does not see it */
typedef struct { int id; bool dead; bool arrived; bool ghost; } agent_t;

typedef struct { int type; int agent; } event_t;

#define OW_MAX_AGENTS 64

/* src/core/world.rs:21-44 `World` */
typedef struct ow_world {
    int width, height;
    tile_t** grid; /* height*width, row major */
    int n_agents;
    agent_t* agents;
    int n_sources; beam_t** beams; /* laser_source_positions order == laser_id order */
    int n_lasers_pos; pos_t* lasers_positions; /* HashSet in the reference: order unspecified; row-major here */
    int n_gems; pos_t* gems_positions;
    int* n_starts; pos_t** random_start_positions; /* per agent */
    int n_voids; pos_t* void_positions;
    int n_exits; pos_t* exits;
    pos_t* agents_positions;
    int n_walls; pos_t* wall_positions;
    int (*available)[5]; int* n_available; /* available_actions, reference order [Stay,N,E,S,W] filtered */
    pos_t* start_positions;
    int panics;                 /* number of reference `panic!`/`expect` sites that would have fired */
    char panic_msg[128];
} ow_world;

static void ow_panic(ow_world* w, const char* msg) {
    w->panics++;
    snprintf(w->panic_msg, sizeof w->panic_msg, "%s", msg);
}

static tile_t* new_tile(tile_kind k) {
    tile_t* t = (tile_t*)calloc(1, sizeof *t);
    t->kind = k; t->agent = -1;
    return t;
}

static tile_t* at(ow_world* w, int i, int j) { return w->grid[i * w->width + j]; }

/* ---------------------------------------------------------------- beams (laser.rs:23-86) */
static bool beam_is_on(const beam_t* b, int offset) { return b->on[offset]; }            /* :41-43 */
static void beam_turn_on(beam_t* b, int offset) {                                          /* :50-55 */
    if (!b->enabled) return;
    for (int k = offset; k < b->len; k++) b->on[k] = true;
}
static void beam_turn_off(beam_t* b, int offset) {                                         /* :57-59 */
    for (int k = offset; k < b->len; k++) b->on[k] = false;
}
static void beam_enable(beam_t* b) { b->enabled = true; beam_turn_on(b, 0); }              /* :69-72 */
static void beam_disable(beam_t* b) { b->enabled = false; beam_turn_off(b, 0); }           /* :74-77 */

/* ---------------------------------------------------------------- Laser tile (laser.rs:100-207) */
static bool laser_is_on(const tile_t* l) { return beam_is_on(l->beam, l->offset); }       /* :132-134 */
static void laser_turn_on(tile_t* l) {                                                     /* :157-162 */
    if (laser_is_on(l)) return;
    beam_turn_on(l->beam, l->offset);
}
static void laser_turn_off(tile_t* l) { beam_turn_off(l->beam, l->offset); }               /* :164-166 */

/* ---------------------------------------------------------------- Tile (tile.rs:20-99) */
static int tile_agent(const tile_t* t) {                                                   /* tile.rs:86-95, laser.rs:204-206 */
    switch (t->kind) {
        case T_WALL: case T_SOURCE: return -1;
        case T_LASER: return tile_agent(t->wrapped);
        default: return t->agent;
    }
}
static bool tile_is_occupied(const tile_t* t) { return tile_agent(t) >= 0; }               /* tile.rs:97-99 */
static bool tile_is_walkable(const tile_t* t) {                                            /* tile.rs:63-73 */
    return !(t->kind == T_WALL || t->kind == T_SOURCE);
}

static void tile_reset(tile_t* t) {                                                        /* tile.rs:75-84 */
    switch (t->kind) {
        case T_GEM: t->collected = false; t->agent = -1; break;                            /* gem.rs:21-24 */
        case T_SOURCE: case T_WALL: break;
        case T_EXIT: case T_FLOOR: case T_VOID: t->agent = -1; break;                      /* void.rs:31-33 */
        case T_LASER: laser_turn_on(t); tile_reset(t->wrapped); break;                     /* laser.rs:168-171 */
    }
}

/* returns 0 = Ok, 1 = Err(TileNotWalkable) */
static int tile_pre_enter(tile_t* t, const agent_t* agent) {                               /* tile.rs:21-27 */
    switch (t->kind) {
        case T_LASER: {                                                                    /* laser.rs:173-182 */
            int res = tile_pre_enter(t->wrapped, agent);
            if (!t->beam->enabled) return res;
            if (!agent->dead && agent->id == t->beam->agent_id) laser_turn_off(t);
            return res;
        }
        case T_WALL: case T_SOURCE: return 1;
        default: return 0;
    }
}

/* returns true and fills *ev when an event is produced */
static bool tile_enter(ow_world* w, tile_t* t, agent_t* agent, event_t* ev) {              /* tile.rs:29-50 */
    switch (t->kind) {
        case T_WALL: case T_SOURCE:
            ow_panic(w, "Cannot enter a wall or a laser source");
            return false;
        case T_EXIT:
            t->agent = agent->id;
            if (!agent->arrived) {
                agent->arrived = true;
                ev->type = EV_AGENT_EXIT; ev->agent = agent->id;
                return true;
            }
            return false;
        case T_FLOOR:
            t->agent = agent->id;
            return false;
        case T_VOID:                                                                       /* void.rs:13-22 */
            t->agent = agent->id;
            if (!agent->dead) {
                agent->dead = true;
                ev->type = EV_AGENT_DIED; ev->agent = agent->id;
                return true;
            }
            return false;
        case T_LASER:                                                                      /* laser.rs:184-197 */
            if (laser_is_on(t) && agent->id != t->beam->agent_id) {
                if (!agent->dead) {
                    agent->dead = true;
                    laser_turn_on(t);
                    ev->type = EV_AGENT_DIED; ev->agent = agent->id;
                    return true;
                }
                return false;
            }
            return tile_enter(w, t->wrapped, agent, ev);
        case T_GEM:                                                                        /* gem.rs:26-35 */
            t->agent = agent->id;
            if (!t->collected) {
                t->collected = true;
                ev->type = EV_GEM_COLLECTED; ev->agent = agent->id;
                return true;
            }
            return false;
    }
    return false;
}

static int tile_leave(ow_world* w, tile_t* t) {                                            /* tile.rs:52-61 */
    switch (t->kind) {
        case T_WALL: case T_SOURCE:
            ow_panic(w, "Cannot leave a wall or a laser source");
            return -1;
        case T_LASER:                                                                      /* laser.rs:199-202 */
            laser_turn_on(t);
            return tile_leave(w, t->wrapped);
        default: {
            int a = t->agent;
            if (a < 0) ow_panic(w, "No agent to leave");
            t->agent = -1;
            return a;
        }
    }
}

/* ---------------------------------------------------------------- parsing (parser_v1.rs, world_config.rs) */
typedef struct {
    int width, height; bool has_width;
    int n_gems; pos_t* gems;
    int n_agents; int* n_starts; pos_t** starts; /* start_positions: Vec<Vec<Position>> */
    int n_voids; pos_t* voids;
    int n_exits; pos_t* exits;
    int n_walls; pos_t* walls;
    int n_lasers; pos_t* laser_pos; int* laser_dir; int* laser_agent; /* laser_configs */
} parsing_data;

#define PUSH(arr, n, v) do { (arr) = realloc((arr), sizeof *(arr) * ((n) + 1)); (arr)[(n)++] = (v); } while (0)

/* Rust `str::parse::<usize>()`: optional '+', then >=1 ASCII digits, nothing else. */
static bool parse_usize(const char* s, int len, int* out) {
    int k = 0;
    if (k < len && s[k] == '+') k++;
    if (k >= len) return false;
    long v = 0;
    for (; k < len; k++) {
        if (!isdigit((unsigned char)s[k])) return false;
        v = v * 10 + (s[k] - '0');
        if (v > 1000000) return false;
    }
    *out = (int)v;
    return true;
}

static int parse_direction_char(char c) {                                                  /* direction.rs:60-76 */
    switch (tolower((unsigned char)c)) {
        case 'n': return DIR_NORTH; case 'e': return DIR_EAST;
        case 's': return DIR_SOUTH; case 'w': return DIR_WEST;
        default: return -1;
    }
}

static void free_parsing(parsing_data* d) {
    free(d->gems); for (int a = 0; a < d->n_agents; a++) free(d->starts[a]);
    free(d->starts); free(d->n_starts); free(d->voids); free(d->exits); free(d->walls);
    free(d->laser_pos); free(d->laser_dir); free(d->laser_agent);
}

/* parser_v1.rs:132-175 `parse` */
static int parse_v1(const char* text, parsing_data* d) {
    memset(d, 0, sizeof *d);
    const char* p = text;
    while (*p) {
        const char* eol = strchr(p, '\n');
        size_t n = eol ? (size_t)(eol - p) : strlen(p);
        const char* ls = p; const char* le = p + n;
        p = eol ? eol + 1 : p + n;
        while (ls < le && isspace((unsigned char)*ls)) ls++;       /* line.trim() */
        while (le > ls && isspace((unsigned char)le[-1])) le--;
        if (ls == le) continue;                                    /* skip empty lines */
        int n_cols = 0;
        const char* q = ls;
        while (q < le) {
            while (q < le && isspace((unsigned char)*q)) q++;
            if (q >= le) break;
            const char* ts = q;
            while (q < le && !isspace((unsigned char)*q)) q++;
            int tl = (int)(q - ts);
            pos_t pos = { d->height, n_cols };
            n_cols++;
            switch (toupper((unsigned char)ts[0])) {
                case '.': break;
                case 'G': PUSH(d->gems, d->n_gems, pos); break;
                case '@': PUSH(d->walls, d->n_walls, pos); break;
                case 'X': PUSH(d->exits, d->n_exits, pos); break;
                case 'V': PUSH(d->voids, d->n_voids, pos); break;
                case 'S': {
                    int agent_id;
                    if (!parse_usize(ts + 1, tl - 1, &agent_id)) return OW_ERR_INVALID_AGENT_ID;
                    /* parser_v1.rs:27-43 add_start_position */
                    while (d->n_agents <= agent_id) {
                        d->starts = realloc(d->starts, sizeof *d->starts * (d->n_agents + 1));
                        d->n_starts = realloc(d->n_starts, sizeof *d->n_starts * (d->n_agents + 1));
                        d->starts[d->n_agents] = NULL; d->n_starts[d->n_agents] = 0;
                        d->n_agents++;
                    }
                    if (d->n_starts[agent_id] != 0) return OW_ERR_DUPLICATE_START;
                    PUSH(d->starts[agent_id], d->n_starts[agent_id], pos);
                    break;
                }
                case 'L': {
                    /* laser_config.rs:21-37 LaserConfig::from_str */
                    int dir = parse_direction_char(ts[tl - 1]);
                    if (dir < 0) return OW_ERR_INVALID_DIRECTION;
                    int agent_id;
                    if (tl < 2 || !parse_usize(ts + 1, tl - 2, &agent_id)) return OW_ERR_INVALID_AGENT_ID;
                    /* parser_v1.rs:22-25 add_laser_source: also a wall */
                    int nl = d->n_lasers + 1;
                    d->laser_pos = realloc(d->laser_pos, sizeof *d->laser_pos * nl);
                    d->laser_dir = realloc(d->laser_dir, sizeof *d->laser_dir * nl);
                    d->laser_agent = realloc(d->laser_agent, sizeof *d->laser_agent * nl);
                    d->laser_pos[nl - 1] = pos; d->laser_dir[nl - 1] = dir; d->laser_agent[nl - 1] = agent_id;
                    d->n_lasers = nl;
                    PUSH(d->walls, d->n_walls, pos);
                    break;
                }
                default: return OW_ERR_INVALID_TILE;
            }
        }
        /* parser_v1.rs:61-76 add_row */
        if (d->has_width) { if (d->width != n_cols) return OW_ERR_INCONSISTENT_DIMENSIONS; }
        else { d->width = n_cols; d->has_width = true; }
        d->height++;
    }
    if (d->height == 0) return OW_ERR_EMPTY_WORLD;                  /* parser_v1.rs:82-84 */
    return OW_OK;
}

static const int DIR_DELTA[4][2] = { {-1, 0}, {0, 1}, {1, 0}, {0, -1} };                   /* direction.rs:20-27 */

static void compute_available_actions(ow_world* w);
void ow_reset(ow_world* w);
void ow_free(ow_world* w);

static bool pos_eq(pos_t a, pos_t b) { return a.i == b.i && a.j == b.j; }

/* world_config.rs:107-122 into_world, :124-147 pre_validate, :176-199 make_grid, :203-250 laser_setup,
 * :149-168 post_validate, then world.rs:48-84 World::new */
static ow_world* build_world(parsing_data* d, int* err) {
    /* pre_validate */
    if (d->n_agents == 0) { *err = OW_ERR_NO_AGENTS; return NULL; }
    if (d->n_exits < d->n_agents) { *err = OW_ERR_NOT_ENOUGH_EXITS; return NULL; }

    ow_world* w = (ow_world*)calloc(1, sizeof *w);
    w->width = d->width; w->height = d->height;
    int H = w->height, W = w->width;
    w->grid = (tile_t**)calloc((size_t)H * W, sizeof *w->grid);
    /* make_grid: floor everywhere, then gems, exits, voids, walls (in that order) */
    for (int c = 0; c < H * W; c++) w->grid[c] = new_tile(T_FLOOR);
#define REPLACE(P, K) do { tile_t** s = &w->grid[(P).i * W + (P).j]; free(*s); *s = new_tile(K); } while (0)
    for (int k = 0; k < d->n_gems; k++) REPLACE(d->gems[k], T_GEM);
    for (int k = 0; k < d->n_exits; k++) REPLACE(d->exits[k], T_EXIT);
    for (int k = 0; k < d->n_voids; k++) REPLACE(d->voids[k], T_VOID);
    for (int k = 0; k < d->n_walls; k++) REPLACE(d->walls[k], T_WALL);

    /* laser_setup */
    bool* is_laser_pos = (bool*)calloc((size_t)H * W, sizeof(bool));
    w->n_sources = d->n_lasers;
    w->beams = (beam_t**)calloc((size_t)d->n_lasers + 1, sizeof *w->beams);
    for (int s = 0; s < d->n_lasers; s++) {
        pos_t sp = d->laser_pos[s];
        int di = DIR_DELTA[d->laser_dir[s]][0], dj = DIR_DELTA[d->laser_dir[s]][1];
        int n_beam = 0; pos_t* beam_positions = NULL;
        int i = sp.i + di, j = sp.j + dj;
        while (i >= 0 && j >= 0 && i < H && j < W) {
            if (!tile_is_walkable(w->grid[i * W + j])) break;
            pos_t bp = { i, j };
            PUSH(beam_positions, n_beam, bp);
            i += di; j += dj;
        }
        for (int k = 0; k < n_beam; k++) is_laser_pos[beam_positions[k].i * W + beam_positions[k].j] = true;
        /* laser_config.rs:39-46 build */
        beam_t* b = (beam_t*)calloc(1, sizeof *b);
        b->len = n_beam; b->on = (bool*)malloc(sizeof(bool) * (n_beam + 1));
        for (int k = 0; k < n_beam; k++) b->on[k] = true;
        b->enabled = true; b->agent_id = d->laser_agent[s]; b->direction = d->laser_dir[s];
        b->laser_id = s; b->source = sp;
        w->beams[s] = b;
        bool is_blocked = false;
        for (int k = 0; k < n_beam; k++) {
            pos_t bp = beam_positions[k];
            if (b->agent_id < d->n_agents && d->n_starts[b->agent_id] == 1 &&
                pos_eq(d->starts[b->agent_id][0], bp)) {
                is_blocked = true;
            }
            tile_t* laser = new_tile(T_LASER);
            laser->wrapped = w->grid[bp.i * W + bp.j];
            laser->beam = b; laser->offset = k;
            if (!is_blocked) {
                for (int a = 0; a < d->n_agents; a++) {
                    if (a == b->agent_id) continue;
                    int m = 0;
                    for (int q = 0; q < d->n_starts[a]; q++)
                        if (!pos_eq(d->starts[a][q], bp)) d->starts[a][m++] = d->starts[a][q];
                    d->n_starts[a] = m;
                }
            }
            w->grid[bp.i * W + bp.j] = laser;
        }
        free(beam_positions);
        tile_t** s_slot = &w->grid[sp.i * W + sp.j];
        /* grid[pos] = Tile::LaserSource(source): the previous tile (a Wall) is dropped */
        free(*s_slot);
        *s_slot = new_tile(T_SOURCE);
        (*s_slot)->beam = b;
    }
    for (int c = 0; c < H * W; c++) if (is_laser_pos[c]) { pos_t p = { c / W, c % W }; PUSH(w->lasers_positions, w->n_lasers_pos, p); }
    free(is_laser_pos);

    /* keep the vectors the World owns */
    w->n_agents = d->n_agents;
    w->n_gems = d->n_gems; w->gems_positions = d->gems; d->gems = NULL;
    w->n_voids = d->n_voids; w->void_positions = d->voids; d->voids = NULL;
    w->n_exits = d->n_exits; w->exits = d->exits; d->exits = NULL;
    w->n_walls = d->n_walls; w->wall_positions = d->walls; d->walls = NULL;
    w->random_start_positions = d->starts; w->n_starts = d->n_starts; d->starts = NULL; d->n_starts = NULL;
    int na = d->n_agents; d->n_agents = 0;
    w->agents = (agent_t*)calloc((size_t)na, sizeof *w->agents);
    for (int a = 0; a < na; a++) w->agents[a].id = a;
    w->agents_positions = (pos_t*)calloc((size_t)na, sizeof(pos_t));
    w->start_positions = (pos_t*)calloc((size_t)na, sizeof(pos_t));
    w->available = calloc((size_t)na, sizeof *w->available);
    w->n_available = (int*)calloc((size_t)na, sizeof(int));

    /* post_validate */
    int total = 0;
    for (int a = 0; a < na; a++) {
        if (w->n_starts[a] == 0) { *err = OW_ERR_AGENT_WITHOUT_START; ow_free(w); return NULL; }
        total += w->n_starts[a];
    }
    if (total < na) { *err = OW_ERR_NOT_ENOUGH_STARTS; ow_free(w); return NULL; }
    *err = OW_OK;
    ow_reset(w);   /* World::new calls reset (world.rs:82) */
    return w;
}

/* core/parsing/mod.rs:14-21 parse (TOML branch out of scope) + world.rs:629-643 try_from */
ow_world* ow_parse(const char* text, int* err) {
    int e = OW_OK;
    if (strchr(text, '=')) { if (err) *err = OW_ERR_TOML_UNSUPPORTED; return NULL; }
    parsing_data d;
    e = parse_v1(text, &d);
    ow_world* w = NULL;
    if (e == OW_OK) w = build_world(&d, &e);
    free_parsing(&d);
    if (err) *err = e;
    return w;
}

static void free_tile(tile_t* t) { if (!t) return; if (t->kind == T_LASER) free_tile(t->wrapped); free(t); }

void ow_free(ow_world* w) {
    if (!w) return;
    for (int c = 0; c < w->width * w->height; c++) free_tile(w->grid[c]);
    free(w->grid);
    for (int s = 0; s < w->n_sources; s++) { free(w->beams[s]->on); free(w->beams[s]); }
    free(w->beams); free(w->lasers_positions); free(w->gems_positions);
    if (w->random_start_positions) for (int a = 0; a < w->n_agents; a++) free(w->random_start_positions[a]);
    free(w->random_start_positions); free(w->n_starts); free(w->void_positions); free(w->exits);
    free(w->agents_positions); free(w->wall_positions); free(w->available); free(w->n_available);
    free(w->start_positions); free(w->agents);
    free(w);
}

/* ---------------------------------------------------------------- World (world.rs) */

/* world.rs:343-363 compute_available_actions */
static void compute_available_actions(ow_world* w) {
    static const int order[4] = { ACT_NORTH, ACT_EAST, ACT_SOUTH, ACT_WEST };
    static const int delta[5][2] = { {-1, 0}, {1, 0}, {0, 1}, {0, -1}, {0, 0} };          /* action.rs:18-26 */
    for (int a = 0; a < w->n_agents; a++) {
        int n = 0;
        w->available[a][n++] = ACT_STAY;
        if (!w->agents[a].dead && !w->agents[a].arrived) {
            for (int k = 0; k < 4; k++) {
                int act = order[k];
                int i = w->agents_positions[a].i + delta[act][0];
                int j = w->agents_positions[a].j + delta[act][1];
                if (i < 0 || j < 0) continue;                      /* position.rs:59-66 Err */
                if (i >= w->height || j >= w->width) continue;     /* world.rs:391-399 at() None */
                tile_t* t = at(w, i, j);
                if (tile_is_walkable(t) && !tile_is_occupied(t)) w->available[a][n++] = act;
            }
        }
        w->n_available[a] = n;
    }
}

/* world.rs:411-432 reset (single-start maps only: sample_different is deterministic, utils/mod.rs:63) */
void ow_reset(ow_world* w) {
    for (int c = 0; c < w->width * w->height; c++) tile_reset(w->grid[c]);
    for (int a = 0; a < w->n_agents; a++) { w->agents[a].dead = false; w->agents[a].arrived = false; w->agents[a].ghost = false; }
    for (int a = 0; a < w->n_agents; a++) {
        w->start_positions[a] = w->random_start_positions[a][0];
        w->agents_positions[a] = w->start_positions[a];
    }
    for (int a = 0; a < w->n_agents; a++) {
        pos_t p = w->agents_positions[a];
        if (tile_pre_enter(at(w, p.i, p.j), &w->agents[a])) ow_panic(w, "The agent should be able to pre-enter");
    }
    for (int a = 0; a < w->n_agents; a++) {
        pos_t p = w->agents_positions[a];
        event_t ev;
        tile_enter(w, at(w, p.i, p.j), &w->agents[a], &ev);
    }
    compute_available_actions(w);
}

/* utils/mod.rs:18-36 find_duplicates_into */
static void find_duplicates(const pos_t* input, int n, bool* result) {
    for (int i = 0; i < n; i++) result[i] = false;
    for (int i = 0; i < n; i++) {
        if (!result[i]) {
            for (int j = i + 1; j < n; j++) {
                if (pos_eq(input[i], input[j])) { result[i] = true; result[j] = true; }
            }
        }
    }
}

/* world.rs:365-378 solve_vertex_conflicts */
static void solve_vertex_conflicts(ow_world* w, pos_t* new_pos) {
    bool scratch[OW_MAX_AGENTS];
    bool conflict = true;
    while (conflict) {
        conflict = false;
        find_duplicates(new_pos, w->n_agents, scratch);
        for (int i = 0; i < w->n_agents; i++) {
            if (scratch[i]) { conflict = true; new_pos[i] = w->agents_positions[i]; }
        }
    }
}

/* world.rs:477-505 move_agents */
static bool move_agents(ow_world* w, const pos_t* new_positions, event_t* events, int* n_events, int cap) {
    for (int a = 0; a < w->n_agents; a++) {
        if (!w->agents[a].dead) {
            pos_t p = w->agents_positions[a];
            tile_leave(w, at(w, p.i, p.j));
        }
    }
    for (int a = 0; a < w->n_agents; a++) {
        pos_t p = new_positions[a];
        if (tile_pre_enter(at(w, p.i, p.j), &w->agents[a]))
            ow_panic(w, "When moving agents, the pre-enter should not fail");
    }
    bool agent_died = false;
    for (int a = 0; a < w->n_agents; a++) {
        pos_t p = new_positions[a];
        event_t ev;
        if (tile_enter(w, at(w, p.i, p.j), &w->agents[a], &ev)) {
            if (ev.type == EV_AGENT_DIED) agent_died = true;
            if (*n_events < cap) events[*n_events] = ev;
            (*n_events)++;
        }
    }
    return agent_died;
}

/* world.rs:435-475 step.  events_out: pairs (type, agent); returns OW_RT_* or 1+agent_id. */
int ow_step(ow_world* w, const uint8_t* actions, int n_actions, uint8_t* events_out, int cap, int* n_events_out) {
    static const int delta[5][2] = { {-1, 0}, {1, 0}, {0, 1}, {0, -1}, {0, 0} };
    if (n_events_out) *n_events_out = 0;
    if (w->n_agents != n_actions) return OW_RT_INVALID_NUMBER_OF_ACTIONS;
    for (int a = 0; a < w->n_agents; a++) {
        bool ok = false;
        for (int k = 0; k < w->n_available[a]; k++) if (w->available[a][k] == actions[a]) ok = true;
        if (!ok) return 1 + a;
    }
    pos_t new_positions[OW_MAX_AGENTS];
    for (int a = 0; a < w->n_agents; a++) {
        new_positions[a].i = w->agents_positions[a].i + delta[actions[a]][0];
        new_positions[a].j = w->agents_positions[a].j + delta[actions[a]][1];
    }
    solve_vertex_conflicts(w, new_positions);
    event_t events[4 * OW_MAX_AGENTS]; int n_events = 0;
    bool agent_died = move_agents(w, new_positions, events, &n_events, 4 * OW_MAX_AGENTS);
    for (int a = 0; a < w->n_agents; a++) w->agents_positions[a] = new_positions[a];
    while (agent_died) agent_died = move_agents(w, new_positions, events, &n_events, 4 * OW_MAX_AGENTS);
    compute_available_actions(w);
    for (int k = 0; k < n_events && k < cap; k++) { events_out[2 * k] = (uint8_t)events[k].type; events_out[2 * k + 1] = (uint8_t)events[k].agent; }
    if (n_events_out) *n_events_out = n_events;
    return OW_RT_OK;
}

/* world.rs:129-139 gems(): gem tiles in gems_positions order, looking through lasers (laser.rs:122-128) */
static tile_t* gem_at(ow_world* w, pos_t p) {
    tile_t* t = at(w, p.i, p.j);
    while (t->kind == T_LASER) t = t->wrapped;
    return t; /* T_GEM by construction */
}

/* world.rs:507-513 get_state.  pos: n_agents (i,j) int32 pairs; gems: n_gems bytes; alive: n_agents bytes */
void ow_get_state(ow_world* w, int32_t* pos, uint8_t* gems, uint8_t* alive) {
    for (int a = 0; a < w->n_agents; a++) { pos[2 * a] = w->agents_positions[a].i; pos[2 * a + 1] = w->agents_positions[a].j; }
    for (int g = 0; g < w->n_gems; g++) gems[g] = gem_at(w, w->gems_positions[g])->collected;
    for (int a = 0; a < w->n_agents; a++) alive[a] = !w->agents[a].dead;
}

/* world.rs:515-597 set_state */
int ow_set_state(ow_world* w, const int32_t* pos, int n_pos, const uint8_t* gems, int n_gems_given,
                 const uint8_t* alive, uint8_t* events_out, int cap, int* n_events_out) {
    if (n_events_out) *n_events_out = 0;
    if (n_gems_given != w->n_gems) return OW_RT_INVALID_NUMBER_OF_GEMS;
    if (n_pos != w->n_agents) return OW_RT_INVALID_NUMBER_OF_AGENTS;
    int A = w->n_agents;
    pos_t req[OW_MAX_AGENTS];
    for (int a = 0; a < A; a++) { req[a].i = pos[2 * a]; req[a].j = pos[2 * a + 1]; }
    bool dup[OW_MAX_AGENTS];
    find_duplicates(req, A, dup);
    for (int a = 0; a < A; a++) if (dup[a]) return OW_RT_INVALID_WORLD_STATE;
    for (int a = 0; a < A; a++)
        if (req[a].i < 0 || req[a].j < 0 || req[a].i >= w->height || req[a].j >= w->width) return OW_RT_OUT_OF_WORLD_POSITION;
    bool ghost_before[OW_MAX_AGENTS];  /* harness bookkeeping: only a successful set_state changes it */
    for (int a = 0; a < A; a++) ghost_before[a] = w->agents[a].ghost;
    /* current_state = self.get_state() */
    int32_t cur_pos[2 * OW_MAX_AGENTS]; uint8_t cur_alive[OW_MAX_AGENTS];
    uint8_t* cur_gems = (uint8_t*)malloc((size_t)w->n_gems + 1);
    ow_get_state(w, cur_pos, cur_gems, cur_alive);

    for (int c = 0; c < w->width * w->height; c++) tile_reset(w->grid[c]);
    /* collect gems: only direct Tile::Gem (world.rs:550-554) */
    for (int g = 0; g < w->n_gems; g++) {
        tile_t* t = at(w, w->gems_positions[g].i, w->gems_positions[g].j);
        if (gems[g] && t->kind == T_GEM) t->collected = true;
    }
    for (int a = 0; a < A; a++) {
        if (tile_pre_enter(at(w, req[a].i, req[a].j), &w->agents[a])) {
            int rc = ow_set_state(w, cur_pos, A, cur_gems, w->n_gems, cur_alive, NULL, 0, NULL);
            if (rc != OW_RT_OK) ow_panic(w, "set_state(current_state).unwrap() failed");
            free(cur_gems);
            for (int q = 0; q < A; q++) w->agents[q].ghost = ghost_before[q];
            return OW_RT_INVALID_AGENT_POSITION;
        }
    }
    for (int a = 0; a < A; a++) w->agents_positions[a] = req[a];
    event_t events[2 * OW_MAX_AGENTS]; int n_events = 0;
    for (int a = 0; a < A; a++) {
        w->agents[a].dead = false; w->agents[a].arrived = false;
        event_t ev;
        if (tile_enter(w, at(w, req[a].i, req[a].j), &w->agents[a], &ev)) events[n_events++] = ev;
        const bool died_here = w->agents[a].dead;
        if (!alive[a]) w->agents[a].dead = true;
        w->agents[a].ghost = w->agents[a].dead && !died_here;
    }
    /* actual_state != *state -> Err(InvalidWorldState) WITHOUT rollback (world.rs:588-594) */
    int32_t act_pos[2 * OW_MAX_AGENTS]; uint8_t act_alive[OW_MAX_AGENTS];
    ow_get_state(w, act_pos, cur_gems, act_alive);
    bool same = true;
    for (int a = 0; a < A; a++) if (act_pos[2 * a] != pos[2 * a] || act_pos[2 * a + 1] != pos[2 * a + 1] || (act_alive[a] != 0) != (alive[a] != 0)) same = false;
    for (int g = 0; g < w->n_gems; g++) if ((cur_gems[g] != 0) != (gems[g] != 0)) same = false;
    free(cur_gems);
    if (!same) { for (int q = 0; q < A; q++) w->agents[q].ghost = ghost_before[q]; return OW_RT_INVALID_WORLD_STATE; }
    compute_available_actions(w);
    if (events_out) for (int k = 0; k < n_events && k < cap; k++) { events_out[2 * k] = (uint8_t)events[k].type; events_out[2 * k + 1] = (uint8_t)events[k].agent; }
    if (n_events_out) *n_events_out = n_events;
    return OW_RT_OK;
}

/* ---------------------------------------------------------------- getters used by tests */
int ow_height(ow_world* w) { return w->height; }
int ow_width(ow_world* w) { return w->width; }
int ow_n_agents(ow_world* w) { return w->n_agents; }
int ow_n_gems(ow_world* w) { return w->n_gems; }
int ow_n_sources(ow_world* w) { return w->n_sources; }
int ow_n_exits(ow_world* w) { return w->n_exits; }
int ow_n_walls(ow_world* w) { return w->n_walls; }
int ow_n_voids(ow_world* w) { return w->n_voids; }
int ow_panics(ow_world* w) { return w->panics; }
const char* ow_panic_msg(ow_world* w) { return w->panic_msg; }

static void copy_pos(const pos_t* src, int n, int32_t* out) { for (int k = 0; k < n; k++) { out[2 * k] = src[k].i; out[2 * k + 1] = src[k].j; } }
void ow_agents_positions(ow_world* w, int32_t* out) { copy_pos(w->agents_positions, w->n_agents, out); }
void ow_start_positions(ow_world* w, int32_t* out) { copy_pos(w->start_positions, w->n_agents, out); }
void ow_exit_positions(ow_world* w, int32_t* out) { copy_pos(w->exits, w->n_exits, out); }
void ow_wall_positions(ow_world* w, int32_t* out) { copy_pos(w->wall_positions, w->n_walls, out); }
void ow_void_positions(ow_world* w, int32_t* out) { copy_pos(w->void_positions, w->n_voids, out); }
void ow_gem_positions(ow_world* w, int32_t* out) { copy_pos(w->gems_positions, w->n_gems, out); }
void ow_agents_flags(ow_world* w, uint8_t* alive, uint8_t* arrived) {
    for (int a = 0; a < w->n_agents; a++) { alive[a] = !w->agents[a].dead; arrived[a] = w->agents[a].arrived; }
}
/* world.rs:265-275 n_gems_collected: direct Gem tiles only */
int ow_n_gems_collected(ow_world* w) {
    int res = 0;
    for (int g = 0; g < w->n_gems; g++) {
        tile_t* t = at(w, w->gems_positions[g].i, w->gems_positions[g].j);
        if (t->kind == T_GEM && t->collected) res++;
    }
    return res;
}
/* available actions as a 5-bit mask per agent (bit = Action value) and as the reference's ordered list */
void ow_available_mask(ow_world* w, uint8_t* mask) {
    for (int a = 0; a < w->n_agents; a++) {
        uint8_t m = 0;
        for (int k = 0; k < w->n_available[a]; k++) m |= (uint8_t)(1u << w->available[a][k]);
        mask[a] = m;
    }
}
int ow_available_list(ow_world* w, int agent, uint8_t* out5) {
    for (int k = 0; k < w->n_available[agent]; k++) out5[k] = (uint8_t)w->available[agent][k];
    return w->n_available[agent];
}
/* sources: (i, j, direction, agent_id, enabled, len) per laser_id */
void ow_sources(ow_world* w, int32_t* out6) {
    for (int s = 0; s < w->n_sources; s++) {
        beam_t* b = w->beams[s];
        out6[6 * s + 0] = b->source.i; out6[6 * s + 1] = b->source.j; out6[6 * s + 2] = b->direction;
        out6[6 * s + 3] = b->agent_id; out6[6 * s + 4] = b->enabled; out6[6 * s + 5] = b->len;
    }
}
/* beam on/off bits of one source, offset order */
void ow_beam_bits(ow_world* w, int laser_id, uint8_t* out) {
    beam_t* b = w->beams[laser_id];
    for (int k = 0; k < b->len; k++) out[k] = b->on[k];
}
/* world.rs:159-172 lasers(): per laser position the outer Laser and, if the wrapped tile is a Laser,
 * that one too (not deeper).  out: (i, j, laser_id, agent_id, is_on, is_enabled) rows; returns count. */
int ow_lasers(ow_world* w, int32_t* out6, int cap) {
    int n = 0;
    for (int k = 0; k < w->n_lasers_pos; k++) {
        pos_t p = w->lasers_positions[k];
        tile_t* t = at(w, p.i, p.j);
        if (t->kind != T_LASER) { ow_panic(w, "lasers(): unreachable"); continue; }
        tile_t* layers[2] = { t, (t->wrapped->kind == T_LASER) ? t->wrapped : NULL };
        for (int q = 0; q < 2; q++) {
            tile_t* l = layers[q];
            if (!l) continue;
            if (n < cap) {
                out6[6 * n + 0] = p.i; out6[6 * n + 1] = p.j; out6[6 * n + 2] = l->beam->laser_id;
                out6[6 * n + 3] = l->beam->agent_id; out6[6 * n + 4] = laser_is_on(l); out6[6 * n + 5] = l->beam->enabled;
            }
            n++;
        }
    }
    return n;
}
/* occupant of a cell (Tile::agent), -1 if none; -2 if out of bounds */
int ow_tile_agent(ow_world* w, int i, int j) {
    if (i < 0 || j < 0 || i >= w->height || j >= w->width) return -2;
    return tile_agent(at(w, i, j));
}
/* Gem.collect() of the bindings (src/bindings/tiles/pygem.rs:52-66 -> tiles/gem.rs:17-19): `collected = true` on the tile AT the
 * position -- World::at_mut, so a gem lying under a beam is a Laser tile there and the call fails (-1; ValueError in Python) */
int ow_gem_collect(ow_world* w, int i, int j) {
    if (i < 0 || j < 0 || i >= w->height || j >= w->width) return -1;
    tile_t* t = at(w, i, j);
    if (t->kind != T_GEM) return -1;
    t->collected = true;
    return 0;
}
/* source mutators (laser_source.rs:37-47; pylaser_source.rs:55-75,107-142 without its start check) */
void ow_source_set_enabled(ow_world* w, int laser_id, int enabled) { if (enabled) beam_enable(w->beams[laser_id]); else beam_disable(w->beams[laser_id]); }
void ow_source_set_agent_id(ow_world* w, int laser_id, int agent_id) { w->beams[laser_id]->agent_id = agent_id; }

/* ---------------------------------------------------------------- World::set_exit_positions (world.rs:195-234)
 * laser.rs:109-115 Laser::set_tile: replaces the INNERMOST wrapped tile, whatever it is */
static void laser_set_tile(tile_t* l, tile_kind kind, int agent) {
    if (l->wrapped->kind == T_LASER) { laser_set_tile(l->wrapped, kind, agent); return; }
    free(l->wrapped);
    l->wrapped = new_tile(kind);
    l->wrapped->agent = agent;
}
/* Returns OW_OK, OW_ERR_NOT_ENOUGH_EXITS (:196-201), or -1 where the reference panics (`other => panic!`, an index out
 * of the grid).  The reference panics HALF WAY through the swap; the checks are made up front here so that the world
 * stays usable, which is also what the product does (include/lle_hip.h lle_map_set_exits). */
int ow_set_exits(ow_world* w, const int32_t* exits_ij, int n_exits) {
    if (n_exits < w->n_agents) return OW_ERR_NOT_ENOUGH_EXITS;
    int W = w->width, H = w->height;
    /* dry run of the two loops on a copy of the innermost kinds: does a `panic!` arm fire? */
    tile_kind* inner = (tile_kind*)malloc(sizeof *inner * (size_t)(H * W));
    for (int c = 0; c < H * W; c++) { tile_t* t = w->grid[c]; while (t->kind == T_LASER) t = t->wrapped; inner[c] = t->kind; }
    int bad = 0;
    for (int k = 0; k < w->n_exits; k++) {
        int c = w->exits[k].i * W + w->exits[k].j;
        if (w->grid[c]->kind != T_LASER && inner[c] != T_EXIT) bad = 1;   /* "Tile is not an exit" */
        inner[c] = T_FLOOR;
    }
    for (int k = 0; k < n_exits && !bad; k++) {
        int i = exits_ij[2 * k], j = exits_ij[2 * k + 1];
        if (i < 0 || j < 0 || i >= H || j >= W) { bad = 1; break; }       /* Vec index out of bounds */
        int c = i * W + j;
        if (w->grid[c]->kind != T_LASER) { if (inner[c] != T_FLOOR) bad = 1; }  /* "Tile is not a floor" */
        else if (inner[c] == T_GEM) bad = 1;  /* the gem would be dropped: World::gems() unwraps None afterwards (world.rs:128-139) */
        inner[c] = T_EXIT;
    }
    free(inner);
    if (bad) return -1;
    /* :203-216 Replace current exits by floor tiles */
    for (int k = 0; k < w->n_exits; k++) {
        tile_t** slot = &w->grid[w->exits[k].i * W + w->exits[k].j];
        tile_t* t = *slot;
        if (t->kind == T_EXIT) { int agent = t->agent; free(t); *slot = new_tile(T_FLOOR); (*slot)->agent = agent; }
        else if (t->kind == T_LASER) laser_set_tile(t, T_FLOOR, tile_agent(t));   /* Floor { agent: laser.agent() } */
    }
    /* :218 self.exits = exits */
    free(w->exits);
    w->exits = (pos_t*)malloc(sizeof(pos_t) * (size_t)(n_exits + 1));
    w->n_exits = n_exits;
    for (int k = 0; k < n_exits; k++) { w->exits[k].i = exits_ij[2 * k]; w->exits[k].j = exits_ij[2 * k + 1]; }
    /* :219-232 Set new exits */
    for (int k = 0; k < n_exits; k++) {
        tile_t** slot = &w->grid[w->exits[k].i * W + w->exits[k].j];
        tile_t* t = *slot;
        if (t->kind == T_FLOOR) { int agent = t->agent; free(t); *slot = new_tile(T_EXIT); (*slot)->agent = agent; }
        else if (t->kind == T_LASER) laser_set_tile(t, T_EXIT, tile_agent(t));    /* Exit { agent: laser.agent() } */
    }
    return OW_OK;
}

/* ---------------------------------------------------------------- layered observation
 * python/lle/observations.py:200-214 (channel layout), :216-237 (_setup, static), :254-266 (observe).
 * Values are {-1,0,1}; stored as int8.  One (C,H,W) slice (the reference tiles it A times).
 * Returns 0, or -1 if a laser colour addresses a layer >= C (numpy IndexError in the reference). */
int ow_layered_obs(ow_world* w, int8_t* obs) {
    int A = w->n_agents, H = w->height, W = w->width;
    int LASER_0 = A, WALL = 2 * A, VOID = WALL + 1, GEM = VOID + 1, EXIT = GEM + 1, C = EXIT + 1;
    memset(obs, 0, (size_t)C * H * W);
#define OBS(c, i, j) obs[((size_t)(c) * H + (i)) * W + (j)]
    for (int k = 0; k < w->n_walls; k++) OBS(WALL, w->wall_positions[k].i, w->wall_positions[k].j) = 1;
    for (int k = 0; k < w->n_voids; k++) OBS(VOID, w->void_positions[k].i, w->void_positions[k].j) = 1;
    for (int k = 0; k < w->n_exits; k++) OBS(EXIT, w->exits[k].i, w->exits[k].j) = 1;
    for (int s = 0; s < w->n_sources; s++) {
        beam_t* b = w->beams[s];
        if (LASER_0 + b->agent_id >= C) return -1;
        OBS(LASER_0 + b->agent_id, b->source.i, b->source.j) = -1;
    }
    /* for laser in world.lasers: outer Laser and, if it directly wraps another Laser, that one (world.rs:159-172) */
    for (int k = 0; k < w->n_lasers_pos; k++) {
        pos_t p = w->lasers_positions[k];
        tile_t* t = at(w, p.i, p.j);
        tile_t* layers[2] = { t, (t->wrapped->kind == T_LASER) ? t->wrapped : NULL };
        for (int q = 0; q < 2; q++) {
            tile_t* l = layers[q];
            if (!l || !laser_is_on(l)) continue;
            if (LASER_0 + l->beam->agent_id >= C) return -1;
            OBS(LASER_0 + l->beam->agent_id, p.i, p.j) = 1;
        }
    }
    for (int g = 0; g < w->n_gems; g++)
        if (!gem_at(w, w->gems_positions[g])->collected) OBS(GEM, w->gems_positions[g].i, w->gems_positions[g].j) = 1;
    for (int a = 0; a < A; a++) OBS(a, w->agents_positions[a].i, w->agents_positions[a].j) = 1;
#undef OBS
    return 0;
}

/* ================================================================== harness helpers
 * Not part of the reference: the counter-based action sampler shared (by definition)
 * with the HIP kernel, the throughput auto-reset policy (mirrors python/lle/env/env.py:253-254) and
 * a batched driver that lays results out in the canonical comparison layout used by tests/. */

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser */
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
static uint32_t rotl32(uint32_t x, uint32_t s) { return (x << (s & 31u)) | (x >> ((32u - s) & 31u)); }
/* 16-bit field of (seed, env, t, agent) (DESIGN.md "Action stream"): a 64-bit key per (seed, t), one 32-bit hash
 * (lowbias32 finaliser) per pair of agents, 16 bits per agent */
uint64_t ow_action_hash(uint64_t seed, uint64_t env, uint64_t t, uint64_t agent) {
    uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ULL * (t + 1));
    uint32_t pair = (uint32_t)(agent >> 1);
    uint32_t h = ((uint32_t)key ^ ((uint32_t)env * 0x9E3779B1u)) + rotl32((uint32_t)(key >> 32) ^ (uint32_t)(env >> 32), 15u) +
                 rotl32(0xC2B2AE35u, 3u * pair + 1u);
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return (h >> (16 * (agent & 1))) & 0xFFFF;
}
/* uniform (up to 2^-16) over the set bits of the 5-bit availability mask: k-th set bit in enum order N,S,E,W,STAY */
int ow_sample_action(uint8_t mask, uint64_t seed, uint64_t env, uint64_t t, uint64_t agent) {
    uint32_t n = (uint32_t)__builtin_popcount(mask & 31u);
    uint32_t k = ((uint32_t)ow_action_hash(seed, env, t, agent) * n) >> 16;
    for (int b = 0; b < 5; b++) if (mask & (1u << b)) { if (k == 0) return b; k--; }
    return ACT_STAY;
}
static bool world_done(ow_world* w) { /* LLE.compute_done: some agent died (by an event), or all arrived */
    bool all = true;
    for (int a = 0; a < w->n_agents; a++) { if (w->agents[a].dead && !w->agents[a].ghost) return true; if (!w->agents[a].arrived) all = false; }
    return all;
}

typedef struct ow_batch { int64_t n; ow_world** w; } ow_batch;

ow_batch* ow_batch_create(const char* text, int64_t n_envs, int* err) {
    ow_batch* b = (ow_batch*)calloc(1, sizeof *b);
    b->n = n_envs; b->w = (ow_world**)calloc((size_t)n_envs, sizeof *b->w);
    for (int64_t e = 0; e < n_envs; e++) {
        b->w[e] = ow_parse(text, err);
        if (!b->w[e]) { for (int64_t q = 0; q < e; q++) ow_free(b->w[q]); free(b->w); free(b); return NULL; }
    }
    return b;
}
void ow_batch_free(ow_batch* b) { if (!b) return; for (int64_t e = 0; e < b->n; e++) ow_free(b->w[e]); free(b->w); free(b); }
ow_world* ow_batch_world(ow_batch* b, int64_t e) { return b->w[e]; }
void ow_batch_reset(ow_batch* b) { for (int64_t e = 0; e < b->n; e++) ow_reset(b->w[e]); }

/* One batched step over envs [e0, e1).
 *   actions: NULL -> sample with (seed, env_offset + e, t); else [n][A] bytes.
 *   auto_reset: reset an env at the start of the step when it is done.
 * Outputs (any may be NULL), canonical layout:
 *   actions_out [n][A], err [n] int32 (0 ok, 1+agent invalid), ev_count [n] u8 (bit 7 = was auto-reset),
 *   events [n][2A][2] u8, obs [n][C*H*W] i8 */
void ow_batch_step_range(ow_batch* b, int64_t e0, int64_t e1, const uint8_t* actions, int auto_reset,
                         uint64_t seed, uint64_t t, int64_t env_offset,
                         uint8_t* actions_out, int32_t* err, uint8_t* ev_count, uint8_t* events, int8_t* obs,
                         int64_t* stats /* [8] or NULL, accumulated */) {
    for (int64_t e = e0; e < e1; e++) {
        ow_world* w = b->w[e];
        int A = w->n_agents;
        int was_reset = 0;
        if (auto_reset && world_done(w)) { ow_reset(w); was_reset = 1; }
        uint8_t act[OW_MAX_AGENTS];
        if (actions) memcpy(act, actions + e * A, (size_t)A);
        else {
            uint8_t mask[OW_MAX_AGENTS];
            ow_available_mask(w, mask);
            for (int a = 0; a < A; a++) act[a] = (uint8_t)ow_sample_action(mask[a], seed, (uint64_t)(env_offset + e), t, (uint64_t)a);
        }
        if (actions_out) memcpy(actions_out + e * A, act, (size_t)A);
        uint8_t ev[4 * OW_MAX_AGENTS * 2]; int n_ev = 0;
        int rc = ow_step(w, act, A, ev, 2 * A, &n_ev);
        if (err) err[e] = rc;
        if (ev_count) ev_count[e] = (uint8_t)(n_ev | (was_reset ? 0x80 : 0));
        if (events) { memset(events + e * 4 * A, 0, (size_t)4 * A); memcpy(events + e * 4 * A, ev, (size_t)2 * (n_ev < 2 * A ? n_ev : 2 * A)); }
        if (obs) { size_t sz = (size_t)(2 * A + 4) * w->height * w->width; ow_layered_obs(w, obs + e * sz); }
        if (stats) {
            stats[0] += 1; stats[1] += A; stats[6] += was_reset;
            if (rc > 0) stats[5] += 1;
            for (int k = 0; k < n_ev; k++) {
                if (ev[2 * k] == EV_GEM_COLLECTED) stats[2]++;
                else if (ev[2 * k] == EV_AGENT_EXIT) stats[3]++;
                else stats[4]++;
            }
        }
    }
}

/* canonical state dump for envs [e0,e1): pos [n][A][2] u8, alive/arrived/occupant [n][A] u8,
 * gems [n][G] u8, beams [n][L][beam_stride] u8 (on/off per offset), avail [n][A] u8 mask */
void ow_batch_dump(ow_batch* b, int64_t e0, int64_t e1, int beam_stride,
                   uint8_t* pos, uint8_t* alive, uint8_t* arrived, uint8_t* occupant,
                   uint8_t* gems, uint8_t* beams, uint8_t* avail) {
    for (int64_t e = e0; e < e1; e++) {
        ow_world* w = b->w[e];
        int A = w->n_agents, G = w->n_gems, L = w->n_sources;
        for (int a = 0; a < A; a++) {
            pos_t p = w->agents_positions[a];
            if (pos) { pos[(e * A + a) * 2] = (uint8_t)p.i; pos[(e * A + a) * 2 + 1] = (uint8_t)p.j; }
            if (alive) alive[e * A + a] = !w->agents[a].dead;
            if (arrived) arrived[e * A + a] = w->agents[a].arrived;
            if (occupant) occupant[e * A + a] = (tile_agent(at(w, p.i, p.j)) == a);
        }
        if (gems) for (int g = 0; g < G; g++) gems[e * G + g] = gem_at(w, w->gems_positions[g])->collected;
        if (beams) for (int s = 0; s < L; s++) {
            uint8_t* row = beams + ((size_t)e * L + s) * beam_stride;
            memset(row, 0, (size_t)beam_stride);
            for (int k = 0; k < w->beams[s]->len && k < beam_stride; k++) row[k] = w->beams[s]->on[k];
        }
        if (avail) ow_available_mask(w, avail + e * A);
    }
}

/* ---------------------------------------------------------------- multi-threaded rollout (cpu_baseline leg of bench.py)
 * Each thread owns a contiguous env range and runs `steps` sampled-action steps with auto-reset,
 * writing the layered observation of every env-step into obs[n][C*H*W] (same bytes the GPU path emits). */
#include <pthread.h>
#include <sched.h>
typedef struct { ow_batch* b; int64_t e0, e1; int steps; uint64_t seed; int8_t* obs; int64_t stats[8]; int cpu; } rollout_job;
static int g_pin_threads = 0;
/* bench.py cpu_baseline: pin the threads of a rollout to CPUs this process may run on, spread evenly over the allowed set
 * (thread k of n on the (k * allowed / n)-th allowed CPU: distinct cores and both sockets also when n < allowed);
 * 0 = leave it to the scheduler */
void ow_set_thread_pinning(int on) { g_pin_threads = on; }
static void* rollout_thread(void* arg) {
    rollout_job* j = (rollout_job*)arg;
    if (j->cpu >= 0) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(j->cpu, &one);
        (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
    }
    int64_t stats[8] = {0};  /* thread-local: the job structs of neighbouring threads share cache lines */
    for (int t = 0; t < j->steps; t++)
        ow_batch_step_range(j->b, j->e0, j->e1, NULL, 1, j->seed, (uint64_t)t, 0, NULL, NULL, NULL, NULL, j->obs, stats);
    for (int q = 0; q < 8; q++) j->stats[q] = stats[q];
    return NULL;
}
void ow_batch_rollout(ow_batch* b, int steps, uint64_t seed, int n_threads, int8_t* obs, int64_t* stats_out) {
    if (n_threads < 1) n_threads = 1;
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof *th);
    rollout_job* jobs = (rollout_job*)calloc((size_t)n_threads, sizeof *jobs);
    cpu_set_t allowed;
    int n_allowed = 0, cpus[CPU_SETSIZE];
    if (g_pin_threads && n_threads > 1 && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[n_allowed++] = c;
    for (int k = 0; k < n_threads; k++) {
        jobs[k].cpu = n_allowed <= 0 ? -1 : n_threads <= n_allowed ? cpus[(int64_t)k * n_allowed / n_threads] : cpus[k % n_allowed];
        jobs[k].b = b; jobs[k].e0 = b->n * k / n_threads; jobs[k].e1 = b->n * (k + 1) / n_threads;
        jobs[k].steps = steps; jobs[k].seed = seed; jobs[k].obs = obs;
        if (n_threads == 1) rollout_thread(&jobs[k]);
        else pthread_create(&th[k], NULL, rollout_thread, &jobs[k]);
    }
    for (int k = 0; k < n_threads; k++) {
        if (n_threads > 1) pthread_join(th[k], NULL);
        if (stats_out) for (int q = 0; q < 8; q++) stats_out[q] += jobs[k].stats[q];
    }
    free(th); free(jobs);
}
