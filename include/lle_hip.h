/*
 * lle_hip.h -- C ABI of lle_amd: a batched, MI355X-native replacement for the `World.step()` hot path of
 * yamoling/lle (reset / step / available_actions / get_state / set_state / layered observation).
 *
 * The reference's boundary for this path is the PyO3 class `lle.world.World`
 * (src/bindings/world/pyworld.rs:42-76) over the Rust `lle::World` (src/core/world.rs:21-44).  This header is what
 * a host in any language (the Python package in lle_amd/, a Rust `extern "C"` block, cgo ...) binds instead:
 * plain pointers and sizes, no torch types.  Each entry point cites the reference interface it replaces.
 *
 * Threading: a batch handle is NOT thread-safe (the reference serialises through a Mutex, pyworld.rs:69-82);
 * distinct handles are independent -- one per GPU of a node, from one thread or several.  Every call on a batch runs with
 * the batch's device current and puts the caller's current device back before it returns.  All device work of a call is enqueued on the `stream` argument
 * (a hipStream_t passed as void*; NULL = the default stream) and is asynchronous unless stated otherwise.
 *
 * Value codes (identical to the reference):
 *   actions  NORTH=0 SOUTH=1 EAST=2 WEST=3 STAY=4            (src/bindings/world/pyaction.rs:13-25)
 *   events   AGENT_EXIT=0 GEM_COLLECTED=1 AGENT_DIED=2        (src/bindings/world/pyevent.rs:9-16)
 *   direction NORTH=0 EAST=1 SOUTH=2 WEST=3                   (src/core/tiles/direction.rs:8-18)
 *   positions are (i, j) = (row, column)                       (src/bindings/world/pyposition.rs:3-9)
 */
#ifndef LLE_HIP_H
#define LLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLE_ABI_VERSION 3  /* 2: beam words (lle_map_info.n_beam_words, lle_laser_tile.word / bit), lle_batch_autotune
                            * 3: lle_batch_create_opt (observation rows in fp16 / bf16 / fp32), lle_debug_launched / _reachable */

/* static limits of this implementation (maps beyond them are rejected at parse/compile time) */
#define LLE_MAX_AGENTS 16
#define LLE_MAX_SOURCES 32
#define LLE_MAX_GEMS 32
#define LLE_MAX_BEAM_LEN 254       /* cells of one beam: any beam a map within LLE_MAX_DIM can hold */
#define LLE_MAX_BEAM_WORDS 32      /* 32-cell words over all beams of a map (a beam of len cells takes ceil(len / 32), at least 1) */
#define LLE_MAX_DIM 255

typedef struct lle_map lle_map;     /* a parsed + compiled map (host object; no GPU needed) */
typedef struct lle_batch lle_batch; /* n_envs lock-stepped worlds of one map, resident on one GPU */

/* ---- status codes of the C functions (whole-call errors) */
enum {
    LLE_OK = 0,
    LLE_ERR_NULL = -1,        /* NULL handle / pointer */
    LLE_ERR_ARG = -2,         /* bad argument (arity, range) */
    LLE_ERR_HIP = -3,         /* a HIP runtime call failed; see lle_last_error() */
    LLE_ERR_UNSUPPORTED = -4, /* map exceeds a static limit of the kernels (LDS footprint, layers ...) */
    LLE_ERR_NO_DEVICE = -5,   /* no HIP device: the product has NO CPU fallback */
    LLE_ERR_ARENA = -6        /* caller-provided arena too small or misaligned */
};

/* ---- parse errors: the v1-map subset of `ParseError` (src/core/parsing/errors.rs:5-74) */
enum {
    LLE_PARSE_OK = 0,
    LLE_PARSE_EMPTY_WORLD = 1,
    LLE_PARSE_NO_AGENTS = 2,
    LLE_PARSE_INVALID_TILE = 3,
    LLE_PARSE_NOT_ENOUGH_EXIT_TILES = 4,
    LLE_PARSE_DUPLICATE_START_TILE = 5,
    LLE_PARSE_INCONSISTENT_DIMENSIONS = 6,
    LLE_PARSE_INVALID_AGENT_ID = 7,
    LLE_PARSE_INVALID_DIRECTION = 8,
    LLE_PARSE_AGENT_WITHOUT_START = 9,
    LLE_PARSE_NOT_ENOUGH_START_TILES = 10,
    LLE_PARSE_TOML_UNSUPPORTED = 11, /* TOML (v2) maps are outside the hot-path scope */
    LLE_PARSE_INVALID_LEVEL = 12,
    LLE_PARSE_LIMIT = 13             /* valid map, but beyond LLE_MAX_* */
};

/* ---- per-environment result codes (buffer LLE_BUF_ERR), the batched form of `RuntimeWorldError`
 * (src/core/errors.rs:6-45).  An env with a non-zero code after `step` is left untouched and emits no
 * event, like the reference (src/core/world.rs:436-453). */
enum {
    LLE_ENV_OK = 0,
    /* 1..LLE_MAX_AGENTS: InvalidAction by agent (code - 1), lowest offending agent id */
    LLE_ENV_INVALID_WORLD_STATE = 0x40,
    LLE_ENV_OUT_OF_WORLD_POSITION = 0x41,
    LLE_ENV_INVALID_AGENT_POSITION = 0x42,
    LLE_ENV_INVALID_COLOUR = 0x43, /* lle_batch_set_sources: a colour >= n_agents ("Agent ID is greater than the number of
                                      agents", pylaser_source.rs:108-112); the env's sources are left unchanged */
    LLE_ENV_COLOUR_CROSSES_START = 0x44 /* lle_batch_set_sources: "Laser source cannot be changed to agent ID c since it would
                                      cross the start position of agent a" (pylaser_source.rs:121-139): a start of another agent
                                      lies on the source's beam; the env's sources are left unchanged */
};

/* ================================================================== maps (host only)
 * replaces World::try_from(&str) / World::get_level (src/core/world.rs:599-643; pyworld.rs:147-200) */

/* Parse a v1 text map.  Returns NULL and sets *parse_error (LLE_PARSE_*) on failure. */
lle_map* lle_map_parse(const char* text, size_t len, int* parse_error);
/* One of the six built-in levels, 1..6 (src/core/levels.rs:1-8). */
lle_map* lle_map_level(int level, int* parse_error);
void lle_map_free(lle_map* map);

typedef struct lle_map_info {
    int32_t height, width, n_agents, n_gems, n_sources;
    int32_t n_layers;        /* C = 2A + 4 (python/lle/observations.py:204-211) */
    int32_t n_exits, n_walls, n_voids, n_laser_tiles;
    int32_t obs_bytes;       /* C*H*W int8 */
    int32_t obs_stride;      /* obs_bytes rounded up to the row alignment, see lle_map_set_row_align (per-env pitch of LLE_BUF_OBS) */
    int32_t max_beam_len, max_cell_layers;
    int32_t obs_supported;   /* 0 if a laser colour addresses a layer >= C (IndexError in the reference) */
    int32_t table_bytes;     /* size of the device table blob */
    /* Beam words: a beam (LaserBeam{Vec<bool>}, src/core/tiles/laser.rs:15-21) is stored as ceil(len / 32) consecutive 32-bit words,
     * the words of source 0 first, then those of source 1, ... (a map with a beam longer than 32 cells is padded to 5 words).
     * LLE_BUF_BEAMS / LLE_BUF_SRC_COLOUR hold n_beam_words entries per env; == n_sources when no beam is longer than 32 cells
     * (every map of the reference's repository).  lle_laser_tile.word / .bit address a tile's bit. */
    int32_t n_beam_words;
    int32_t dyn_row_bytes;   /* bytes of a row that dynamic state can change, in whole 128-byte lines: what LLE_STEP_INCREMENTAL_OBS writes per
                              * env and step (== obs_stride where nothing can be skipped: rows that are not whole lines) */
} lle_map_info;
int lle_map_get_info(const lle_map* map, lle_map_info* out);

/* position lists (World properties start_pos/exit_pos/wall_pos/void_pos/gems positions, pyworld.rs:46-56,394-399) */
enum { LLE_POS_START = 0, LLE_POS_EXIT = 1, LLE_POS_WALL = 2, LLE_POS_VOID = 3, LLE_POS_GEM = 4 };
/* Writes up to `cap` (i, j) pairs; returns the total count (or a negative status). */
int lle_map_positions(const lle_map* map, int which, int32_t* out_ij, int cap);

/* laser sources in laser_id order (World.laser_sources, pyworld.rs:351-362) */
typedef struct lle_source_info { int32_t i, j, direction, agent_id, enabled, length, laser_id; } lle_source_info;
int lle_map_sources(const lle_map* map, lle_source_info* out, int cap);
/* LaserSource.enable/disable/set_agent_id (src/core/tiles/laser_source.rs:37-47).  -1 = leave unchanged.
 * Only changes the host object: call lle_batch_update_sources() to push it to a live batch. */
int lle_map_set_source(lle_map* map, int laser_id, int enabled, int agent_id);
/* World.exit_pos = [...] (setter: src/bindings/world/pyworld.rs:203-209 -> World::set_exit_positions,
 * src/core/world.rs:195-234): the current exits become Floor tiles and the `n_exits` given (i, j) cells Exit tiles -- under a
 * beam the innermost tile of the Laser stack is swapped (Laser::set_tile, src/core/tiles/laser.rs:109-115).  The occupant of
 * a swapped tile stays, agents keep their `arrived` flags: no dynamic state changes.  The list is kept as given (order,
 * duplicates under a beam) and is what lle_map_positions(LLE_POS_EXIT) returns from now on.
 *   fewer exits than agents  -> LLE_ERR_ARG, *parse_error = LLE_PARSE_NOT_ENOUGH_EXIT_TILES (ParseError::NotEnoughExitTiles)
 *   a cell that is not a Floor (wall, source, gem, void, the same direct cell twice), a gem under a beam, a position out of the
 *   world -> LLE_ERR_ARG with *parse_error = 0 and the reason in lle_last_error().  The reference PANICS there, half way
 *   through the swap, which poisons the world's mutex for good (pyworld.rs:205 `lock().unwrap()`); here the map is untouched.
 * Only changes the host object: call lle_batch_update_map() to push it to a live batch. */
int lle_map_set_exits(lle_map* map, const int32_t* exits_ij, int n_exits, int* parse_error);
/* An independent copy of a map (sources, exits, row alignment and head lines included). */
lle_map* lle_map_clone(const lle_map* map);
/* 1 if source `laser_id` may take colour `agent_id` without a possible start of another agent on its beam (the check of
 * the binding's LaserSource.set_colour, src/bindings/tiles/pylaser_source.rs:121-139), 0 if not, negative on bad arguments. */
int lle_map_colour_allowed(const lle_map* map, int laser_id, int agent_id);
/* The beam of source `laser_id` right after World.reset (src/core/world.rs:411-432) when the source is enabled and has
 * colour `agent_id`: bit k = the tile at offset k is on (all on, cut from where that agent's start lies inside the beam:
 * Laser::pre_enter, src/core/tiles/laser.rs:173-182).  What LLE_STEP_RECOLOUR_RESETS stores as the env's reset beams.
 * Beams longer than 63 cells: LLE_ERR_UNSUPPORTED (the result is one 64-bit mask).  Negative status on bad arguments. */
int64_t lle_map_reset_beam(const lle_map* map, int laser_id, int agent_id);

/* Pitch of an observation row (lle_map_info.obs_stride, the env stride of LLE_BUF_OBS and of every layered-style
 * lle_obs_desc): C*H*W rounded up to `align` bytes (16, 32, 64, 128 or 256).  The first C*H*W bytes of a row are the
 * reference's tensor (python/lle/observations.py:254-266), the rest is zero.  128 keeps the rows of neighbouring
 * environments in separate cache lines, so that every line goes to HBM once and whole.  0 -- the default -- picks 128
 * when that pads the row by at most 1/32 of its size (level 6: 1 872 -> 1 920 B) and 16 otherwise.  Always read the
 * pitch from lle_map_info.obs_stride / the buffer descriptors.  Call before lle_batch_create (a live batch keeps the
 * pitch it was created with). */
int lle_map_set_row_align(lle_map* map, int align);
/* The HEAD of a row: a run of whole 128-byte lines that hold no byte an agent, a beam or a gem can change -- behind
 * the agent layers, no laser tile, no gem -- i.e. the same bytes in every environment after every step.  The step
 * kernel stores these lines BEFORE its state machine runs (the memory system starts earlier; lle_amd/csrc/
 * step_kernel.hpp; launches of about one to two rounds of workgroups only, where it pays).  `lines` = the most lines
 * taken from the longest such run, 0..8 (0 = none), or -1 = automatic, the default: a fifth of the row.  The
 * LLE_HEAD_LINES environment variable overrides the default.  Rows that are not line-aligned have no head.  Results do
 * not depend on it. */
int lle_map_set_head_lines(lle_map* map, int lines);
/* The head chosen for the map's current sources: byte offset inside a row and length (0 = no head). */
int lle_map_row_head(const lle_map* map, int32_t* first_byte, int32_t* n_bytes);

/* The same for batches with per-environment sources (lle_batch_set_sources): the lines that hold no byte an agent, a gem or a
 * laser of ANY colour below n_agents can change (WALL / VOID / EXIT planes).  0 bytes when a source of the map itself has a
 * colour >= n_agents (quirk Q5: its layer aliases those planes). */
/* Which 128-byte lines of a row dynamic state can change (what LLE_STEP_INCREMENTAL_OBS writes): out_lines[l] = 1 / 0 for line l of the
 * obs_stride / 128 lines; returns the number of lines (all dynamic where the row is not a whole number of lines).  Host side. */
int lle_map_row_dynamic_lines(const lle_map* map, uint8_t* out_lines, int cap);
int lle_map_row_head_env_sources(const lle_map* map, int32_t* first_byte, int32_t* n_bytes);
/* ... and the SECOND run of such lines behind the first (level 6: bytes 1280-1535 are the first run, 1792-1919 the second): the two together
 * are the head lines the map asks for (lle_map_set_head_lines); 0 bytes: none. */
int lle_map_row_head_env_sources_second(const lle_map* map, int32_t* first_byte, int32_t* n_bytes);

/* static description of World.lasers (src/core/world.rs:159-172): per laser position the outer layer and, if
 * nested, the second one; `offset` = the tile's index in the beam of `laser_id`; its on / off bit is bit `bit` of word `word`
 * of the env's LLE_BUF_BEAMS record (word == laser_id, bit == offset unless a beam of the map is longer than 32 cells). */
typedef struct lle_laser_tile { int32_t i, j, laser_id, offset, layer, word, bit; } lle_laser_tile;
int lle_map_laser_tiles(const lle_map* map, lle_laser_tile* out, int cap);

/* World.world_string (pyworld.rs:212-218): v1 text with the current source colours.  Returns needed size. */
size_t lle_map_world_string(const lle_map* map, char* buf, size_t cap);

/* ================================================================== batches (device) */

enum {
    LLE_BUF_POS = 0,   /* u8  [n][A][2]   (i, j) of every agent               World.agents_positions */
    LLE_BUF_BITS,      /* u64 [n]         alive bits 0-15 | arrived 16-31 | occupant 32-47 | 48-63: dead by set_state without an
                          AgentDied event (LLE.compute_done counts events: such an agent does not end the episode) */
    LLE_BUF_GEMS,      /* u32 [n]         bit g = gem g collected (parse order; World.gems) */
    LLE_BUF_BEAMS,     /* u32 [n][Lw]     Lw = lle_map_info.n_beam_words; bit k of word source_first_word + w = the beam is on at offset 32 w + k
                                          (LaserBeam, laser.rs:15-21; Lw == n_sources and word == laser_id unless a beam is longer than 32 cells) */
    LLE_BUF_AVAIL,     /* u8  [n][A]      bit a = Action a available (World.available_actions) */
    LLE_BUF_ACTIONS,   /* u8  [n][A]      joint action taken by the last step (input, or sampled output) */
    LLE_BUF_ERR,       /* u8  [n]         LLE_ENV_* of the last step / set_state */
    LLE_BUF_EVCOUNT,   /* u8  [n]         number of events of the last step (bit 7: env was auto-reset first) */
    LLE_BUF_EVENTS,    /* u8  [n][2A]     entry = type << 4 | agent, in the reference's order */
    LLE_BUF_DONE,      /* u8  [n]         1 if any agent is dead or all have arrived (LLE.compute_done, env.py:253-254) */
    LLE_BUF_OBS,       /* i8  [n][obs_stride]  first C*H*W elements = layered observation (C,H,W); f16 / bf16 / f32 elements in a batch
                          created with lle_batch_options.obs_dtype (read elem_bytes from the descriptor) */
    LLE_BUF_STATS,     /* i64 [n_blocks][8] per-block partial counters (see lle_batch_stats) */
    LLE_BUF_REQ_POS,   /* u8  [n][A][2]   set_state request */
    LLE_BUF_REQ_GEMS,  /* u32 [n] */
    LLE_BUF_REQ_ALIVE, /* u16 [n] */
    LLE_BUF_REWARD,    /* u8  [n][4]      per-step (gems collected, exits, deaths, all agents arrived) of the last step:
                          the inputs of the reference's reward strategies (python/lle/env/reward_strategy.py:58-109) */
    LLE_BUF_SRC_COLOUR,  /* u8  [n][Lw]    colour (agent_id) per beam word of this env (the words of a source carry the same one); valid after lle_batch_set_sources */
    LLE_BUF_SRC_ENABLED, /* u32 [n]        bit l = source l of this env enabled;          valid after lle_batch_set_sources */
    LLE_BUF_COUNT
};

/* The per-agent buffers (POS, AVAIL, ACTIONS, EVENTS, REQ_POS) are laid out with an env pitch of 4, 8 or 16 agents
 * (the smallest that holds the map's A), so that a record is a whole number of dwords: always address them through
 * `stride` below. */
typedef struct lle_buffer_desc {
    void* ptr;            /* device pointer */
    int64_t arena_offset; /* byte offset inside the arena */
    int64_t bytes;
    int32_t elem_bytes;
    int32_t ndim;
    int64_t shape[3];
    int64_t stride[3];    /* in elements */
} lle_buffer_desc;

/* Bytes of device memory a batch needs (so a host can allocate the arena itself, 256-B aligned). */
int64_t lle_batch_arena_bytes(const lle_map* map, int64_t n_envs);

/* Create n_envs worlds of `map` on HIP device `device_id` and reset them (World::new calls reset, world.rs:82).
 * arena: caller-owned device memory of >= lle_batch_arena_bytes() bytes, or NULL to let the handle hipMalloc it.
 * Returns NULL on failure (see lle_last_status()/lle_last_error()).  There is no CPU fallback. */
lle_batch* lle_batch_create(const lle_map* map, int64_t n_envs, int device_id, void* arena, int64_t arena_bytes,
                            void* stream);
void lle_batch_free(lle_batch* b);

int lle_batch_get_buffer(const lle_batch* b, int which, lle_buffer_desc* out);
int64_t lle_batch_n_envs(const lle_batch* b);

/* Exact checkpoint of the dynamic state (positions, alive/arrived/occupant bits, gems, beam masks, availability; with
 * per-environment sources also every env's colours and flags): unlike World.get_state/set_state (world_state.rs:5-9)
 * nothing is re-derived, so it is valid mid-episode, corpses included.  The MAP is not part of it: a snapshot restored
 * after lle_batch_update_map (exits moved) comes back under the current exits, LLE_BUF_OBS / LLE_BUF_DONE are rebuilt and
 * the per-env reset records recomputed.  `dst_dev` / `src_dev`: device memory of lle_batch_snapshot_bytes() bytes. */
int64_t lle_batch_snapshot_bytes(const lle_batch* b);
int lle_batch_snapshot(lle_batch* b, void* dst_dev, void* stream);
int lle_batch_restore(lle_batch* b, const void* src_dev, void* stream);

/* World.reset (src/core/world.rs:411-432) for every env, or for envs whose byte in env_mask (device, u8[n]) != 0. */
int lle_batch_reset(lle_batch* b, const uint8_t* env_mask_dev, void* stream);

/* World.step (src/core/world.rs:435-475) + Layered.observe (python/lle/observations.py:254-266) for every env. */
enum {
    LLE_STEP_SAMPLE_ACTIONS = 1, /* ignore `actions`: draw uniformly from each agent's available actions with the
                                    counter-based sampler of DESIGN.md (seed, env, t, agent); writes LLE_BUF_ACTIONS */
    LLE_STEP_AUTO_RESET = 2,     /* reset an env at the start of the step when LLE_BUF_DONE says it is over */
    LLE_STEP_NO_OBS = 4,         /* skip the observation write */
    LLE_STEP_INCREMENTAL_OBS = 16, /* lle_batch_step / lle_batch_step_outputs (single steps, the map's own sources): write only the 128-byte lines
                                    of each row of LLE_BUF_OBS that dynamic state can change -- agent layers, lines with a laser tile or a gem.
                                    The other lines (WALL / VOID / EXIT planes, beam-less parts of the laser planes: a third of level 6's row)
                                    hold the same bytes after every step and are already in the buffer from the last full write (create, reset,
                                    observe, any source / exit update, any step without this flag), so the buffer's CONTENT is the same; the
                                    caller promises not to have written into LLE_BUF_OBS itself.  Ignored (full rows) by fused rollouts into
                                    rings and on rows that are not whole 128-byte lines; with per-environment sources every laser plane is dynamic. */
    LLE_STEP_RECOLOUR_RESETS = 8 /* with LLE_STEP_AUTO_RESET, batches with per-environment sources: an env that is reset also draws a
                                    fresh colour for each of its sources -- LLE.reset with randomize_lasers (python/lle/env/
                                    env.py:189-203: world.reset() under the colours the env had, then the new colours on the live
                                    world).  Uniform over the colours the source may take (lle_map_colour_allowed; all of
                                    [0, n_agents) on the maps where the reference's own draw cannot fail), drawn with the action
                                    sampler's hash keyed by (seed ^ 0xC01055EED, env_offset + env, t, laser_id); the new colours
                                    are in LLE_BUF_SRC_COLOUR after the step.  Maps with a cell of more than two laser layers:
                                    LLE_ERR_UNSUPPORTED (lle_batch_reset_sources serves those). */
};
/* actions_dev: device u8 [n][A], or NULL to use LLE_BUF_ACTIONS as already filled by the host. */
int lle_batch_step(lle_batch* b, const uint8_t* actions_dev, uint32_t flags, uint64_t seed, uint64_t t,
                   int64_t env_offset, void* stream);

/* Fused rollout: `n_steps` consecutive steps in ONE launch.  Actions: sampled on the device (LLE_STEP_SAMPLE_ACTIONS;
 * step j uses time index t0 + j) and WRITTEN to the action ring, or -- without that flag -- READ from it: the caller
 * fills slot (ring_pos + j) % ring_slots of `ring->actions` with the joint actions of step j beforehand (an open-loop
 * plan, a replayed trajectory, an action repeated k times: with ring == NULL every step takes LLE_BUF_ACTIONS).  An env
 * whose action is refused at some step keeps its state for that step, like lle_batch_step, and goes on with the next.
 * Identical results to n_steps calls of lle_batch_step;
 * the state stays in registers between steps and the waves drift apart, so the integer work of one step overlaps
 * the observation stream of another.  Per-step outputs go to slot (ring_pos + j) % ring_slots of caller-provided
 * trajectory rings (device memory), or in place (LLE_BUF_OBS / ACTIONS / REWARD, each step overwriting the last)
 * when `ring` is NULL.  Events / err / evcount / done always reflect the last step. */
typedef struct lle_rollout_ring {
    int32_t ring_slots;   /* R >= 1 */
    int32_t pad;
    uint64_t ring_pos;    /* slot of the first step */
    int8_t* obs;          /* [R][n_envs][obs_stride] elements of the batch's observation type (int8 unless lle_batch_options.obs_dtype says otherwise) */
    uint8_t* actions;     /* [R][n_envs][agent pitch] (pitch: lle_buffer_desc.stride[0] of LLE_BUF_ACTIONS); output with
                             LLE_STEP_SAMPLE_ACTIONS, input without */
    uint32_t* reward;     /* [R][n_envs]  gems | exits << 8 | deaths << 16 | all_arrived << 24 */
} lle_rollout_ring;
int lle_batch_rollout(lle_batch* b, uint32_t n_steps, uint32_t flags, uint64_t seed, uint64_t t0, int64_t env_offset,
                      const lle_rollout_ring* ring, void* stream);

/* World.set_state (src/core/world.rs:515-597) with the reference's semantics (incl. its lossy re-derivation of
 * beams and its rollback rules) from LLE_BUF_REQ_*; per-env result in LLE_BUF_ERR, events in LLE_BUF_EVENTS. */
int lle_batch_set_state(lle_batch* b, void* stream);

/* Push the map's current source colours / enabled flags to the device tables and apply
 * LaserBeam::enable/disable (laser.rs:69-77) to the beam masks of every env; rewrites the observation. */
int lle_batch_update_sources(lle_batch* b, const lle_map* map, void* stream);

/* Push another compilation of map `map_index` of the batch (0 for a one-map batch) to the device: the same map after
 * lle_map_set_exits and / or lle_map_set_source.  The tables (cell kinds, static observation with its EXIT plane, row head)
 * and the reset state of the map's envs follow -- an agent whose start is an exit now arrives at reset, world.rs:411-432 --,
 * the dynamic state of live envs does not (world.rs:195-234 keeps occupants and flags), LLE_BUF_OBS is rewritten.  With
 * per-environment sources every env keeps ITS colours / flags and gets its reset state recomputed; the map's own source
 * colours / flags must then be unchanged (likewise in a batch of several maps), else LLE_ERR_ARG.  LLE_ERR_ARG too when the
 * map is not a recompilation of the batch's map (other tiles, other row pitch). */
int lle_batch_update_map(lle_batch* b, int map_index, const lle_map* map, void* stream);

/* Per-ENVIRONMENT sources: what LLE.reset does with randomize_lasers (python/lle/env/env.py:198-200:
 * `source.set_colour(random.randint(0, n_agents - 1))`) and LaserSource.enable / disable, for every env at once.
 *   colours_dev  u8  [n_envs][L]  new colour of every source, or NULL (keep)   -> LaserBeam::set_agent_id (laser.rs:84-86)
 *   enabled_dev  u32 [n_envs]     bit l = source l enabled, or NULL (keep)     -> LaserBeam::enable / disable where the
 *                                 flag CHANGES (pylaser_source.rs:55-75): enable re-lights the whole beam, disable
 *                                 clears it (laser.rs:69-77)
 *   env_mask_dev u8  [n_envs] or NULL: envs to touch.
 * Refused per env, with NONE of its sources changed: a colour >= n_agents (LLE_BUF_ERR = LLE_ENV_INVALID_COLOUR) and a
 * colour that would put the start of another agent on the source's beam (LLE_ENV_COLOUR_CROSSES_START: the check of the
 * binding's LaserSource.set_colour, pylaser_source.rs:121-139, on the tiles World.lasers() exposes; which (source, colour)
 * pairs pass is a property of the map: lle_map_colour_allowed).  The reference raises ValueError there having ALREADY
 * recoloured the core source and every source before it in the loop of env.py:198-200; a batch cannot raise per env, and
 * leaving the env untouched is the state a caller can reason about.  Rewrites LLE_BUF_OBS.
 * From the first call on the batch keeps colours and flags per env (LLE_BUF_SRC_*): reset, step (auto-reset restarts an
 * env from ITS reset state), set_state and every observation builder use them; lle_batch_update_sources then
 * broadcasts the map's sources to every env. */
int lle_batch_set_sources(lle_batch* b, const uint8_t* colours_dev, const uint32_t* enabled_dev, const uint8_t* env_mask_dev,
                          void* stream);

/* LLE.reset with randomize_lasers in one launch (python/lle/env/env.py:189-203): lle_batch_reset of the selected envs
 * followed by lle_batch_set_sources on the same envs -- World::reset under the sources the env HAD, then the new
 * colours / flags on the live world, exactly the state the two calls in a row leave (events cleared, LLE_BUF_ERR =
 * LLE_ENV_INVALID_COLOUR for a refused env, which is reset all the same).  `env_mask_dev` may be LLE_BUF_DONE itself.
 * flags: 0, or LLE_STEP_NO_OBS when the caller steps next and does not read the observation in between. */
int lle_batch_reset_sources(lle_batch* b, const uint8_t* colours_dev, const uint32_t* enabled_dev, const uint8_t* env_mask_dev,
                            uint32_t flags, void* stream);

/* Rebuild LLE_BUF_OBS from the current state. */
int lle_batch_observe(lle_batch* b, void* stream);

/* ---- batches of SEVERAL maps (e.g. one generated map per block of environments; SURVEY.md section 8(d), config 5 variant)
 * Map m owns the envs [m * envs_per_map, (m + 1) * envs_per_map); n_envs = n_maps * envs_per_map.  The maps must agree on
 * height, width and the numbers of agents, sources and gems (one tensor shape, one kernel instantiation); walls, exits,
 * starts, beams and colours are free.  envs_per_map: any positive number, down to ONE map per environment (the reference's WorldBuilder hands every env
 * its own map; a multiple of 8 was required until the end of round 5, of 16 before).  A wavefront serves one map: from 64 envs per map on the launches keep
 * their full shape; below that the workgroups -- and, below a wavefront's worth, the wavefronts -- are narrowed, which costs throughput (INTEGRATION.md
 * section 6; bench.py cfg5_multi_map).  Every entry point works on such a batch except lle_batch_update_sources (use lle_batch_set_sources). */
int64_t lle_batch_arena_bytes_multi(const lle_map* const* maps, int n_maps, int64_t envs_per_map);
lle_batch* lle_batch_create_multi(const lle_map* const* maps, int n_maps, int64_t envs_per_map, int device_id, void* arena,
                                  int64_t arena_bytes, void* stream);
int lle_batch_n_maps(const lle_batch* b);

/* ---- batch options: the element type of the layered observation.
 * The reference's Layered.observe returns FLOAT32 (python/lle/observations.py:223: np.zeros(..., dtype=np.float32), values -1 / 0 / 1),
 * and a learner's first layer usually wants fp16 / bf16.  A caller of an int8 batch pays a second pass over the tensor for the cast (bench.py
 * consumer_loop: step 19.9 us, int8 -> fp16 cast 85.9 us at level 6 x 65 536).  With obs_dtype set, every kernel that writes LLE_BUF_OBS -- the step
 * kernel, reset, observe, the source / exit updates -- and the fused rollout's observation ring WIDEN AT THE STORE: the row is built as int8 in
 * LDS as before and leaves the chip in the caller's type, once.  LLE_BUF_OBS then holds [n][obs_stride] ELEMENTS of that type
 * (lle_buffer_desc.elem_bytes = 1 / 2 / 2 / 4; stride and shape in elements, unchanged), lle_rollout_ring.obs likewise
 * [R][n][obs_stride] elements.  Content: exactly the int8 tensor, cast (tests/test_gpu_obs_dtype.py).  The other layered-style observations
 * (lle_batch_observe_as of LLE_OBS_LAYERED / _LAYERED_PADDED / _PERSPECTIVE / _PARTIAL, the partial observation of lle_batch_step_outputs) come in the
 * same type since the end of round 5 -- lle_obs_desc.elem_bytes says so, its strides stay in elements, `bytes` is the buffer to hand --; the state
 * vector (LLE_OBS_STATE / _NORMALIZED_STATE) is float32 whatever the batch.
 * `opt` == NULL or obs_dtype == LLE_DTYPE_I8: what lle_batch_create / lle_batch_create_multi give.  n_maps == 1: one map (any envs_per_map). */
enum { LLE_DTYPE_I8 = 0, LLE_DTYPE_F16 = 1, LLE_DTYPE_BF16 = 2, LLE_DTYPE_F32 = 3 };
typedef struct lle_batch_options {
    uint32_t struct_bytes;   /* sizeof(lle_batch_options): lets the struct grow */
    int32_t obs_dtype;       /* LLE_DTYPE_* of LLE_BUF_OBS and of the observation rings */
    int32_t reserved[6];     /* zero */
} lle_batch_options;
int64_t lle_batch_arena_bytes_opt(const lle_map* const* maps, int n_maps, int64_t envs_per_map, const lle_batch_options* opt);
lle_batch* lle_batch_create_opt(const lle_map* const* maps, int n_maps, int64_t envs_per_map, int device_id, void* arena, int64_t arena_bytes,
                                const lle_batch_options* opt, void* stream);
int lle_batch_obs_dtype(const lle_batch* b);  /* LLE_DTYPE_* of the batch */

/* ---- the other observation builders of python/lle/observations.py, from the same device state -------------------
 * kind / param                      reference generator (file:line)                       element, logical shape per env
 * LLE_OBS_LAYERED           0       Layered            observations.py:274-276            i8  (C, H, W), C = 2A+4
 * LLE_OBS_LAYERED_PADDED    p >= 0  LayeredPadded      observations.py:196-266            i8  (2(A+p)+4, H, W)
 * LLE_OBS_PERSPECTIVE       0       AgentZeroPerspective observations.py:372-395          i8  (A, C, H, W)   (one slice per agent)
 * LLE_OBS_PARTIAL            k odd  PartialGenerator   observations.py:312-369            i8  (A, 2A+3, k, k)
 * LLE_OBS_STATE             0       StateGenerator(normalize=False) observations.py:137-159 f32 (3A+G,)
 * LLE_OBS_NORMALIZED_STATE  0       StateGenerator(normalize=True)                          f32 (3A+G,)
 * "flattened" (FlattenedLayered, observations.py:279-290) is LLE_OBS_LAYERED read as one row of C*H*W values.
 * The reference returns the tensor tiled n_agents times (np.tile) for every kind but PARTIAL and PERSPECTIVE; one copy
 * per env is written here (expose the tiling as a broadcast view).  int8 holds the values {-1, 0, 1} exactly.
 * Rows are padded to a multiple of 16 bytes: lle_obs_desc.stride gives the element strides (env first). */
enum lle_obs_kind {
    LLE_OBS_LAYERED = 0, LLE_OBS_LAYERED_PADDED = 1, LLE_OBS_PERSPECTIVE = 2, LLE_OBS_PARTIAL = 3, LLE_OBS_STATE = 4,
    LLE_OBS_NORMALIZED_STATE = 5
};

typedef struct lle_obs_desc {
    int32_t kind, param;
    int32_t elem_bytes;  /* the layered-style kinds: element size of the batch's observation type (1 = int8 unless lle_batch_options.obs_dtype); state kinds: 4 (float32) */
    int32_t ndim;        /* dimensions incl. the env axis: shape[0] = n_envs */
    int64_t shape[6];
    int64_t stride[6];   /* in elements */
    int64_t bytes;       /* size of the output buffer for all envs */
    int32_t supported;   /* 0: the reference raises IndexError for this map (a laser colour has no layer) */
    int32_t pad;
} lle_obs_desc;

/* Shape, strides and size of the buffer lle_batch_observe_as writes for (kind, param). */
int lle_batch_obs_desc(lle_batch* b, int kind, int param, lle_obs_desc* out);

/* Write the observation of every env to `out_dev` (device memory, 256-byte aligned, >= desc.bytes).
 * LLE_OBS_PARTIAL with k = 3, 5, 7 on a batch of 4 096 environments and more: the FIRST call for a window size times four variants of the writer on this
 * batch (window tables or bitmap x two batch sizes; identical bytes; about 24 launches and ONE synchronisation of `stream`, so not inside a stream
 * capture) and later calls launch the fastest; LLE_PARTIAL_NO_TRIAL=1 in the environment skips the trial (profiles/r05_partial.md). */
int lle_batch_observe_as(lle_batch* b, int kind, int param, void* out_dev, int64_t out_bytes, void* stream);

/* LLE.available_actions (python/lle/env/env.py:146-163): u8 bools [n_envs][A][5] in Action value order N,S,E,W,STAY.
 * walkable_lasers = 0 drops the actions that lead onto an active laser of another colour. */
int lle_batch_available_actions(lle_batch* b, int walkable_lasers, uint8_t* out_dev, void* stream);

/* Everything LLE.step returns besides the observation, for every env, in ONE launch (python/lle/env/env.py:165-187:
 * `Step(obs, state, available_actions, reward, done, ...)`).  Each output is optional (NULL = not wanted):
 *   state      f32 [n][3A+G]  WorldState.as_array (src/bindings/world/pyworld_state.rs:79-101), divided by (H, W) per
 *                             coordinate when normalize_state != 0 (python/lle/observations.py:145-175)
 *   reward     f32 [n][1]     reward_kind 0: SingleObjective.compute_reward (python/lle/env/reward_strategy.py:58-75)
 *              f32 [n][4]     reward_kind 1: MultiObjective.compute_reward  (reward_strategy.py:90-109)
 *   done       u8  [n]        LLE.compute_done (env.py:253-254)
 *   available  u8  [n][A][5]  LLE.available_actions (env.py:146-163), walkable_lasers as in lle_batch_available_actions
 *   alive, arrived u8 [n][A]  the is-alive / has-arrived entries of Step.info (env.py:174-176)
 * All pointers are device memory. */
typedef struct lle_env_outputs {
    float* state;
    float* reward;
    uint8_t* done;
    uint8_t* available;
    uint8_t* alive;
    uint8_t* arrived;
    int32_t normalize_state;
    int32_t reward_kind;
    int32_t walkable_lasers;
    int32_t partial_k;       /* window size of `partial` (odd, 1..15) */
    int8_t* partial;         /* lle_batch_step_outputs only: the partial k x k observation (LLE_OBS_PARTIAL, partial_k; layout of
                              * lle_batch_obs_desc) written by the step launch itself INSTEAD of the layered one -- what
                              * `LLE(obs_type="partial7x7").step` returns, in one launch.  NULL: not wanted. */
} lle_env_outputs;
int lle_batch_env_outputs(lle_batch* b, const lle_env_outputs* out, void* stream);

/* LLE.step in ONE launch (python/lle/env/env.py:165-187): lle_batch_step with the outputs of lle_batch_env_outputs written
 * by the step kernel itself, from registers (the separate launch costs 4.9 us at 65 536 envs, all of it launch boundary).
 * Same results as the two calls in a row.  `out->available` needs walkable_lasers != 0 (LLE_ERR_UNSUPPORTED otherwise: the
 * mask without moves into foreign beams reads the neighbours' laser stacks -- use lle_batch_env_outputs for it).
 * The struct travels in the launch's kernel arguments (round 4): other buffers every step cost nothing extra.
 * With `out->partial` set the launch writes the partial observation (python/lle/observations.py:312-369) from the state machine's
 * own records instead of LLE_BUF_OBS: 38-41 us -> one launch for a step of `LLE(obs_type="partial7x7")` (maps with at most 8 beam
 * words, the map's own sources; LLE_ERR_UNSUPPORTED otherwise: lle_batch_observe_as behind the step). */
int lle_batch_step_outputs(lle_batch* b, const uint8_t* actions_dev, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset,
                           const lle_env_outputs* out, void* stream);

/* Sum the per-block counters (synchronises `stream`):
 * out[0] env_steps, [1] agent_steps, [2] gems, [3] exits, [4] deaths, [5] invalid, [6] auto_resets, [7] reward_sum */
int lle_batch_stats(lle_batch* b, int64_t out[8], int reset_counters, void* stream);

/* ================================================================== multi-GPU: the end-of-batch reduction
 * Environments are independent (no cross-env state anywhere in `World`, src/core/world.rs:21-44), so a batch shards over the
 * GPUs of a node by env range with NO data-path exchange: rank r owns the envs [r*n, (r+1)*n) and samples with
 * env_offset = r*n.  The one collective is the sum of the eight rollout counters at the end of a batch -- 64 bytes over
 * RCCL / xGMI, latency-bound (the reference itself has no collective: a single-process library).
 * librccl is dlopen'ed on first use (LLE_RCCL_LIB overrides the search); without it these calls return LLE_ERR_UNSUPPORTED.
 *
 * One process per GPU:   rank 0 calls lle_comm_unique_id() and ships the 128 bytes to the other ranks by any side channel
 *                        (a file, a pipe, MPI, the launcher's store); every rank then calls lle_comm_create().
 * One process, N GPUs:   lle_comm_create_all() (ncclCommInitAll) gives one communicator per device; reduce with
 *                        lle_batch_stats_allreduce_group(), which posts every rank's call inside one RCCL group. */
typedef struct lle_comm lle_comm;
#define LLE_COMM_ID_BYTES 128
enum { LLE_COMM_SUM = 0, LLE_COMM_MAX = 1 };
int lle_comm_unique_id(uint8_t id[LLE_COMM_ID_BYTES]);
lle_comm* lle_comm_create(const uint8_t id[LLE_COMM_ID_BYTES], int n_ranks, int rank, int device_id);
/* out[k] = the communicator of device_ids[k] (NULL: devices 0..n_devices-1), rank k of n_devices. */
int lle_comm_create_all(lle_comm** out, int n_devices, const int* device_ids);
void lle_comm_free(lle_comm* c);
int lle_comm_rank(const lle_comm* c, int* rank, int* n_ranks);
/* lle_batch_stats summed over every rank of the communicator; every rank gets the total.  Collective: all ranks call it.
 * Enqueued on `stream` behind the batch's launches; synchronises `stream`. */
int lle_batch_stats_allreduce(lle_batch* b, lle_comm* c, int64_t out[8], int reset_counters, void* stream);
/* The same for ONE process that owns all n ranks (batches[k] and comms[k] on the same device, streams[k] the stream of
 * handle k or streams == NULL for the default streams). */
int lle_batch_stats_allreduce_group(lle_batch* const* batches, lle_comm* const* comms, void* const* streams, int n, int64_t out[8],
                                    int reset_counters);
/* In-place all-reduce of `count` int64 values in device memory (op: LLE_COMM_SUM / LLE_COMM_MAX), asynchronous on `stream`:
 * for a host's own end-of-batch numbers (e.g. the slowest rank's elapsed nanoseconds). */
int lle_comm_allreduce_i64(lle_comm* c, int64_t* buf_dev, int count, int op, void* stream);
/* The same for ONE process that owns all n ranks: bufs_dev[k] on the device of comms[k]; the n calls are posted inside one group.
 * Asynchronous on streams[k] (NULL: the default streams). */
int lle_comm_allreduce_i64_group(lle_comm* const* comms, int64_t* const* bufs_dev, void* const* streams, int n, int count, int op);

/* The sampler used by LLE_STEP_SAMPLE_ACTIONS (host copy, for harnesses): the 16-bit field f of (seed, env, t, agent);
 * the action taken is the k-th available one in enum order with k = (f * popcount(avail)) >> 16.  See DESIGN.md. */
uint64_t lle_action_hash(uint64_t seed, uint64_t env, uint64_t t, uint64_t agent);

int lle_abi_version(void);
int lle_last_status(void);
const char* lle_last_error(void);
/* Name + dynamic-LDS bytes + envs-per-wave of the step kernel a batch launches (for profiling reports). */
int lle_batch_kernel_info(const lle_batch* b, char* name_buf, size_t cap, int32_t* lds_bytes, int32_t* envs_per_wave);
/* Profiling aid: one step (flags as lle_batch_step) that also writes, per wavefront, eight s_memrealtime stamps
 * (10 ns ticks: entry, tables in LDS, state requested, logic done, state stored, observation stores issued, drained)
 * to stamps_dev [n_waves][8] u64. */
int lle_batch_step_stamped(lle_batch* b, uint32_t flags, uint64_t seed, uint64_t t, uint64_t* stamps_dev, void* stream);
/* Profiling aid: overwrite LLE_BUF_OBS with the step kernel's store pattern and nothing else -- the same rows per wavefront,
 * store instructions, store policy and workgroup -> block mapping, every 32-bit word = `value`, no state machine.  Its
 * duration is the write ceiling of THIS box for this batch shape (bench.py `fill_ceiling`): boxes differ by up to 15 % once
 * the rows of a launch exceed the Infinity Cache.  Call lle_batch_observe() afterwards to get the observation back. */
int lle_batch_probe_row_fill(lle_batch* b, uint32_t value, void* stream);
/* ---- launch rules measured on the batch itself (no reference counterpart: the reference has no launch to tune).
 * The step launcher's rules -- environments per wavefront, row heads ahead of the state machine, `sc1` or plain stores,
 * split rows, the alternating walk of outputs larger than the Infinity Cache -- have defaults fitted on the builder's
 * boxes (NOTEBOOK.md section 4).  lle_batch_autotune times the alternatives that exist for THIS batch on ITS OWN arena
 * (its plain single step with sampled actions and auto-reset, HIP events on `stream`, about budget_ms of GPU time in total;
 * <= 0: 20 ms) and keeps the fastest of each in the handle; later launches of the batch follow them.  The trials are real
 * steps: the call ends with World.reset of every environment and the counters at zero -- call it right after
 * lle_batch_create (or lle_batch_set_sources), not in the middle of an episode.  Results never depend on these choices.
 * The LLE_* environment overrides (NOTEBOOK.md section 7) still win; they are read ONCE per process, never on the launch
 * path -- lle_tuning_refresh() reads them again (tests and tuning tools change them mid-process). */
typedef struct lle_tuning_info {
    int32_t envs_per_wave;    /* environments per wavefront of the step kernel */
    int32_t row_heads;        /* 1: a plain single step stores the rows' static head lines ahead of the state machine */
    int32_t write_through;    /* 1: `sc1` stores of the observation rows */
    int32_t split_rows;       /* 1: every row split over the wavefronts of a workgroup (big observations) */
    int32_t alternating_walk; /* 1: successive launches walk the environments alternately up and down */
    int32_t rotate_rows;      /* 1: every wavefront starts its stream at another one of its rows (whole-row streams only) */
    int32_t autotuned;        /* 1: lle_batch_autotune has run on this handle (else: the default rules) */
    int32_t head_group;       /* 1 / 2 / 4: wavefronts of a workgroup whose row heads ONE of them stores (row_heads == 1 only; default 1) */
} lle_tuning_info;
int lle_batch_autotune(lle_batch* b, double budget_ms, void* stream);
/* The rules a plain single step of this batch is launched with, and (log_buf, optional) the trial log of lle_batch_autotune. */
int lle_batch_tuning(const lle_batch* b, lle_tuning_info* out, char* log_buf, size_t cap);
void lle_tuning_refresh(void);

/* ---- debug registry: which kernel instantiations has this process LAUNCHED, and which can the dispatch REACH?
 * The library is 570-odd template instantiations of a few kernels (step_kernel<G, LM, MODE, ML1, LX>: lanes per environment, beam
 * registers, launch mode, single-layer maps, exact source count; world_kernel<AM, LM, MODE>; the observers).  A compiler fault in ONE
 * of them is only seen by a test that launches that one: tests/test_gpu_instantiations.py drives every reachable instantiation against
 * the oracle and fails when lle_debug_reachable() names a kernel that lle_debug_launched() does not.
 * Both write newline-separated names (rocprofv3's spelling without spaces, e.g. "step_kernel<4,4,6,true,3>"), sorted, NUL-terminated,
 * truncated to `cap`; they return the bytes needed.  `reachable` walks the launchers' own dispatch code with launches suppressed, for
 * every (agents 1..16, beam words 0..32, crossing beams or not, mode): it cannot drift from the dispatch.  Host side, thread-safe. */
size_t lle_debug_launched(char* buf, size_t cap);
size_t lle_debug_reachable(char* buf, size_t cap);
void lle_debug_reset_launched(void);

/* Placement aid: the step kernel's store pattern (rows_per_wave rows of row_bytes per wavefront, XCD-contiguous blocks) over ANY device
 * buffer of n_rows x row_bytes bytes, zeros.  Past the Infinity Cache the write rate of a buffer depends on where the allocation landed
 * (profiles/r04_alloc_probe.md): a host that allocates a trajectory ring or an observer output of that size times this on a few
 * candidate buffers and keeps the fastest (lle_amd.placement). */
int lle_probe_fill_rows(void* out_dev, int64_t n_rows, int64_t row_bytes, int rows_per_wave, void* stream);

/* Profiling aid: a consumer's first touch of an observation buffer -- `bytes` int8 values at rows_dev (any of the observation
 * outputs, on the current device) converted to fp16 into out_f16_dev (2 x bytes), the way the first layer of a policy reads
 * `obs` between two steps (python/lle/env/env.py:165-189).  bench.py's `consumer_loop` alternates it with the step. */
int lle_probe_read_rows(const void* rows_dev, void* out_f16_dev, int64_t bytes, void* stream);

/* Diagnostic knob: step with the one-environment-per-lane kernel at 8, 16, 32 or 64 environments per wavefront
 * instead of the default one-lane-per-agent step kernel (64 / G environments per wavefront). */
int lle_batch_set_envs_per_wave(lle_batch* b, int envs_per_wave);

#ifdef __cplusplus
}
#endif
#endif /* LLE_HIP_H */
