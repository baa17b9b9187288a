// tables.h -- static map tables shared by the host map compiler and the HIP kernels.
//
// One map compiles to a flat "blob": a MapHeader followed by 16-byte aligned sections.  The kernel copies the
// sections into LDS verbatim (same offsets), so the host compiler alone defines the layout.
//
// Beam WORDS.  A beam (`LaserBeam{Vec<bool>}`, laser.rs:15-21) is stored as ceil(len / 32) consecutive 32-bit words: word w of
// source s is index source_word[s] + w, bit k of it = cell 32 w + k of the beam.  Every table below, every kernel and LLE_BUF_BEAMS
// speak of WORDS ("beam b" = word b): MapHeader.L is their number, MapHeader.n_sources the number of sources.  A map whose
// beams all fit one word (every map of the reference's repository, every BASELINE configuration) has L == n_sources and
// word == laser_id -- nothing changes for it.  A longer beam is CHAINED: chain_mask bit b says that word b continues the beam of
// word b - 1, and a re-light / cut that reaches the end of a word fills / clears every following word of the chain
// (LaserBeam::turn_on / turn_off run to the end of the Vec, laser.rs:50-59).  Chained maps always carry at least five words (the
// compiler pads with empty ones), so that they take the kernels' LDS-record form of the beam masks (step_lanes.hpp BM), the only
// one that walks chains: the register form of maps with at most four sources stays as it was.
//   cell_lay [HW] u64 : up to 4 laser layers of the cell, OUTERMOST first (a later source wraps an earlier one,
//                       reference src/core/parsing/world_config.rs:223-247), 16 bits each:
//                         bit 0 valid | bits 1-5 beam word | bits 6-10 bit within the word | bits 11-15 colour
//                       colour 31 = "no agent has this colour" (reference colours >= n_agents are legal, Q5)
//   cell_meta[HW] u32 : bits 0-2 kind | bits 3-8 gem index (source cells: first beam word) | bits 9-12 static walk mask (bit = Action N,S,E,W:
//                       neighbour in bounds and not Wall/LaserSource, reference world.rs:351-356, tile.rs:63-73)
//                       | bits 13-15 number of layers   (accessors: meta_*.  Since the end of round 5 the gem-index fields and the laser references of
//                       cell_meta, dyn and elems are wide enough for 64 gems / beam words; the layer entries of cell_lay above still carry 5-bit
//                       words, and the state's masks are u32: NOTEBOOK.md section 10 has the rest of the work list)
//   dyn      [D]  u64 : the observation bytes that depend on dynamic state other than agent positions:
//                         bits 0-19 byte index in the (C,H,W) int8 observation | bits 20-27 base value (int8)
//                         bits 28-29 number of laser refs (0-2) | bits 30-40 ref0 (beam word:6, offset:5)
//                         bits 41-51 ref1 | bits 52-58 gem index (127 = none)
//                       value = base; any ref on -> 1; gem present and not collected -> 1
//                       (write order of reference python/lle/observations.py:216-266)
//   template [obs_stride] i8 : static observation (walls, voids, exits, -1 at sources), dyn bytes at their base
// Second section, used when every environment has its own source colours / enabled flags (lle_batch_set_sources):
//   bare     [obs_stride] i8 : walls, voids, exits only
//   elems    [E]  u32 : what depends on colours or dynamic state: bits 0-15 cell | 16-21 beam word or gem index |
//                       22-26 offset | 27-28 type (0 source: -1 on layer LASER_0 + colour; 1 laser tile exposed by
//                       World.lasers(): 1 on that layer when the beam bit is on; 2 gem: 1 on GEM when not collected)
#pragma once
#include <stdint.h>

namespace lle {

enum CellKind : uint32_t { K_FLOOR = 0, K_WALL = 1, K_VOID = 2, K_EXIT = 3, K_GEM = 4, K_SOURCE = 5 };

constexpr int MAX_AGENTS = 16;
constexpr int MAX_SOURCES = 32;
constexpr int MAX_GEMS = 32;
constexpr int MAX_BEAM_LEN = 32;      // bits of one beam WORD (a beam is a chain of words)
constexpr int MAX_CELL_LAYERS = 4;
constexpr uint32_t NO_GEM = 127;       // dyn entries: no gem on this byte (7-bit field)
constexpr uint32_t NO_INDEX = 63;      // cell_meta index field of a cell that is neither a gem nor a source
// Window tables of the partial k x k observation (python/lle/observations.py:312-369) for k = 3, 5, 7, built on the host per map and window size
// (map_compile.cpp Map::window_table) and uploaded when a batch first writes that window:
//   sets [HW][2] u64 : per observer cell p two 64-bit sets over the k x k window centred at p, bit wi * k + wj = window cell (wi, wj) --
//                      [0] a wall or a source there (the WALL layer's byte), [1] the cells whose bytes go through the cell tables (a gem, an exit,
//                      a laser tile, a source).  A bit's index is the byte's offset inside a layer of the window: the writers turn wall bits into
//                      bytes with find-first-set and a store, instead of cutting the sets out of a bitmap of the map per environment and
//                      evaluating EVERY non-empty cell against the cell tables (partial_stream.hpp);
//   cell_lay [HW] u64, cell_meta [HW] u32 : copies of the map's cell tables, so that the observer kernel takes ONE table into LDS -- 28 B per
//                      cell in all: with a separate (KiB-padded) copy of the cell tables next to the sets a workgroup of level 6's 7 x 7
//                      writer is 42.6 KB, three per CU; with this table 40.8 KB, four (profiles/r05_partial.md).
// Padded to whole 1-KiB rows (the LDS copy's unit); the step kernel's writer (MODE 9) has the cell tables already and takes the sets only.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr bool win_sets_serve(int k) { return k == 3 || k == 5 || k == 7; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t win_table_bytes(uint32_t HW) { return (HW * 28u + 1023u) & ~1023u; }  // the whole table, as uploaded (stride per map)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t win_table_used(uint32_t HW) { return (HW * 28u + 15u) & ~15u; }       // ... what of it a kernel keeps in LDS
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t win_sets_bytes(uint32_t HW) { return (HW * 16u + 1023u) & ~1023u; }   // the sets alone, in whole rows (MODE 9)
constexpr uint32_t NO_COLOUR = 31;
constexpr uint32_t TMPL_NEG_MAX = 64;  // -1 bytes of a static observation that the bit form lists (MapHeader.off_tmpl_bits)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t tmpl_bits_bytes(uint32_t n_chunks) { return (n_chunks * 2u + 15u) & ~15u; }  // the u16s, ahead of the list
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t packed_meta_bytes(uint32_t HW) { return (HW * 2u + 15u) & ~15u; }   // packed image: the u16 cell words
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t packed_idx_bytes(uint32_t n_lay) { return (n_lay * 2u + 15u) & ~15u; }  // ... the cells of the layer words
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t tmpl_section_bytes(uint32_t n_chunks) { return (tmpl_bits_bytes(n_chunks) + TMPL_NEG_MAX * 4u + 127u) & ~127u; }

struct MapHeader {
    uint32_t magic;          // 'LLE1'
    uint32_t H, W, A, G, L, C;
    uint32_t HW;
    uint32_t obs_bytes;      // C*HW
    uint32_t obs_stride;     // obs_bytes rounded up to 16
    uint32_t n_chunks;       // obs_stride / 16
    uint32_t D;              // number of dyn entries
    uint32_t max_layers;     // max laser layers on any cell
    uint32_t enabled_mask;   // bit b: source b enabled
    uint32_t blob_bytes;     // header + sections
    uint32_t off_cell_lay, off_cell_meta, off_dyn, off_template;  // byte offsets from blob start
    uint32_t lds_table_bytes;  // bytes [off_cell_lay, blob_bytes) copied to LDS
    uint32_t obs_supported;
    uint32_t direct_gems;    // bit g: gem g is a direct Tile::Gem (no laser layer on its cell)
    uint32_t blob_capacity;  // blob_bytes with the largest possible dyn table (any recolouring of the sources fits)
    uint32_t lds_split_table_bytes;  // bytes [off_cell_lay, off_template) in whole 1-KiB rows: the LDS tables of a split-row launch
    uint16_t start[MAX_AGENTS];        // start cell of each agent, i | j << 8
    uint32_t beam_full[MAX_SOURCES];   // (1 << len) - 1
    uint8_t beam_len[MAX_SOURCES];
    uint8_t beam_colour[MAX_SOURCES];  // min(colour, 31)
    uint16_t gem_cell[MAX_GEMS];       // cell of each gem, i | j << 8
    // per-environment sources (kernels instantiated with PES = true): a second LDS section right behind the first,
    // [off_bare, off_bare + ext_bytes): the static observation WITHOUT the sources' -1 marks, and the element list
    uint32_t off_bare, off_elems, n_elems, ext_bytes;
    // bit c: source s may take colour c -- no possible start of an agent other than c lies on the tiles World::lasers()
    // exposes for that source (the check of the binding's LaserSource.set_colour, src/bindings/tiles/pylaser_source.rs:121-139)
    uint16_t colour_ok[MAX_SOURCES];
    // The "head" of a row: chunks [head_lo, head_lo + head_n) (whole 128-byte lines, at most 64 chunks) hold no byte that
    // an agent, a beam or a gem can change -- identical in every environment at every step.  The default step kernel
    // stores them BEFORE the state machine runs (step_kernel.hpp); head_n = 0: no such run of lines (or unaligned rows).
    uint32_t head_lo, head_n;
    // Inside the per-env-sources section: per source, [colour_ok | beam mask after World::reset when the source has colour
    // 0, 1, ..., A-1] (A + 1 words): what the step kernel needs to re-colour an environment that it resets
    // (STEP_RECOLOUR_RESETS).  Exact for maps without a cell of more than two laser layers (`recolour_exact`): there the
    // binding's colour check keeps every beam of another colour off every start, so nothing but the beams depends on the colours.
    uint32_t off_recolour, recolour_exact;
    // The head of a row when every environment has its own source colours (step_kernel MODE 8): lines that stay the same
    // under ANY colouring with colours below n_agents -- behind the agent AND laser layers, no gem byte (WALL / VOID / EXIT
    // planes); chunks [pes_head_lo, pes_head_lo + pes_head_n) of the BARE template.  0 when a source of the map itself has
    // a colour >= n_agents (its layer then aliases those planes, quirk Q5).
    uint32_t pes_head_lo, pes_head_n;
    // beam words (top of this file): L counts words; sources and their chains
    uint32_t n_sources;                  // laser sources of the map (LaserSource tiles); == L unless a beam is longer than 32 cells
    uint32_t chain_mask;                 // bit b: word b continues the beam of word b - 1
    uint32_t word_mask;                  // bit b: word b belongs to a source (0: padding of a chained map)
    uint8_t word_source[MAX_SOURCES];    // source (laser_id) of word b
    uint8_t source_word[MAX_SOURCES];    // first word of source s
    // The DYNAMIC chunks of a row (u16 chunk indices, ascending, behind the template in the table section): the 16-byte chunks of the
    // 128-byte lines that an agent, a beam or a gem can change.  Every other line of LLE_BUF_OBS holds the same bytes after every
    // step (the head lines above are a run of those), so a single step that rewrites the rows IN PLACE may leave them alone
    // (STEP_INCREMENTAL_OBS): level 6 writes 10 of its 15 lines.  n_dyn_chunks == n_chunks: nothing to skip (or unaligned rows).
    uint32_t off_dyn_chunks, n_dyn_chunks;
    // the same under per-environment source colours (inside the per-env-sources section, like off_recolour): a laser byte may sit on
    // any of the A laser planes, so every line below 2A * HW is dynamic, and behind them the lines with a gem byte (pes_head_* is a run of the rest)
    uint32_t off_pes_dyn_chunks, n_pes_dyn_chunks;
    // a SECOND run of such lines (level 6: lines 10-11 are the first run, line 14 -- the end of the EXIT plane -- the second): together the
    // `head_lines` the map asks for; chunks, behind the first run (pes_head2_lo > pes_head_lo + pes_head_n), 0: none
    uint32_t pes_head2_lo, pes_head2_n;
    // The static observation once more, as BITS, between the header and the cell tables (outside every LDS copy; at the same offset in every map of
    // these dimensions): [u16 per 16-byte chunk of the row: bit i = byte i of the chunk is 1 | u32 byte index of every -1 (the sources' marks), tmpl_neg_n
    // of them], bits_bytes in all (whole 128-byte lines).  The split-row launch's wavefronts build their slices of the row from it instead of copying
    // them from `template`: an eighth of the bytes, at an address that does not wait for the header -- what a workgroup reads per map, and in how many
    // dependent round trips, is what a batch of many maps with few environments each pays on top of one map (profiles/r05_multi_map.md).
    // off_tmpl_bits == 0: the template holds another value than -1 / 0 / 1 or more than TMPL_NEG_MAX marks; the wavefronts copy it.
    uint32_t off_tmpl_bits, tmpl_neg_n, bits_bytes;
    // The table section [off_cell_lay, off_template) once more, PACKED (behind the per-env-sources section; never copied verbatim):
    //   [cell_meta as u16[HW] (its 15 bits) | cells u16[packed_n_lay]: the cells with a laser layer | cell_lay u64[packed_n_lay] of those cells |
    //    the bytes [off_dyn, off_template) as they are (dyn, dynamic chunks)], each part padded to 16 B; packed_bytes in all.
    // In a batch of thousands of maps with a few environments each, every workgroup reads its map's tables from memory, once, ahead of its first store --
    // and pays for the bytes (profiles/r05_multi_map.md): the step kernels of such a launch (LAUNCH_PACKED_TABLES) read this image and expand it into the
    // same LDS layout the verbatim copy gives.  Config 5's shape: 13.3 KB -> 4.1 KB.  off_packed == 0: none (a cell_meta word beyond 16 bits).
    uint32_t off_packed, packed_bytes, packed_n_lay;
    uint32_t packed_cap;   // bytes reserved for the image behind off_bare + ext_bytes: its size under any colouring of the sources (capi.cpp table_stride)
    uint32_t head_pad[6];  // (pads the header to 640 B: the table sections behind it start on a 128-byte line)
};
static_assert(sizeof(MapHeader) % 128 == 0, "the sections start on a 128-byte line (the LDS copy loads 1 KiB per wave instruction)");

constexpr uint32_t MAP_MAGIC = 0x31454C4Cu;

// Observation kinds (mirror include/lle_hip.h LLE_OBS_*; reference python/lle/observations.py:38-60)
enum ObsKind : int {
    OBS_LAYERED = 0, OBS_LAYERED_PADDED = 1, OBS_PERSPECTIVE = 2, OBS_PARTIAL = 3, OBS_STATE = 4, OBS_NORMALIZED_STATE = 5
};

// A "view": the tables of a layered-style observation with another channel layout (agent padding, agent-zero
// perspective), compiled from the same map.  Blob = ViewHeader, dyn table, static template (same entry formats as
// the map blob), padded to whole 1-KiB rows; the view kernel copies all of it to LDS.
struct ViewHeader {
    uint32_t magic;          // 'LLV1'
    uint32_t A, L, H, W, HW, C;
    uint32_t obs_bytes, obs_stride, n_chunks, D;
    uint32_t off_dyn, off_template, blob_bytes;
    uint32_t supported;      // 0: some laser colour has no layer (the reference raises IndexError)
    uint32_t pad;
    uint8_t agent_layer[MAX_AGENTS];  // layer that shows agent a
    // for batches with per-environment sources (the layer of a laser byte depends on the env's colours):
    uint32_t off_bare;       // static observation of this view without the sources' -1 marks
    uint32_t gem_layer, n_laser, pad2;
    uint8_t laser_layer[48]; // layer of laser colour c, c < n_laser
};
static_assert(sizeof(ViewHeader) % 16 == 0, "sections must stay 16-byte aligned");
constexpr uint32_t VIEW_MAGIC = 0x31564C4Cu;

// Agents per env record in the per-agent buffers (pos, avail, actions, events): the agent bound of the lane-per-env
// kernel instantiation that serves the map, so that a record is a whole number of dwords.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int agent_stride_of(int A, int L) { return (A <= 4 && L <= 4) ? 4 : ((A <= 8 && L <= 8) ? 8 : 16); }

// ---- cell_lay / cell_meta field accessors: EVERY reader of the two cell tables goes through these (and the compiler packs through lay_pack /
// meta_pack), so the widths of the beam-word, bit, colour and gem-index fields are written down once -- the limits of MAX_SOURCES / MAX_GEMS live here
constexpr uint32_t LAY_VALID = 1u;
#if defined(__HIPCC__)
#define LLE_TAB_FN __host__ __device__ inline constexpr
#else
#define LLE_TAB_FN inline constexpr
#endif
LLE_TAB_FN uint32_t lay_pack(uint32_t beam, uint32_t off, uint32_t colour) { return LAY_VALID | (beam << 1) | (off << 6) | (colour << 11); }
LLE_TAB_FN uint32_t lay_entry(uint64_t lay, int k) { return (uint32_t)(lay >> (16 * k)) & 0xFFFFu; }  // layer k of a cell's u64 (0 = outermost)
LLE_TAB_FN uint32_t lay_word(uint32_t e) { return (e >> 1) & 31u; }    // beam word of a 16-bit layer entry
LLE_TAB_FN uint32_t lay_bit(uint32_t e) { return (e >> 6) & 31u; }     // bit of the cell within that word
LLE_TAB_FN uint32_t lay_colour(uint32_t e) { return e >> 11; }         // colour (the entry's top five bits; NO_COLOUR: nobody's)
LLE_TAB_FN uint32_t meta_pack(uint32_t kind, uint32_t index, uint32_t walk, uint32_t layers) { return kind | (index << 3) | (walk << 9) | (layers << 13); }  // 16 bits: the packed image keeps u16
LLE_TAB_FN uint32_t meta_kind(uint32_t m) { return m & 7u; }           // CellKind
LLE_TAB_FN uint32_t meta_index(uint32_t m) { return (m >> 3) & 63u; }  // gem index of a gem cell / first beam word of a source cell (6 bits: room for 64 of either)
LLE_TAB_FN uint32_t meta_walk(uint32_t m) { return (m >> 9) & 15u; }   // static walk mask (bit = Action N, S, E, W)
LLE_TAB_FN uint32_t meta_layers(uint32_t m) { return (m >> 13) & 7u; } // number of laser layers of the cell
// ---- dyn entries (u64; layout at the top of this file) and their laser references (11 bits: beam word | bit << 6)
LLE_TAB_FN uint64_t dyn_pack(uint32_t byte_index, uint8_t base, uint32_t n_refs, uint32_t ref0, uint32_t ref1, uint32_t gem) {
    return (uint64_t)byte_index | ((uint64_t)base << 20) | ((uint64_t)n_refs << 28) | ((uint64_t)ref0 << 30) | ((uint64_t)ref1 << 41) | ((uint64_t)gem << 52);
}
LLE_TAB_FN uint32_t dyn_index(uint64_t e) { return (uint32_t)e & 0xFFFFFu; }        // byte index in the (C, H, W) observation
LLE_TAB_FN int32_t dyn_base(uint64_t e) { return (int32_t)(int8_t)(uint8_t)(e >> 20); }  // value when nothing is lit
LLE_TAB_FN uint32_t dyn_refs(uint64_t e) { return (uint32_t)(e >> 28) & 3u; }        // number of laser references (0-2)
LLE_TAB_FN uint32_t dyn_ref0(uint64_t e) { return (uint32_t)(e >> 30) & 0x7FFu; }
LLE_TAB_FN uint32_t dyn_ref1(uint64_t e) { return (uint32_t)(e >> 41) & 0x7FFu; }
LLE_TAB_FN uint32_t dyn_gem(uint64_t e) { return (uint32_t)(e >> 52) & 127u; }       // gem index, NO_GEM: none
LLE_TAB_FN uint32_t ref_pack(uint32_t word, uint32_t bit) { return word | (bit << 6); }   // 11 bits: beam word 6 | bit 5
LLE_TAB_FN uint32_t ref_word(uint32_t r) { return r & 63u; }
LLE_TAB_FN uint32_t ref_bit(uint32_t r) { return r >> 6; }
LLE_TAB_FN uint32_t gem_bit(uint32_t gem) { return gem & 31u; }                      // bit of a gem in the record's ~gems word
// ---- elems (per-environment sources; u32: cell | beam word or gem index << 16 | bit << 22 | type << 27)
LLE_TAB_FN uint32_t elem_pack(uint32_t cell, uint32_t index, uint32_t bit, uint32_t type) { return cell | (index << 16) | (bit << 22) | (type << 27); }
LLE_TAB_FN uint32_t elem_cell(uint32_t e) { return e & 0xFFFFu; }
LLE_TAB_FN uint32_t elem_index(uint32_t e) { return (e >> 16) & 63u; }
LLE_TAB_FN uint32_t elem_bit(uint32_t e) { return (e >> 22) & 31u; }
LLE_TAB_FN uint32_t elem_type(uint32_t e) { return (e >> 27) & 3u; }

// ---- per-env error codes (mirror include/lle_hip.h)
constexpr uint8_t ENV_OK = 0;
constexpr uint8_t ENV_INVALID_WORLD_STATE = 0x40;
constexpr uint8_t ENV_OUT_OF_WORLD_POSITION = 0x41;
constexpr uint8_t ENV_INVALID_AGENT_POSITION = 0x42;
constexpr uint8_t ENV_INVALID_COLOUR = 0x43;
constexpr uint8_t ENV_COLOUR_CROSSES_START = 0x44;

// ---- step flags (mirror include/lle_hip.h)
// STEP_RECOLOUR_RESETS: an env that STEP_AUTO_RESET resets draws a fresh colour for each of its sources (LLE.reset with
// randomize_lasers, python/lle/env/env.py:189-203); batches with per-env sources only.
constexpr uint64_t RECOLOUR_SALT = 0xC01055EEDULL;  // seed ^ salt keys the colour draws (a stream of their own)
constexpr uint32_t STEP_SAMPLE_ACTIONS = 1, STEP_AUTO_RESET = 2, STEP_NO_OBS = 4, STEP_RECOLOUR_RESETS = 8;
// STEP_INCREMENTAL_OBS: a single step in place writes only the lines of a row that dynamic state can change (MapHeader.n_dyn_chunks);
// the others already hold their bytes from the last full write (reset, observe, any earlier step).  The buffer's content is the same.
constexpr uint32_t STEP_INCREMENTAL_OBS = 16;
constexpr uint32_t LAUNCH_PER_ENV_SOURCES = 0x10000;  // internal: the batch keeps colours / enabled flags per env
constexpr uint32_t LAUNCH_FILL_DEFAULTS = 0x20000;    // internal (MODE_ENV_SOURCES): take them from the map header
constexpr uint32_t LAUNCH_GENERAL = 0x80000;          // internal: the general step_kernel instantiation (per-env sources / several maps)
constexpr uint32_t LAUNCH_SINGLE_LAYER = 0x100000;    // internal: no cell has more than one laser layer (step_kernel ML1 instantiation)
constexpr uint32_t LAUNCH_ROLLOUT = 0x200000;         // internal: step_kernel MODE 1 (fused rollout / stamps; one map, map-wide sources)
constexpr uint32_t LAUNCH_ARRAYS_INVALID = 0x40000;   // internal (MODE_ENV_SOURCES): first call, nothing stored per env yet
constexpr uint32_t LAUNCH_RESET_FIRST = 0x800000;      // internal (MODE_ENV_SOURCES): World::reset of the env, then the source update
constexpr uint32_t LAUNCH_SPLIT_ROWS = 0x1000000;      // internal: step_kernel splits every observation row over the wavefronts of a workgroup
constexpr uint32_t LAUNCH_REVERSE = 0x2000000;         // internal: the workgroups serve the blocks of environments from the last to the first (obs_stream.hpp)
constexpr uint32_t LAUNCH_HEAD_GROUP2 = 0x8000000;     // internal (HEAD kernels): every 2nd wavefront of a workgroup stores the row heads of itself and its neighbour,
constexpr uint32_t LAUNCH_HEAD_GROUP4 = 0x10000000;    //   (4: the first wavefront those of the whole workgroup) -- the others go straight to their state machines
constexpr uint32_t LAUNCH_POST_FIRST = 0x20000000;     // internal: every wavefront writes its small outputs BEFORE its observation stream (step_kernel.hpp post_first), ...
constexpr uint32_t LAUNCH_POST_LAST = 0x40000000;      //   ... or every one after it (default: the last quarter of the grid before, the rest after)
constexpr uint32_t LAUNCH_ROTATE_ROWS = 0x4000000;     // internal: every wavefront starts its rows at another one of them (obs_stream.hpp row_rotation)
// The element type of the batch's own layered rows (LLE_BUF_OBS and the observation rings; lle_batch_options.obs_dtype): the kernels keep the row
// as int8 in LDS and WIDEN AT THE STORE (obs_stream.hpp stream_wide) -- what the reference returns is float32 (python/lle/observations.py:223),
// what a learner's first layer reads is usually fp16 / bf16; the values are -1, 0, 1 in every type.  Two bits of the launch flags.
enum ObsElem : uint32_t { OBS_I8 = 0, OBS_F16 = 1, OBS_BF16 = 2, OBS_F32 = 3 };
constexpr uint32_t LAUNCH_PACKED_TABLES = 0x8000;     // internal (step_kernel, several maps): the workgroups expand MapHeader.off_packed instead of copying the table section
constexpr uint32_t LAUNCH_OBS_ELEM_SHIFT = 13, LAUNCH_OBS_ELEM_MASK = 3u << LAUNCH_OBS_ELEM_SHIFT;  // internal
#if defined(__HIPCC__)
__host__ __device__
#endif
inline constexpr uint32_t obs_elem_shift(uint32_t et) { return et == OBS_I8 ? 0u : (et == OBS_F32 ? 2u : 1u); }  // log2(bytes per element)
constexpr uint32_t STEP_PUBLIC_FLAGS = 0x1Fu;        // the LLE_STEP_* bits a caller may pass; everything above is the library's own
constexpr uint32_t LAUNCH_DRY_RUN = 0x80000000u;      // internal, host side only: walk the dispatch, note the instantiation, launch nothing (kernels.h debug registry)
constexpr uint32_t LAUNCH_WRITE_THROUGH = 0x400000;    // internal: observation rows are stored `sc1` (stream_store, obs_stream.hpp)
// A launch writes its rows through L2 while all of them fit the Infinity Cache (256 MB, MI355X_MICROARCH.md); beyond
// that plain write-back stores are faster (measured break-even between 245 MB and 490 MB per launch).
constexpr uint64_t WRITE_THROUGH_MAX_BYTES = 256ull << 20;
constexpr uint32_t ELEM_SOURCE = 0, ELEM_TILE = 1, ELEM_GEM = 2;

// LLE_BUF_BITS: alive 0-15 | arrived 16-31 | occupant 32-47 | ghost 48-63.  "Ghost" = flagged dead by set_state without an
// AgentDied event (a dead agent forced onto a tile that does not kill it): World state is `alive = false`, but LLE.compute_done
// (python/lle/env/env.py:253-254) counts death EVENTS since the last reset / set_state, so such an agent does not end the
// episode.  done = (alive | ghost) != all || arrived == all.  Cleared by reset; deaths by event never set it.
constexpr uint32_t GHOST_SHIFT = 48;

// ---- event codes
constexpr uint32_t EV_EXIT = 0, EV_GEM = 1, EV_DIED = 2;

// lle_env_outputs (include/lle_hip.h) as the kernels see it: everything LLE.step returns besides the observation.
struct EnvOutputs {
    float* state;
    float* reward;
    uint8_t* done;
    uint8_t* available;
    uint8_t* alive;
    uint8_t* arrived;
    int8_t* partial;       // step_kernel MODE 9 only: the partial k x k observation of every env, written by the step launch
    uint8_t normalize_state, reward_kind, walkable_lasers, per_env_sources;
    uint32_t partial_k;    // window size of `partial`
};

// Per-env-sources kernels: every wavefront's private LDS area ends with this many bytes for the seven counters of a launch that the fused
// rollout (MODE 3) would otherwise carry in registers across its steps: 8 words per environment, 64 / 2 environments at most (one-agent
// maps keep the registers).
constexpr uint32_t PES_WAVE_EXTRA_BYTES = 32u * 8u * 4u;

// Per-launch arguments.
struct LaunchArgs {
    uint32_t flags;            // STEP_*
    uint32_t envs_per_wave;    // 1..64, power of two
    uint64_t seed, t;
    int64_t env_offset;        // global id of env 0 (multi-GPU shards sample as one big batch)
    int64_t env_base;          // first environment of this launch (block b handles env_base + b * envs_per_wave ...)
    int64_t env_limit;         // one past the last environment of this launch
    const uint8_t* env_mask;   // reset: optional u8[n]
    const uint8_t* actions_in; // step: optional u8[n][A]
    uint32_t old_enabled;      // update_sources: enabled mask before the update
    uint32_t partial_k;        // step_kernel MODE 9: window size of the partial observation this launch writes (0: none)
    // fused rollout (step_kernel only): n_steps consecutive steps per launch; per-step observation / actions / reward
    // counts go to slot (ring_pos + step) % ring_slots of caller-provided trajectory rings (ring_slots = 0: in place)
    uint32_t n_steps, ring_slots;
    uint64_t ring_pos;
    int64_t ring_env_count;    // envs per ring slot (= the batch's n_envs)
    int8_t* ring_obs;          // [ring_slots][n][obs_stride]
    uint8_t* ring_actions;     // [ring_slots][n][agent stride]
    uint32_t* ring_reward;     // [ring_slots][n]
    uint64_t* stamps;          // profiling aid: [n_blocks][8] s_memrealtime stamps (10 ns ticks) of lane 0, or NULL
    const uint8_t* colours_in;   // set_sources: optional u8[n][L] new colours
    const uint32_t* enabled_in;  // set_sources: optional u32[n] new enabled masks
    // batches of several maps (lle_batch_create_multi): map m owns the envs [m * envs_per_map, (m + 1) * envs_per_map);
    // its tables are BatchPtrs.tables + m * table_stride and its reset record BatchPtrs.init[m].  envs_per_map = 0:
    // one map.  map_override = m + 1 forces map m (the hidden env that computes a map's reset record).
    int64_t envs_per_map;
    uint32_t table_stride, map_override;
    uint32_t n_sources;        // host side only: MapHeader.L, for the launcher's choice of instantiation
    uint32_t partial_E;        // step_kernel MODE 9: environments per batch of the partial writer (partial_stream.hpp)
    const uint8_t* win_sets;   // step_kernel MODE 9: the window tables of partial_k (win_table_bytes(HW) apart per map; the sets lead each), or NULL
    // step_kernel only: write LLE.step's other outputs in the same launch (lle_batch_step_outputs).  `env_out` non-NULL says so; the
    // struct itself travels in the kernel arguments (`out`, since round 4: a caller that hands other tensors every step pays no
    // upload for it) and is read from the kernarg segment with scalar loads where it is used (step_kernel.hpp kernarg_env_out):
    // its pointers never occupy registers during the state machine.
    const EnvOutputs* env_out;
    EnvOutputs out;
};

// State of a freshly reset environment (identical for every env of a map: v1 maps have one start per agent).
// Filled on the device by resetting a hidden environment; read by the auto-reset path of the step kernel.
struct InitRecord {
    uint16_t pos[MAX_AGENTS];
    uint64_t bits;
    uint32_t gems;
    uint32_t beams[MAX_SOURCES];
    uint8_t avail[MAX_AGENTS];
};

// Device pointers of one batch (kernel argument, passed by value).
struct BatchPtrs {
    const InitRecord* init;  // reset state (see InitRecord)
    const uint8_t* tables;   // device blob (MapHeader + sections)
    uint16_t* pos;           // [n][A]   i | j << 8
    uint64_t* bits;          // [n]
    uint32_t* gems;          // [n]
    uint32_t* beams;         // [n][L]
    uint8_t* avail;          // [n][A]
    uint8_t* actions;        // [n][A]
    uint8_t* err;            // [n]
    uint8_t* evcount;        // [n]
    uint8_t* events;         // [n][2A]
    uint8_t* done;           // [n]
    uint32_t* reward;        // [n] gems | exits << 8 | deaths << 16 | all_arrived << 24 of the last step
    int8_t* obs;             // [n][obs_stride]
    int64_t* stats;          // [n_blocks][8]
    const uint16_t* req_pos; // [n][A]
    const uint32_t* req_gems;
    const uint16_t* req_alive;
    int64_t n_envs;
    // per-environment sources (lle_batch_set_sources): colour of every source (4 per dword, stride src_stride bytes),
    // enabled mask, and the env's own reset state (what World.reset gives with those colours / flags), which the
    // auto-reset path copies instead of the shared InitRecord
    uint8_t* src_colour;     // [n][src_stride]
    uint32_t* src_enabled;   // [n]
    uint16_t* init_pos;      // [n][A]
    uint64_t* init_bits;     // [n]
    uint32_t* init_gems;     // [n]
    uint32_t* init_beams;    // [n][L]
    uint8_t* init_avail;     // [n][A]
};

// bytes of one env's colour record: L rounded up to whole dwords
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int src_stride_of(int L) { return L <= 4 ? 4 : (L <= 8 ? 8 : (L <= 16 ? 16 : 32)); }

// which map owns an env, for the observer kernels (same meaning as the LaunchArgs fields)
struct MapSel { int64_t envs_per_map; uint32_t table_stride, pad; };

// map index of the wave whose first env is env0 (see LaunchArgs.envs_per_map)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t map_index_of(const LaunchArgs& K, int64_t env0) {
    if (K.map_override) return K.map_override - 1u;
    if (!K.envs_per_map) return 0u;
    // (a 64-bit division is a long software sequence on the device, and it sits in front of the first table load)
    if ((((uint64_t)env0 | (uint64_t)K.envs_per_map) >> 32) == 0) return (uint32_t)env0 / (uint32_t)K.envs_per_map;
    return (uint32_t)((uint64_t)env0 / (uint64_t)K.envs_per_map);
}


}  // namespace lle
