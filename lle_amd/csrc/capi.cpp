// capi.cpp -- the extern "C" boundary declared in include/lle_hip.h.
//
// Host side of the batched World: owns the device arena (or adopts a caller-provided one), uploads the
// compiled map tables and launches the kernels of kernels.hip.  There is no CPU execution path: every
// dynamic entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/lle_hip.h"
#include "capi_internal.hpp"
#include "kernels.h"
#include "map_compile.hpp"
#include "step_logic.hpp"

using namespace lle;

struct lle_map { Map m; };

namespace {
thread_local std::string g_error;
thread_local int g_status = LLE_OK;

int fail(int status, const std::string& msg) {
    g_status = status;
    g_error = msg;
    return status;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(LLE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Every entry point of a batch runs with the batch's device current and puts the caller's device back afterwards: a host
// that owns one handle per GPU of the node (BASELINE north_star; lle_hip.h "distinct handles are independent") calls them
// with whatever device happens to be current, and a Python host's torch must not find its current device changed.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err;
    explicit DeviceScope(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_DEVICE_OF(b)                                                                                              \
    DeviceScope device_scope_((b)->device);                                                                          \
    if (device_scope_.err != hipSuccess) return fail(LLE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.err))

constexpr int64_t ALIGN = 256;
int64_t align_up(int64_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

struct Layout {
    int64_t off[LLE_BUF_COUNT];
    int64_t bytes[LLE_BUF_COUNT];
    int64_t off_tables;
    int64_t off_init;
    int64_t table_stride;     // bytes between the table blobs of consecutive maps
    int64_t off_env_init[5];  // per-env reset state (pos, bits, gems, beams, avail), used with per-env sources
    int64_t off_env_out;      // device copy of the EnvOutputs of lle_batch_step_outputs (64 B)
    int64_t off_stats_sum;    // the eight counters summed over the per-wavefront slots (lle_batch_stats: 64 B come back, not the slots)
    int64_t total;
    int64_t n_stat_blocks;
};

// `h`: the common header (dimensions; table sizes = the largest of the batch's maps); n_maps blobs and reset records
Layout make_layout(const MapHeader& h, int64_t n, int64_t n_maps = 1, uint32_t obs_et = OBS_I8) {
    Layout l{};
    const int64_t L = h.L;
    const int64_t A = agent_stride((int)h.A, (int)h.L);  // per-agent buffers are strided by the kernel's agent bound
    const int64_t n_pad = (n + 1 + 63) / 64 * 64;  // + one hidden env (slot n) used to compute the reset state on the device
    // one slot of LLE_BUF_STATS per wavefront.  A wavefront serves ONE map, so in a batch of several maps it takes at most the largest power of two that
    // divides envs_per_map (1 when every env has its own map: a slot per environment)
    int64_t min_epw = MIN_ENVS_PER_WAVE;
    if (n_maps > 1)
        while (min_epw > 1 && (n / n_maps) % min_epw != 0) min_epw >>= 1;
    l.n_stat_blocks = std::max<int64_t>(MIN_STAT_SLOTS, (n + min_epw - 1) / min_epw);
    int64_t sz[LLE_BUF_COUNT];
    sz[LLE_BUF_POS] = n_pad * A * 2;
    sz[LLE_BUF_BITS] = n_pad * 8;
    sz[LLE_BUF_GEMS] = n_pad * 4;
    sz[LLE_BUF_BEAMS] = n_pad * (L > 0 ? L : 1) * 4;
    sz[LLE_BUF_AVAIL] = n_pad * A;
    sz[LLE_BUF_ACTIONS] = n_pad * A;
    sz[LLE_BUF_ERR] = n_pad;
    sz[LLE_BUF_EVCOUNT] = n_pad;
    sz[LLE_BUF_EVENTS] = n_pad * 2 * A;
    sz[LLE_BUF_DONE] = n_pad;
    sz[LLE_BUF_OBS] = (n * (int64_t)h.obs_stride) << obs_elem_shift(obs_et);  // (rows of the batch's element type, lle_batch_options.obs_dtype)
    sz[LLE_BUF_STATS] = l.n_stat_blocks * 8 * 8;
    sz[LLE_BUF_REQ_POS] = n_pad * A * 2;
    sz[LLE_BUF_REQ_GEMS] = n_pad * 4;
    sz[LLE_BUF_REQ_ALIVE] = n_pad * 2;
    sz[LLE_BUF_REWARD] = n_pad * 4;
    sz[LLE_BUF_SRC_COLOUR] = n_pad * src_stride_of((int)h.L);
    sz[LLE_BUF_SRC_ENABLED] = n_pad * 4;
    int64_t off = 0;
    l.off_tables = off;
    l.table_stride = align_up((int64_t)h.blob_capacity + h.ext_bytes + h.packed_cap);
    off = align_up(off + n_maps * l.table_stride);
    l.off_init = off;
    off = align_up(off + n_maps * (int64_t)sizeof(InitRecord));
    for (int k = 0; k < LLE_BUF_COUNT; k++) {
        l.off[k] = off;
        l.bytes[k] = sz[k];
        off = align_up(off + sz[k]);
    }
    const int64_t init_sz[5] = {sz[LLE_BUF_POS], sz[LLE_BUF_BITS], sz[LLE_BUF_GEMS], sz[LLE_BUF_BEAMS], sz[LLE_BUF_AVAIL]};
    for (int k = 0; k < 5; k++) {
        l.off_env_init[k] = off;
        off = align_up(off + init_sz[k]);
    }
    l.off_env_out = off;
    off = align_up(off + (int64_t)sizeof(EnvOutputs));
    l.off_stats_sum = off;
    off = align_up(off + 64);
    l.total = off;
    return l;
}
}  // namespace

struct lle_batch {
    bool lane_per_env_step;  // tuning/diagnostic: step with world_kernel (one env per lane) instead of step_kernel
    MapHeader hdr;
    int64_t n_envs;
    int device;
    uint8_t* arena;
    bool owns_arena;
    Layout layout;
    BatchPtrs ptrs;
    uint32_t envs_per_wave;
    bool per_env_sources;  // lle_batch_set_sources was called: colours / enabled flags / reset state live per env
    // batches of several maps (lle_batch_create_multi): map m owns envs [m * envs_per_map, (m + 1) * envs_per_map);
    // `hdr` then holds the common dimensions and the LARGEST table sizes.  One map: maps.size() == 1, envs_per_map = 0.
    std::vector<Map> maps;
    int64_t envs_per_map;
    uint32_t worst_table_bytes;  // largest table section any recolouring of any map can need (LDS check)
    // observation views (layered-padded / perspective): compiled from the maps, and the device blobs compiled so far
    // (one per map, `stride` bytes apart), keyed by (kind, param); dropped when the sources change
    struct View { ViewHeader hdr; uint8_t* dev; uint32_t stride; };
    std::map<std::pair<int, int>, View> views;
    // lle_batch_step_outputs: what the device copy of the EnvOutputs holds (re-uploaded only when the caller's struct changes)
    // outputs larger than the Infinity Cache: the launches that rewrite one walk the environments alternately up and down
    // (obs_stream.hpp xcd_block_dir); per output buffer, the direction of its next launch (the batch's own rows: a member, no lookup)
    std::map<const void*, bool> walk_dir;
    bool obs_walk_dir = false;
    // what lle_batch_autotune chose for this batch's step launches (kernels.h StepTune) and the log of its trials
    StepTune tune{};
    std::string tune_log;
    // the window sets of the partial k x k observation (tables.h), k = 3, 5, 7: one table per map, uploaded when the batch first writes that
    // window; dropped with the views when a map changes (exits)
    uint8_t* win_sets[3] = {nullptr, nullptr, nullptr};
    // lle_batch_observe_as(LLE_OBS_PARTIAL, k): the variant of the lane kernel measured fastest on this batch (window sets or bitmap, environments
    // per batch), chosen at the first call for that k; 0 = not decided
    struct PartialChoice { uint8_t decided = 0, use_sets = 0, E = 0; } partial_choice[16];
    // element type of LLE_BUF_OBS and of the observation rings (tables.h ObsElem; lle_batch_options.obs_dtype): the kernels widen at the store
    uint32_t obs_et = OBS_I8;
    uint64_t row_pitch() const { return (uint64_t)hdr.obs_stride << obs_elem_shift(obs_et); }  // bytes between the rows of two environments
    uint64_t rows_bytes() const { return (uint64_t)n_envs * row_pitch(); }                     // ... of one launch's rows
};

// environments per wavefront of the batch's step launches (kernels.hip step_envs_per_wave; small blocks of a multi-map batch with split rows
// take fewer so that a map's block fills a workgroup)
static uint32_t batch_step_epw(const lle_batch* b, const StepTune& tune) {
    const bool split_block = b->envs_per_map != 0 && step_splits_rows(b->hdr, b->per_env_sources, tune);
    uint32_t e = step_envs_per_wave(b->n_envs, (int)b->hdr.A, tune, split_block ? b->envs_per_map : 0);
    // a wavefront serves one map (the launcher narrows it the same way; done HERE so that the check against the counter slots sees the final number)
    if (b->envs_per_map) {
        const uint32_t cap = 64u / (uint32_t)step_group((int)b->hdr.A);
        e = e < cap ? e : cap;
        while (e > 1 && b->envs_per_map % (int64_t)e != 0) e >>= 1;
    }
    return e;
}

// Whether alternating the walk can pay: the rows of one launch must exceed what the 256 MB Infinity Cache keeps of them
// (LLE_PINGPONG=0 / 1 forces it either way; read per launch, the parity tests run both in one process).
static bool pingpong_pays(const lle_batch* b, uint64_t row_bytes_per_launch) {
    if (tuning().pingpong >= 0) return tuning().pingpong == 1;  // LLE_PINGPONG (read once: kernels.h Tuning)
    if (b->tune.walk >= 0) return b->tune.walk == 1 && row_bytes_per_launch > (256ull << 20);  // (the batch's own A/B: lle_batch_autotune)
    return row_bytes_per_launch > (256ull << 20);
}
// the direction of the launch that is about to rewrite `out` (bytes of it), and the flip for the next one
static bool next_walk_reversed(lle_batch* b, const void* out, uint64_t bytes) {
    if (!pingpong_pays(b, bytes)) return false;
    if (out == b->ptrs.obs) {  // the step path
        const bool r = b->obs_walk_dir;
        b->obs_walk_dir = !r;
        return r;
    }
    if (b->walk_dir.size() > 64) b->walk_dir.clear();  // (callers that hand a fresh buffer every time)
    bool& d = b->walk_dir[out];
    const bool r = d;
    d = !d;
    return r;
}

namespace lle {
int capi_fail(int status, const std::string& msg) { return fail(status, msg); }
int capi_ok() { g_status = LLE_OK; return LLE_OK; }
int capi_batch_device(const lle_batch* b) { return b->device; }
int capi_batch_stats_to_device(lle_batch* b, int64_t* out8_dev, int reset_counters, void* stream) {
    ON_DEVICE_OF(b);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_stats_sum(b->ptrs.stats, b->layout.n_stat_blocks, out8_dev, st));
    if (reset_counters) HIP_TRY(hipMemsetAsync(b->ptrs.stats, 0, (size_t)b->layout.n_stat_blocks * 64, st));
    return LLE_OK;
}
int capi_batch_reset_counters(lle_batch* b, void* stream) {
    ON_DEVICE_OF(b);
    HIP_TRY(hipMemsetAsync(b->ptrs.stats, 0, (size_t)b->layout.n_stat_blocks * 64, (hipStream_t)stream));
    return LLE_OK;
}
}  // namespace lle

extern "C" {

int lle_abi_version(void) { return LLE_ABI_VERSION; }
int lle_last_status(void) { return g_status; }
const char* lle_last_error(void) { return g_error.c_str(); }

uint64_t lle_action_hash(uint64_t seed, uint64_t env, uint64_t t, uint64_t agent) {
    return action_field(action_hash_pair(action_step_key(seed, t), env, (uint32_t)(agent >> 1)), (uint32_t)agent);
}

// ------------------------------------------------------------------------------------------------ maps
lle_map* lle_map_parse(const char* text, size_t len, int* parse_error) {
    if (!text) { if (parse_error) *parse_error = LLE_PARSE_EMPTY_WORLD; fail(LLE_ERR_NULL, "text is NULL"); return nullptr; }
    lle_map* m = new (std::nothrow) lle_map();
    if (!m) return nullptr;
    int rc = parse_map(text, len, m->m);
    if (parse_error) *parse_error = rc;
    if (rc != LLE_PARSE_OK) { delete m; fail(LLE_ERR_ARG, "map parse error " + std::to_string(rc)); return nullptr; }
    g_status = LLE_OK;
    return m;
}

lle_map* lle_map_level(int level, int* parse_error) {
    if (level < 1 || level > 6) { if (parse_error) *parse_error = LLE_PARSE_INVALID_LEVEL; fail(LLE_ERR_ARG, "level must be 1..6"); return nullptr; }
    return lle_map_parse(LEVEL_TEXT[level - 1], std::strlen(LEVEL_TEXT[level - 1]), parse_error);
}

void lle_map_free(lle_map* map) { delete map; }

int lle_map_get_info(const lle_map* map, lle_map_info* out) {
    if (!map || !out) return fail(LLE_ERR_NULL, "NULL argument");
    const Map& m = map->m;
    out->height = m.H; out->width = m.W; out->n_agents = m.n_agents(); out->n_gems = (int)m.gems.size();
    out->n_sources = (int)m.sources.size(); out->n_layers = m.n_layers();
    out->n_exits = (int)m.exits.size(); out->n_walls = (int)m.walls.size(); out->n_voids = (int)m.voids.size();
    out->n_laser_tiles = m.n_laser_tiles();
    out->obs_bytes = (int)m.header.obs_bytes; out->obs_stride = (int)m.header.obs_stride;
    int mb = 0;
    for (auto& s : m.sources) mb = std::max(mb, (int)s.beam.size());
    out->max_beam_len = mb; out->max_cell_layers = (int)m.header.max_layers;
    out->obs_supported = (int)m.header.obs_supported; out->table_bytes = (int)m.header.blob_bytes;
    out->n_beam_words = m.n_words();
    out->dyn_row_bytes = (int)(m.header.n_dyn_chunks * 16u);
    return LLE_OK;
}

int lle_map_positions(const lle_map* map, int which, int32_t* out_ij, int cap) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const Map& m = map->m;
    std::vector<Pos> tmp;
    const std::vector<Pos>* v = nullptr;
    switch (which) {
        case LLE_POS_START: for (auto& s : m.starts) tmp.push_back(s[0]); v = &tmp; break;
        case LLE_POS_EXIT: v = &m.exits; break;
        case LLE_POS_WALL: v = &m.walls; break;
        case LLE_POS_VOID: v = &m.voids; break;
        case LLE_POS_GEM: v = &m.gems; break;
        default: return fail(LLE_ERR_ARG, "unknown position list");
    }
    for (int k = 0; k < (int)v->size() && k < cap && out_ij; k++) { out_ij[2 * k] = (*v)[k].i; out_ij[2 * k + 1] = (*v)[k].j; }
    return (int)v->size();
}

int lle_map_sources(const lle_map* map, lle_source_info* out, int cap) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const Map& m = map->m;
    for (int k = 0; k < (int)m.sources.size() && k < cap && out; k++) {
        const Source& s = m.sources[k];
        out[k] = lle_source_info{s.pos.i, s.pos.j, s.direction, s.agent_id, s.enabled ? 1 : 0, (int)s.beam.size(), s.laser_id};
    }
    return (int)m.sources.size();
}

int lle_map_set_source(lle_map* map, int laser_id, int enabled, int agent_id) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    Map& m = map->m;
    if (laser_id < 0 || laser_id >= (int)m.sources.size()) return fail(LLE_ERR_ARG, "laser_id out of range");
    if (enabled >= 0) m.sources[laser_id].enabled = enabled != 0;
    if (agent_id >= 0) m.sources[laser_id].agent_id = agent_id;
    m.compile();
    return LLE_OK;
}

int lle_map_set_exits(lle_map* map, const int32_t* exits_ij, int n_exits, int* parse_error) {
    if (parse_error) *parse_error = LLE_PARSE_OK;
    if (!map || (!exits_ij && n_exits > 0)) return fail(LLE_ERR_NULL, "NULL argument");
    if (n_exits < 0) return fail(LLE_ERR_ARG, "n_exits is negative");
    std::vector<Pos> ex((size_t)n_exits);
    for (int k = 0; k < n_exits; k++) ex[(size_t)k] = Pos{exits_ij[2 * k], exits_ij[2 * k + 1]};
    std::string why;
    const int rc = map->m.set_exits(ex, why);
    if (rc == LLE_PARSE_NOT_ENOUGH_EXIT_TILES) {
        if (parse_error) *parse_error = rc;
        return fail(LLE_ERR_ARG, "Not enough exit tiles: " + std::to_string(map->m.n_agents()) + " starts, " + std::to_string(n_exits) + " exits");
    }
    if (rc != LLE_PARSE_OK) return fail(LLE_ERR_ARG, why);
    g_status = LLE_OK;
    return LLE_OK;
}

lle_map* lle_map_clone(const lle_map* map) {
    if (!map) { fail(LLE_ERR_NULL, "NULL map"); return nullptr; }
    lle_map* m = new (std::nothrow) lle_map(*map);
    if (m) g_status = LLE_OK;
    return m;
}

int lle_map_colour_allowed(const lle_map* map, int laser_id, int agent_id) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const Map& m = map->m;
    if (laser_id < 0 || laser_id >= (int)m.sources.size()) return fail(LLE_ERR_ARG, "laser_id out of range");
    if (agent_id < 0 || agent_id >= m.n_agents()) return fail(LLE_ERR_ARG, "Agent ID is greater than the number of agents");
    return (m.header.colour_ok[m.source_word[(size_t)laser_id]] >> agent_id) & 1;
}

int lle_map_set_row_align(lle_map* map, int align) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    if (align != 0 && align != 16 && align != 32 && align != 64 && align != 128 && align != 256)
        return fail(LLE_ERR_ARG, "row alignment must be 0 (automatic), 16, 32, 64, 128 or 256");
    map->m.row_align = (uint32_t)align;
    map->m.compile();
    return LLE_OK;
}

int lle_map_set_head_lines(lle_map* map, int lines) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    if (lines < -1 || lines > 8) return fail(LLE_ERR_ARG, "head lines must be -1 (automatic) or 0..8");
    map->m.head_lines = lines;
    map->m.compile();
    return LLE_OK;
}

int lle_map_row_head(const lle_map* map, int32_t* first_byte, int32_t* n_bytes) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    if (first_byte) *first_byte = (int32_t)(map->m.header.head_lo * 16u);
    if (n_bytes) *n_bytes = (int32_t)(map->m.header.head_n * 16u);
    return LLE_OK;
}

int lle_map_row_dynamic_lines(const lle_map* map, uint8_t* out_lines, int cap) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const MapHeader& h = map->m.header;
    const int n_lines = (int)((h.obs_stride + 127u) / 128u);
    for (int l = 0; l < n_lines && l < cap && out_lines; l++) out_lines[l] = h.n_dyn_chunks >= h.n_chunks ? 1 : 0;
    if (h.n_dyn_chunks < h.n_chunks && out_lines) {
        const uint16_t* tab = reinterpret_cast<const uint16_t*>(map->m.blob.data() + h.off_dyn_chunks);
        for (uint32_t i = 0; i < h.n_dyn_chunks; i++)
            if ((int)(tab[i] / 8u) < cap) out_lines[tab[i] / 8u] = 1;
    }
    return n_lines;
}

int lle_map_row_head_env_sources(const lle_map* map, int32_t* first_byte, int32_t* n_bytes) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    if (first_byte) *first_byte = (int32_t)(map->m.header.pes_head_lo * 16u);
    if (n_bytes) *n_bytes = (int32_t)(map->m.header.pes_head_n * 16u);
    return LLE_OK;
}

int lle_map_row_head_env_sources_second(const lle_map* map, int32_t* first_byte, int32_t* n_bytes) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    if (first_byte) *first_byte = (int32_t)(map->m.header.pes_head2_lo * 16u);
    if (n_bytes) *n_bytes = (int32_t)(map->m.header.pes_head2_n * 16u);
    return LLE_OK;
}

int64_t lle_map_reset_beam(const lle_map* map, int laser_id, int agent_id) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const MapHeader& h = map->m.header;
    if (laser_id < 0 || laser_id >= (int)h.n_sources || agent_id < 0 || agent_id >= (int)h.A) return fail(LLE_ERR_ARG, "laser_id / agent_id out of range");
    if (map->m.sources[(size_t)laser_id].beam.size() > 63) return fail(LLE_ERR_UNSUPPORTED, "lle_map_reset_beam returns one 64-bit mask: beams of at most 63 cells");
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(map->m.blob.data() + h.off_recolour);
    int64_t mask = 0;  // the words of the source, low cells first
    for (int w = h.source_word[laser_id], k = 0; w < (int)h.L && h.word_source[w] == laser_id && ((h.word_mask >> w) & 1u) && k < 2; w++, k++)
        mask |= (int64_t)tab[(size_t)w * (h.A + 1u) + 1u + (uint32_t)agent_id] << (32 * k);
    return mask;
}

int lle_map_laser_tiles(const lle_map* map, lle_laser_tile* out, int cap) {
    if (!map) return fail(LLE_ERR_NULL, "NULL map");
    const Map& m = map->m;
    int n = 0;
    for (int c = 0; c < m.H * m.W; c++) {
        const auto& layers = m.cell_layers[c];
        for (int k = 0; k < (int)layers.size() && k < 2; k++) {
            if (out && n < cap)
                out[n] = lle_laser_tile{c / m.W, c % m.W, layers[k].laser_id, layers[k].offset, k, m.word_of(layers[k].laser_id, layers[k].offset),
                                        Map::bit_of(layers[k].offset)};
            n++;
        }
    }
    return n;
}

size_t lle_map_world_string(const lle_map* map, char* buf, size_t cap) {
    if (!map) return 0;
    std::string s = map->m.world_string();
    if (buf && cap > 0) {
        size_t n = std::min(cap - 1, s.size());
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}

// ------------------------------------------------------------------------------------------------ batches
int64_t lle_batch_arena_bytes(const lle_map* map, int64_t n_envs) {
    if (!map || n_envs <= 0) return fail(LLE_ERR_ARG, "bad arguments");
    return make_layout(map->m.header, n_envs).total;
}

static void bind_ptrs(lle_batch* b) {
    uint8_t* base = b->arena;
    const Layout& l = b->layout;
    BatchPtrs& p = b->ptrs;
    p.tables = base + l.off_tables;
    p.init = reinterpret_cast<const InitRecord*>(base + l.off_init);
    p.pos = reinterpret_cast<uint16_t*>(base + l.off[LLE_BUF_POS]);
    p.bits = reinterpret_cast<uint64_t*>(base + l.off[LLE_BUF_BITS]);
    p.gems = reinterpret_cast<uint32_t*>(base + l.off[LLE_BUF_GEMS]);
    p.beams = reinterpret_cast<uint32_t*>(base + l.off[LLE_BUF_BEAMS]);
    p.avail = base + l.off[LLE_BUF_AVAIL];
    p.actions = base + l.off[LLE_BUF_ACTIONS];
    p.err = base + l.off[LLE_BUF_ERR];
    p.evcount = base + l.off[LLE_BUF_EVCOUNT];
    p.events = base + l.off[LLE_BUF_EVENTS];
    p.done = base + l.off[LLE_BUF_DONE];
    p.obs = reinterpret_cast<int8_t*>(base + l.off[LLE_BUF_OBS]);
    p.stats = reinterpret_cast<int64_t*>(base + l.off[LLE_BUF_STATS]);
    p.req_pos = reinterpret_cast<const uint16_t*>(base + l.off[LLE_BUF_REQ_POS]);
    p.req_gems = reinterpret_cast<const uint32_t*>(base + l.off[LLE_BUF_REQ_GEMS]);
    p.req_alive = reinterpret_cast<const uint16_t*>(base + l.off[LLE_BUF_REQ_ALIVE]);
    p.reward = reinterpret_cast<uint32_t*>(base + l.off[LLE_BUF_REWARD]);
    p.n_envs = b->n_envs;
    p.src_colour = base + l.off[LLE_BUF_SRC_COLOUR];
    p.src_enabled = reinterpret_cast<uint32_t*>(base + l.off[LLE_BUF_SRC_ENABLED]);
    p.init_pos = reinterpret_cast<uint16_t*>(base + l.off_env_init[0]);
    p.init_bits = reinterpret_cast<uint64_t*>(base + l.off_env_init[1]);
    p.init_gems = reinterpret_cast<uint32_t*>(base + l.off_env_init[2]);
    p.init_beams = reinterpret_cast<uint32_t*>(base + l.off_env_init[3]);
    p.init_avail = base + l.off_env_init[4];
}

#ifdef LLE_STAMP_SINGLE
// diagnostic build only (tools/lle_prof.py stamps --outputs): the next step launches of the process stamp into this buffer
static uint64_t* g_debug_stamps = nullptr;
extern "C" void lle_debug_set_stamps(void* stamps_dev) { g_debug_stamps = (uint64_t*)stamps_dev; }
#endif
static int launch(lle_batch* b, int mode, LaunchArgs K, void* stream) {
#ifdef LLE_STAMP_SINGLE
    if (mode == KMODE_STEP && g_debug_stamps && !K.stamps && !(K.flags & STEP_RECOLOUR_RESETS)) K.stamps = g_debug_stamps;
#endif
    K.envs_per_wave = b->envs_per_wave;
    K.env_base = 0;
    K.env_limit = b->n_envs;
    if (b->per_env_sources) K.flags |= LAUNCH_PER_ENV_SOURCES;
    K.envs_per_map = b->envs_per_map;
    K.table_stride = (uint32_t)b->layout.table_stride;
    K.flags = (K.flags & ~LAUNCH_OBS_ELEM_MASK) | (b->obs_et << LAUNCH_OBS_ELEM_SHIFT);  // every kernel that writes the batch's rows widens alike
    if (mode == KMODE_STEP && (K.flags & STEP_RECOLOUR_RESETS)) {
        if (!(K.flags & STEP_AUTO_RESET)) return fail(LLE_ERR_ARG, "LLE_STEP_RECOLOUR_RESETS re-colours the envs LLE_STEP_AUTO_RESET resets: pass both");
        if (!b->per_env_sources)
            return fail(LLE_ERR_ARG, "LLE_STEP_RECOLOUR_RESETS needs per-environment sources: call lle_batch_set_sources / lle_batch_reset_sources once first");
        if (b->lane_per_env_step) return fail(LLE_ERR_ARG, "LLE_STEP_RECOLOUR_RESETS is served by the default step kernel only");
        if (K.n_steps > 1 || K.ring_slots || K.stamps) return fail(LLE_ERR_UNSUPPORTED, "LLE_STEP_RECOLOUR_RESETS: single steps only (lle_batch_step, lle_batch_step_outputs)");
        for (const Map& m : b->maps)
            if (!m.header.recolour_exact)
                return fail(LLE_ERR_UNSUPPORTED, "LLE_STEP_RECOLOUR_RESETS: a cell of the map carries more than two laser layers "
                                                 "(use lle_batch_reset_sources, which resets such an env in full)");
    }
    if (mode == KMODE_STEP && !b->lane_per_env_step) {
        K.envs_per_wave = batch_step_epw(b, b->tune);
        // every wavefront owns one slot of LLE_BUF_STATS (kernel_common.hpp flush_stats): never more wavefronts than slots
        if ((b->n_envs + K.envs_per_wave - 1) / K.envs_per_wave > b->layout.n_stat_blocks)
            return fail(LLE_ERR_ARG, "internal: more wavefronts than counter slots (envs_per_wave below " + std::to_string(MIN_ENVS_PER_WAVE) + ")");
        // single steps that rewrite the rows in place (a fused rollout rewrites them inside one launch, a ring never revisits a slot in time)
        if (K.n_steps <= 1 && !K.ring_slots && !K.stamps && !(K.flags & STEP_NO_OBS) &&
            next_walk_reversed(b, b->ptrs.obs, b->rows_bytes()))
            K.flags |= LAUNCH_REVERSE;
        HIP_TRY(launch_step_kernel(b->hdr, b->ptrs, K, (hipStream_t)stream, b->tune));
    }
    else {
        if (b->envs_per_map)  // (narrowed like the launcher will: the lane-per-env STEP flushes its counters per wavefront too)
            while (K.envs_per_wave > 1 && b->envs_per_map % (int64_t)K.envs_per_wave != 0) K.envs_per_wave >>= 1;
        if (mode == KMODE_STEP && (b->n_envs + K.envs_per_wave - 1) / K.envs_per_wave > b->layout.n_stat_blocks)
            return fail(LLE_ERR_ARG, "internal: more wavefronts than counter slots");
        if (mode != KMODE_SET_STATE && next_walk_reversed(b, b->ptrs.obs, b->rows_bytes())) K.flags |= LAUNCH_REVERSE;
        HIP_TRY(launch_world_kernel(mode, b->hdr, b->ptrs, K, (hipStream_t)stream));
    }
    g_status = LLE_OK;
    return LLE_OK;
}

// World::reset is deterministic for v1 maps, so its result is the same for every env: run it once on the hidden env
// (slot n_envs) with the current tables and keep the result as the InitRecord the auto-reset path copies from.
static int refresh_init_record(lle_batch* b, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = b->n_envs, L = b->hdr.L, A = agent_stride((int)b->hdr.A, (int)b->hdr.L);
    for (size_t m = 0; m < b->maps.size(); m++) {
        LaunchArgs K{};
        K.envs_per_wave = MIN_ENVS_PER_WAVE;
        K.env_base = b->n_envs;
        K.env_limit = b->n_envs + 1;
        K.flags = STEP_NO_OBS;
        K.table_stride = (uint32_t)b->layout.table_stride;
        K.map_override = (uint32_t)m + 1u;
        HIP_TRY(launch_world_kernel(KMODE_RESET, b->hdr, b->ptrs, K, st));
        uint8_t* rec = b->arena + b->layout.off_init + (int64_t)m * (int64_t)sizeof(InitRecord);
        HIP_TRY(hipMemcpyAsync(rec + offsetof(InitRecord, pos), b->ptrs.pos + n * A, (size_t)A * 2, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(rec + offsetof(InitRecord, bits), b->ptrs.bits + n, 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(rec + offsetof(InitRecord, gems), b->ptrs.gems + n, 4, hipMemcpyDeviceToDevice, st));
        if (L > 0)
            HIP_TRY(hipMemcpyAsync(rec + offsetof(InitRecord, beams), b->ptrs.beams + n * L, (size_t)L * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(rec + offsetof(InitRecord, avail), b->ptrs.avail + n * A, (size_t)A, hipMemcpyDeviceToDevice, st));
    }
    return LLE_OK;
}

static int create_impl(lle_batch* b, void* arena, int64_t arena_bytes, void* stream) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(LLE_ERR_NO_DEVICE, "no HIP device: lle_amd has no CPU execution path");
    if (b->device < 0 || b->device >= n_dev) return fail(LLE_ERR_ARG, "device_id out of range");
    ON_DEVICE_OF(b);
    MapHeader worst = b->hdr;
    worst.lds_table_bytes = b->worst_table_bytes;
    const uint32_t lds = kernel_lds_bytes(worst, 1);
    if (lds > 160 * 1024) return fail(LLE_ERR_UNSUPPORTED, "map tables need " + std::to_string(lds) + " B of LDS per wavefront (> 160 KiB)");
    b->layout = make_layout(b->hdr, b->n_envs, (int64_t)b->maps.size(), b->obs_et);
    if (arena) {
        if (arena_bytes < b->layout.total || (reinterpret_cast<uintptr_t>(arena) % ALIGN) != 0)
            return fail(LLE_ERR_ARENA, "arena too small or not 256-byte aligned");
        b->arena = static_cast<uint8_t*>(arena);
        b->owns_arena = false;
    } else {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)b->layout.total));
        b->arena = static_cast<uint8_t*>(p);
        b->owns_arena = true;
    }
    bind_ptrs(b);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(b->arena, 0, (size_t)b->layout.off[LLE_BUF_OBS], st));
    HIP_TRY(hipMemsetAsync(b->arena + b->layout.off[LLE_BUF_STATS], 0,
                           (size_t)(b->layout.total - b->layout.off[LLE_BUF_STATS]), st));
    for (size_t m = 0; m < b->maps.size(); m++)
        HIP_TRY(hipMemcpyAsync(b->arena + b->layout.off_tables + (int64_t)m * b->layout.table_stride, b->maps[m].blob.data(),
                               b->maps[m].blob.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the blobs are host vectors: make the copies complete before returning
    int rc = refresh_init_record(b, stream);
    if (rc != LLE_OK) return rc;
    LaunchArgs K{};
    return launch(b, KMODE_RESET, K, stream);
}

// common header of a set of maps: the dimensions must agree; table sizes are the largest
static int common_header(const lle_map* const* maps, int n_maps, MapHeader* out, uint32_t* worst_table_bytes) {
    MapHeader h = maps[0]->m.header;
    uint32_t worst = h.lds_table_bytes + (h.blob_capacity - h.blob_bytes);  // table section with the largest possible dyn table
    for (int m = 1; m < n_maps; m++) {
        const MapHeader& o = maps[m]->m.header;
        if (o.H != h.H || o.W != h.W || o.A != h.A || o.L != h.L || o.G != h.G || o.n_sources != h.n_sources)
            return fail(LLE_ERR_ARG, "the maps of a batch must agree on height, width and the numbers of agents, sources (and beam words) and gems");
        if (std::memcmp(o.word_source, h.word_source, sizeof h.word_source) != 0 || o.chain_mask != h.chain_mask)
            return fail(LLE_ERR_ARG, "the maps of a batch must agree on which beams are longer than 32 cells (the layout of their beam words)");
        if (o.obs_stride != h.obs_stride) return fail(LLE_ERR_ARG, "the maps of a batch must agree on the row alignment (lle_map_set_row_align)");
        worst = std::max(worst, o.lds_table_bytes + (o.blob_capacity - o.blob_bytes));
        h.lds_table_bytes = std::max(h.lds_table_bytes, o.lds_table_bytes);
        h.blob_capacity = std::max(h.blob_capacity, o.blob_capacity);
        h.ext_bytes = std::max(h.ext_bytes, o.ext_bytes);
        h.packed_cap = std::max(h.packed_cap, o.packed_cap);
        if (!o.off_packed) h.off_packed = 0;  // (the common header: non-zero only when every map carries the packed image)
        h.lds_split_table_bytes = std::max(h.lds_split_table_bytes, o.lds_split_table_bytes);
        h.n_elems = std::max(h.n_elems, o.n_elems);
        h.obs_supported = h.obs_supported && o.obs_supported;
        h.max_layers = std::max(h.max_layers, o.max_layers);
    }
    *out = h;
    if (worst_table_bytes) *worst_table_bytes = worst;
    return LLE_OK;
}

static bool options_ok(const lle_batch_options* opt) {
    if (!opt) return true;
    if (opt->struct_bytes < 8u) { fail(LLE_ERR_ARG, "lle_batch_options.struct_bytes must be set to sizeof(lle_batch_options)"); return false; }
    if (opt->obs_dtype < LLE_DTYPE_I8 || opt->obs_dtype > LLE_DTYPE_F32) { fail(LLE_ERR_ARG, "obs_dtype must be LLE_DTYPE_I8, _F16, _BF16 or _F32"); return false; }
    return true;
}

static lle_batch* create_batch(const lle_map* const* maps, int n_maps, int64_t n_envs, int device_id, void* arena, int64_t arena_bytes,
                               void* stream, const lle_batch_options* opt = nullptr) {
    if (!options_ok(opt)) return nullptr;
    lle_batch* b = new (std::nothrow) lle_batch();
    if (!b) return nullptr;
    if (opt) b->obs_et = (uint32_t)opt->obs_dtype;
    if (common_header(maps, n_maps, &b->hdr, &b->worst_table_bytes) != LLE_OK) { delete b; return nullptr; }
    for (int m = 0; m < n_maps; m++) b->maps.push_back(maps[m]->m);
    b->envs_per_map = n_maps > 1 ? n_envs / n_maps : 0;
    b->n_envs = n_envs;
    b->device = device_id;
    b->arena = nullptr;
    b->owns_arena = false;
    b->lane_per_env_step = false;
    b->per_env_sources = false;
    // enough waves to cover the 256 CUs several times over, at most 32 envs per wave (measured best on level 6)
    b->envs_per_wave = 32;
    while (b->envs_per_wave > MIN_ENVS_PER_WAVE && n_envs / b->envs_per_wave < 2048) b->envs_per_wave /= 2;
    if (create_impl(b, arena, arena_bytes, stream) != LLE_OK) {
        if (b->owns_arena && b->arena) (void)hipFree(b->arena);
        delete b;
        return nullptr;
    }
    return b;
}

lle_batch* lle_batch_create(const lle_map* map, int64_t n_envs, int device_id, void* arena, int64_t arena_bytes, void* stream) {
    if (!map) { fail(LLE_ERR_NULL, "NULL map"); return nullptr; }
    if (n_envs <= 0) { fail(LLE_ERR_ARG, "n_envs must be positive"); return nullptr; }
    return create_batch(&map, 1, n_envs, device_id, arena, arena_bytes, stream);
}

int64_t lle_batch_arena_bytes_multi(const lle_map* const* maps, int n_maps, int64_t envs_per_map) {
    if (!maps || n_maps <= 0 || envs_per_map <= 0) return fail(LLE_ERR_ARG, "bad arguments");
    for (int m = 0; m < n_maps; m++)
        if (!maps[m]) return fail(LLE_ERR_NULL, "NULL map");
    MapHeader h;
    int rc = common_header(maps, n_maps, &h, nullptr);
    if (rc != LLE_OK) return rc;
    return make_layout(h, envs_per_map * n_maps, n_maps).total;
}

lle_batch* lle_batch_create_multi(const lle_map* const* maps, int n_maps, int64_t envs_per_map, int device_id, void* arena,
                                  int64_t arena_bytes, void* stream) {
    if (!maps || n_maps <= 0) { fail(LLE_ERR_ARG, "no maps"); return nullptr; }
    for (int m = 0; m < n_maps; m++)
        if (!maps[m]) { fail(LLE_ERR_NULL, "NULL map"); return nullptr; }
    if (envs_per_map <= 0) {
        fail(LLE_ERR_ARG, "envs_per_map must be positive");
        return nullptr;
    }
    return create_batch(maps, n_maps, envs_per_map * n_maps, device_id, arena, arena_bytes, stream);
}

int64_t lle_batch_arena_bytes_opt(const lle_map* const* maps, int n_maps, int64_t envs_per_map, const lle_batch_options* opt) {
    if (!maps || n_maps <= 0 || envs_per_map <= 0) return fail(LLE_ERR_ARG, "bad arguments");
    for (int m = 0; m < n_maps; m++)
        if (!maps[m]) return fail(LLE_ERR_NULL, "NULL map");
    if (!options_ok(opt)) return LLE_ERR_ARG;
    MapHeader h;
    int rc = common_header(maps, n_maps, &h, nullptr);
    if (rc != LLE_OK) return rc;
    return make_layout(h, envs_per_map * n_maps, n_maps, opt ? (uint32_t)opt->obs_dtype : (uint32_t)OBS_I8).total;
}

lle_batch* lle_batch_create_opt(const lle_map* const* maps, int n_maps, int64_t envs_per_map, int device_id, void* arena, int64_t arena_bytes,
                                const lle_batch_options* opt, void* stream) {
    if (!maps || n_maps <= 0) { fail(LLE_ERR_ARG, "no maps"); return nullptr; }
    for (int m = 0; m < n_maps; m++)
        if (!maps[m]) { fail(LLE_ERR_NULL, "NULL map"); return nullptr; }
    if (envs_per_map <= 0) {
        fail(LLE_ERR_ARG, "envs_per_map must be positive");
        return nullptr;
    }
    return create_batch(maps, n_maps, envs_per_map * n_maps, device_id, arena, arena_bytes, stream, opt);
}

int lle_batch_obs_dtype(const lle_batch* b) { return b ? (int)b->obs_et : LLE_ERR_NULL; }

int lle_batch_n_maps(const lle_batch* b) { return b ? (int)b->maps.size() : 0; }

static void drop_views(lle_batch* b) {
    for (auto& kv : b->views)
        if (kv.second.dev) (void)hipFree(kv.second.dev);
    b->views.clear();
    for (auto& w : b->win_sets) {
        if (w) (void)hipFree(w);
        w = nullptr;
    }
}

// Device copy of the window tables of window size k (tables.h): win_table_bytes(HW) per map; NULL for the other sizes.
static int get_win_sets(lle_batch* b, int k, hipStream_t st, const uint8_t** out) {
    *out = nullptr;
    if (!win_sets_serve(k) || getenv("LLE_PARTIAL_NO_SETS")) return LLE_OK;  // (LLE_PARTIAL_NO_SETS: the bitmap path for every size -- A/B and cross-check)
    uint8_t*& dev = b->win_sets[(k - 3) / 2];
    if (!dev) {
        const size_t each = win_table_bytes(b->hdr.HW);
        std::vector<uint8_t> all(each * b->maps.size() + 1024);  // (+ a row: a kernel's whole-row copy of a part of the last table may read past it)
        for (size_t m = 0; m < b->maps.size(); m++) {
            const std::vector<uint8_t> one = b->maps[m].window_table(k);
            std::memcpy(all.data() + m * each, one.data(), each);
        }
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, all.size()));
        hipError_t e = hipMemcpyAsync(p, all.data(), all.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (a host temporary)
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(LLE_ERR_HIP, std::string("window sets upload: ") + hipGetErrorString(e));
        }
        dev = static_cast<uint8_t*>(p);
    }
    *out = dev;
    return LLE_OK;
}

void lle_batch_free(lle_batch* b) {
    if (!b) return;
    drop_views(b);
    if (b->owns_arena && b->arena) (void)hipFree(b->arena);
    delete b;
}

int64_t lle_batch_n_envs(const lle_batch* b) { return b ? b->n_envs : 0; }

int lle_batch_get_buffer(const lle_batch* b, int which, lle_buffer_desc* out) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    if (which < 0 || which >= LLE_BUF_COUNT) return fail(LLE_ERR_ARG, "unknown buffer");
    const int64_t n = b->n_envs, A = b->hdr.A, L = b->hdr.L, As = agent_stride((int)b->hdr.A, (int)b->hdr.L);
    lle_buffer_desc d{};
    d.ptr = b->arena + b->layout.off[which];
    d.arena_offset = b->layout.off[which];
    d.bytes = b->layout.bytes[which];
    auto set = [&](int elem, int ndim, int64_t s0, int64_t s1, int64_t s2, int64_t t0, int64_t t1, int64_t t2) {
        d.elem_bytes = elem; d.ndim = ndim;
        d.shape[0] = s0; d.shape[1] = s1; d.shape[2] = s2;
        d.stride[0] = t0; d.stride[1] = t1; d.stride[2] = t2;
    };
    switch (which) {
        case LLE_BUF_POS: case LLE_BUF_REQ_POS: set(1, 3, n, A, 2, 2 * As, 2, 1); break;
        case LLE_BUF_BITS: set(8, 1, n, 1, 1, 1, 1, 1); break;
        case LLE_BUF_GEMS: case LLE_BUF_REQ_GEMS: set(4, 1, n, 1, 1, 1, 1, 1); break;
        case LLE_BUF_BEAMS: set(4, 2, n, L, 1, L, 1, 1); break;
        case LLE_BUF_AVAIL: case LLE_BUF_ACTIONS: set(1, 2, n, A, 1, As, 1, 1); break;
        case LLE_BUF_ERR: case LLE_BUF_EVCOUNT: case LLE_BUF_DONE: set(1, 1, n, 1, 1, 1, 1, 1); break;
        case LLE_BUF_EVENTS: set(1, 2, n, 2 * A, 1, 2 * As, 1, 1); break;
        case LLE_BUF_OBS: set(1 << obs_elem_shift(b->obs_et), 2, n, b->hdr.obs_bytes, 1, b->hdr.obs_stride, 1, 1); break;  // (C*H*W ELEMENTS per row, pitch in elements)
        case LLE_BUF_STATS: set(8, 2, b->layout.n_stat_blocks, 8, 1, 8, 1, 1); break;
        case LLE_BUF_REQ_ALIVE: set(2, 1, n, 1, 1, 1, 1, 1); break;
        case LLE_BUF_REWARD: set(1, 2, n, 4, 1, 4, 1, 1); break;
        case LLE_BUF_SRC_COLOUR: set(1, 2, n, L, 1, src_stride_of((int)L), 1, 1); break;
        case LLE_BUF_SRC_ENABLED: set(4, 1, n, 1, 1, 1, 1, 1); break;
    }
    *out = d;
    return LLE_OK;
}

// the dynamic state is the contiguous arena range [pos, actions): pos, bits, gems, beams, avail; followed in a snapshot
// by the per-env sources and reset states (meaningful once lle_batch_set_sources has been called)
static int64_t state_bytes(const lle_batch* b) { return b->layout.off[LLE_BUF_ACTIONS] - b->layout.off[LLE_BUF_POS]; }
// (up to, not including, the device mirror of the EnvOutputs descriptor of lle_batch_step_outputs: pointers are not state)
static int64_t sources_bytes(const lle_batch* b) { return b->layout.off_env_out - b->layout.off[LLE_BUF_SRC_COLOUR]; }
int64_t lle_batch_snapshot_bytes(const lle_batch* b) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    return state_bytes(b) + sources_bytes(b) + 256;
}
int lle_batch_snapshot(lle_batch* b, void* dst_dev, void* stream) {
    if (!b || !dst_dev) return fail(LLE_ERR_NULL, "NULL argument");
    ON_DEVICE_OF(b);
    uint8_t* dst = static_cast<uint8_t*>(dst_dev);
    const uint64_t mode = b->per_env_sources ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(dst, &mode, 8, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // `mode` is a stack variable
    HIP_TRY(hipMemcpyAsync(dst + 256, b->arena + b->layout.off[LLE_BUF_POS], (size_t)state_bytes(b), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    if (b->per_env_sources)
        HIP_TRY(hipMemcpyAsync(dst + 256 + state_bytes(b), b->arena + b->layout.off[LLE_BUF_SRC_COLOUR], (size_t)sources_bytes(b),
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return LLE_OK;
}
int lle_batch_restore(lle_batch* b, const void* src_dev, void* stream) {
    if (!b || !src_dev) return fail(LLE_ERR_NULL, "NULL argument");
    ON_DEVICE_OF(b);
    const uint8_t* src = static_cast<const uint8_t*>(src_dev);
    uint64_t mode = 0;
    HIP_TRY(hipMemcpyAsync(&mode, src, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if ((mode != 0) != b->per_env_sources)
        return fail(LLE_ERR_ARG, "snapshot and batch disagree on per-environment sources");
    HIP_TRY(hipMemcpyAsync(b->arena + b->layout.off[LLE_BUF_POS], src + 256, (size_t)state_bytes(b), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    if (b->per_env_sources)
        HIP_TRY(hipMemcpyAsync(b->arena + b->layout.off[LLE_BUF_SRC_COLOUR], src + 256 + state_bytes(b), (size_t)sources_bytes(b),
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
    LaunchArgs K{};
    // bring the observation (and `done`) in line with the restored state.  With per-environment sources also every env's own
    // reset record, from its restored colours / flags under the CURRENT tables: the snapshot may predate an exit change
    // (lle_batch_update_map), and an agent whose start is an exit now arrives at reset
    return launch(b, b->per_env_sources ? KMODE_ENV_SOURCES : KMODE_OBSERVE, K, stream);
}

int lle_batch_reset(lle_batch* b, const uint8_t* env_mask_dev, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    K.env_mask = env_mask_dev;
    return launch(b, KMODE_RESET, K, stream);
}

int lle_batch_step(lle_batch* b, const uint8_t* actions_dev, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset,
                   void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    K.flags = flags & STEP_PUBLIC_FLAGS; K.seed = seed; K.t = t; K.env_offset = env_offset; K.actions_in = actions_dev;  // (the bits above are the library's own)
    return launch(b, KMODE_STEP, K, stream);
}

static int obs_desc(const lle_batch* b, int kind, int param, lle_obs_desc* d);  // (below)

int lle_batch_step_outputs(lle_batch* b, const uint8_t* actions_dev, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset,
                           const lle_env_outputs* out, void* stream) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    if (out->reward_kind != 0 && out->reward_kind != 1) return fail(LLE_ERR_ARG, "reward_kind must be 0 (single objective) or 1 (multi objective)");
    if (out->available && !out->walkable_lasers)
        return fail(LLE_ERR_UNSUPPORTED, "the fused step writes LLE.available_actions with walkable_lasers only: use lle_batch_env_outputs");
    if (b->lane_per_env_step) return fail(LLE_ERR_ARG, "the fused outputs are written by the default step kernel only");
    ON_DEVICE_OF(b);
    EnvOutputs O{};
    O.state = out->state; O.reward = out->reward; O.done = out->done; O.available = out->available; O.alive = out->alive; O.arrived = out->arrived;
    O.normalize_state = out->normalize_state ? 1 : 0; O.reward_kind = (uint8_t)out->reward_kind; O.walkable_lasers = out->walkable_lasers ? 1 : 0;
    O.per_env_sources = b->per_env_sources ? 1 : 0;
    uint32_t partial_E = 0;
    if (out->partial) {
        // the partial k x k observation written by the step launch itself (step_kernel MODE 9, partial_stream.hpp)
        lle_obs_desc d;
        int rc = obs_desc(b, LLE_OBS_PARTIAL, out->partial_k, &d);
        if (rc != LLE_OK) return rc;
        if (!d.supported) return fail(LLE_ERR_UNSUPPORTED, "a laser colour has no layer in this observation (the reference raises IndexError)");
        if ((reinterpret_cast<uintptr_t>(out->partial) % 16) != 0) return fail(LLE_ERR_ARENA, "the partial observation buffer must be 16-byte aligned");
        partial_E = step_partial_batch(b->hdr, out->partial_k, b->per_env_sources);
        if (!partial_E)
            return fail(LLE_ERR_UNSUPPORTED, "the step launch writes the partial observation for maps with at most 8 beam words and the map's own sources "
                                             "(not per-environment ones): use lle_batch_observe_as(LLE_OBS_PARTIAL) behind the step");
        O.partial = out->partial;
        O.partial_k = (uint32_t)out->partial_k;
    }
    LaunchArgs K{};
    // (the struct rides in the kernel arguments: nothing to upload, whatever the caller hands from one step to the next)
    K.flags = flags & STEP_PUBLIC_FLAGS; K.seed = seed; K.t = t; K.env_offset = env_offset; K.actions_in = actions_dev;
    K.out = O; K.env_out = reinterpret_cast<const EnvOutputs*>(b->arena + b->layout.off_env_out);  // (non-NULL: "outputs wanted"; never read)
    K.partial_k = O.partial ? O.partial_k : 0u; K.partial_E = partial_E;
    if (O.partial) {  // (the launcher uses the window sets where they do not cost the launch a workgroup per CU: kernels.hip launch_step_kernel)
        int rc = get_win_sets(b, (int)O.partial_k, (hipStream_t)stream, &K.win_sets);
        if (rc != LLE_OK) return rc;
    }
    if (O.partial && (flags & STEP_RECOLOUR_RESETS)) return fail(LLE_ERR_UNSUPPORTED, "the partial observation of the step launch: the map's own sources only");
    return launch(b, KMODE_STEP, K, stream);
}

int lle_batch_rollout(lle_batch* b, uint32_t n_steps, uint32_t flags, uint64_t seed, uint64_t t0, int64_t env_offset,
                      const lle_rollout_ring* ring, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    if (n_steps == 0 || n_steps > 4096) return fail(LLE_ERR_ARG, "n_steps must be 1..4096");
    if (b->lane_per_env_step) return fail(LLE_ERR_ARG, "the fused rollout runs on the default step kernel only");
    LaunchArgs K{};
    K.flags = flags & STEP_PUBLIC_FLAGS; K.seed = seed; K.t = t0; K.env_offset = env_offset; K.n_steps = n_steps;
    if (ring) {
        if (ring->ring_slots < 1 || !ring->obs || !ring->actions || !ring->reward) return fail(LLE_ERR_ARG, "incomplete ring");
        K.ring_slots = (uint32_t)ring->ring_slots; K.ring_pos = ring->ring_pos % (uint64_t)ring->ring_slots; K.ring_env_count = b->n_envs;
        K.ring_obs = ring->obs; K.ring_actions = ring->actions; K.ring_reward = ring->reward;
    }
    return launch(b, KMODE_STEP, K, stream);
}

int lle_batch_set_state(lle_batch* b, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    return launch(b, KMODE_SET_STATE, K, stream);
}

static int set_sources(lle_batch* b, const uint8_t* colours_dev, const uint32_t* enabled_dev, const uint8_t* env_mask_dev,
                       uint32_t flags, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    if (kernel_lds_bytes(b->hdr, 1, true) > 160 * 1024)
        return fail(LLE_ERR_UNSUPPORTED, "map tables with per-environment sources exceed the LDS of a workgroup");
    if (!b->per_env_sources) {
        // first call: every env gets the map's sources and the shared reset state as its own
        b->per_env_sources = true;
        LaunchArgs K0{};
        K0.flags = LAUNCH_FILL_DEFAULTS | LAUNCH_ARRAYS_INVALID;
        int rc = launch(b, KMODE_ENV_SOURCES, K0, stream);
        if (rc != LLE_OK) { b->per_env_sources = false; return rc; }
    }
    LaunchArgs K{};
    K.flags = flags;
    K.colours_in = colours_dev;
    K.enabled_in = enabled_dev;
    K.env_mask = env_mask_dev;
    return launch(b, KMODE_ENV_SOURCES, K, stream);
}

int lle_batch_set_sources(lle_batch* b, const uint8_t* colours_dev, const uint32_t* enabled_dev, const uint8_t* env_mask_dev,
                          void* stream) {
    return set_sources(b, colours_dev, enabled_dev, env_mask_dev, 0u, stream);
}

int lle_batch_reset_sources(lle_batch* b, const uint8_t* colours_dev, const uint32_t* enabled_dev, const uint8_t* env_mask_dev,
                            uint32_t flags, void* stream) {
    if (flags & ~(uint32_t)LLE_STEP_NO_OBS) return fail(LLE_ERR_ARG, "lle_batch_reset_sources takes LLE_STEP_NO_OBS or 0");
    return set_sources(b, colours_dev, enabled_dev, env_mask_dev, LAUNCH_RESET_FIRST | (flags & LLE_STEP_NO_OBS), stream);
}

int lle_batch_observe(lle_batch* b, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    return launch(b, KMODE_OBSERVE, K, stream);
}

// ---- the other observation builders (observers.hip)
static int obs_desc(const lle_batch* b, int kind, int param, lle_obs_desc* d) {
    const MapHeader& h = b->hdr;
    const int64_t n = b->n_envs, A = h.A, H = h.H, W = h.W;
    *d = lle_obs_desc{};
    d->kind = kind; d->param = param; d->supported = 1;
    auto set = [&](int elem, std::initializer_list<int64_t> shape, int64_t env_pitch_elems) {
        d->elem_bytes = elem;
        d->ndim = (int32_t)shape.size();
        int k = 0;
        for (int64_t v : shape) d->shape[k++] = v;
        int64_t st = 1;
        for (int q = d->ndim - 1; q >= 1; q--) { d->stride[q] = st; st *= d->shape[q]; }
        d->stride[0] = env_pitch_elems;
        d->bytes = n * env_pitch_elems * elem;
    };
    // the layered-style observations (-1 / 0 / 1) come in the batch's element type (lle_batch_options.obs_dtype): every kernel that writes them widens
    // at the store; the state vector is float32 whatever the batch
    const int el = 1 << obs_elem_shift(b->obs_et);
    switch (kind) {
        case LLE_OBS_LAYERED:
            if (param != 0) return fail(LLE_ERR_ARG, "LLE_OBS_LAYERED takes no parameter");
            set(el, {n, (int64_t)h.C, H, W}, h.obs_stride);
            d->supported = (int32_t)h.obs_supported;
            return LLE_OK;
        case LLE_OBS_LAYERED_PADDED: {
            if (param < 0 || A + param > 32) return fail(LLE_ERR_ARG, "padding out of range (n_agents + padding <= 32)");
            const int64_t C = 2 * (A + param) + 4;
            if (C * H * W >= (1 << 20)) return fail(LLE_ERR_UNSUPPORTED, "padded observation too large");
            set(el, {n, C, H, W}, (int64_t)b->maps[0].row_pitch_of((uint32_t)(C * H * W)));
            for (const Map& mp : b->maps)
                for (const Source& s : mp.sources)
                    if (s.agent_id >= C - (A + param)) d->supported = 0;
            return LLE_OK;
        }
        case LLE_OBS_PERSPECTIVE: {
            if (param != 0) return fail(LLE_ERR_ARG, "LLE_OBS_PERSPECTIVE takes no parameter");
            set(el, {n, A, (int64_t)h.C, H, W}, A * (int64_t)h.obs_stride);
            d->stride[1] = h.obs_stride;  // one padded layered row per observer
            d->supported = (int32_t)h.obs_supported;
            return LLE_OK;
        }
        case LLE_OBS_PARTIAL: {
            if (param < 1 || param > 15 || param % 2 == 0) return fail(LLE_ERR_ARG, "square size must be odd, 1..15");
            set(el, {n, A, 2 * A + 3, (int64_t)param, (int64_t)param}, partial_pitch((int)A, param));
            for (const Map& mp : b->maps)
                for (const Source& s : mp.sources)
                    if (s.agent_id > (int)A + 1) d->supported = 0;  // LASER_0 + colour must be a layer (< 2A+3)
            return LLE_OK;
        }
        case LLE_OBS_STATE:
        case LLE_OBS_NORMALIZED_STATE:
            if (param != 0) return fail(LLE_ERR_ARG, "the state observation takes no parameter");
            set(4, {n, 3 * A + (int64_t)h.G}, 3 * A + (int64_t)h.G);
            return LLE_OK;
        default:
            return fail(LLE_ERR_ARG, "unknown observation kind");
    }
}

int lle_batch_obs_desc(lle_batch* b, int kind, int param, lle_obs_desc* out) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    int rc = obs_desc(b, kind, param, out);
    if (rc == LLE_OK) g_status = LLE_OK;
    return rc;
}

// Device copy of the view blob(s) for (kind, param), one per map, `stride` bytes apart.
// (OBS_PERSPECTIVE, -1) = per map, the blobs of all observers back to back.
static int get_view(lle_batch* b, int kind, int param, hipStream_t st, const lle_batch::View** out) {
    auto key = std::make_pair(kind, param);
    auto it = b->views.find(key);
    if (it == b->views.end()) {
        std::vector<std::vector<uint8_t>> blobs;
        size_t stride = 0;
        for (const Map& mp : b->maps) {
            std::vector<uint8_t> blob;
            if (kind == LLE_OBS_PERSPECTIVE && param < 0) {
                for (int k = 0; k < (int)b->hdr.A; k++) {
                    std::vector<uint8_t> one = mp.compile_view(kind, k);
                    blob.insert(blob.end(), one.begin(), one.end());
                }
            } else {
                blob = mp.compile_view(kind, param);
            }
            stride = std::max(stride, blob.size());
            blobs.push_back(std::move(blob));
        }
        lle_batch::View v{};
        // the header the launcher sizes LDS with: the largest blob of the set
        for (const auto& blob : blobs) {
            ViewHeader vh;
            std::memcpy(&vh, blob.data(), sizeof vh);
            if (vh.blob_bytes >= v.hdr.blob_bytes) v.hdr = vh;
        }
        v.stride = (uint32_t)stride;
        std::vector<uint8_t> all(stride * blobs.size(), 0);
        for (size_t m = 0; m < blobs.size(); m++) std::memcpy(all.data() + m * stride, blobs[m].data(), blobs[m].size());
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, all.size()));
        v.dev = static_cast<uint8_t*>(p);
        hipError_t e = hipMemcpyAsync(v.dev, all.data(), all.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the blob is a host temporary
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(LLE_ERR_HIP, std::string("view upload: ") + hipGetErrorString(e));
        }
        it = b->views.emplace(key, v).first;
    }
    *out = &it->second;
    return LLE_OK;
}

int lle_batch_observe_as(lle_batch* b, int kind, int param, void* out_dev, int64_t out_bytes, void* stream) {
    if (!b || !out_dev) return fail(LLE_ERR_NULL, "NULL argument");
    lle_obs_desc d;
    int rc = obs_desc(b, kind, param, &d);
    if (rc != LLE_OK) return rc;
    if (!d.supported) return fail(LLE_ERR_UNSUPPORTED, "a laser colour has no layer in this observation (the reference raises IndexError)");
    if (b->per_env_sources && kind == LLE_OBS_LAYERED) {  // the env-coloured layered writer of the world kernel, into `out_dev`
        ON_DEVICE_OF(b);
        BatchPtrs P = b->ptrs;
        P.obs = static_cast<int8_t*>(out_dev);
        LaunchArgs K{};
        K.envs_per_wave = b->envs_per_wave;
        K.env_limit = b->n_envs;
        K.flags = LAUNCH_PER_ENV_SOURCES | (b->obs_et << LAUNCH_OBS_ELEM_SHIFT);
        if (next_walk_reversed(b, out_dev, (uint64_t)d.bytes)) K.flags |= LAUNCH_REVERSE;
        K.envs_per_map = b->envs_per_map;
        K.table_stride = (uint32_t)b->layout.table_stride;
        HIP_TRY(launch_world_kernel(KMODE_OBSERVE, b->hdr, P, K, (hipStream_t)stream));
        g_status = LLE_OK;
        return LLE_OK;
    }
    if (out_bytes < d.bytes || (reinterpret_cast<uintptr_t>(out_dev) % 16) != 0)
        return fail(LLE_ERR_ARENA, "output buffer too small or not 16-byte aligned");
    ON_DEVICE_OF(b);
    hipStream_t st = (hipStream_t)stream;
    const MapHeader& h = b->hdr;
    const bool pes = b->per_env_sources;
    const MapSel M{b->envs_per_map, (uint32_t)b->layout.table_stride, 0u};
    const bool reverse = next_walk_reversed(b, out_dev, (uint64_t)d.bytes);  // (every launch of this call in the same direction)
    const uint32_t sh = obs_elem_shift(b->obs_et);  // (log2 of the element size of the layered-style outputs)
    switch (kind) {
        case LLE_OBS_LAYERED:
        case LLE_OBS_PERSPECTIVE: {
            // observer k of the perspective = the layered tensor with two layer pairs swapped; k = 0 is Layered itself
            const int n_views = kind == LLE_OBS_PERSPECTIVE ? (int)h.A : 1;
            const lle_batch::View* v;
            rc = get_view(b, LLE_OBS_PERSPECTIVE, 0, st, &v);
            if (rc != LLE_OK) return rc;
            if (n_views > 1 && view_kernel_fits(v->hdr, (uint32_t)n_views, pes, h.n_elems)) {
                // small maps: one launch, the rows of all observers of an env written back to back
                rc = get_view(b, LLE_OBS_PERSPECTIVE, -1, st, &v);
                if (rc != LLE_OK) return rc;
                HIP_TRY(launch_view_observe(v->hdr, b->ptrs, v->dev, (uint32_t)n_views, static_cast<int8_t*>(out_dev),
                                            ((int64_t)n_views * h.obs_stride) << sh, (int64_t)h.obs_stride << sh, b->n_envs, pes, h.n_elems, M, v->stride, reverse, st, b->obs_et));
                break;
            }
            for (int k = 0; k < n_views; k++) {  // big rows: one launch per observer, rows strided by A * obs_stride
                rc = get_view(b, LLE_OBS_PERSPECTIVE, k, st, &v);
                if (rc != LLE_OK) return rc;
                HIP_TRY(launch_view_observe(v->hdr, b->ptrs, v->dev, 1u, static_cast<int8_t*>(out_dev) + (((int64_t)k * h.obs_stride) << sh),
                                            ((int64_t)n_views * h.obs_stride) << sh, 0, b->n_envs, pes, h.n_elems, M, v->stride, reverse, st, b->obs_et));
            }
            break;
        }
        case LLE_OBS_LAYERED_PADDED: {
            const lle_batch::View* v;
            rc = get_view(b, kind, param, st, &v);
            if (rc != LLE_OK) return rc;
            HIP_TRY(launch_view_observe(v->hdr, b->ptrs, v->dev, 1u, static_cast<int8_t*>(out_dev), (int64_t)v->hdr.obs_stride << sh, 0,
                                        b->n_envs, pes, h.n_elems, M, v->stride, reverse, st, b->obs_et));
            break;
        }
        case LLE_OBS_PARTIAL: {
            uint32_t n_entities = 0;  // what a k x k window can show besides agents (observers.hip partial_project_kernel)
            for (const Map& mp : b->maps)
                n_entities = std::max<uint32_t>(n_entities, (uint32_t)(mp.walls.size() + mp.exits.size() + mp.gems.size() + mp.n_laser_tiles() +
                                                                    mp.sources.size()));
            const uint8_t* win = nullptr;
            rc = get_win_sets(b, param, st, &win);
            if (rc != LLE_OK) return rc;
            // Window sets cut a third of the lane kernel's vector instructions (2 464 -> 1 876 per wavefront at 7 x 7 on level 6) but cost LDS --
            // 24 B per cell against a bitmap of a few hundred bytes -- and whether that drops a workgroup per CU depends on the map, the
            // window and the environments per batch together (profiles/r05_partial.md).  So the first call for a window size times the four
            // neighbours of the rule on THIS batch (sets or bitmap x the rule's E or half of it; identical bytes, a few hundred microseconds
            // once, one synchronisation of `stream`) and later calls launch the winner.  The LLE_PARTIAL_* overrides switch the trial off.
            lle_batch::PartialChoice& ch = b->partial_choice[param & 15];
            const Tuning& tn = tuning();
            const bool overridden = tn.partial_e || tn.partial_kernel || tn.partial_project >= 0 || tn.partial_batches || getenv("LLE_PARTIAL_NO_SETS") || getenv("LLE_PARTIAL_NO_TRIAL");
            if (!ch.decided && win && !overridden && b->n_envs >= 4096) {
                uint32_t rule = 0;
                HIP_TRY(launch_partial_observe(h, b->ptrs, static_cast<int8_t*>(out_dev), param, b->n_envs, b->per_env_sources, M, n_entities, reverse, st, nullptr, 0, &rule, b->obs_et));
                hipEvent_t e0, e1;
                HIP_TRY(hipEventCreate(&e0));
                if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(LLE_ERR_HIP, "hipEventCreate"); }
                float best = 0.f;
                hipError_t err = hipSuccess;
                for (int v = 0; v < 4 && rule && err == hipSuccess; v++) {
                    const bool sets = (v & 1) != 0;
                    const uint32_t E = (v & 2) ? rule / 2u : rule;
                    if (!E) continue;
                    for (int i = 0; i < 6 && err == hipSuccess; i++) {  // two warm-ups, four timed
                        if (i == 2) err = hipEventRecord(e0, st);
                        if (err == hipSuccess)
                            err = launch_partial_observe(h, b->ptrs, static_cast<int8_t*>(out_dev), param, b->n_envs, b->per_env_sources, M, n_entities, reverse, st, sets ? win : nullptr, E, nullptr, b->obs_et);
                    }
                    float ms = 0.f;
                    if (err == hipSuccess) err = hipEventRecord(e1, st);
                    if (err == hipSuccess) err = hipEventSynchronize(e1);
                    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
                    if (err == hipSuccess && (!ch.decided || ms < best)) { best = ms; ch.decided = 1; ch.use_sets = sets ? 1 : 0; ch.E = (uint8_t)E; }
                }
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                if (err != hipSuccess) return fail(LLE_ERR_HIP, std::string("partial observer trial: ") + hipGetErrorString(err));
            }
            const bool use = ch.decided ? ch.use_sets != 0 : false;  // (undecided -- small batches, overrides: the bitmap form and the rule's E, as before round 5; LLE_PARTIAL_SETS=1 forces the sets)
            const bool force_sets = getenv("LLE_PARTIAL_SETS") != nullptr;
            HIP_TRY(launch_partial_observe(h, b->ptrs, static_cast<int8_t*>(out_dev), param, b->n_envs, b->per_env_sources, M, n_entities, reverse, st,
                                           (use || force_sets) ? win : nullptr, ch.decided ? ch.E : 0u, nullptr, b->obs_et));
            break;
        }
        default:
            HIP_TRY(launch_state_observe(h, b->ptrs, static_cast<float*>(out_dev), kind == LLE_OBS_NORMALIZED_STATE, b->n_envs, st));
            break;
    }
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_batch_available_actions(lle_batch* b, int walkable_lasers, uint8_t* out_dev, void* stream) {
    if (!b || !out_dev) return fail(LLE_ERR_NULL, "NULL argument");
    ON_DEVICE_OF(b);
    const MapSel M{b->envs_per_map, (uint32_t)b->layout.table_stride, 0u};
    HIP_TRY(launch_avail(b->hdr, b->ptrs, out_dev, walkable_lasers, b->n_envs, b->per_env_sources, M, (hipStream_t)stream));
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_batch_env_outputs(lle_batch* b, const lle_env_outputs* out, void* stream) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    if (out->reward_kind != 0 && out->reward_kind != 1) return fail(LLE_ERR_ARG, "reward_kind must be 0 (single objective) or 1 (multi objective)");
    ON_DEVICE_OF(b);
    const MapSel M{b->envs_per_map, (uint32_t)b->layout.table_stride, 0u};
    EnvOutputs O{};
    O.state = out->state; O.reward = out->reward; O.done = out->done; O.available = out->available; O.alive = out->alive; O.arrived = out->arrived;
    O.normalize_state = out->normalize_state ? 1 : 0; O.reward_kind = (uint8_t)out->reward_kind; O.walkable_lasers = out->walkable_lasers ? 1 : 0;
    O.per_env_sources = b->per_env_sources ? 1 : 0;
    if (out->partial) return fail(LLE_ERR_ARG, "lle_env_outputs.partial is written by lle_batch_step_outputs only (here: lle_batch_observe_as)");
    HIP_TRY(launch_env_outputs(b->hdr, b->ptrs, O, b->n_envs, M, (hipStream_t)stream));
    g_status = LLE_OK;
    return LLE_OK;
}

// A live batch takes another compilation of one of its maps: the sources' colours / flags (lle_map_set_source) and the
// exits (lle_map_set_exits) may differ, nothing else.  `broadcast`: with per-environment sources every env takes the map's
// sources (lle_batch_update_sources); otherwise every env keeps its own and only its reset state is recomputed.
static int push_map(lle_batch* b, int map_index, const lle_map* map, bool broadcast, void* stream) {
    if (!b || !map) return fail(LLE_ERR_NULL, "NULL argument");
    if (map_index < 0 || map_index >= (int)b->maps.size()) return fail(LLE_ERR_ARG, "map_index out of range");
    const MapHeader& nh = map->m.header;
    const MapHeader& oh = b->maps[(size_t)map_index].header;
    if (nh.H != oh.H || nh.W != oh.W || nh.A != oh.A || nh.L != oh.L || nh.G != oh.G) return fail(LLE_ERR_ARG, "map does not match the batch");
    // the rows of LLE_BUF_OBS keep the pitch the batch was created with (lle_map_set_row_align after lle_batch_create)
    if (nh.obs_stride != oh.obs_stride || nh.n_chunks != oh.n_chunks)
        return fail(LLE_ERR_ARG, "the map's row alignment differs from the batch's (lle_map_set_row_align: a live batch keeps the pitch it was created with)");
    if (nh.blob_capacity != oh.blob_capacity || nh.blob_bytes > nh.blob_capacity || nh.ext_bytes != oh.ext_bytes)
        return fail(LLE_ERR_ARG, "map does not match the batch");
    for (size_t s = 0; s < map->m.sources.size(); s++) {
        const Source &a = map->m.sources[s], &o = b->maps[(size_t)map_index].sources[s];
        if (!(a.pos == o.pos) || a.direction != o.direction || a.beam.size() != o.beam.size()) return fail(LLE_ERR_ARG, "map does not match the batch (another map's sources)");
    }
    if (map->m.kind.size() != b->maps[(size_t)map_index].kind.size()) return fail(LLE_ERR_ARG, "map does not match the batch");
    for (size_t c = 0; c < map->m.kind.size(); c++) {
        // walls, gems and sources stay where they are; floor <-> exit may swap anywhere (World::set_exit_positions, world.rs:195-234);
        // a void takes part only UNDER A BEAM, where Laser::set_tile (laser.rs:109-115) replaces the innermost tile whatever it is --
        // a plain void that turned into floor (or the reverse) is another map: live agents would stand on a void without a death event
        const uint8_t a = map->m.kind[c], o = b->maps[(size_t)map_index].kind[c];
        if (a == o) continue;
        const bool fe = (a == K_FLOOR || a == K_EXIT) && (o == K_FLOOR || o == K_EXIT);
        const bool under_beam = !map->m.cell_layers[c].empty() && !b->maps[(size_t)map_index].cell_layers[c].empty();
        const bool with_void = under_beam && (a == K_FLOOR || a == K_EXIT || a == K_VOID) && (o == K_FLOOR || o == K_EXIT || o == K_VOID);
        if (!fe && !with_void) return fail(LLE_ERR_ARG, "map does not match the batch (another map's tiles)");
    }
    // several maps: the common header (largest table sizes) is worked out BEFORE anything is uploaded or replaced, so that a
    // refusal leaves host and device as they were
    MapHeader new_common = nh;
    uint32_t new_worst = b->worst_table_bytes;
    if (b->maps.size() > 1) {
        std::vector<lle_map> tmp(b->maps.size());
        std::vector<const lle_map*> ptrs;
        for (size_t m = 0; m < b->maps.size(); m++) { tmp[m].m = (int)m == map_index ? map->m : b->maps[m]; ptrs.push_back(&tmp[m]); }
        int rc = common_header(ptrs.data(), (int)ptrs.size(), &new_common, &new_worst);
        if (rc != LLE_OK) return rc;
    }
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    K.old_enabled = oh.enabled_mask;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(b->arena + b->layout.off_tables + (int64_t)map_index * b->layout.table_stride, map->m.blob.data(),
                           map->m.blob.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    b->maps[(size_t)map_index] = map->m;
    b->hdr = new_common;
    b->worst_table_bytes = new_worst;
    drop_views(b);  // their tables depend on the colours and the exits (the stream is idle: synchronised above)
    for (auto& c : b->partial_choice) c = lle_batch::PartialChoice();
    int rc = refresh_init_record(b, stream);
    if (rc != LLE_OK) return rc;
    if (b->per_env_sources) {
        // every env's own reset state follows the new tables; broadcast: and every env takes the map's sources
        // (enable / disable applied per env, where the flag changes)
        if (broadcast) K.flags = LAUNCH_FILL_DEFAULTS;
        return launch(b, KMODE_ENV_SOURCES, K, stream);
    }
    // one map: applies enable / disable where a flag changed; both rewrite the observation
    return launch(b, b->maps.size() == 1 ? KMODE_SOURCES : KMODE_OBSERVE, K, stream);
}

int lle_batch_update_sources(lle_batch* b, const lle_map* map, void* stream) {
    if (!b || !map) return fail(LLE_ERR_NULL, "NULL argument");
    if (b->maps.size() != 1)
        return fail(LLE_ERR_UNSUPPORTED, "lle_batch_update_sources serves one-map batches; use lle_batch_set_sources per environment");
    return push_map(b, 0, map, true, stream);
}

int lle_batch_update_map(lle_batch* b, int map_index, const lle_map* map, void* stream) {
    if (!b || !map) return fail(LLE_ERR_NULL, "NULL argument");
    if (map_index < 0 || map_index >= (int)b->maps.size()) return fail(LLE_ERR_ARG, "map_index out of range");
    // per-environment sources are kept: the map's source colours / flags must then be the ones the batch was last told about;
    // likewise in a batch of several maps (the enable / disable pass below compares ONE pair of masks for every env)
    if (b->per_env_sources || b->maps.size() > 1) {
        const Map& cur = b->maps[(size_t)map_index];
        for (size_t s = 0; s < cur.sources.size() && s < map->m.sources.size(); s++)
            if (cur.sources[s].agent_id != map->m.sources[s].agent_id || cur.sources[s].enabled != map->m.sources[s].enabled)
                return fail(LLE_ERR_ARG, "lle_batch_update_map: with per-environment sources or several maps the map's source colours / flags must be "
                                         "unchanged (use lle_batch_update_sources to broadcast, lle_batch_set_sources per environment)");
    }
    return push_map(b, map_index, map, false, stream);
}

int lle_batch_stats(lle_batch* b, int64_t out[8], int reset_counters, void* stream) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    ON_DEVICE_OF(b);
    hipStream_t st = (hipStream_t)stream;
    int64_t* sum = reinterpret_cast<int64_t*>(b->arena + b->layout.off_stats_sum);
    int rc = capi_batch_stats_to_device(b, sum, reset_counters, stream);
    if (rc != LLE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, sum, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_batch_kernel_info(const lle_batch* b, char* name_buf, size_t cap, int32_t* lds_bytes, int32_t* envs_per_wave) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    if (name_buf && cap) {
        if (b->lane_per_env_step) std::snprintf(name_buf, cap, "%s", kernel_variant_name(kernel_variant((int)b->hdr.A, (int)b->hdr.L)));
        else std::snprintf(name_buf, cap, "step_kernel<%d,%d>", step_group((int)b->hdr.A), step_lm((int)b->hdr.L));
    }
    if (lds_bytes) {
        const bool pes = b->per_env_sources;
        if (!b->lane_per_env_step && step_splits_rows(b->hdr, pes, b->tune)) {
            const uint32_t cap = 64u / (uint32_t)step_group((int)b->hdr.A), e = batch_step_epw(b, b->tune);
            *lds_bytes = (int32_t)split_lds_bytes(b->hdr, 4, e < cap ? e : cap);
        } else {
            *lds_bytes = (int32_t)kernel_lds_bytes(b->hdr, kernel_waves_per_wg(b->hdr, pes), pes);
        }
    }
    if (envs_per_wave) *envs_per_wave = b->lane_per_env_step ? (int32_t)b->envs_per_wave : (int32_t)batch_step_epw(b, b->tune);
    return LLE_OK;
}

// Profiling aid (not part of the stable ABI surface documented in the header's main section): one step with
// per-wave s_memrealtime stamps written to `stamps_dev` ([n_blocks][8] u64).
int lle_batch_step_stamped(lle_batch* b, uint32_t flags, uint64_t seed, uint64_t t, uint64_t* stamps_dev, void* stream) {
    if (!b || !stamps_dev) return fail(LLE_ERR_NULL, "NULL argument");
    ON_DEVICE_OF(b);
    LaunchArgs K{};
    K.flags = flags & STEP_PUBLIC_FLAGS; K.seed = seed; K.t = t; K.stamps = stamps_dev;
    return launch(b, KMODE_STEP, K, stream);
}

int lle_batch_probe_row_fill(lle_batch* b, uint32_t value, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    ON_DEVICE_OF(b);
    const uint32_t epw = b->lane_per_env_step ? b->envs_per_wave : batch_step_epw(b, b->tune);
    // the same alternation as the step launches it stands in for
    const bool reverse = next_walk_reversed(b, b->ptrs.obs, b->rows_bytes());
    HIP_TRY(launch_row_fill_probe(b->ptrs.obs, b->n_envs, (uint32_t)b->row_pitch(), epw, value, reverse,
                                  rotate_rows_pays(b->tune, b->rows_bytes(), (uint32_t)b->row_pitch()), (hipStream_t)stream));
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_probe_fill_rows(void* out_dev, int64_t n_rows, int64_t row_bytes, int rows_per_wave, void* stream) {
    if (!out_dev) return fail(LLE_ERR_NULL, "NULL argument");
    if (n_rows <= 0 || row_bytes <= 0 || row_bytes % 16 != 0 || row_bytes > (1 << 24) || (reinterpret_cast<uintptr_t>(out_dev) % 16) != 0)
        return fail(LLE_ERR_ARG, "rows of a positive multiple of 16 bytes, 16-byte aligned");
    if (rows_per_wave < 1 || rows_per_wave > 64 || (rows_per_wave & (rows_per_wave - 1))) return fail(LLE_ERR_ARG, "rows_per_wave must be a power of two, 1..64");
    const uint64_t bytes = (uint64_t)n_rows * (uint64_t)row_bytes;
    HIP_TRY(launch_row_fill_probe(static_cast<int8_t*>(out_dev), n_rows, (uint32_t)row_bytes, (uint32_t)rows_per_wave, 0u, false,
                                  rotate_rows_pays(StepTune(), bytes, (uint32_t)row_bytes), (hipStream_t)stream));
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_probe_read_rows(const void* rows_dev, void* out_f16_dev, int64_t bytes, void* stream) {
    if (!rows_dev || !out_f16_dev) return fail(LLE_ERR_NULL, "NULL argument");
    if (bytes <= 0 || bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(rows_dev) % 16) != 0 || (reinterpret_cast<uintptr_t>(out_f16_dev) % 16) != 0)
        return fail(LLE_ERR_ARG, "bytes must be a positive multiple of 16 and both buffers 16-byte aligned");
    HIP_TRY(launch_cast_rows(static_cast<const int8_t*>(rows_dev), out_f16_dev, bytes, (hipStream_t)stream));
    g_status = LLE_OK;
    return LLE_OK;
}

void lle_tuning_refresh(void) { tuning_refresh(); }

size_t lle_debug_launched(char* buf, size_t cap) { return debug_list(false, buf, cap); }
size_t lle_debug_reachable(char* buf, size_t cap) { return debug_list(true, buf, cap); }
void lle_debug_reset_launched(void) { debug_reset_launched(); }

// One timed trial of the batch's plain single step (sampled actions + auto-reset) under `t`: us per launch by HIP events.
static int time_step_trial(lle_batch* b, const StepTune& t, int launches, hipStream_t st, hipEvent_t e0, hipEvent_t e1, uint64_t* t_idx, double* us) {
    const StepTune keep = b->tune;
    b->tune = t;
    int rc = LLE_OK;
    auto one = [&]() {
        LaunchArgs K{};
        K.flags = STEP_SAMPLE_ACTIONS | STEP_AUTO_RESET; K.seed = 0x7E57ull; K.t = (*t_idx)++;
        return launch(b, KMODE_STEP, K, st);
    };
    for (int i = 0; i < 4 && rc == LLE_OK; i++) rc = one();  // (an even count: the alternating walk keeps its phase)
    if (rc == LLE_OK && hipEventRecord(e0, st) != hipSuccess) rc = fail(LLE_ERR_HIP, "hipEventRecord");
    for (int i = 0; i < launches && rc == LLE_OK; i++) rc = one();
    if (rc == LLE_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(LLE_ERR_HIP, "hipEventSynchronize");
    float ms = 0.f;
    if (rc == LLE_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(LLE_ERR_HIP, "hipEventElapsedTime");
    *us = (double)ms * 1e3 / launches;
    b->tune = keep;
    return rc;
}

int lle_batch_autotune(lle_batch* b, double budget_ms, void* stream) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    if (b->lane_per_env_step) return fail(LLE_ERR_ARG, "lle_batch_autotune tunes the default step kernel (lle_batch_set_envs_per_wave selected the diagnostic one)");
    if (!(budget_ms > 0.0)) budget_ms = 20.0;
    ON_DEVICE_OF(b);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(LLE_ERR_HIP, "hipEventCreate"); }
    const bool pes = b->per_env_sources;
    const MapHeader& h = b->hdr;
    const uint64_t row_bytes = b->rows_bytes();
    // the alternatives, one coordinate at a time (each keeps the best of the ones before it): environments per wavefront, row
    // heads, store policy, split rows, the alternating walk -- only those that exist for this batch
    const uint32_t cap = 64u / (uint32_t)step_group((int)h.A);
    // environments per wavefront: the default rule's choice and its two neighbours (the rule already shrinks the wavefronts of small batches:
    // sweeping {cap, cap / 2, cap / 4} kept a one-agent map's 4 096-env batch at 16 environments per wavefront -- 64 workgroups for 256 CUs --
    // where the rule says 4; round 5).  Below MIN_ENVS_PER_WAVE only while the wavefronts still fit LLE_BUF_STATS' slots (step_envs_per_wave).
    std::vector<uint32_t> epws;
    {
        const uint32_t rule = batch_step_epw(b, StepTune());
        for (uint32_t e : {rule * 2u, rule, rule / 2u})
            if (e >= 1 && e <= cap && (e >= MIN_ENVS_PER_WAVE || (b->n_envs + e - 1) / e <= (int64_t)MIN_STAT_SLOTS)) epws.push_back(e);
    }
    const bool can_heads = step_has_row_heads(h, pes) && b->obs_et == OBS_I8, can_split = step_can_split_rows(h, pes), can_walk = row_bytes > (256ull << 20);
    const int n_trials = (int)epws.size() + (can_heads ? 4 : 0) + 2 + (can_split ? 2 : 0) + (can_walk ? 2 : 0) + 2;
    uint64_t t_idx = 1u << 20;
    StepTune best = b->tune;
    double probe_us = 0.0;
    int rc = time_step_trial(b, best, 4, st, e0, e1, &t_idx, &probe_us);  // how long a launch is: sizes the trials
    const double per_trial_us = budget_ms * 1e3 / n_trials;
    int launches = probe_us > 0 ? (int)(per_trial_us / probe_us) : 16;
    launches = launches < 6 ? 6 : (launches > 400 ? 400 : launches);
    launches &= ~1;
    char line[256];
    std::string log;
    // `dflt`: what the default rule picks for this coordinate.  An alternative replaces it only when it measured at least 3 % faster: a trial
    // is a few dozen launches, and a pick inside the noise can cost more than it gains (round 5, under the profiler: head_group 2 "won" by
    // 21.82 against 21.87 us and ran the headline 8 % slower than the default).
    auto sweep = [&](const char* what, std::vector<int> values, int dflt, auto&& set) {
        if (rc != LLE_OK || values.size() < 2) return;
        double best_us = 0.0, dflt_us = 0.0;
        int best_v = values[0];
        std::snprintf(line, sizeof line, "%s:", what);
        log += line;
        for (size_t i = 0; i < values.size() && rc == LLE_OK; i++) {
            StepTune t = best;
            set(t, values[i]);
            double us = 0.0;
            rc = time_step_trial(b, t, launches, st, e0, e1, &t_idx, &us);
            std::snprintf(line, sizeof line, " %d=%.2fus", values[i], us);
            log += line;
            if (i == 0 || us < best_us) { best_us = us; best_v = values[i]; }
            if (values[i] == dflt) dflt_us = us;
        }
        if (dflt_us > 0.0 && best_v != dflt && best_us > 0.97 * dflt_us) best_v = dflt;  // (not clearly better: keep the rule)
        set(best, best_v);
        std::snprintf(line, sizeof line, " -> %d; ", best_v);
        log += line;
    };
    // the default rules' answers for this batch (lle_batch_tuning with nothing tuned)
    lle_tuning_info rules{};
    {
        const StepTune keep = b->tune;
        const std::string keep_log = b->tune_log;
        b->tune = StepTune();
        (void)lle_batch_tuning(b, &rules, nullptr, 0);
        b->tune = keep;
        b->tune_log = keep_log;
    }
    sweep("envs_per_wave", std::vector<int>(epws.begin(), epws.end()), rules.envs_per_wave, [](StepTune& t, int v) { t.epw = (uint8_t)v; });
    if (can_heads) {
        // (the rule's answer depends on the number of wavefronts, i.e. on the environments per wavefront just chosen)
        const uint32_t nw = (uint32_t)((b->n_envs + (best.epw ? best.epw : rules.envs_per_wave) - 1) / (best.epw ? best.epw : rules.envs_per_wave));
        const bool general = pes || b->envs_per_map != 0;
        const int heads_rule = (general ? nw >= 2048u : (nw >= 2048u && nw <= 12288u)) ? 1 : 0;
        sweep("row_heads", {0, 1}, heads_rule, [](StepTune& t, int v) { t.heads = (int8_t)v; });
    }
    if (can_heads && best.heads == 1) sweep("head_group", {1, 2}, 1, [](StepTune& t, int v) { t.head_group = (int8_t)v; });
    sweep("write_through", {0, 1}, rules.write_through, [](StepTune& t, int v) { t.write_through = (int8_t)v; });
    if (can_split) sweep("split_rows", {0, 1}, rules.split_rows, [](StepTune& t, int v) { t.split = (int8_t)v; });
    if (can_walk) sweep("alternating_walk", {0, 1}, rules.alternating_walk, [](StepTune& t, int v) { t.walk = (int8_t)v; });
    if (!can_split || !step_splits_rows(h, pes, best)) sweep("rotate_rows", {0, 1}, rules.rotate_rows, [](StepTune& t, int v) { t.rotate = (int8_t)v; });
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != LLE_OK) return rc;
    b->tune = best;
    std::snprintf(line, sizeof line, "%d launches per trial, %d trials", launches, n_trials);
    b->tune_log = log + line;
    // the trials were real steps: back to World.reset of every env, counters at zero
    HIP_TRY(hipMemsetAsync(b->ptrs.stats, 0, (size_t)b->layout.n_stat_blocks * 64, st));
    LaunchArgs K{};
    return launch(b, KMODE_RESET, K, stream);
}

int lle_batch_tuning(const lle_batch* b, lle_tuning_info* out, char* log_buf, size_t cap) {
    if (!b || !out) return fail(LLE_ERR_NULL, "NULL argument");
    const bool pes = b->per_env_sources;
    const MapHeader& h = b->hdr;
    const uint64_t row_bytes = b->rows_bytes();
    const uint32_t epw = batch_step_epw(b, b->tune);
    const uint32_t n_waves = (uint32_t)((b->n_envs + epw - 1) / epw);
    out->envs_per_wave = (int32_t)epw;
    out->split_rows = step_splits_rows(h, pes, b->tune) ? 1 : 0;
    out->write_through = write_through_pays(row_bytes, (uint32_t)b->row_pitch(), b->tune.write_through) ? 1 : 0;
    out->alternating_walk = pingpong_pays(b, row_bytes) ? 1 : 0;
    out->rotate_rows = (!out->split_rows && rotate_rows_pays(b->tune, row_bytes, (uint32_t)b->row_pitch())) ? 1 : 0;
    // (what a plain single step of this batch gets: the same conditions as launch_step_kernel)
    const bool general = pes || b->envs_per_map != 0;
    StepTune t = b->tune;
    int heads = 0;
    if (step_has_row_heads(h, pes) && !out->split_rows && b->obs_et == OBS_I8) {
        const int forced = tuning().row_heads >= 0 ? tuning().row_heads : (int)t.heads;
        heads = forced >= 0 ? forced : ((general ? n_waves >= 2048u : (n_waves >= 2048u && n_waves <= 12288u)) ? 1 : 0);
    }
    out->row_heads = heads;
    out->head_group = heads ? (tuning().head_group ? tuning().head_group : (t.head_group ? (int)t.head_group : 1)) : 1;
    out->autotuned = b->tune_log.empty() ? 0 : 1;
    if (log_buf && cap) std::snprintf(log_buf, cap, "%s", b->tune_log.c_str());
    g_status = LLE_OK;
    return LLE_OK;
}

int lle_batch_set_envs_per_wave(lle_batch* b, int epw) {
    if (!b) return fail(LLE_ERR_NULL, "NULL batch");
    if (epw < (int)MIN_ENVS_PER_WAVE || epw > 64 || (epw & (epw - 1))) return fail(LLE_ERR_ARG, "envs_per_wave must be 8, 16, 32 or 64");
    b->envs_per_wave = (uint32_t)epw;
    b->lane_per_env_step = true;  // the knob belongs to the lane-per-env step kernel
    return LLE_OK;
}

}  // extern "C"
