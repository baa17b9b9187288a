// map_compile.cpp -- see map_compile.hpp.
#include "map_compile.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <map>

#include "../../include/lle_hip.h"

namespace lle {

// The six built-in levels (data of the reference's resources/levels/lvl1..6, whitespace-normalised;
// src/core/levels.rs:1-8).  All are 12 rows x 13 columns.
const char* const LEVEL_TEXT[6] = {
    // level 1
    ". . . . . . . S0 . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . G . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . X . . . . .\n"
    ". . . . . . . . . . . . .\n",
    // level 2
    ". . . . . . S1 S0 . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . G . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . X X . . . . .\n"
    ". . . . . . . . . . . . .\n",
    // level 3
    ". . . . . . . S0 S1 . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    "L0E . . . . . . . . . . . .\n"
    ". . . . . . . . . . G . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . X X . . . .\n"
    ". . . . . . . . . . . . .\n",
    // level 4
    ". . . . . . . S0 S1 . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    "L0E . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . L1W\n"
    ". . . . . . . . . . G . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . X X . . . .\n"
    ". . . . . . . . . . . . .\n",
    // level 5
    "G . L2S . S0 S1 S2 S3 . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    "@ @ . . . . . . . . . . .\n"
    ". . . . . . . @ @ @ @ @ @\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . L1W\n"
    ". . . . . . . @ G . . . .\n"
    ". G . . . . . @ @ @ @ @ @\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . X X . .\n"
    ". . . . G . . . . X X . G\n",
    // level 6
    "G L2S . . S0 S1 S2 S3 . . . . .\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . .\n"
    "@ @ . . . . . . . . . . .\n"
    "L0E . . . . . . @ @ @ @ @ @\n"
    ". . . . . . . . . . . . .\n"
    ". . . . . . . . . . . . L1W\n"
    ". . . . . . . @ . G . . .\n"
    ". . . . . . . @ @ @ @ @ @\n"
    ". . . . . . . . . . . . .\n"
    ". . . . G . . . . . X X .\n"
    ". . . . . . . . . . X X G\n",
};

namespace {

const int DIR_DELTA[4][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}};  // N E S W (direction.rs:20-27)

// Rust `str::parse::<usize>()`
bool parse_usize(const std::string& s, int& out) {
    size_t k = 0;
    if (k < s.size() && s[k] == '+') k++;
    if (k >= s.size()) return false;
    long v = 0;
    for (; k < s.size(); k++) {
        if (!std::isdigit((unsigned char)s[k])) return false;
        v = v * 10 + (s[k] - '0');
        if (v > 100000) return false;
    }
    out = (int)v;
    return true;
}

std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out;
    size_t k = 0;
    while (k < line.size()) {
        while (k < line.size() && std::isspace((unsigned char)line[k])) k++;
        size_t s = k;
        while (k < line.size() && !std::isspace((unsigned char)line[k])) k++;
        if (k > s) out.push_back(line.substr(s, k - s));
    }
    return out;
}

}  // namespace

int parse_map(const char* text, size_t len, Map& m) {
    std::string all(text, len);
    if (all.find('=') != std::string::npos) return LLE_PARSE_TOML_UNSUPPORTED;

    // ---- tokenise (parser_v1.rs:132-175)
    struct RawSource { Pos pos; int dir; int agent; };
    std::vector<RawSource> raw_sources;
    int height = 0, width = -1;
    size_t p = 0;
    while (p <= all.size()) {
        size_t e = all.find('\n', p);
        if (e == std::string::npos) e = all.size();
        std::vector<std::string> toks = split_ws(all.substr(p, e - p));
        p = e + 1;
        if (toks.empty()) continue;
        for (int col = 0; col < (int)toks.size(); col++) {
            const std::string& t = toks[col];
            Pos pos{height, col};
            switch (std::toupper((unsigned char)t[0])) {
                case '.': break;
                case 'G': m.gems.push_back(pos); break;
                case '@': m.walls.push_back(pos); break;
                case 'X': m.exits.push_back(pos); break;
                case 'V': m.voids.push_back(pos); break;
                case 'S': {
                    int id;
                    if (!parse_usize(t.substr(1), id)) return LLE_PARSE_INVALID_AGENT_ID;
                    if (id >= 4096) return LLE_PARSE_LIMIT;
                    if ((int)m.starts.size() <= id) m.starts.resize(id + 1);
                    if (!m.starts[id].empty()) return LLE_PARSE_DUPLICATE_START_TILE;
                    m.starts[id].push_back(pos);
                    break;
                }
                case 'L': {
                    int dir;
                    switch (std::tolower((unsigned char)t.back())) {
                        case 'n': dir = 0; break;
                        case 'e': dir = 1; break;
                        case 's': dir = 2; break;
                        case 'w': dir = 3; break;
                        default: return LLE_PARSE_INVALID_DIRECTION;  // the reference panics (laser_config.rs:22)
                    }
                    int agent;
                    if (t.size() < 2 || !parse_usize(t.substr(1, t.size() - 2), agent)) return LLE_PARSE_INVALID_AGENT_ID;
                    raw_sources.push_back({pos, dir, agent});
                    m.walls.push_back(pos);
                    break;
                }
                default: return LLE_PARSE_INVALID_TILE;
            }
        }
        if (width < 0) width = (int)toks.size();
        else if (width != (int)toks.size()) return LLE_PARSE_INCONSISTENT_DIMENSIONS;
        height++;
    }
    if (height == 0) return LLE_PARSE_EMPTY_WORLD;
    m.H = height;
    m.W = width;

    // ---- pre_validate (world_config.rs:124-147)
    if (m.starts.empty()) return LLE_PARSE_NO_AGENTS;
    if (m.exits.size() < m.starts.size()) return LLE_PARSE_NOT_ENOUGH_EXIT_TILES;

    // ---- make_grid (world_config.rs:176-199): later writes win: gems, exits, voids, walls
    const int HW = m.H * m.W;
    m.kind.assign(HW, K_FLOOR);
    for (auto& q : m.gems) m.kind[q.i * m.W + q.j] = K_GEM;
    for (auto& q : m.exits) m.kind[q.i * m.W + q.j] = K_EXIT;
    for (auto& q : m.voids) m.kind[q.i * m.W + q.j] = K_VOID;
    for (auto& q : m.walls) m.kind[q.i * m.W + q.j] = K_WALL;
    m.gem_index.assign(HW, -1);
    for (size_t g = 0; g < m.gems.size(); g++) m.gem_index[m.gems[g].i * m.W + m.gems[g].j] = (int)g;

    // ---- laser_setup (world_config.rs:203-250).  Sources in parse order; each beam stops at the first
    // non-walkable tile (Wall or an already placed LaserSource; later sources are still Walls at this point).
    m.cell_layers.assign(HW, {});
    for (size_t s = 0; s < raw_sources.size(); s++) {
        const RawSource& rs = raw_sources[s];
        Source src;
        src.pos = rs.pos; src.direction = rs.dir; src.agent_id = rs.agent; src.enabled = true; src.laser_id = (int)s;
        int i = rs.pos.i + DIR_DELTA[rs.dir][0], j = rs.pos.j + DIR_DELTA[rs.dir][1];
        while (i >= 0 && j >= 0 && i < m.H && j < m.W) {
            uint8_t k = m.kind[i * m.W + j];
            if (k == K_WALL || k == K_SOURCE) break;
            src.beam.push_back({i, j});
            i += DIR_DELTA[rs.dir][0];
            j += DIR_DELTA[rs.dir][1];
        }
        bool is_blocked = false;
        for (size_t k = 0; k < src.beam.size(); k++) {
            const Pos bp = src.beam[k];
            if (src.agent_id < (int)m.starts.size() && m.starts[src.agent_id].size() == 1 && m.starts[src.agent_id][0] == bp)
                is_blocked = true;
            // the new Laser wraps whatever is there: it becomes the OUTERMOST layer
            auto& layers = m.cell_layers[bp.i * m.W + bp.j];
            layers.insert(layers.begin(), CellLayer{(int)s, (int)k});
            if (!is_blocked) {
                for (size_t a = 0; a < m.starts.size(); a++) {
                    if ((int)a == src.agent_id) continue;
                    auto& st = m.starts[a];
                    st.erase(std::remove(st.begin(), st.end(), bp), st.end());
                }
            }
        }
        m.kind[rs.pos.i * m.W + rs.pos.j] = K_SOURCE;
        m.sources.push_back(std::move(src));
    }

    // ---- post_validate (world_config.rs:149-168)
    size_t total = 0;
    for (auto& st : m.starts) {
        if (st.empty()) return LLE_PARSE_AGENT_WITHOUT_START;
        total += st.size();
    }
    if (total < m.starts.size()) return LLE_PARSE_NOT_ENOUGH_START_TILES;

    // ---- static limits of the kernels
    if (m.H > LLE_MAX_DIM || m.W > LLE_MAX_DIM || m.n_agents() > LLE_MAX_AGENTS || (int)m.gems.size() > LLE_MAX_GEMS ||
        (int)m.sources.size() > LLE_MAX_SOURCES)
        return LLE_PARSE_LIMIT;
    for (auto& s : m.sources)
        if ((int)s.beam.size() > LLE_MAX_BEAM_LEN) return LLE_PARSE_LIMIT;
    if (!m.layout_words()) return LLE_PARSE_LIMIT;  // more than LLE_MAX_BEAM_WORDS 32-cell words over all beams
    for (auto& l : m.cell_layers)
        if ((int)l.size() > MAX_CELL_LAYERS) return LLE_PARSE_LIMIT;  // impossible: one beam per travel direction
    if ((int64_t)m.n_layers() * HW >= (1 << 20)) return LLE_PARSE_LIMIT;  // 16-bit chunk ids, 20-bit byte indices

    m.compile();
    return LLE_PARSE_OK;
}

bool Map::layout_words() {
    source_word.clear();
    word_source.clear();
    chain_mask = 0;
    bool chained = false;
    for (size_t s = 0; s < sources.size(); s++) {
        const int words = std::max(1, ((int)sources[s].beam.size() + MAX_BEAM_LEN - 1) / MAX_BEAM_LEN);
        source_word.push_back((int)word_source.size());
        for (int w = 0; w < words; w++) {
            if (w > 0 && word_source.size() < 32) chain_mask |= 1u << word_source.size();
            word_source.push_back((int)s);
        }
        chained = chained || words > 1;
    }
    // a chained map takes the kernels' LDS-record form of the beam masks (more than four words), the only one that walks chains
    while (chained && word_source.size() < 5) word_source.push_back(-1);
    return (int)word_source.size() <= MAX_SOURCES;
}

int Map::word_len(int b) const {
    const int s = word_source[(size_t)b];
    if (s < 0) return 0;
    const int first = (b - source_word[(size_t)s]) * MAX_BEAM_LEN, len = (int)sources[(size_t)s].beam.size();
    return std::max(0, std::min(MAX_BEAM_LEN, len - first));
}

int Map::n_laser_tiles() const {
    int n = 0;
    for (auto& l : cell_layers) n += (int)std::min<size_t>(l.size(), 2);
    return n;
}

// Static template + dynamic-byte table of a layered-style observation whose channels are given by `lm`
// (python/lle/observations.py:216-266; write order: WALL, VOID, EXIT, -1 at sources | lasers on, gems | agents).
// Returns false when some laser colour has no layer (the reference raises IndexError there).
bool Map::build_obs_tables(const LayerMap& lm, std::vector<int8_t>& tmpl, std::vector<uint64_t>& dyn_tab) const {
    const int HW = H * W, G = (int)gems.size();
    bool supported = true;
    const uint32_t obs_stride = row_pitch_of((uint32_t)(lm.C * HW));
    tmpl.assign(obs_stride, 0);
    auto at = [&](int layer, Pos q) -> int8_t& { return tmpl[(size_t)layer * HW + q.i * W + q.j]; };
    for (auto& q : walls) at(lm.wall, q) = 1;
    for (auto& q : voids) at(lm.void_, q) = 1;
    for (auto& q : exits) at(lm.exit, q) = 1;
    for (auto& s : sources) {
        if (s.agent_id >= lm.n_laser) { supported = false; continue; }
        at(lm.laser[s.agent_id], s.pos) = -1;
    }
    // dynamic bytes: laser tiles that World.lasers() exposes (outer layer + the one directly below,
    // world.rs:159-172), then uncollected gems
    struct Dyn { int n_refs = 0; uint32_t ref[2] = {0, 0}; uint32_t gem = NO_GEM; };
    std::map<uint32_t, Dyn> dyn;
    for (int c = 0; c < HW; c++) {
        const auto& layers = cell_layers[c];
        for (size_t k = 0; k < layers.size() && k < 2; k++) {
            const Source& src = sources[layers[k].laser_id];
            if (src.agent_id >= lm.n_laser) { supported = false; continue; }
            uint32_t idx = (uint32_t)(lm.laser[src.agent_id] * HW + c);
            Dyn& d = dyn[idx];
            d.ref[d.n_refs++] = ref_pack((uint32_t)word_of(layers[k].laser_id, layers[k].offset), (uint32_t)bit_of(layers[k].offset));
        }
    }
    for (int g = 0; g < G; g++) dyn[(uint32_t)(lm.gem * HW + gems[g].i * W + gems[g].j)].gem = (uint32_t)g;
    dyn_tab.clear();
    for (auto& kv : dyn) {
        const Dyn& d = kv.second;
        uint64_t e = dyn_pack(kv.first, (uint8_t)tmpl[kv.first], d.n_refs, d.ref[0], d.ref[1], d.gem);
        dyn_tab.push_back(e);
    }
    return supported;
}

// View blob of a layered-style observation with another channel layout (tables.h ViewHeader):
//   OBS_LAYERED_PADDED (param = padding p): LayeredPadded, observations.py:196-214 -- n_agents + p agent layers and as
//     many laser layers, so colours in [A, A+p) get a layer of their own instead of aliasing;
//   OBS_PERSPECTIVE (param = observer k): AgentZeroPerspective, observations.py:372-395 -- the Layered tensor with
//     layers A0 <-> A0+k and LASER_0 <-> LASER_0+k swapped (a permutation of the finished layers).
std::vector<uint8_t> Map::compile_view(int kind, int param) const {
    const int A = n_agents(), HW = H * W;
    LayerMap lm;
    if (kind == OBS_LAYERED_PADDED) {
        const int Ap = A + param;
        lm.C = 2 * Ap + 4;
        for (int a = 0; a < A; a++) lm.agent[a] = a;
        lm.n_laser = lm.C - Ap;
        for (int c = 0; c < lm.n_laser; c++) lm.laser[c] = Ap + c;
        lm.wall = 2 * Ap; lm.void_ = lm.wall + 1; lm.gem = lm.wall + 2; lm.exit = lm.wall + 3;
    } else {  // OBS_PERSPECTIVE (k = 0 is the plain layered tensor)
        const int k = param;
        lm.C = 2 * A + 4;
        auto sigma = [&](int l) { return l == 0 ? k : (l == k ? 0 : (l == A ? A + k : (l == A + k ? A : l))); };
        for (int a = 0; a < A; a++) lm.agent[a] = sigma(a);
        lm.n_laser = lm.C - A;
        for (int c = 0; c < lm.n_laser; c++) lm.laser[c] = sigma(A + c);
        lm.wall = sigma(2 * A); lm.void_ = sigma(2 * A + 1); lm.gem = sigma(2 * A + 2); lm.exit = sigma(2 * A + 3);
    }
    std::vector<int8_t> tmpl;
    std::vector<uint64_t> dyn_tab;
    ViewHeader v{};
    v.magic = VIEW_MAGIC;
    v.supported = build_obs_tables(lm, tmpl, dyn_tab) ? 1u : 0u;
    v.A = (uint32_t)A; v.L = (uint32_t)n_words(); v.H = (uint32_t)H; v.W = (uint32_t)W; v.HW = (uint32_t)HW;
    v.C = (uint32_t)lm.C;
    v.obs_bytes = (uint32_t)(lm.C * HW);
    v.obs_stride = row_pitch_of(v.obs_bytes);
    v.n_chunks = v.obs_stride / 16;
    v.D = (uint32_t)dyn_tab.size();
    for (int a = 0; a < A; a++) v.agent_layer[a] = (uint8_t)lm.agent[a];
    std::vector<int8_t> bare(v.obs_stride, 0);
    {
        auto bat = [&](int layer, Pos q) -> int8_t& { return bare[(size_t)layer * HW + q.i * W + q.j]; };
        for (auto& q : walls) bat(lm.wall, q) = 1;
        for (auto& q : voids) bat(lm.void_, q) = 1;
        for (auto& q : exits) bat(lm.exit, q) = 1;
    }
    v.gem_layer = (uint32_t)lm.gem;
    v.n_laser = (uint32_t)std::min(lm.n_laser, 48);
    for (uint32_t c = 0; c < v.n_laser; c++) v.laser_layer[c] = (uint8_t)lm.laser[c];
    size_t off = sizeof(ViewHeader);
    v.off_dyn = (uint32_t)off; off = (off + dyn_tab.size() * 8 + 15) & ~(size_t)15;
    v.off_template = (uint32_t)off; off += tmpl.size();
    v.off_bare = (uint32_t)off; off += bare.size();
    off = (off + 1023) & ~(size_t)1023;  // the kernel copies the whole blob to LDS in 1-KiB rows
    v.blob_bytes = (uint32_t)off;
    std::vector<uint8_t> out(off, 0);
    if (!dyn_tab.empty()) std::memcpy(out.data() + v.off_dyn, dyn_tab.data(), dyn_tab.size() * 8);
    std::memcpy(out.data() + v.off_template, tmpl.data(), tmpl.size());
    std::memcpy(out.data() + v.off_bare, bare.data(), bare.size());
    std::memcpy(out.data(), &v, sizeof v);
    return out;
}

void Map::compile() {
    const int A = n_agents(), G = (int)gems.size(), L = n_words(), C = n_layers(), HW = H * W;  // L: beam WORDS (tables.h)
    MapHeader h{};
    h.n_sources = (uint32_t)sources.size();
    h.chain_mask = chain_mask;
    for (int b = 0; b < L; b++) {
        h.word_source[b] = (uint8_t)std::max(word_source[(size_t)b], 0);
        if (word_source[(size_t)b] >= 0) h.word_mask |= 1u << b;
    }
    for (size_t s = 0; s < sources.size(); s++) h.source_word[s] = (uint8_t)source_word[s];
    h.magic = MAP_MAGIC;
    h.H = H; h.W = W; h.A = A; h.G = G; h.L = L; h.C = C; h.HW = HW;
    h.obs_bytes = (uint32_t)(C * HW);
    h.obs_stride = row_pitch_of(h.obs_bytes);
    h.n_chunks = h.obs_stride / 16;
    h.obs_supported = 1;
    for (int a = 0; a < A; a++) h.start[a] = (uint16_t)(starts[a][0].i | (starts[a][0].j << 8));
    for (int g = 0; g < G; g++) {
        h.gem_cell[g] = (uint16_t)(gems[g].i | (gems[g].j << 8));
        if (cell_layers[gems[g].i * W + gems[g].j].empty()) h.direct_gems |= 1u << g;
    }
    for (int b = 0; b < L; b++) {  // per word; the words of a chain carry their source's colour and flag
        if (word_source[(size_t)b] < 0) continue;  // (padding: empty, disabled)
        const Source& src = sources[(size_t)word_source[(size_t)b]];
        const int len = word_len(b);
        h.beam_len[b] = (uint8_t)len;
        h.beam_full[b] = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
        h.beam_colour[b] = (uint8_t)std::min(src.agent_id, (int)NO_COLOUR);
        if (src.enabled) h.enabled_mask |= 1u << b;
    }

    // which colours a source may take (pylaser_source.rs:121-139): its exposed tiles (outer layer and the one directly
    // below, world.rs:159-172) must not hold a possible start of an agent of another colour
    for (int b = 0; b < L; b++) {
        const int s = word_source[(size_t)b];
        if (s < 0) continue;
        uint32_t on_beam = 0;  // agents with a start on the exposed tiles of source s
        for (int a = 0; a < A; a++)
            for (const Pos& st : starts[a]) {
                const auto& layers = cell_layers[st.i * W + st.j];
                for (size_t k = 0; k < layers.size() && k < 2; k++)
                    if (layers[k].laser_id == s) on_beam |= 1u << a;
            }
        uint32_t ok = 0;
        for (int c = 0; c < A; c++)
            if ((on_beam & ~(1u << c)) == 0) ok |= 1u << c;
        h.colour_ok[b] = (uint16_t)ok;
    }

    // ---- cell tables
    std::vector<uint64_t> cell_lay(HW, 0);
    std::vector<uint32_t> cell_meta(HW, 0);
    static const int ACT_DELTA[4][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}};  // N S E W (action.rs:18-26)
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const int c = i * W + j;
            const auto& layers = cell_layers[c];
            h.max_layers = std::max<uint32_t>(h.max_layers, (uint32_t)layers.size());
            uint64_t lay = 0;
            for (size_t k = 0; k < layers.size(); k++) {
                const Source& src = sources[layers[k].laser_id];
                uint32_t colour = (uint32_t)std::min(src.agent_id, (int)NO_COLOUR);
                if ((int)colour >= A) colour = NO_COLOUR;
                lay |= (uint64_t)lay_pack((uint32_t)word_of(layers[k].laser_id, layers[k].offset), (uint32_t)bit_of(layers[k].offset), colour) << (16 * k);
            }
            cell_lay[c] = lay;
            uint32_t walk = 0;
            for (int d = 0; d < 4; d++) {
                int ni = i + ACT_DELTA[d][0], nj = j + ACT_DELTA[d][1];
                if (ni < 0 || nj < 0 || ni >= H || nj >= W) continue;
                uint8_t k = kind[ni * W + nj];
                if (k != K_WALL && k != K_SOURCE) walk |= 1u << d;
            }
            uint32_t gi = gem_index[c] >= 0 ? (uint32_t)gem_index[c] : NO_INDEX;
            if (kind[c] == K_SOURCE)  // the field holds the (first) beam word of a source cell (read by the partial observer: its colour)
                for (const Source& src : sources)
                    if (src.pos.i == i && src.pos.j == j) gi = (uint32_t)source_word[(size_t)src.laser_id];
            cell_meta[c] = meta_pack(kind[c], gi, walk, (uint32_t)layers.size());
        }

    // ---- layered observation tables (static template + dynamic bytes) with the Layered channel order
    std::vector<int8_t> tmpl;
    std::vector<uint64_t> dyn_tab;
    {
        LayerMap lm;
        lm.C = C;
        for (int a = 0; a < A; a++) lm.agent[a] = a;
        lm.n_laser = C - A;  // LASER_0 + colour is a plain index: colours >= A alias WALL/VOID/GEM/EXIT (Q5)
        for (int c = 0; c < lm.n_laser; c++) lm.laser[c] = A + c;
        lm.wall = 2 * A; lm.void_ = 2 * A + 1; lm.gem = 2 * A + 2; lm.exit = 2 * A + 3;
        if (!build_obs_tables(lm, tmpl, dyn_tab)) h.obs_supported = 0;
    }
    h.D = (uint32_t)dyn_tab.size();
    // ---- the head of a row (tables.h): the longest run of 128-byte lines without a dynamic byte -- behind the agent
    // layers (an agent can stand on any walkable cell), no dyn entry -- cut to `head_lines` lines
    if (h.obs_stride % 128u == 0 && h.obs_supported) {
        const uint32_t n_lines = h.obs_stride / 128u;
        std::vector<uint8_t> dynamic_line(n_lines, 0);
        for (uint32_t l = 0; l < n_lines && l * 128u < (uint32_t)(A * HW); l++) dynamic_line[l] = 1;
        for (uint64_t e : dyn_tab) dynamic_line[((uint32_t)e & 0xFFFFFu) / 128u] = 1;
        uint32_t best_lo = 0, best_n = 0;
        for (uint32_t l = 0; l < n_lines;) {
            if (dynamic_line[l]) { l++; continue; }
            uint32_t e = l;
            while (e < n_lines && !dynamic_line[e]) e++;
            if (e - l > best_n) { best_lo = l; best_n = e - l; }
            l = e;
        }
        const uint32_t want = head_lines < 0 ? std::max(1u, (n_lines + 2u) / 5u) : (uint32_t)head_lines;
        best_n = std::min(best_n, std::min(want, 8u));
        h.head_lo = best_lo * 8u;
        h.head_n = best_n * 8u;
    }

    // ---- the dynamic chunks of a row (tables.h off_dyn_chunks): every chunk of a line that holds an agent-layer byte or a dyn entry
    std::vector<uint16_t> dyn_chunks;
    if (h.obs_stride % 128u == 0 && h.obs_supported && h.n_chunks <= 0xFFFFu) {
        const uint32_t n_lines = h.obs_stride / 128u;
        std::vector<uint8_t> dynamic_line(n_lines, 0);
        for (uint32_t l = 0; l < n_lines && l * 128u < (uint32_t)(A * HW); l++) dynamic_line[l] = 1;
        for (uint64_t e : dyn_tab) dynamic_line[((uint32_t)e & 0xFFFFFu) / 128u] = 1;
        for (uint32_t l = 0; l < n_lines; l++)
            if (dynamic_line[l])
                for (uint32_t c = 0; c < 8; c++) dyn_chunks.push_back((uint16_t)(l * 8u + c));
    } else {
        for (uint32_t c = 0; c < h.n_chunks && c <= 0xFFFFu; c++) dyn_chunks.push_back((uint16_t)c);
    }
    h.n_dyn_chunks = h.n_chunks <= 0xFFFFu ? (uint32_t)dyn_chunks.size() : h.n_chunks;

    // the same for environments with their own source colours (tables.h pes_head_*): a laser byte may then sit on any of the
    // A laser layers, so the whole of [0, 2A * HW) is dynamic; behind it only the gem bytes are
    if (h.obs_stride % 128u == 0 && h.obs_supported) {
        bool plain_colours = true;
        for (auto& src : sources) plain_colours = plain_colours && src.agent_id < A;
        const uint32_t n_lines = h.obs_stride / 128u;
        std::vector<uint8_t> dynamic_line(n_lines, 0);
        for (uint32_t l = 0; l < n_lines && l * 128u < (uint32_t)(2 * A * HW); l++) dynamic_line[l] = 1;
        for (int g = 0; g < G; g++) dynamic_line[(uint32_t)((2 * A + 2) * HW + gems[g].i * W + gems[g].j) / 128u] = 1;
        uint32_t best_lo = 0, best_n = 0;
        for (uint32_t l = 0; l < n_lines && plain_colours;) {
            if (dynamic_line[l]) { l++; continue; }
            uint32_t e = l;
            while (e < n_lines && !dynamic_line[e]) e++;
            if (e - l > best_n) { best_lo = l; best_n = e - l; }
            l = e;
        }
        const uint32_t want = std::min(head_lines < 0 ? std::max(1u, (n_lines + 2u) / 5u) : (uint32_t)head_lines, 8u);
        const uint32_t run1 = best_n;  // (uncut: the second run lies outside it)
        best_n = std::min(best_n, want);
        // what the first run leaves of the wanted lines comes from the next longest run of static lines (one 64-lane store covers both)
        uint32_t lo2 = 0, n2 = 0;
        for (uint32_t l = 0; l < n_lines && plain_colours && best_n != 0 && best_n < want;) {
            if (dynamic_line[l] || (l >= best_lo && l < best_lo + run1)) { l++; continue; }
            uint32_t e = l;
            while (e < n_lines && !dynamic_line[e] && !(e >= best_lo && e < best_lo + run1)) e++;
            if (e - l > n2) { lo2 = l; n2 = e - l; }
            l = e;
        }
        n2 = std::min(n2, want - best_n);
        if (n2 != 0 && lo2 < best_lo) { std::swap(lo2, best_lo); std::swap(n2, best_n); }  // in row order
        h.pes_head_lo = best_lo * 8u;
        h.pes_head_n = best_n * 8u;
        h.pes_head2_lo = n2 ? lo2 * 8u : 0u;
        h.pes_head2_n = n2 * 8u;
    }

    // ---- assemble blob
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = sizeof(MapHeader);
    // the static observation as bits (tables.h off_tmpl_bits), right behind the header: at the same offset in every map of these dimensions
    std::vector<uint16_t> tmpl_bits(h.n_chunks, 0);
    std::vector<uint32_t> tmpl_neg;
    {
        bool bits_ok = true;
        for (size_t i = 0; i < tmpl.size(); i++) {
            if (tmpl[i] == 1) tmpl_bits[i / 16] |= (uint16_t)(1u << (i % 16));
            else if (tmpl[i] == -1) tmpl_neg.push_back((uint32_t)i);
            else if (tmpl[i] != 0) bits_ok = false;
        }
        bits_ok = bits_ok && tmpl_neg.size() <= TMPL_NEG_MAX;
        h.off_tmpl_bits = bits_ok ? (uint32_t)off : 0u;
        h.tmpl_neg_n = bits_ok ? (uint32_t)tmpl_neg.size() : 0u;
        h.bits_bytes = tmpl_section_bytes(h.n_chunks);
        off += h.bits_bytes;
    }
    h.off_cell_lay = (uint32_t)off; off = align16(off + cell_lay.size() * 8);
    h.off_cell_meta = (uint32_t)off; off = align16(off + cell_meta.size() * 4);
    h.off_dyn = (uint32_t)off; off = align16(off + dyn_tab.size() * 8);
    h.off_dyn_chunks = (uint32_t)off; off = align16(off + dyn_chunks.size() * 2);  // (ahead of the template: split-row launches keep it in LDS too)
    h.off_template = (uint32_t)off; off = align16(off + tmpl.size());
    // the kernel copies [off_cell_lay, blob_bytes) to LDS in rows of 64 lanes x 16 B: pad to whole rows
    off = h.off_cell_lay + ((off - h.off_cell_lay + 1023) & ~(size_t)1023);
    h.blob_bytes = (uint32_t)off;
    // recolouring can split merged dyn entries: at most one per exposed laser tile plus one per gem.  The capacity
    // is the blob size with that largest dyn table, so it does not depend on the current colours.
    {
        size_t cap = h.off_dyn;
        cap = align16(cap + ((size_t)n_laser_tiles() + (size_t)G) * 8);
        cap = align16(cap + (size_t)h.n_chunks * 2);  // (the dynamic chunks: at most every chunk of the row)
        cap = align16(cap + tmpl.size());
        h.blob_capacity = (uint32_t)(h.off_cell_lay + ((cap - h.off_cell_lay + 1023) & ~(size_t)1023));
    }
    h.lds_table_bytes = h.blob_bytes - h.off_cell_lay;
    h.lds_split_table_bytes = (h.off_template - h.off_cell_lay + 1023u) & ~1023u;  // (<= lds_table_bytes: the copy stays inside the blob)
    // ---- second section (per-environment sources), at an offset that does not depend on the colours
    std::vector<int8_t> bare(h.obs_stride, 0);
    {
        auto bat = [&](int layer, Pos q) -> int8_t& { return bare[(size_t)layer * HW + q.i * W + q.j]; };
        for (auto& q : walls) bat(2 * A, q) = 1;
        for (auto& q : voids) bat(2 * A + 1, q) = 1;
        for (auto& q : exits) bat(2 * A + 3, q) = 1;
    }
    std::vector<uint32_t> elems;
    for (auto& s : sources) elems.push_back(elem_pack((uint32_t)(s.pos.i * W + s.pos.j), (uint32_t)source_word[(size_t)s.laser_id], 0u, ELEM_SOURCE));
    for (int c = 0; c < HW; c++)
        for (size_t k = 0; k < cell_layers[c].size() && k < 2; k++)
            elems.push_back(elem_pack((uint32_t)c, (uint32_t)word_of(cell_layers[c][k].laser_id, cell_layers[c][k].offset),
                                      (uint32_t)bit_of(cell_layers[c][k].offset), ELEM_TILE));
    for (int g = 0; g < G; g++) elems.push_back(elem_pack((uint32_t)(gems[g].i * W + gems[g].j), (uint32_t)g, 0u, ELEM_GEM));
    h.off_bare = h.blob_capacity;
    h.off_elems = h.off_bare + h.obs_stride;
    h.n_elems = (uint32_t)elems.size();
    // re-colouring table (tables.h off_recolour): World::reset turns every enabled beam fully on, then the agent standing on
    // its start INSIDE a beam of its own colour cuts that beam from there (pre_enter, laser.rs:173-182)
    std::vector<uint32_t> recolour((size_t)L * (A + 1), 0u);
    for (int b = 0; b < L; b++) {
        const int s = word_source[(size_t)b];
        if (s < 0) continue;
        recolour[(size_t)b * (A + 1)] = h.colour_ok[b];
        for (int c = 0; c < A; c++) {
            uint32_t beam = h.beam_full[b];
            const Pos st = starts[c][0];
            for (const auto& lay : cell_layers[st.i * W + st.j])
                if (lay.laser_id == s) {  // the cut runs to the end of the beam: this word from the start's bit on, later words whole
                    const int w = word_of(s, lay.offset);
                    if (w == b) beam &= (1u << bit_of(lay.offset)) - 1u;
                    else if (w < b) beam = 0u;
                }
            recolour[(size_t)b * (A + 1) + 1 + c] = beam;
        }
    }
    // the dynamic chunks of a row under per-environment colours (tables.h off_pes_dyn_chunks)
    std::vector<uint16_t> pes_dyn_chunks;
    {
        bool plain_colours = h.obs_stride % 128u == 0 && h.obs_supported && h.n_chunks <= 0xFFFFu;
        for (auto& src : sources) plain_colours = plain_colours && src.agent_id < A;
        const uint32_t n_lines = h.obs_stride / 128u;
        std::vector<uint8_t> dynamic_line(n_lines ? n_lines : 1, plain_colours ? 0 : 1);
        for (uint32_t l = 0; plain_colours && l < n_lines && l * 128u < (uint32_t)(2 * A * HW); l++) dynamic_line[l] = 1;
        for (int g = 0; plain_colours && g < G; g++) dynamic_line[(uint32_t)((2 * A + 2) * HW + gems[g].i * W + gems[g].j) / 128u] = 1;
        if (plain_colours) {
            for (uint32_t l = 0; l < n_lines; l++)
                if (dynamic_line[l])
                    for (uint32_t c = 0; c < 8; c++) pes_dyn_chunks.push_back((uint16_t)(l * 8u + c));
        }
        h.n_pes_dyn_chunks = plain_colours ? (uint32_t)pes_dyn_chunks.size() : h.n_chunks;
        if (!plain_colours) pes_dyn_chunks.clear();
    }
    h.off_recolour = h.off_elems + h.n_elems * 4u;
    // (the step kernel draws one colour per WORD: exact only where a word is a source)
    h.recolour_exact = (h.max_layers <= 2 && chain_mask == 0) ? 1u : 0u;
    h.off_pes_dyn_chunks = (h.off_recolour + (uint32_t)recolour.size() * 4u + 15u) & ~15u;
    // (sized for the largest table -- every chunk of the row --, so that ext_bytes does not depend on the colours)
    h.ext_bytes = ((h.off_pes_dyn_chunks - h.off_bare) + h.n_chunks * 2u + 1023u) & ~1023u;
    off = (size_t)h.blob_capacity + h.ext_bytes;
    // ---- the packed image of the table section (tables.h off_packed), sized for the largest dyn table like blob_capacity
    std::vector<uint16_t> pk_meta(HW), pk_idx;
    std::vector<uint64_t> pk_lay;
    bool pk_ok = HW <= 0xFFFF;
    for (int c = 0; c < HW; c++) {
        pk_ok = pk_ok && cell_meta[c] <= 0xFFFFu;
        pk_meta[c] = (uint16_t)cell_meta[c];
        if (cell_lay[c]) { pk_idx.push_back((uint16_t)c); pk_lay.push_back(cell_lay[c]); }
    }
    const uint32_t pk_tail = h.off_template - h.off_dyn;  // (dyn + dynamic chunks, verbatim)
    h.packed_n_lay = pk_ok ? (uint32_t)pk_idx.size() : 0u;
    h.packed_bytes = pk_ok ? packed_meta_bytes((uint32_t)HW) + packed_idx_bytes(h.packed_n_lay) + ((h.packed_n_lay * 8u + 15u) & ~15u) + pk_tail : 0u;
    h.off_packed = pk_ok ? (uint32_t)off : 0u;
    {   // capacity: every cell with a layer, the largest dyn table, every chunk dynamic -- the same for any colouring of the sources
        const size_t cap_tail = (((size_t)n_laser_tiles() + (size_t)G) * 8 + 15) / 16 * 16 + ((size_t)h.n_chunks * 2 + 15) / 16 * 16;
        size_t n_lay_cells = 0;
        for (auto& l : cell_layers) n_lay_cells += l.empty() ? 0 : 1;
        h.packed_cap = (uint32_t)((packed_meta_bytes((uint32_t)HW) + packed_idx_bytes((uint32_t)n_lay_cells) + ((n_lay_cells * 8 + 15) & ~(size_t)15) + cap_tail + 127) & ~(size_t)127);
        off += h.packed_cap;
    }
    blob.assign(off, 0);
    if (h.off_packed) {
        uint8_t* q = blob.data() + h.off_packed;
        std::memcpy(q, pk_meta.data(), pk_meta.size() * 2); q += packed_meta_bytes((uint32_t)HW);
        if (!pk_idx.empty()) std::memcpy(q, pk_idx.data(), pk_idx.size() * 2);
        q += packed_idx_bytes(h.packed_n_lay);
        if (!pk_lay.empty()) std::memcpy(q, pk_lay.data(), pk_lay.size() * 8);
        q += (h.packed_n_lay * 8u + 15u) & ~15u;
        // (the tail is copied below, once the sections it repeats are in place)
    }
    if (h.off_tmpl_bits) {
        std::memcpy(blob.data() + h.off_tmpl_bits, tmpl_bits.data(), tmpl_bits.size() * 2);
        if (!tmpl_neg.empty()) std::memcpy(blob.data() + h.off_tmpl_bits + tmpl_bits_bytes(h.n_chunks), tmpl_neg.data(), tmpl_neg.size() * 4);
    }
    std::memcpy(blob.data() + h.off_bare, bare.data(), bare.size());
    if (!elems.empty()) std::memcpy(blob.data() + h.off_elems, elems.data(), elems.size() * 4);
    if (!recolour.empty()) std::memcpy(blob.data() + h.off_recolour, recolour.data(), recolour.size() * 4);
    if (!pes_dyn_chunks.empty()) std::memcpy(blob.data() + h.off_pes_dyn_chunks, pes_dyn_chunks.data(), pes_dyn_chunks.size() * 2);
    std::memcpy(blob.data() + h.off_cell_lay, cell_lay.data(), cell_lay.size() * 8);
    std::memcpy(blob.data() + h.off_cell_meta, cell_meta.data(), cell_meta.size() * 4);
    if (!dyn_tab.empty()) std::memcpy(blob.data() + h.off_dyn, dyn_tab.data(), dyn_tab.size() * 8);
    std::memcpy(blob.data() + h.off_template, tmpl.data(), tmpl.size());
    if (!dyn_chunks.empty()) std::memcpy(blob.data() + h.off_dyn_chunks, dyn_chunks.data(), dyn_chunks.size() * 2);
    if (h.off_packed) std::memcpy(blob.data() + h.off_packed + h.packed_bytes - pk_tail, blob.data() + h.off_dyn, pk_tail);
    std::memcpy(blob.data(), &h, sizeof h);
    header = h;
}

// The window table of the partial k x k observation (tables.h): [sets u64[HW][2] | cell_lay u64[HW] | cell_meta u32[HW]], whole 1-KiB rows.
std::vector<uint8_t> Map::window_table(int k) const {
    const int HW = H * W, centre = k / 2;
    std::vector<uint8_t> out(win_table_bytes((uint32_t)HW), 0);
    uint64_t* sets = reinterpret_cast<uint64_t*>(out.data());
    for (int pi = 0; pi < H; pi++)
        for (int pj = 0; pj < W; pj++)
            for (int wi = 0; wi < k; wi++)
                for (int wj = 0; wj < k; wj++) {
                    const int i = pi - centre + wi, j = pj - centre + wj;
                    if (i < 0 || j < 0 || i >= H || j >= W) continue;  // (outside the map: nothing, observations.py:331-340)
                    const int c = i * W + j;
                    const uint64_t bit = 1ull << (wi * k + wj);
                    uint64_t* e = sets + (size_t)(pi * W + pj) * 2;
                    if (kind[c] == K_WALL || kind[c] == K_SOURCE) e[0] |= bit;   // wall_pos holds the sources too (parser_v1.rs:22-25)
                    if (kind[c] == K_GEM || kind[c] == K_EXIT || kind[c] == K_SOURCE || !cell_layers[c].empty()) e[1] |= bit;
                }
    std::memcpy(out.data() + (size_t)HW * 16, blob.data() + header.off_cell_lay, (size_t)HW * 8);
    std::memcpy(out.data() + (size_t)HW * 24, blob.data() + header.off_cell_meta, (size_t)HW * 4);
    return out;
}

// World::set_exit_positions, src/core/world.rs:195-234.  The reference swaps tile objects: `Exit{agent}` -> `Floor{agent}`
// for every current exit, then `Floor{agent}` -> `Exit{agent}` for every new one; under a beam `Laser::set_tile`
// (laser.rs:109-115) replaces the INNERMOST tile whatever it is.  Here the innermost tile is `kind[cell]` and the
// occupant is dynamic state (one bit per agent), so the swap is a change of `kind` and nothing else.
int Map::set_exits(const std::vector<Pos>& new_exits, std::string& why) {
    if ((int)new_exits.size() < n_agents()) return LLE_PARSE_NOT_ENOUGH_EXIT_TILES;  // :196-201
    std::vector<uint8_t> k = kind;
    // :203-216 -- a current exit is `Tile::Exit` or a `Tile::Laser` (whose innermost tile becomes Floor); anything else
    // panics, which the parser and this function never leave behind
    for (const Pos& q : exits) k[q.i * W + q.j] = K_FLOOR;
    // :219-232
    for (const Pos& q : new_exits) {
        if (q.i < 0 || q.j < 0 || q.i >= H || q.j >= W) {
            why = "exit position (" + std::to_string(q.i) + ", " + std::to_string(q.j) + ") is out of the world (the reference panics)";
            return -1;
        }
        const int c = q.i * W + q.j;
        if (cell_layers[c].empty()) {
            if (k[c] != K_FLOOR) {  // `other => panic!("Tile is not a floor")`: wall, source, gem, void, or an exit given twice
                why = "tile (" + std::to_string(q.i) + ", " + std::to_string(q.j) + ") is not a floor (the reference panics)";
                return -1;
            }
        } else if (k[c] == K_GEM) {
            // Laser::set_tile would replace the gem by the exit without a word; World::gems() (world.rs:128-139) then
            // unwraps None on its next call.  Refused here.
            why = "tile (" + std::to_string(q.i) + ", " + std::to_string(q.j) + ") holds a gem under a laser (the reference drops the gem and panics later)";
            return -1;
        }
        k[c] = K_EXIT;  // (under a beam: whatever was innermost -- floor, a void -- is replaced, laser.rs:109-115)
    }
    kind = std::move(k);
    exits = new_exits;
    compile();
    return LLE_PARSE_OK;
}

std::string Map::world_string() const {
    // parser_v1.rs:100-130
    std::vector<std::vector<std::string>> res(H, std::vector<std::string>(W, " . "));
    for (size_t a = 0; a < starts.size(); a++) res[starts[a][0].i][starts[a][0].j] = "S" + std::to_string(a) + " ";
    for (auto& q : gems) res[q.i][q.j] = " G ";
    for (auto& q : walls) res[q.i][q.j] = " @ ";
    for (auto& q : exits) res[q.i][q.j] = " X ";
    for (auto& q : voids) res[q.i][q.j] = " V ";
    static const char* DIRS = "NESW";
    for (auto& s : sources) res[s.pos.i][s.pos.j] = "L" + std::to_string(s.agent_id) + DIRS[s.direction];
    std::string out;
    for (int i = 0; i < H; i++) {
        if (i) out += "\n";
        for (int j = 0; j < W; j++) {
            if (j) out += " ";
            out += res[i][j];
        }
    }
    return out;
}

}  // namespace lle
