// map_compile.hpp -- host-side map compiler: v1 text -> static tables for the HIP kernels.
//
// Replaces, for the hot path, the reference's build-time pipeline
//   parser_v1::parse (src/core/parsing/parser_v1.rs:132-175)
//   -> WorldConfig::into_world (src/core/parsing/world_config.rs:107-122): pre_validate, make_grid (:176-199),
//      laser_setup (:203-250, beam tracing + start pruning), post_validate.
// Instead of a grid of boxed tiles it emits flat tables (tables.h).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "tables.h"

namespace lle {

struct Pos {
    int i, j;
    bool operator==(const Pos& o) const { return i == o.i && j == o.j; }
};

struct Source {
    Pos pos;
    int direction;  // N=0 E=1 S=2 W=3
    int agent_id;   // colour (mutable: LaserSource.set_agent_id)
    bool enabled;
    int laser_id;
    std::vector<Pos> beam;  // cells in offset order
};

struct CellLayer { int laser_id, offset; };

struct Map {
    int H = 0, W = 0;
    std::vector<Pos> gems, exits, voids, walls;       // parse order (walls include sources, parser_v1.rs:22-25)
    std::vector<std::vector<Pos>> starts;             // per agent, after start pruning
    std::vector<Source> sources;                      // laser_id order
    std::vector<std::vector<CellLayer>> cell_layers;  // [HW], outermost first
    std::vector<uint8_t> kind;                        // [HW] CellKind of the innermost tile
    std::vector<int> gem_index;                       // [HW] or -1

    // Pitch of an observation row in HBM (and of its LDS template): C*H*W rounded up to `row_align` bytes.  16 is the
    // store width; 128 = one cache line, so that the rows of neighbouring environments never share a line (a batch
    // whose rows do not fit the Infinity Cache then writes whole lines to HBM only; lle_map_set_row_align).
    // 0 = automatic (the default): whole lines when that pads the row by at most 1/32 (level 6: 1 872 -> 1 920 B,
    // +2.6 % bytes, measured 2 % FASTER inside the Infinity Cache and 5-9 % faster past it), 16 otherwise (level 1:
    // 936 B would become 1 024, +9.4 %).
    uint32_t row_align = 0;
    uint32_t row_pitch_of(uint32_t bytes) const {
        if (row_align == 0) {
            const uint32_t lines = (bytes + 127u) & ~127u;
            return (uint64_t)(lines - bytes) * 32u <= bytes ? lines : ((bytes + 15u) & ~15u);
        }
        return (bytes + row_align - 1u) / row_align * row_align;
    }

    // Lines of a row that the default step kernel stores ahead of the state machine (tables.h head_lo / head_n), at most
    // 8; -1 = automatic: a fifth of the row (measured best on levels 3, 5, 6 and generated 8- and 12-agent maps: with
    // less the state machine is not covered, with more the head stores themselves hold the wavefront up);
    // LLE_HEAD_LINES overrides the default.
    static int default_head_lines() {
        const char* e = std::getenv("LLE_HEAD_LINES");
        return e && *e ? std::atoi(e) : -1;
    }
    int head_lines = default_head_lines();

    // Beam words (tables.h): a beam of len cells is ceil(len / 32) consecutive 32-bit words (at least one); chained maps are padded
    // to five words.  Filled by layout_words() (parse time; the beams never change afterwards).
    std::vector<int> source_word;   // first word of source s
    std::vector<int> word_source;   // source of word b (-1: padding)
    uint32_t chain_mask = 0;        // bit b: word b continues word b - 1
    int n_words() const { return (int)word_source.size(); }
    bool layout_words();            // false: more than MAX_SOURCES words
    int word_of(int laser_id, int offset) const { return source_word[(size_t)laser_id] + offset / 32; }
    static int bit_of(int offset) { return offset % 32; }
    int word_len(int b) const;      // cells of word b

    int n_agents() const { return (int)starts.size(); }
    int n_layers() const { return 2 * n_agents() + 4; }
    int n_laser_tiles() const;

    // channel layout of a layered-style observation: layer index of each agent, of each laser colour, and of
    // the four fixed layers
    struct LayerMap {
        int C = 0, n_laser = 0;
        int agent[MAX_AGENTS] = {0};
        int laser[2 * MAX_AGENTS + 40] = {0};
        int wall = 0, void_ = 0, gem = 0, exit = 0;
    };
    bool build_obs_tables(const LayerMap& lm, std::vector<int8_t>& tmpl, std::vector<uint64_t>& dyn_tab) const;
    std::vector<uint8_t> compile_view(int kind, int param) const;  // ViewHeader + dyn + template (tables.h)
    std::vector<uint8_t> window_table(int k) const;                // the window table of the partial k x k observation (tables.h), k = 3, 5, 7

    // compiled form
    MapHeader header{};
    std::vector<uint8_t> blob;  // header + sections
    void compile();             // (re)builds header + blob from the fields above

    std::string world_string() const;  // parser_v1.rs:100-130 to_v1_string

    // World::set_exit_positions (src/core/world.rs:195-234): the current exits become Floor, the given cells Exit; the
    // occupant of a tile survives the swap, so nothing dynamic changes.  Returns LLE_PARSE_OK,
    // LLE_PARSE_NOT_ENOUGH_EXIT_TILES (fewer exits than agents, :196-201), or -1 with `why` set where the reference
    // PANICS half way through the swap (`other => panic!`, :213,230; an index out of the grid): there the map is left
    // untouched instead.  Recompiles the tables.
    int set_exits(const std::vector<Pos>& new_exits, std::string& why);
};

// Returns LLE_PARSE_* (0 = ok).
int parse_map(const char* text, size_t len, Map& out);
extern const char* const LEVEL_TEXT[6];

}  // namespace lle
