// fuzz_lanes.cpp -- TEST-ONLY differential fuzzer, built with -fsanitize=address,undefined (tests/test_lanes_fuzz_sanitized.py):
// the lane-per-agent state machine of the step kernel (lle_amd/csrc/step_lanes.hpp through tests/hostsim) against the CPU
// oracle (oracle/lle_oracle.c), every buffer after every step, over the maps listed in a file.
//
//   fuzz_lanes <maps.txt> <envs> <steps> <engine 0|1|2>
// maps.txt: maps separated by a line "===".  Per map: rollouts with sampled actions, the first half of the steps without
// auto-reset (corpses pile up: quirks Q1 / Q2), the second half with.  Exit code 0 = no difference, 1 = a difference
// (printed), other = a sanitizer report.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/lle_hip.h"
#include "../../lle_amd/csrc/map_compile.hpp"

struct hs_batch;
struct ow_batch;
struct ow_world;
extern "C" {
hs_batch* hs_create(const char* text, int64_t n, int* parse_error);
void hs_free(hs_batch* b);
void hs_set_engine(hs_batch* b, int engine);
int64_t hs_lane_passes(hs_batch* b);
void hs_step(hs_batch* b, const uint8_t* actions, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset);
void* hs_buffer(hs_batch* b, int which);
ow_batch* ow_batch_create(const char* text, int64_t n_envs, int* err);
void ow_batch_free(ow_batch* b);
ow_world* ow_batch_world(ow_batch* b, int64_t e);
void ow_batch_step_range(ow_batch* b, int64_t e0, int64_t e1, const uint8_t* actions, int auto_reset, uint64_t seed, uint64_t t,
                         int64_t env_offset, uint8_t* actions_out, int32_t* err, uint8_t* ev_count, uint8_t* events, int8_t* obs,
                         int64_t* stats);
void ow_batch_dump(ow_batch* b, int64_t e0, int64_t e1, int beam_stride, uint8_t* pos, uint8_t* alive, uint8_t* arrived,
                   uint8_t* occupant, uint8_t* gems, uint8_t* beams, uint8_t* avail);
int ow_n_agents(ow_world* w);
int ow_n_gems(ow_world* w);
int ow_n_sources(ow_world* w);
int ow_height(ow_world* w);
int ow_width(ow_world* w);
int ow_panics(ow_world* w);
int hs_set_exits(hs_batch* b, const int32_t* exits_ij, int n_exits);
int ow_set_exits(ow_world* w, const int32_t* exits_ij, int n_exits);
}

static int fail(const char* what, size_t map_idx, uint64_t t, int64_t env, const std::string& text) {
    std::printf("MISMATCH in '%s' (map %zu, step %llu, env %lld):\n%s\n", what, map_idx, (unsigned long long)t, (long long)env, text.c_str());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: fuzz_lanes maps.txt envs steps engine\n"); return 2; }
    std::ifstream in(argv[1]);
    const int64_t n = std::atoll(argv[2]);
    const uint64_t steps = (uint64_t)std::atoll(argv[3]);
    const int engine = std::atoi(argv[4]);
    std::vector<std::string> maps(1);
    for (std::string line; std::getline(in, line);) {
        if (line == "===") maps.emplace_back();
        else maps.back() += line + "\n";
    }
    if (maps.back().empty()) maps.pop_back();
    int64_t env_steps = 0, deaths = 0, passes = 0, exits_taken = 0, exits_refused = 0, long_beams = 0;
    for (size_t m = 0; m < maps.size(); m++) {
        const std::string& text = maps[m];
        int perr = 0, oerr = 0;
        hs_batch* hb = hs_create(text.c_str(), n, &perr);
        ow_batch* ob = ow_batch_create(text.c_str(), n, &oerr);
        if (!hb || !ob) {
            if (!hb && !ob) continue;  // both refuse the map (the parse errors themselves are compared by the KATs)
            return fail("parse: one side refused the map", m, 0, 0, text);
        }
        hs_set_engine(hb, engine);
        ow_world* w0 = ow_batch_world(ob, 0);
        const int A = ow_n_agents(w0), G = ow_n_gems(w0), L = ow_n_sources(w0), H = ow_height(w0), W = ow_width(w0);
        const int C = 2 * A + 4, Ls = L ? L : 1;
        const size_t row = (size_t)C * H * W;
        lle::Map cm;  // the product's map compiler: row pitch of the host simulator's observation buffer, layout of the beam words
        if (lle::parse_map(text.c_str(), text.size(), cm) != 0) return fail("parse: the map compiler refused the map", m, 0, 0, text);
        const size_t pitch = (size_t)cm.header.obs_stride;
        int bs = 32;  // cells per beam in the oracle's dump: the longest beam (beams of more than 32 cells are chains of words, tables.h)
        for (const auto& src : cm.sources) bs = std::max(bs, (int)src.beam.size());
        const int Lw = cm.n_words() ? cm.n_words() : 1;
        long_beams += bs > 32;
        std::vector<uint8_t> o_act(n * A), o_evc(n), o_ev(n * 4 * A), o_pos(n * A * 2), o_alive(n * A), o_arr(n * A), o_occ(n * A),
            o_gems(n * (G ? G : 1)), o_beams(n * Ls * bs), o_avail(n * A);
        std::vector<int32_t> o_err(n);
        std::vector<int8_t> o_obs(n * row);
        uint64_t lcg = 0x9E3779B97F4A7C15ull * (m + 1);
        for (uint64_t t = 0; t < steps; t++) {
            const int auto_reset = t >= steps / 2;
            const uint64_t seed = 1000 + m;
            if (t % 16 == 15) {
                // World.exit_pos = [...] (world.rs:195-234) with RANDOM cells, legal or not: both sides must take or refuse the same
                // lists (a refusal leaves either untouched), and go on identically afterwards
                int32_t ij[2 * 20];
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                const int cnt = A - 1 + (int)((lcg >> 60) & 3u) < 0 ? 0 : A - 1 + (int)((lcg >> 60) & 3u);
                for (int k = 0; k < cnt && k < 20; k++) {
                    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                    ij[2 * k] = (int32_t)((lcg >> 33) % (uint64_t)(H + 1));      // (H, W themselves: out of the world)
                    ij[2 * k + 1] = (int32_t)((lcg >> 13) % (uint64_t)(W + 1));
                }
                const int r_hs = hs_set_exits(hb, ij, cnt < 20 ? cnt : 20);
                int r_ow = 0;
                for (int64_t e = 0; e < n; e++) r_ow = ow_set_exits(ow_batch_world(ob, e), ij, cnt < 20 ? cnt : 20);
                if ((r_hs == 0) != (r_ow == 0) || (r_hs > 0) != (r_ow > 0)) return fail("set_exits: one side took the list, the other refused it", m, t, 0, text);
                exits_taken += r_hs == 0;
                exits_refused += r_hs != 0;
            }
            ow_batch_step_range(ob, 0, n, nullptr, auto_reset, seed, t, 3, o_act.data(), o_err.data(), o_evc.data(), o_ev.data(), o_obs.data(), nullptr);
            ow_batch_dump(ob, 0, n, bs, o_pos.data(), o_alive.data(), o_arr.data(), o_occ.data(), G ? o_gems.data() : nullptr,
                          L ? o_beams.data() : nullptr, o_avail.data());
            hs_step(hb, nullptr, LLE_STEP_SAMPLE_ACTIONS | (auto_reset ? LLE_STEP_AUTO_RESET : 0), seed, t, 3);
            const uint16_t* pos = (const uint16_t*)hs_buffer(hb, LLE_BUF_POS);
            const uint64_t* bits = (const uint64_t*)hs_buffer(hb, LLE_BUF_BITS);
            const uint32_t* gems = (const uint32_t*)hs_buffer(hb, LLE_BUF_GEMS);
            const uint32_t* beams = (const uint32_t*)hs_buffer(hb, LLE_BUF_BEAMS);
            const uint8_t* avail = (const uint8_t*)hs_buffer(hb, LLE_BUF_AVAIL);
            const uint8_t* actions = (const uint8_t*)hs_buffer(hb, LLE_BUF_ACTIONS);
            const uint8_t* err = (const uint8_t*)hs_buffer(hb, LLE_BUF_ERR);
            const uint8_t* evc = (const uint8_t*)hs_buffer(hb, LLE_BUF_EVCOUNT);
            const uint8_t* ev = (const uint8_t*)hs_buffer(hb, LLE_BUF_EVENTS);
            const int8_t* obs = (const int8_t*)hs_buffer(hb, LLE_BUF_OBS);
            for (int64_t e = 0; e < n; e++) {
                if ((int32_t)err[e] != o_err[e]) return fail("err", m, t, e, text);
                if (evc[e] != o_evc[e]) return fail("event count / auto-reset flag", m, t, e, text);
                for (int k = 0; k < (evc[e] & 0x7F); k++) {
                    const uint8_t byte = ev[e * 2 * A + k];
                    if ((byte >> 4) != o_ev[e * 4 * A + 2 * k] || (byte & 15) != o_ev[e * 4 * A + 2 * k + 1]) return fail("events", m, t, e, text);
                    deaths += (byte >> 4) == 2;
                }
                for (int a = 0; a < A; a++) {
                    const uint16_t p = pos[e * A + a];
                    if ((p & 0xFF) != o_pos[(e * A + a) * 2] || (p >> 8) != o_pos[(e * A + a) * 2 + 1]) return fail("pos", m, t, e, text);
                    if (((bits[e] >> a) & 1) != o_alive[e * A + a]) return fail("alive", m, t, e, text);
                    if (((bits[e] >> (16 + a)) & 1) != o_arr[e * A + a]) return fail("arrived", m, t, e, text);
                    if (((bits[e] >> (32 + a)) & 1) != o_occ[e * A + a]) return fail("occupant", m, t, e, text);
                    if (avail[e * A + a] != o_avail[e * A + a]) return fail("avail", m, t, e, text);
                    if (actions[e * A + a] != o_act[e * A + a]) return fail("actions", m, t, e, text);
                }
                for (int g = 0; g < G; g++)
                    if (((gems[e] >> g) & 1) != o_gems[e * G + g]) return fail("gems", m, t, e, text);
                for (int s = 0; s < L; s++)
                    for (int k = 0; k < (int)cm.sources[(size_t)s].beam.size(); k++)
                        if (((beams[e * Lw + cm.word_of(s, k)] >> lle::Map::bit_of(k)) & 1) != o_beams[((size_t)e * L + s) * bs + k]) return fail("beams", m, t, e, text);
                if (std::memcmp(obs + e * pitch, o_obs.data() + e * row, row) != 0) return fail("obs", m, t, e, text);
            }
            env_steps += n;
        }
        for (int64_t e = 0; e < n; e++)
            if (ow_panics(ow_batch_world(ob, e))) return fail("the oracle reached a reference panic site", m, steps, e, text);
        passes += hs_lane_passes(hb);
        hs_free(hb);
        ow_batch_free(ob);
    }
    std::printf("OK maps=%zu env_steps=%lld deaths=%lld exits_taken=%lld exits_refused=%lld long_beam_maps=%lld lane_passes=%lld engine=%d\n", maps.size(), (long long)env_steps, (long long)deaths,
                (long long)exits_taken, (long long)exits_refused, (long long)long_beams,
                (long long)passes, engine);
    return 0;
}
