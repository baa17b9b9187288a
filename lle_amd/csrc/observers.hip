// observers.hip -- gfx950 kernels of the observation builders other than the layered tensor the step kernel emits
// (reference python/lle/observations.py, python/lle/env/env.py:146-163).
//
//   view_observe_kernel    layered-padded / agent-zero perspective: the layered streamer of obs_stream.hpp run on a
//                          "view" blob (other channel layout, map_compile.cpp compile_view) -- HBM-write bound
//   partial_observe_kernel k x k windows centred on each agent, assembled in LDS per environment and streamed as one
//                          16-byte-aligned row -- HBM-write bound
//   state_observe_kernel   [i0, j0, ..., gems, alive] as f32, optionally normalised -- one thread per element
//   avail_kernel           availability as bools, optionally without moves into foreign active lasers
//
// All of them read the packed state the step kernel keeps (pos, bits, gems, beams, avail) and write to a caller buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "obs_stream.hpp"
#include "observers_logic.hpp"
#include "partial_stream.hpp"
#include "tables.h"

// (kernels.h debug registry: one relaxed load per launch)
#define LLE_NOTE_OBS(ID)                                                                                 \
    do {                                                                                                 \
        static std::atomic<bool> noted_{false};                                                          \
        if (!noted_.load(std::memory_order_relaxed)) {                                                   \
            noted_.store(true, std::memory_order_relaxed);                                               \
            lle::debug_note(lle::debug_key(lle::DBG_OBSERVER, (ID) & 15, 0, (ID) >> 4, false, -1), false); \
        }                                                                                                \
    } while (0)

namespace lle {

constexpr uint32_t OBS_ENVS_PER_WAVE = 16;

// tables of the map that owns env (batches of several maps: lle_batch_create_multi; envs_per_map = 0: one map)
__device__ __forceinline__ const uint8_t* tables_of(const BatchPtrs& P, const MapSel& M, int64_t env) {
    return P.tables + (M.envs_per_map ? (uint64_t)env / (uint64_t)M.envs_per_map : 0ull) * M.table_stride;
}
constexpr uint32_t OBS_LDS_LIMIT = 160 * 1024;

// ---------------------------------------------------------------------------------------------- layered views
// `views` = n_views view blobs of equal size back to back (one for layered-padded; one per observer for the
// perspective when they fit in LDS together).  LDS: [the blobs] [element list of the map, per-env sources only] then per
// wave, per view [patchable template copy | OBS_ENVS_PER_WAVE records].  Row of (env, view v) = out + env * row_pitch +
// v * view_pitch: with several views the wave writes the rows of one environment back to back.
// pes: the batch keeps source colours per environment -- the copy starts from the view's bare static observation and
// every env writes its laser / gem bytes through the view's colour -> layer table (write_observations_env).
__global__ void __launch_bounds__(256) view_observe_kernel(BatchPtrs P, const uint8_t* __restrict__ views, uint32_t n_views,
                                                           int8_t* __restrict__ out, int64_t row_pitch, int64_t view_pitch,
                                                           int64_t env_base, int64_t env_limit, int pes, MapSel M, uint32_t views_stride, int wt, uint32_t walk,
                                                           uint32_t epw /* environments per wavefront: OBS_ENVS_PER_WAVE, or fewer when a map owns fewer */,
                                                           uint32_t et /* element type of the rows (tables.h ObsElem): row_pitch / view_pitch are in BYTES */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, walk);  // the block of environments this workgroup serves, and the launch's direction (obs_stream.hpp)
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
    const int64_t wg_env0 = env_base + (int64_t)(blk * waves_per_wg) * epw;
    const uint8_t* __restrict__ map_tables = tables_of(P, M, wg_env0);   // this workgroup's map and its views
    views += (M.envs_per_map ? (uint64_t)wg_env0 / (uint64_t)M.envs_per_map : 0ull) * views_stride;
    const ViewHeader* __restrict__ gh = reinterpret_cast<const ViewHeader*>(views);
    const MapHeader* __restrict__ mh = reinterpret_cast<const MapHeader*>(map_tables);
    const uint32_t blob_bytes = gh->blob_bytes;
    copy_tables_to_lds(views, lds, blob_bytes * n_views, lane, wave_in_wg, waves_per_wg);
    const uint32_t n_elems = pes ? mh->n_elems : 0u, elem_bytes = (n_elems * 4u + 15u) & ~15u;
    uint32_t* elems = reinterpret_cast<uint32_t*>(lds + blob_bytes * n_views);
    if (pes) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(map_tables + mh->off_elems);
        for (uint32_t i = threadIdx.x; i < n_elems; i += blockDim.x) elems[i] = src[i];
    }
    __syncthreads();
    const ViewHeader* vh0 = reinterpret_cast<const ViewHeader*>(lds);
    const int A = (int)vh0->A, L = (int)vh0->L, W = (int)vh0->W;
    const int64_t As = agent_stride_of(A, L);
    const int CW = pes ? src_stride_of(L) / 4 : 0;
    const uint32_t obs_stride = vh0->obs_stride, n_chunks = vh0->n_chunks;
    const uint32_t scr_stride = (uint32_t)(L + A + 2 + CW) | 1u;
    // private to a wavefront: ONE patchable row (re-initialised from the view's pristine template when the wave turns to
    // the next view) and the hand-over records of its environments, one set per view (the agents' byte indices differ)
    const uint32_t rec_bytes = OBS_ENVS_PER_WAVE * scr_stride * 4u;
    uint8_t* priv = lds + blob_bytes * n_views + elem_bytes + wave_in_wg * (obs_stride + n_views * rec_bytes);
    int8_t* tmpl = reinterpret_cast<int8_t*>(priv);
    const int64_t env0 = env_base + (int64_t)wave_id * epw;
    int64_t n_here = env_limit - env0;
    n_here = n_here < 0 ? 0 : (n_here > (int64_t)epw ? (int64_t)epw : n_here);
    // hand-over records [0 | beam masks | ~gem bits | byte index of each agent | colour words], all loads in flight together
    const uint32_t per_env = (uint32_t)(L + A + 2 + CW);
    for (uint32_t idx = lane; idx < (uint32_t)n_here * per_env; idx += 64) {
        const uint32_t k = idx / per_env, f = idx - k * per_env;
        const int64_t env = env0 + k;
        uint32_t v = 0, cell = 0;
        const bool is_agent = f > (uint32_t)L + 1u && f < (uint32_t)(L + 2 + A);
        if (f >= 1 && f <= (uint32_t)L) v = P.beams[env * L + (f - 1)];
        else if (f == (uint32_t)L + 1u) v = ~P.gems[env];
        else if (is_agent) cell = cell_of((uint32_t)P.pos[env * As + (f - (uint32_t)L - 2u)], W);
        else if (f >= (uint32_t)(L + 2 + A)) v = reinterpret_cast<const uint32_t*>(P.src_colour)[env * CW + (f - (uint32_t)(L + 2 + A))];
        for (uint32_t q = 0; q < n_views; q++) {
            const ViewHeader* vh = reinterpret_cast<const ViewHeader*>(lds + q * blob_bytes);
            uint32_t* scratch = reinterpret_cast<uint32_t*>(priv + obs_stride + q * rec_bytes);
            scratch[k * scr_stride + f] = is_agent ? (uint32_t)vh->agent_layer[f - (uint32_t)L - 2u] * vh->HW + cell : v;
        }
    }
    wave_sync();
    // one view at a time, all the wave's environments per view: the lane's dyn entry is decoded once per view and the
    // patch / stream / unpatch of consecutive environments pipeline (env-major order -- the views of an environment back
    // to back, one call per row -- measured 108.7 us for level 6 / 65 536 envs x 4 observers, this order 99.9)
    if (n_here > 0)
        for (uint32_t q = 0; q < n_views; q++) {
            const ViewHeader* vh = reinterpret_cast<const ViewHeader*>(lds + q * blob_bytes);
            const uint32_t* scratch = reinterpret_cast<const uint32_t*>(priv + obs_stride + q * rec_bytes);
            {
                const uint4* pristine = reinterpret_cast<const uint4*>(lds + q * blob_bytes + (pes ? vh->off_bare : vh->off_template));
                uint4* mine = reinterpret_cast<uint4*>(tmpl);
                for (uint32_t c = lane; c < n_chunks; c += 64) mine[c] = pristine[c];
                wave_sync();
            }
            // (the store policy and the element width become template arguments here, outside the loops over the environments: obs_stream.hpp dispatch_stream)
            const uint32_t sflags = (wt ? LAUNCH_WRITE_THROUGH : 0u) | (et << LAUNCH_OBS_ELEM_SHIFT);
            if (pes) {
                const int8_t* bare = reinterpret_cast<const int8_t*>(lds + q * blob_bytes + vh->off_bare);
                dispatch_stream<true>(sflags, [&](auto wt_, auto wide_) {
                    constexpr bool WT = decltype(wt_)::value, WIDE = decltype(wide_)::value;
                    write_observations_env<WT, false, false, WIDE>(A, L, vh->HW, n_elems, n_chunks, (uint64_t)row_pitch, elems, bare, tmpl, scratch, scr_stride,
                                                                   out + (int64_t)q * view_pitch, env0, n_here, lane, vh->laser_layer, vh->gem_layer, 0u, 0u, 0u,
                                                                   nullptr, 0u, 0u, 0u, et);
                });
            } else {
                const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + q * blob_bytes + vh->off_dyn);
                dispatch_stream<true>(sflags, [&](auto wt_, auto wide_) {
                    constexpr bool WT = decltype(wt_)::value, WIDE = decltype(wide_)::value;
                    write_observations<WT, false, false, WIDE>(A, L, vh->D, n_chunks, (uint64_t)row_pitch, dyn, tmpl, scratch, scr_stride,
                                                               out + (int64_t)q * view_pitch, env0, n_here, lane, 0u, 0u, 0u, nullptr, 0u, et);
                });
            }
        }
}

// ---------------------------------------------------------------------------------------------- partial k x k
// LDS: [map tables (cell stacks, cell meta, ...)] [unit table: (agent, wi, wj) of every window cell] then per wave
// [row buffer (pitch bytes) | records of its envs].  record of an env: pos as u16[As] | gems u32 | beams u32[L].
// Per environment the wave clears its row with 16-byte LDS stores, every lane evaluates its window cells (only
// non-zero bytes are written), and the row is streamed as 16 B per lane.
__global__ void __launch_bounds__(256) partial_observe_kernel(BatchPtrs P, int8_t* __restrict__ out, int k, uint32_t pitch,
                                                              int64_t env_base, int64_t env_limit, int per_env_sources, MapSel M,
                                                              uint32_t epw, uint32_t walk, uint32_t et) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, walk);
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
    const uint8_t* __restrict__ tables = tables_of(P, M, env_base + (int64_t)(blk * waves_per_wg) * epw);
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    const int A = (int)hdr->A, L = (int)hdr->L;
    const int64_t As = agent_stride_of(A, L);
    const uint32_t tab_bytes = hdr->lds_table_bytes, tab_off = hdr->off_cell_lay;
    copy_tables_to_lds(tables + tab_off, lds, tab_bytes, lane, wave_in_wg, waves_per_wg);
    const uint32_t kk = (uint32_t)(k * k), units = (uint32_t)A * kk, row_bytes = units * (uint32_t)(2 * A + 3);
    const uint32_t unit_bytes = (units * 4u + 15u) & ~15u;
    uint32_t* unit_tab = reinterpret_cast<uint32_t*>(lds + tab_bytes);
    for (uint32_t u = threadIdx.x; u < units; u += blockDim.x) {
        // [row offset of the cell's byte in layer 0 of observer a : 20 | a : 4 | wi : 4 | wj : 4]  (A <= 16, k <= 15)
        const uint32_t a = u / kk, w = u - a * kk, wi = w / (uint32_t)k, wj = w - wi * (uint32_t)k;
        unit_tab[u] = (a * (uint32_t)(2 * A + 3) * kk + w) | (a << 20) | (wi << 24) | (wj << 28);
    }
    __syncthreads();
    ObsTables T;
    T.cell_lay = reinterpret_cast<const uint64_t*>(lds);
    T.cell_meta = reinterpret_cast<const uint32_t*>(lds + (hdr->off_cell_meta - tab_off));
    T.beam_colour = hdr->beam_colour;
    T.A = A; T.H = (int)hdr->H; T.W = (int)hdr->W;
    const uint32_t rec_dwords = (uint32_t)(As / 2 + 1 + L);
    const uint32_t occ_bytes = ((hdr->HW + 1u) / 2u * 4u + 15u) & ~15u;  // which agents stand on a cell: u16 per cell
    const uint32_t priv_bytes = pitch + ((epw * rec_dwords * 4u + 15u) & ~15u) + occ_bytes;
    int8_t* row = reinterpret_cast<int8_t*>(lds + tab_bytes + unit_bytes + wave_in_wg * priv_bytes);
    uint32_t* recs = reinterpret_cast<uint32_t*>(row + pitch);
    uint32_t* occ = reinterpret_cast<uint32_t*>(row + pitch + ((epw * rec_dwords * 4u + 15u) & ~15u));
    const int64_t env0 = env_base + (int64_t)wave_id * epw;
    int64_t n_here = env_limit - env0;
    n_here = n_here < 0 ? 0 : (n_here > (int64_t)epw ? (int64_t)epw : n_here);
    for (uint32_t idx = lane; idx < (uint32_t)n_here * rec_dwords; idx += 64) {
        const uint32_t e = idx / rec_dwords, f = idx - e * rec_dwords;
        const int64_t env = env0 + e;
        uint32_t v;
        if (f < (uint32_t)(As / 2)) v = reinterpret_cast<const uint32_t*>(P.pos)[env * (As / 2) + f];
        else if (f == (uint32_t)(As / 2)) v = P.gems[env];
        else v = P.beams[env * L + (f - (uint32_t)(As / 2) - 1u)];
        recs[idx] = v;
    }
    wave_sync();
    const uint32_t n_chunks = pitch / 16;
    (void)row_bytes;
    uint4* row16 = reinterpret_cast<uint4*>(row);
    for (int64_t e = 0; e < n_here; e++) {
        const uint32_t* rec = recs + (uint32_t)e * rec_dwords;
        const uint16_t* pos = reinterpret_cast<const uint16_t*>(rec);
        const uint32_t gems = rec[As / 2];
        const uint32_t* beams = rec + As / 2 + 1;
        if (per_env_sources) T.beam_colour = P.src_colour + (env0 + e) * src_stride_of(L);  // this env's colours
        for (uint32_t c = lane; c < n_chunks; c += 64) row16[c] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t c = lane; c < occ_bytes / 16u; c += 64) reinterpret_cast<uint4*>(occ)[c] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();  // LDS operations of a wave execute in order: the writes below land after the clears
        if ((int)lane < A) {  // dead agents included (agents_positions), so two agents may share a cell: a bit each
            const uint32_t cell = (uint32_t)(pos[lane] & 0xFFu) * (uint32_t)T.W + (uint32_t)(pos[lane] >> 8);
            atomicOr(&occ[cell >> 1], (1u << lane) << (16u * (cell & 1u)));
        }
        wave_sync();
        const int centre = k / 2;
        for (uint32_t u = lane; u < units; u += 64) {
            const uint32_t t = unit_tab[u];
            const int a = (int)((t >> 20) & 15u), wi = (int)((t >> 24) & 15u), wj = (int)(t >> 28);
            const uint32_t pa = pos[a];
            const int i = (int)(pa & 0xFFu) - centre + wi, j = (int)(pa >> 8) - centre + wj;
            if (i < 0 || j < 0 || i >= T.H || j >= T.W) continue;
            const int cell = i * T.W + j;
            uint32_t here = (occ[cell >> 1] >> (16 * (cell & 1))) & 0xFFFFu;
            const uint32_t meta = T.cell_meta[cell];
            const uint64_t lay = T.cell_lay[cell];
            int8_t* cp = row + (t & 0xFFFFFu);
            while (here) {  // agents first (observations.py:345-346)
                const int a2 = __ffs((int)here) - 1;
                here &= here - 1u;
                cp[a2 * (int)kk] = 1;
            }
            if (meta_kind(meta) == K_FLOOR && lay == 0ull) continue;  // nothing else can be on the cell
            // gems, exits, walls, lasers that are on, -1 at sources: the reference's order (partial_cell, observers_logic.hpp)
            const uint32_t kind = meta_kind(meta), idx = meta_index(meta);
            const int WALL = A, LASER_0 = A + 1, GEM = 2 * A + 1, EXIT = 2 * A + 2;
            if (kind == K_GEM && !((gems >> idx) & 1u)) cp[GEM * (int)kk] = 1;
            if (kind == K_EXIT) cp[EXIT * (int)kk] = 1;
            if (kind == K_WALL || kind == K_SOURCE) cp[WALL * (int)kk] = 1;
            for (int q = 0; q < 2; q++) {
                const uint32_t e2 = lay_entry(lay, (int)q);
                if (!(e2 & LAY_VALID)) break;
                const uint32_t beam = lay_word(e2), off = lay_bit(e2);
                if ((beams[beam] >> off) & 1u) cp[(LASER_0 + (int)T.beam_colour[beam]) * (int)kk] = 1;
            }
            if (kind == K_SOURCE) cp[(LASER_0 + (int)T.beam_colour[idx]) * (int)kk] = -1;
        }
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + (((uint64_t)(env0 + e) * pitch) << obs_elem_shift(et)));
        // plain stores: written through (stream_store<true>) this kernel measured 5-10 % slower at every size
        if (et == OBS_I8) stream_whole_row<false>(dst, row16, n_chunks, lane);
        else stream_wide<false>(dst, reinterpret_cast<const int8_t*>(row16), n_chunks, et, lane);  // (the batch's element type: widened at the store, obs_stream.hpp)
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------- partial k x k, by projection
// The window kernel above evaluates every cell of every agent's window against the cell tables: A * k * k units of ~75
// vector instructions each, most of them on empty floor (7x7 on level 6: 2 394 VALU per wavefront of 8 envs, vector-issue
// bound at 3.1 TB/s).  A row is almost all zeros, though: its non-zero bytes are the map's ENTITIES seen through the
// windows -- walls (sources included), exits, uncollected gems, lit laser tiles (the two layers World.lasers() exposes),
// -1 at sources, agents.  This kernel projects instead: one lane per (entity, observer) pair tests whether the entity's
// cell lies in the observer's window and, if so, writes its one byte.  (n_entities + A) * A pairs of ~30 instructions
// (level 6: 54 x 4 = 216 pairs against 196 window cells at 2.5 x the cost each).  All writes of one environment commute:
// two entities share a byte only where a laser colour A / A + 1 aliases GEM / EXIT, and then both write 1 (a source, the
// only -1, sits on a wall cell that holds nothing else) -- so any lane order reproduces the reference's write order
// (observations.py:343-359).  The launcher picks the cheaper kernel per (map, k).
//
// The entity table is built once per workgroup, straight from the map's cell tables in global memory (nothing else of
// them is needed afterwards, so they are not copied to LDS): u32 = i | j << 8 | type << 16 | index << 19 | offset << 24.
// LDS: [count | entity table] then per wave [row (pitch bytes) | records of its envs: pos u16[As] | gems | beams[L] | colour words].
enum : uint32_t { PE_WALL = 0, PE_EXIT = 1, PE_GEM = 2, PE_TILE = 3, PE_SOURCE = 4, PE_AGENT = 5 };
__global__ void __launch_bounds__(256) partial_project_kernel(BatchPtrs P, int8_t* __restrict__ out, int k, uint32_t pitch,
                                                              int64_t env_base, int64_t env_limit, int per_env_sources, MapSel M,
                                                              uint32_t epw, uint32_t ent_cap, uint32_t walk, uint32_t et) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, walk);
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
    const uint8_t* __restrict__ tables = tables_of(P, M, env_base + (int64_t)(blk * waves_per_wg) * epw);
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    const int A = (int)hdr->A, L = (int)hdr->L, W = (int)hdr->W;
    const int64_t As = agent_stride_of(A, L);
    uint32_t* ent_count = reinterpret_cast<uint32_t*>(lds);
    uint32_t* ents = reinterpret_cast<uint32_t*>(lds) + 4;
    if (threadIdx.x == 0) *ent_count = 0u;
    __syncthreads();
    {
        const uint64_t* __restrict__ glay = reinterpret_cast<const uint64_t*>(tables + hdr->off_cell_lay);
        const uint32_t* __restrict__ gmeta = reinterpret_cast<const uint32_t*>(tables + hdr->off_cell_meta);
        for (uint32_t c = threadIdx.x; c < hdr->HW; c += blockDim.x) {
            const uint32_t meta = gmeta[c], kind = meta_kind(meta), idx = meta_index(meta);
            const uint64_t lay = glay[c];
            const uint32_t i = c / (uint32_t)W, ij = i | ((c - i * (uint32_t)W) << 8);
            uint32_t e[4], n = 0;
            if (kind == K_WALL || kind == K_SOURCE) e[n++] = ij | (PE_WALL << 16);   // wall_pos holds the sources too
            if (kind == K_SOURCE) e[n++] = ij | (PE_SOURCE << 16) | (idx << 19);      // idx = laser id of a source cell
            if (kind == K_EXIT) e[n++] = ij | (PE_EXIT << 16);
            if (kind == K_GEM) e[n++] = ij | (PE_GEM << 16) | (idx << 19);
            for (int q = 0; q < 2; q++) {                                            // World.lasers(): two layers per cell
                const uint32_t e2 = lay_entry(lay, (int)q);
                if (!(e2 & LAY_VALID)) break;
                e[n++] = ij | (PE_TILE << 16) | (lay_word(e2) << 19) | (lay_bit(e2) << 24);
            }
            if (n) {
                const uint32_t at = atomicAdd(ent_count, n);
                for (uint32_t q = 0; q < n; q++)
                    if (at + q < ent_cap) ents[at + q] = e[q];
            }
        }
    }
    __syncthreads();
    const uint32_t n_ent = *ent_count < ent_cap ? *ent_count : ent_cap;
    const int CW = src_stride_of(L) / 4;
    const uint32_t rec_dwords = (uint32_t)(As / 2 + 1 + L + CW);
    const uint32_t ent_bytes = (16u + ent_cap * 4u + 15u) & ~15u;
    const uint32_t priv_bytes = pitch + ((epw * rec_dwords * 4u + 15u) & ~15u);
    int8_t* row = reinterpret_cast<int8_t*>(lds + ent_bytes + wave_in_wg * priv_bytes);
    uint32_t* recs = reinterpret_cast<uint32_t*>(row + pitch);
    const int64_t env0 = env_base + (int64_t)wave_id * epw;
    int64_t n_here = env_limit - env0;
    n_here = n_here < 0 ? 0 : (n_here > (int64_t)epw ? (int64_t)epw : n_here);
    const uint32_t* __restrict__ map_colw = reinterpret_cast<const uint32_t*>(hdr->beam_colour);
    for (uint32_t idx = lane; idx < (uint32_t)n_here * rec_dwords; idx += 64) {
        const uint32_t e = idx / rec_dwords, f = idx - e * rec_dwords;
        const int64_t env = env0 + e;
        uint32_t v;
        if (f < (uint32_t)(As / 2)) v = reinterpret_cast<const uint32_t*>(P.pos)[env * (As / 2) + f];
        else if (f == (uint32_t)(As / 2)) v = P.gems[env];
        else if (f < (uint32_t)(As / 2 + 1 + L)) v = P.beams[env * L + (f - (uint32_t)(As / 2) - 1u)];
        else {
            const uint32_t q = f - (uint32_t)(As / 2 + 1 + L);  // colours, 4 per word: the env's own or the map's
            v = per_env_sources ? reinterpret_cast<const uint32_t*>(P.src_colour)[env * CW + q] : map_colw[q];
        }
        recs[idx] = v;
    }
    wave_sync();
    const uint32_t n_chunks = pitch / 16, kk = (uint32_t)(k * k), layers = (uint32_t)(2 * A + 3);
    const uint32_t logA = A <= 1 ? 0u : (A <= 2 ? 1u : (A <= 4 ? 2u : (A <= 8 ? 3u : 4u)));
    const uint32_t n_pairs = (n_ent + (uint32_t)A) << logA;
    const int centre = k / 2;
    // A lane's pairs p = lane + 64 q are the same for every environment, and so is its observer a = lane mod 2^logA:
    // what does not depend on the environment is decoded ONCE per wavefront -- the entity word, and for entities whose
    // layer is static (everything but lasers under per-env colours) the byte offset of the layer in the observer's block.
    constexpr int QM = 4;  // pairs per lane kept in registers (level 6: 216 pairs = 4 per lane); the rest take the loop below
    const uint32_t a = lane & ((1u << logA) - 1u);
    const bool a_ok = a < (uint32_t)A;
    // per pair, static: packed cell of the entity, byte offset of (observer a, layer) in the row, the value to write, and
    // how the dynamic parts are read -- LDS offsets inside an env record for the agent position / beam mask / colour byte
    // (a harmless in-record offset when the pair needs none), bit to test, and masks that select among them arithmetically:
    // the per-environment loop below has no data-dependent branch and issues its LDS reads together.
    uint32_t q_cell[QM], q_base[QM], q_posoff[QM], q_beamoff[QM], q_coloff[QM], q_bit[QM], q_agent[QM], q_gem[QM], q_tile[QM], q_always[QM],
        q_laser[QM];
    int32_t q_val[QM];
    const uint8_t* map_colour = hdr->beam_colour;
    const uint32_t beams_at = (uint32_t)(As / 2 + 1) * 4u, colour_at = (uint32_t)(As / 2 + 1 + L) * 4u;  // byte offsets in a record
#pragma unroll
    for (int q = 0; q < QM; q++) {
        const uint32_t p = lane + 64u * (uint32_t)q, en = p >> logA;
        const bool valid = p < n_pairs && a_ok, is_agent = en >= n_ent;
        const uint32_t ent = is_agent ? ((en - n_ent) << 19) | ((uint32_t)PE_AGENT << 16) : ents[valid ? en : 0u];
        const uint32_t type = (ent >> 16) & 7u, idx = (ent >> 19) & 31u, off = (ent >> 24) & 31u;
        const bool laser = type == PE_TILE || type == PE_SOURCE;
        const uint32_t col = (laser && !per_env_sources) ? (uint32_t)map_colour[idx] : 0u;
        const uint32_t layer = type == PE_WALL ? (uint32_t)A : type == PE_EXIT ? (uint32_t)(2 * A + 2) : type == PE_GEM ? (uint32_t)(2 * A + 1)
                             : type == PE_AGENT ? idx : (uint32_t)(A + 1) + col;
        q_cell[q] = ent & 0xFFFFu;
        q_base[q] = (a * layers + layer) * kk;
        q_val[q] = type == PE_SOURCE ? -1 : 1;
        q_posoff[q] = (type == PE_AGENT ? idx : 0u) * 2u;
        q_beamoff[q] = beams_at + (type == PE_TILE ? idx : 0u) * 4u;
        q_coloff[q] = colour_at + (laser ? idx : 0u);
        q_bit[q] = type == PE_GEM ? idx : off;
        q_agent[q] = type == PE_AGENT ? 0xFFFFFFFFu : 0u;
        q_gem[q] = (valid && type == PE_GEM) ? 1u : 0u;
        q_tile[q] = (valid && type == PE_TILE) ? 1u : 0u;
        q_always[q] = (valid && type != PE_GEM && type != PE_TILE) ? 1u : 0u;
        q_laser[q] = (laser && per_env_sources) ? kk : 0u;   // colour -> byte offset multiplier (0: the layer is static)
    }
    uint4* row16 = reinterpret_cast<uint4*>(row);
    for (int64_t e = 0; e < n_here; e++) {
        const uint8_t* rec8 = reinterpret_cast<const uint8_t*>(recs + (uint32_t)e * rec_dwords);
        const uint16_t* pos = reinterpret_cast<const uint16_t*>(rec8);
        const uint32_t gems = reinterpret_cast<const uint32_t*>(rec8)[As / 2];
        for (uint32_t c = lane; c < n_chunks; c += 64) row16[c] = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t pa = pos[a_ok ? a : 0u];
        uint32_t r_pos[QM], r_beam[QM], r_col[QM];
#pragma unroll
        for (int q = 0; q < QM; q++) {
            r_pos[q] = *reinterpret_cast<const uint16_t*>(rec8 + q_posoff[q]);
            r_beam[q] = *reinterpret_cast<const uint32_t*>(rec8 + q_beamoff[q]);
            r_col[q] = rec8[q_coloff[q]];
        }
        const int oi = (int)(pa & 0xFFu) - centre, oj = (int)(pa >> 8) - centre;  // the window's origin
        wave_sync();  // LDS operations of a wave execute in order: the writes below land after the clears
#pragma unroll
        for (int q = 0; q < QM; q++) {
            const uint32_t pe = (r_pos[q] & q_agent[q]) | (q_cell[q] & ~q_agent[q]);
            const int dy = (int)(pe & 0xFFu) - oi, dx = (int)(pe >> 8) - oj;
            const uint32_t on = (q_gem[q] & ~(gems >> q_bit[q])) | (q_tile[q] & (r_beam[q] >> q_bit[q])) | q_always[q];
            const uint32_t at = q_base[q] + r_col[q] * q_laser[q] + (uint32_t)(dy * k + dx);
            const bool in = (on & 1u) && (uint32_t)dy < (uint32_t)k && (uint32_t)dx < (uint32_t)k;
            if (in) row[at] = (int8_t)q_val[q];
        }
        const int8_t* colour = reinterpret_cast<const int8_t*>(rec8 + colour_at);
        const uint32_t* beams = reinterpret_cast<const uint32_t*>(rec8 + beams_at);
        for (uint32_t p = lane + 64u * QM; p < n_pairs; p += 64) {  // maps with more pairs than QM per lane
            const uint32_t en = p >> logA;
            const bool is_agent = en >= n_ent;
            const uint32_t ent = is_agent ? ((en - n_ent) << 19) | ((uint32_t)PE_AGENT << 16) : ents[en];
            const uint32_t type = (ent >> 16) & 7u, idx = (ent >> 19) & 31u, off = (ent >> 24) & 31u;
            const uint32_t pe = type == PE_AGENT ? (uint32_t)pos[idx] : (ent & 0xFFFFu);
            const int dy = (int)(pe & 0xFFu) - oi, dx = (int)(pe >> 8) - oj;
            const uint32_t beam_bits = beams[type == PE_TILE ? idx : 0u];
            const uint32_t col = (uint32_t)(uint8_t)colour[(type == PE_TILE || type == PE_SOURCE) ? idx : 0u];
            const bool on = type == PE_GEM ? !((gems >> idx) & 1u) : (type == PE_TILE ? ((beam_bits >> off) & 1u) != 0 : true);
            const uint32_t layer = type == PE_WALL ? (uint32_t)A : type == PE_EXIT ? (uint32_t)(2 * A + 2) : type == PE_GEM ? (uint32_t)(2 * A + 1)
                                 : type == PE_AGENT ? idx : (uint32_t)(A + 1) + col;
            const bool in = a_ok && (uint32_t)dy < (uint32_t)k && (uint32_t)dx < (uint32_t)k && on;
            if (in) row[(a * layers + layer) * kk + (uint32_t)(dy * k + dx)] = type == PE_SOURCE ? (int8_t)-1 : (int8_t)1;
        }
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + (((uint64_t)(env0 + e) * pitch) << obs_elem_shift(et)));
        if (et == OBS_I8) stream_whole_row<false>(dst, row16, n_chunks, lane);
        else stream_wide<false>(dst, reinterpret_cast<const int8_t*>(row16), n_chunks, et, lane);
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------- partial k x k, one lane per (env, observer)
// Both kernels above spend the whole wavefront on ONE environment at a time: 36 window cells (3x3, four agents) leave half
// of the lanes idle, every environment pays its own clear -> patch -> stream chain, and the instruction count per
// environment (150-300 vector instructions) is what bounds small maps (~20 us at 65 536 envs whatever the window).
// Here a wavefront takes a BATCH of E environments and a lane is one (environment, observer) pair -- or, when E x observers
// is below 64, one share of that observer's window rows (S lanes per observer).  What makes a lane cheap:
//   * a non-empty bitmap of the map, one bit per cell (wall, source, exit, gem, laser tile), built once per workgroup with
//     8 empty cells of margin on every side: a window row is k bits of it at (i0 + wi, j0), no bounds test anywhere;
//   * the lane walks only the SET bits of its rows (level 6: 1-2 of the 9 cells of a 3x3 window, ~8 of 49 at 7x7) and
//     evaluates those cells against the cell tables like the window kernel -- reference order observations.py:343-359:
//     agents, gems, exits, walls, lasers that are on, -1 at sources; all writes of an environment commute (see above);
//   * agents are pairs (observer, other agent): A tests per lane instead of an occupancy map.
// The E rows are contiguous in the output, so the wavefront clears, patches and streams them as ONE block of E x pitch
// bytes: 1 KiB per store instruction whatever the row size (a 400-byte row alone fills 25 lanes of 64).
// LDS: [cell_lay | cell_meta (whole KiB rows)] [bitmap] then per wave [E rows] [E records: pos u16[As] | gems | beams[L] | colour words].
struct PartialDims { int32_t A, L, H, W; uint32_t off_cell_meta, max_layers; };  // common to the maps of a batch; off_cell_meta relative to off_cell_lay
__global__ void __launch_bounds__(256) partial_lanes_kernel(BatchPtrs P, int8_t* __restrict__ out, int k, uint32_t pitch, int64_t env_base,
                                                            int64_t env_limit, int per_env_sources, MapSel M, uint32_t E, uint32_t batches,
                                                            uint32_t tab_bytes, int wt, PartialDims D, uint32_t tab_off, uint32_t walk,
                                                            const uint8_t* __restrict__ win_sets, uint32_t win_bytes, uint32_t et) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, walk);
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
    const uint32_t epw = E * batches;  // environments per wavefront
    const uint8_t* __restrict__ tables = tables_of(P, M, env_base + (int64_t)(blk * waves_per_wg) * epw);
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    // the dimensions come with the kernel arguments (the maps of a batch agree on them): no header read stands in front of
    // the first loads -- the wavefront's records and the workgroup's tables are requested together, one round trip in all
    const int A = D.A, L = D.L, H = D.H, W = D.W;
    const int64_t As = agent_stride_of(A, L);
    const int CW = src_stride_of(L) / 4;
    const uint32_t rec_dwords = (uint32_t)(As / 2 + 1 + L + CW);
    const int64_t envw = env_base + (int64_t)wave_id * epw;
    int64_t n_all = env_limit - envw;
    n_all = n_all < 0 ? 0 : (n_all > (int64_t)epw ? (int64_t)epw : n_all);
    const uint32_t* __restrict__ map_colw = reinterpret_cast<const uint32_t*>(hdr->beam_colour);
    // Record dword (e, f) of the wavefront -- pos u16[As] | gems | beams[L] | colour words -- is requested by lane idx = f * epw + e
    // (epw is a power of two: no division; neighbouring lanes read neighbouring environments of one array) and lands at
    // recs_all[e * rec_dwords + f].  The four arrays are told apart with selects, not branches.
    const uint32_t lg_epw = 31u - (uint32_t)__builtin_clz(epw);
    const uint32_t rec_total = rec_dwords << lg_epw;
    auto rec_word = [&](uint32_t idx, uint32_t& at) -> uint32_t {
        const uint32_t f = idx >> lg_epw, e = idx & (epw - 1u);
        const bool ok = idx < rec_total && e < (uint32_t)n_all;
        const int64_t env = envw + (ok ? e : 0u);
        const uint32_t fp = (uint32_t)(As / 2), fb = fp + 1u, fc = fb + (uint32_t)L;
        const uint32_t* base = f < fp ? reinterpret_cast<const uint32_t*>(P.pos) : (f == fp ? P.gems : (f < fc ? P.beams
                             : (per_env_sources ? reinterpret_cast<const uint32_t*>(P.src_colour) : map_colw)));
        const int64_t off = f < fp ? env * (As / 2) + f : (f == fp ? env : (f < fc ? env * L + (f - fb)
                          : (per_env_sources ? env * CW + (f - fc) : (int64_t)(f - fc))));
        at = ok ? e * rec_dwords + f : 0xFFFFFFFFu;
        return ok ? base[off] : 0u;
    };
    constexpr int RV = 4;  // record dwords per lane requested ahead of the table copy (the rest, if any, behind it)
    uint32_t rv[RV], rat[RV];
#pragma unroll
    for (int q = 0; q < RV; q++) rv[q] = rec_word(lane + 64u * (uint32_t)q, rat[q]);
    // Two LDS layouts ahead of the wavefronts' private areas:
    //   win_sets != NULL (k = 3, 5, 7): ONE window table of this window size and map -- [sets | cell_lay | cell_meta], tables.h --, copied in whole
    //     1-KiB rows (`win_bytes` = the table's stride per map); `tab_bytes` = what of it is kept, and the tail of the copy's last row lands in
    //     the first wavefront's row block, which that wavefront clears before use (behind the barrier below);
    //   else: the map's cell tables (tab_bytes, whole rows) and the non-empty bitmap of the map, built here from them.
    const uint32_t HWc = (uint32_t)(H * W);
    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds + (win_sets ? HWc * 16u : 0u));
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (win_sets ? HWc * 24u : D.off_cell_meta));
    const uint32_t RW = partial_bitmap_row_words((uint32_t)W), bm_words = (uint32_t)(H + 16) * RW;
    const uint32_t bm_bytes = win_sets ? 0u : (bm_words * 4u + 15u) & ~15u;
    uint32_t* bm = reinterpret_cast<uint32_t*>(lds + tab_bytes);
    if (win_sets) {
        const uint64_t map_idx = M.envs_per_map ? (uint64_t)(env_base + (int64_t)(blk * waves_per_wg) * epw) / (uint64_t)M.envs_per_map : 0ull;
        copy_tables_to_lds(win_sets + map_idx * win_bytes, lds, win_bytes, lane, wave_in_wg, waves_per_wg);
        __syncthreads();
    } else {
        copy_tables_to_lds(tables + tab_off, lds, tab_bytes, lane, wave_in_wg, waves_per_wg);
        for (uint32_t w = threadIdx.x; w < bm_words; w += blockDim.x) bm[w] = 0u;
        __syncthreads();  // (also: the table copy has landed)
        partial_bitmap_fill(bm, cell_lay, cell_meta, H, W);
        __syncthreads();
    }
    const uint64_t* sets = win_sets ? reinterpret_cast<const uint64_t*>(lds) : nullptr;
    // ---- who this lane is: environment slot e of the batch, observer a, share s of S of the window rows
    const uint32_t logA = A <= 1 ? 0u : (A <= 2 ? 1u : (A <= 4 ? 2u : (A <= 8 ? 3u : 4u)));
    const uint32_t S = 64u / (E << logA);                 // lanes per (env, observer); the launcher keeps E << logA <= 64
    const uint32_t e_slot = lane / (S << logA), a = (lane / S) & ((1u << logA) - 1u), s = lane % S;
    const uint32_t kk = (uint32_t)(k * k), layers = (uint32_t)(2 * A + 3), n_chunks = pitch / 16u;
    const uint32_t rec_bytes = (epw * rec_dwords * 4u + 15u) & ~15u;
    const uint32_t priv_bytes = E * pitch + rec_bytes + 16u;
    int8_t* rows = reinterpret_cast<int8_t*>(lds + tab_bytes + bm_bytes + wave_in_wg * priv_bytes);
    uint32_t* recs_all = reinterpret_cast<uint32_t*>(rows + E * pitch);
    uint4* rows16 = reinterpret_cast<uint4*>(rows);
    const uint32_t colour_at = (uint32_t)(As / 2 + 1 + L) * 4u;  // byte offset of the colour bytes in a record
#pragma unroll
    for (int q = 0; q < RV; q++)
        if (rat[q] != 0xFFFFFFFFu) recs_all[rat[q]] = rv[q];
    for (uint32_t idx = lane + 64u * RV; idx < rec_total; idx += 64) {
        uint32_t at;
        const uint32_t v = rec_word(idx, at);
        if (at != 0xFFFFFFFFu) recs_all[at] = v;
    }
    int8_t* dummy = rows + E * pitch + rec_bytes;   // 16 bytes nobody reads: where the writes of a cell that do not apply go
    const PartialGeo G{A, W, k, kk, S, RW, D.max_layers > 1u};   // (two_layers, uniform: maps without crossing beams never look at a second layer)
    const uint64_t share = sets ? partial_share_mask((uint32_t)k, S, s) : 0ull;

    for (uint32_t batch = 0; batch < batches; batch++) {
        const int64_t env0 = env_base + (int64_t)wave_id * epw + (int64_t)batch * E;
        int64_t n_here = env_limit - env0;
        n_here = n_here < 0 ? 0 : (n_here > (int64_t)E ? (int64_t)E : n_here);
        if (n_here == 0) break;
        const uint32_t* recs = recs_all + batch * E * rec_dwords;
        for (uint32_t c = lane; c < (uint32_t)n_here * n_chunks; c += 64) rows16[c] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();  // LDS operations of a wavefront execute in order: everything below lands after the clears
        const bool live = e_slot < (uint32_t)n_here && a < (uint32_t)A;
        const PartialRecPacked R{reinterpret_cast<const uint8_t*>(recs + (live ? e_slot : 0u) * rec_dwords), (uint32_t)(As / 2), colour_at};
        int8_t* mine = rows + __umul24(live ? e_slot : 0u, pitch) + __umul24(a, layers * kk);   // observer a's block of this env's row
        // the lane's agents and its share of the window's non-empty cells (partial_stream.hpp: shared with the step kernel's writer)
        partial_window(G, R, live, a, s, mine, dummy, cell_lay, cell_meta, bm, sets, share);
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + (((uint64_t)env0 * pitch) << obs_elem_shift(et)));
        if (et != OBS_I8) {  // (the batch's element type: the block of rows widened at the store, obs_stream.hpp stream_wide)
            if (wt) stream_wide<true>(dst, rows, (uint32_t)n_here * n_chunks, et, lane);
            else stream_wide<false>(dst, rows, (uint32_t)n_here * n_chunks, et, lane);
        } else if (wt) stream_row<true>(dst, rows16, 0u, (uint32_t)n_here * n_chunks, lane);
        else stream_row<false>(dst, rows16, 0u, (uint32_t)n_here * n_chunks, lane);
        wave_sync();  // the next batch clears the rows: after these reads (in order, same wavefront)
    }
}

// ---------------------------------------------------------------------------------------------- state vector
__global__ void __launch_bounds__(256) state_observe_kernel(BatchPtrs P, float* __restrict__ out, int normalize, int64_t n_envs) {
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(P.tables);  // dimensions are common to all maps
    const int A = (int)hdr->A, G = (int)hdr->G, L = (int)hdr->L;
    const int64_t As = agent_stride_of(A, L);
    const int len = 3 * A + G;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_envs * len) return;
    const int64_t env = idx / len;
    const int e = (int)(idx - env * len);
    out[idx] = state_elem(A, G, (int)hdr->H, (int)hdr->W, P.pos + env * As, P.gems[env], (uint32_t)P.bits[env] & 0xFFFFu, e, normalize != 0);
}

// ---------------------------------------------------------------------------------------------- availability bools
__global__ void __launch_bounds__(256) avail_kernel(BatchPtrs P, uint8_t* __restrict__ out, int walkable_lasers, int64_t n_envs,
                                                    int per_env_sources, MapSel M) {
    const MapHeader* __restrict__ hdr0 = reinterpret_cast<const MapHeader*>(P.tables);
    const int A = (int)hdr0->A, L = (int)hdr0->L;
    const int64_t As = agent_stride_of(A, L);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_envs * A) return;
    const int64_t env = idx / A;
    const uint8_t* __restrict__ tables = tables_of(P, M, env);
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    const int a = (int)(idx - env * A);
    ObsTables T;
    T.cell_lay = reinterpret_cast<const uint64_t*>(tables + hdr->off_cell_lay);
    T.cell_meta = reinterpret_cast<const uint32_t*>(tables + hdr->off_cell_meta);
    T.beam_colour = per_env_sources ? P.src_colour + env * src_stride_of(L) : hdr->beam_colour;
    T.A = A; T.H = (int)hdr->H; T.W = (int)hdr->W;
    const uint32_t m = avail_bools(T, P.pos + env * As, P.beams + env * L, a, (uint32_t)P.avail[env * As + a], walkable_lasers != 0);
    uint8_t* o = out + idx * 5;
    for (int act = 0; act < 5; act++) o[act] = (uint8_t)((m >> act) & 1u);
}

// ---------------------------------------------------------------------------------------------- LLE.step outputs
// One thread per (env, agent): its availability bools and alive / arrived flags, a share of the state vector
// (elements a, a + A, ...), and -- agent 0 -- the env's reward and done.  One launch instead of the four or more a
// host class needs when it assembles `Step` from the separate entry points.
__global__ void __launch_bounds__(256) env_outputs_kernel(BatchPtrs P, EnvOutputs O, int64_t n_envs, MapSel M) {
    const MapHeader* __restrict__ hdr0 = reinterpret_cast<const MapHeader*>(P.tables);
    const int A = (int)hdr0->A, L = (int)hdr0->L, G = (int)hdr0->G;
    const int64_t As = agent_stride_of(A, L);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_envs * A) return;
    const int64_t env = idx / A;
    const int a = (int)(idx - env * A);
    const uint64_t bits = P.bits[env];
    const uint32_t alive = (uint32_t)bits & 0xFFFFu, arrived = (uint32_t)(bits >> 16) & 0xFFFFu;
    if (O.alive) O.alive[idx] = (uint8_t)((alive >> a) & 1u);
    if (O.arrived) O.arrived[idx] = (uint8_t)((arrived >> a) & 1u);
    if (O.available) {
        const uint8_t* __restrict__ tables = tables_of(P, M, env);
        const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
        ObsTables T;
        T.cell_lay = reinterpret_cast<const uint64_t*>(tables + hdr->off_cell_lay);
        T.cell_meta = reinterpret_cast<const uint32_t*>(tables + hdr->off_cell_meta);
        T.beam_colour = O.per_env_sources ? P.src_colour + env * src_stride_of(L) : hdr->beam_colour;
        T.A = A; T.H = (int)hdr->H; T.W = (int)hdr->W;
        const uint32_t m = avail_bools(T, P.pos + env * As, P.beams + env * L, a, (uint32_t)P.avail[env * As + a], O.walkable_lasers != 0);
        uint8_t* o = O.available + idx * 5;
        for (int act = 0; act < 5; act++) o[act] = (uint8_t)((m >> act) & 1u);
    }
    if (O.state) {
        const int len = 3 * A + G;
        const uint32_t gems = P.gems[env];
        for (int e = a; e < len; e += A)
            O.state[env * len + e] = state_elem(A, G, (int)hdr0->H, (int)hdr0->W, P.pos + env * As, gems, alive, e, O.normalize_state != 0);
    }
    if (a == 0) {
        if (O.done) O.done[env] = P.done[env];
        if (O.reward) {
            // counts of the last step: gems | exits << 8 | deaths << 16 | all-arrived bonus << 24
            const uint32_t r = P.reward[env];
            const float gem = (float)(r & 255u), ex = (float)((r >> 8) & 255u), died = (float)((r >> 16) & 255u), bonus = (float)(r >> 24);
            if (O.reward_kind == 0) {
                O.reward[env] = gem + ex - died + bonus;  // reward_strategy.py:58-75 (the death override never fires)
            } else {  // reward_strategy.py:90-109: [gem, exit, death, done]; a death zeroes the others
                const bool dead = died > 0.f;
                float* o = O.reward + env * 4;
                o[0] = dead ? 0.f : gem; o[1] = dead ? 0.f : ex; o[2] = -died; o[3] = dead ? 0.f : bonus;
            }
        }
    }
}

// ---- ceiling probe (lle_probe_row_fill; bench.py `fill_ceiling`): the write pattern of the step kernel's observation stream
// and nothing else -- wavefront w owns rows [w * rows_per_wave, (w + 1) * rows_per_wave), 16 B per lane and instruction, the
// same stores (stream_store) and the same workgroup -> block mapping (xcd_block).  What this reaches on a box is what a
// row-owning writer can reach there; the step kernel is read against it (NOTEBOOK.md section 4 "Two kinds of box").
template <bool WT>
__global__ void __launch_bounds__(256) row_fill_probe_kernel(int8_t* __restrict__ out, int64_t n_rows, uint32_t n_chunks, uint32_t rows_per_wave,
                                                             uint32_t value, uint32_t flags) {
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, flags);
    const uint32_t lane = threadIdx.x & 63u, wave = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t r0 = (int64_t)wave * rows_per_wave;
    const uint4 v = {value, value, value, value};
    const int64_t left = n_rows - r0, n_here = left < (int64_t)rows_per_wave ? left : (int64_t)rows_per_wave;
    const uint32_t rot = row_rotation(wave, rows_per_wave, n_here, flags);  // (the step kernel's own row order)
    for (uint32_t r = 0; (int64_t)r < n_here; r++) {
        uint4* dst = reinterpret_cast<uint4*>(out + (r0 + rotated(r, rot, (uint32_t)n_here)) * (int64_t)n_chunks * 16);
        for (uint32_t c = lane; c < n_chunks; c += 64u) stream_store<WT>(dst + c, v);
    }
}

// A consumer's first touch of the observation (bench.py consumer_loop): every int8 of the rows converted to fp16 into a separate
// buffer -- what the first layer of a policy does with `obs` before the next step (python/lle/env/env.py:165-189: the caller of
// LLE.step reads the observation, then steps again).  16 bytes in, 32 bytes out per lane and iteration, grid-stride.
__global__ void __launch_bounds__(256) cast_rows_kernel(const int8_t* __restrict__ rows, _Float16* __restrict__ out, int64_t n_chunks) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(rows)[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        half8 lo, hi;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            lo[k] = (_Float16)(int)(int8_t)(w[k >> 2] >> (8 * (k & 3)));
            hi[k] = (_Float16)(int)(int8_t)(w[2 + (k >> 2)] >> (8 * (k & 3)));
        }
        reinterpret_cast<half8*>(out)[2 * c] = lo;
        reinterpret_cast<half8*>(out)[2 * c + 1] = hi;
    }
}
hipError_t launch_cast_rows(const int8_t* rows, void* out_f16, int64_t bytes, hipStream_t stream) {
    LLE_NOTE_OBS(OBSK_CAST_ROWS);
    hipLaunchKernelGGL(cast_rows_kernel, dim3(256 * 8), dim3(256), 0, stream, rows, static_cast<_Float16*>(out_f16), bytes / 16);
    return hipGetLastError();
}

// The eight rollout counters: every wavefront of a step launch owns a slot of 8 x i64 (no atomics on the hot path); this sums
// the slots.  One workgroup of 1 024 threads, thread t takes counter t & 7 of the slots t >> 3, t >> 3 + 128, ...: 64-byte
// rows read whole by 8 neighbouring lanes.  8 192 slots = 512 KB, L2-resident right after a rollout.
__global__ void __launch_bounds__(1024) stats_sum_kernel(const int64_t* __restrict__ stats, int64_t n_blocks, int64_t* __restrict__ out8) {
    __shared__ int64_t part[16][8];
    const uint32_t k = threadIdx.x & 7u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    int64_t acc = 0;
    for (int64_t slot = threadIdx.x >> 3; slot < n_blocks; slot += 128) acc += stats[slot * 8 + k];
    // lanes with the same k inside the wavefront: strides of 8
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)acc, o, 64), hi = __shfl_xor((uint32_t)((uint64_t)acc >> 32), o, 64);
        acc += (int64_t)(((uint64_t)hi << 32) | lo);
    }
    if (lane < 8) part[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 8) {
        int64_t v = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) v += part[w][threadIdx.x];
        out8[threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------- launchers
hipError_t launch_row_fill_probe(int8_t* out, int64_t n_rows, uint32_t row_bytes, uint32_t rows_per_wave, uint32_t value, bool reverse, bool rotate,
                                 hipStream_t stream) {
    const uint32_t flags = (reverse ? LAUNCH_REVERSE : 0u) | (rotate ? LAUNCH_ROTATE_ROWS : 0u);
    const uint32_t n_chunks = row_bytes / 16u, wpw = 4;
    const int64_t n_waves = (n_rows + rows_per_wave - 1) / rows_per_wave;
    const dim3 grid((uint32_t)((n_waves + wpw - 1) / wpw)), block(64 * wpw);
    if (write_through_pays((uint64_t)n_rows * row_bytes, row_bytes)) {
        LLE_NOTE_OBS(OBSK_ROW_FILL_WT);
        hipLaunchKernelGGL(row_fill_probe_kernel<true>, grid, block, 0, stream, out, n_rows, n_chunks, rows_per_wave, value, flags);
    } else {
        LLE_NOTE_OBS(OBSK_ROW_FILL_PLAIN);
        hipLaunchKernelGGL(row_fill_probe_kernel<false>, grid, block, 0, stream, out, n_rows, n_chunks, rows_per_wave, value, flags);
    }
    return hipGetLastError();
}

hipError_t launch_stats_sum(const int64_t* stats, int64_t n_blocks, int64_t* out8, hipStream_t stream) {
    LLE_NOTE_OBS(OBSK_STATS_SUM);
    hipLaunchKernelGGL(stats_sum_kernel, dim3(1), dim3(1024), 0, stream, stats, n_blocks, out8);
    return hipGetLastError();
}

// LDS bytes of the view kernel for n_views views per launch and wpw wavefronts per workgroup
static uint32_t view_lds(const ViewHeader& v, uint32_t n_views, uint32_t wpw, bool pes, uint32_t n_elems) {
    const uint32_t scr_stride = (v.L + v.A + 2 + (pes ? (uint32_t)src_stride_of((int)v.L) / 4u : 0u)) | 1u;
    const uint32_t elem_bytes = pes ? ((n_elems * 4u + 15u) & ~15u) : 0u;
    return n_views * v.blob_bytes + elem_bytes + wpw * (v.obs_stride + n_views * OBS_ENVS_PER_WAVE * scr_stride * 4u);
}

bool view_kernel_fits(const ViewHeader& v, uint32_t n_views, bool pes, uint32_t n_elems) {
    return view_lds(v, n_views, 1, pes, n_elems) <= OBS_LDS_LIMIT;
}

// workgroups must not straddle two maps: envs_per_map is a multiple of OBS_ENVS_PER_WAVE (narrower workgroups below 64 envs per map)
// (a map may own as few as 8 environments since round 5: the wavefronts then take fewer environments each)
static uint32_t obs_envs_per_wave(const MapSel& M) {
    uint32_t epw = OBS_ENVS_PER_WAVE;
    while (epw > 1 && M.envs_per_map && M.envs_per_map % (int64_t)epw != 0) epw >>= 1;
    return epw;
}
static uint32_t cap_wpw(uint32_t wpw, const MapSel& M, uint32_t epw = OBS_ENVS_PER_WAVE) {
    while (wpw > 1 && M.envs_per_map && M.envs_per_map % (int64_t)(wpw * epw) != 0) wpw >>= 1;
    return wpw;
}

hipError_t launch_view_observe(const ViewHeader& v, const BatchPtrs& P, const uint8_t* views_dev, uint32_t n_views, int8_t* out,
                               int64_t row_pitch, int64_t view_pitch, int64_t n_envs, bool pes, uint32_t n_elems, MapSel M,
                               uint32_t views_stride, bool reverse, hipStream_t stream, uint32_t et) {
    const uint32_t epw = obs_envs_per_wave(M);
    uint32_t wpw = cap_wpw(4, M, epw);
    while (wpw > 1 && view_lds(v, n_views, wpw, pes, n_elems) > OBS_LDS_LIMIT) wpw >>= 1;
    const uint32_t lds = view_lds(v, n_views, wpw, pes, n_elems);
    if (lds > OBS_LDS_LIMIT) return hipErrorInvalidValue;
    static LdsGrant granted;  // per device (kernels.h)
    hipError_t e = granted.ensure(reinterpret_cast<const void*>(&view_observe_kernel), lds);
    if (e != hipSuccess) return e;
    const uint32_t n_waves = (uint32_t)((n_envs + epw - 1) / epw);
    LLE_NOTE_OBS(OBSK_VIEW);
    hipLaunchKernelGGL(view_observe_kernel, dim3((n_waves + wpw - 1) / wpw), dim3(64 * wpw), lds, stream, P, views_dev, n_views, out,
                       row_pitch, view_pitch, (int64_t)0, n_envs, pes ? 1 : 0, M, views_stride,
                       write_through_pays((uint64_t)n_envs * (uint64_t)(n_views > 1 ? view_pitch * n_views : row_pitch), (uint32_t)v.obs_stride << obs_elem_shift(et)) ? 1 : 0,
                       reverse ? LAUNCH_REVERSE : 0u, epw, et);
    return hipGetLastError();
}

uint32_t partial_pitch(int A, int k) { return ((uint32_t)(A * (2 * A + 3) * k * k) + 15u) & ~15u; }

// The cheaper of the two partial kernels for this (map, k).  Measured at 65 536 envs (us per launch, window / projection):
// level 6 (216 pairs; 36 / 100 / 196 window cells) 3x3 19.0 / 23.6, 5x5 28.3 / 25.7, 7x7 44.8 / 28.7; config 5 (2 112 pairs;
// 72 / 200 / 392 cells) 46 / 166, 102 / 170, 168 / 193: about 0.07 us per pair against 0.13-0.3 us per window cell, with
// the projection's table build on top.  LLE_PARTIAL_PROJECT=0 / 1 forces it (tests, tuning).
static bool partial_projects(const MapHeader& h, int k, uint32_t n_entities) {
    if (tuning().partial_project >= 0) return tuning().partial_project == 1;
    const uint32_t a_pad = h.A <= 1 ? 1u : (h.A <= 2 ? 2u : (h.A <= 4 ? 4u : (h.A <= 8 ? 8u : 16u)));
    const uint64_t pairs = (uint64_t)(n_entities + h.A) * a_pad, cells = (uint64_t)h.A * (uint32_t)(k * k);
    return pairs <= cells * 3u / 2u + 80u;
}

hipError_t launch_partial_observe(const MapHeader& h, const BatchPtrs& P, int8_t* out, int k, int64_t n_envs, bool per_env_sources,
                                  MapSel M, uint32_t n_entities, bool reverse, hipStream_t stream, const uint8_t* win_sets, uint32_t force_E,
                                  uint32_t* rule_E, uint32_t et) {
    const uint32_t walk = reverse ? LAUNCH_REVERSE : 0u;
    const uint32_t pitch_l = partial_pitch((int)h.A, k);
    int force_old = -1;  // LLE_PARTIAL_KERNEL=window / project: one of the two round-1/2 kernels (kept as cross-checks)
    {   // ---- the lane-per-(env, observer) kernel (partial_lanes_kernel): every map and window size
        const int which = tuning().partial_kernel;  // LLE_PARTIAL_KERNEL: "lanes" (default) | "window" | "project" | "auto" (the round-2 choice)
        const bool old_forced = tuning().partial_project >= 0;
        if (!old_forced && (which == 0 || which == 1)) {
            const uint32_t a_pad = h.A <= 1 ? 1u : (h.A <= 2 ? 2u : (h.A <= 4 ? 4u : (h.A <= 8 ? 8u : 16u)));
            // S = 64 / (E * a_pad) lanes share an observer's rows; a lane's rows must fit its 64-bit set: 8 rows (k <= 8), else 4
            const uint32_t s_min = k <= 8 ? 1u : ((uint32_t)k + 3u) / 4u;
            uint32_t e_max = 64u / a_pad;
            while (e_max > 1 && 64u / (e_max * a_pad) < s_min) e_max >>= 1;
            // E, the environments per batch (S = 64 / (E * a_pad) lanes then share an observer's window rows).  Measured at 65 536
            // envs (tools/lle_prof.py partial --sweep, profiles/r03_partial.md): the best E keeps the batch's block of rows at
            // 6-16 KiB per wavefront -- level 6 (4 observers) 3x3 / 5x5 / 7x7: E = 16 / 8 / 4; 32x32 with 8 observers: 4 / 4 / 2.
            // Rule: the largest E whose rows stay within 16 KiB; above 9 KiB half of that when the lanes stay busy (S <= k and
            // at least 3/4 of the lanes of an observer have a row in every pass).  LLE_PARTIAL_E: tuning override.
            uint32_t E = e_max;
            while (E > 1 && E * pitch_l > 16384u) E >>= 1;
            if (E > 1 && E * pitch_l > 9216u) {
                const uint32_t S2 = 64u / ((E / 2u) * a_pad), rl = ((uint32_t)k + S2 - 1u) / S2;
                if (S2 <= (uint32_t)k && 4u * (uint32_t)k >= 3u * S2 * rl) E >>= 1;
            }
            while (E > 1 && M.envs_per_map && M.envs_per_map % (int64_t)E != 0) E >>= 1;  // (a batch of E environments belongs to one map)
            if (rule_E) *rule_E = E;  // (what the rule says: lle_batch_observe_as times its neighbours once per batch and window size)
            if (force_E >= 1 && force_E <= e_max && !(force_E & (force_E - 1))) E = force_E;
            if (const uint32_t v = (uint32_t)tuning().partial_e) {
                if (v >= 1 && v <= e_max && !(v & (v - 1))) E = v;
            }
            const uint32_t As_l = (uint32_t)agent_stride_of((int)h.A, (int)h.L);
            const uint32_t rec_dwords = As_l / 2 + 1 + h.L + (uint32_t)src_stride_of((int)h.L) / 4u;
            uint32_t tab = (h.off_dyn - h.off_cell_lay + 1023u) & ~1023u;
            if (tab > h.lds_table_bytes) tab = h.lds_table_bytes;
            // (ahead of the wavefronts' areas: the window table of this size -- sets and cell tables in one, tables.h --, or the cell tables and the
            // map's non-empty bitmap)
            const uint32_t win_bytes = win_sets ? win_table_bytes(h.HW) : 0u;
            if (win_sets) tab = win_table_used(h.HW);
            const uint32_t bm_bytes = win_sets ? 0u : partial_bitmap_bytes(h.H, h.W);
            // batches per wavefront: a workgroup copies the tables and builds the bitmap before its first row, so a launch should
            // be ONE round of workgroups (about four per CU): 65 536 envs -> 16 environments per wavefront, i.e. batches = 16 / E
            uint32_t batches = 1;
            while (batches < 16 && (int64_t)E * batches * 2 * 4096 <= n_envs) batches *= 2;
            if (const uint32_t v = (uint32_t)tuning().partial_batches) {
                if (v >= 1 && v <= 64) batches = v;
            }
            uint32_t wpw = 4;
            while (wpw > 1 && M.envs_per_map && M.envs_per_map % (int64_t)(wpw * E * batches) != 0) wpw >>= 1;
            while (batches > 1 && M.envs_per_map && M.envs_per_map % (int64_t)(wpw * E * batches) != 0) batches >>= 1;
            uint32_t priv = E * pitch_l + ((E * batches * rec_dwords * 4u + 15u) & ~15u) + 16u;
            while (wpw > 1 && tab + bm_bytes + wpw * priv > OBS_LDS_LIMIT) wpw >>= 1;
            uint32_t lds = tab + bm_bytes + wpw * priv;
            if (lds < win_bytes) lds = win_bytes;  // (the table is copied in whole rows: the allocation holds the last one)
            const bool fits = lds <= OBS_LDS_LIMIT && (!M.envs_per_map || M.envs_per_map % (int64_t)(wpw * E * batches) == 0);
            if (fits) {
                static LdsGrant granted_l;
                hipError_t e = granted_l.ensure(reinterpret_cast<const void*>(&partial_lanes_kernel), lds);
                if (e != hipSuccess) return e;
                const uint32_t epw = E * batches;
                const uint32_t n_waves = (uint32_t)((n_envs + epw - 1) / epw);
                // written through (`sc1`): the batch's rows are one contiguous block, so lines are shared only at its two ends;
                // measured better at every size incl. 489 MB (32x32 7x7: 114 -> 99 us).  LLE_PARTIAL_WT=0 / 1: tuning override
                int wt = 1;
                if (tuning().partial_wt >= 0) wt = tuning().partial_wt;
                const PartialDims D{(int32_t)h.A, (int32_t)h.L, (int32_t)h.H, (int32_t)h.W, h.off_cell_meta - h.off_cell_lay, h.max_layers};
                LLE_NOTE_OBS(OBSK_PARTIAL_LANES);
                hipLaunchKernelGGL(partial_lanes_kernel, dim3((n_waves + wpw - 1) / wpw), dim3(64 * wpw), lds, stream, P, out, k, pitch_l,
                                   (int64_t)0, n_envs, per_env_sources ? 1 : 0, M, E, batches, tab, wt, D, h.off_cell_lay, walk, win_sets, win_bytes, et);
                return hipGetLastError();
            }
        }
        if (which == 2) force_old = 0;
        if (which == 3) force_old = 1;
    }

    const uint32_t pitch = partial_pitch((int)h.A, k);
    const uint32_t As = (uint32_t)agent_stride_of((int)h.A, (int)h.L);
    if (force_old == 1 || (force_old < 0 && partial_projects(h, k, n_entities))) {
        uint32_t epw = 16;  // level 6 7x7: 4 -> 36.8 us, 8 -> 30.2, 16 -> 28.7 (table build and static decode are per wavefront)
        if (const uint32_t v = (uint32_t)tuning().partial_epw) {
            if (v >= 1 && v <= OBS_ENVS_PER_WAVE && !(v & (v - 1))) epw = v;
        }
        while (epw > 1 && M.envs_per_map && M.envs_per_map % (int64_t)epw != 0) epw >>= 1;
        const uint32_t ent_cap = n_entities;  // an upper bound computed by the host from the map(s)
        const uint32_t rec_dwords = As / 2 + 1 + h.L + (uint32_t)src_stride_of((int)h.L) / 4u;
        const uint32_t shared = (16u + ent_cap * 4u + 15u) & ~15u, priv = pitch + ((epw * rec_dwords * 4u + 15u) & ~15u);
        uint32_t wpw = cap_wpw(4, M, epw);
        while (wpw > 1 && shared + wpw * priv > OBS_LDS_LIMIT) wpw >>= 1;
        const uint32_t lds = shared + wpw * priv;
        if (lds > OBS_LDS_LIMIT) return hipErrorInvalidValue;
        static LdsGrant granted_p;
        hipError_t e = granted_p.ensure(reinterpret_cast<const void*>(&partial_project_kernel), lds);
        if (e != hipSuccess) return e;
        const uint32_t n_waves = (uint32_t)((n_envs + epw - 1) / epw);
        LLE_NOTE_OBS(OBSK_PARTIAL_PROJECT);
        hipLaunchKernelGGL(partial_project_kernel, dim3((n_waves + wpw - 1) / wpw), dim3(64 * wpw), lds, stream, P, out, k, pitch,
                           (int64_t)0, n_envs, per_env_sources ? 1 : 0, M, epw, ent_cap, walk, et);
        return hipGetLastError();
    }
    // envs per wavefront: the chain of one env (clear, agents, cells, stream) is latency, so fewer envs per wave = more
    // waves in flight; measured at 65 536 envs, level 6 7x7: 16 -> 58 us, 8 -> 46, 4 -> 45, 2 -> 48; 32x32 maps (21 KB of
    // tables per workgroup) 7x7: 166 / 172 / 185 / 209 us, 3x3: 54 / 48 / 52 / 63.  LLE_PARTIAL_EPW: tuning override.
    uint32_t epw = 8;
    if (const uint32_t v = (uint32_t)tuning().partial_epw) {
        if (v >= 1 && v <= OBS_ENVS_PER_WAVE && !(v & (v - 1))) epw = v;
    }
    while (epw > 1 && M.envs_per_map && M.envs_per_map % (int64_t)epw != 0) epw >>= 1;
    const uint32_t priv = pitch + ((epw * (As / 2 + 1 + h.L) * 4u + 15u) & ~15u) + (((h.HW + 1u) / 2u * 4u + 15u) & ~15u);
    const uint32_t shared = h.lds_table_bytes + (((uint32_t)(h.A * k * k) * 4u + 15u) & ~15u);
    uint32_t wpw = cap_wpw(4, M, epw);
    while (wpw > 1 && shared + wpw * priv > OBS_LDS_LIMIT) wpw >>= 1;
    const uint32_t lds = shared + wpw * priv;
    if (lds > OBS_LDS_LIMIT) return hipErrorInvalidValue;
    static LdsGrant granted;
    hipError_t e = granted.ensure(reinterpret_cast<const void*>(&partial_observe_kernel), lds);
    if (e != hipSuccess) return e;
    const uint32_t n_waves = (uint32_t)((n_envs + epw - 1) / epw);
    LLE_NOTE_OBS(OBSK_PARTIAL_WINDOW);
    hipLaunchKernelGGL(partial_observe_kernel, dim3((n_waves + wpw - 1) / wpw), dim3(64 * wpw), lds, stream, P, out, k, pitch,
                       (int64_t)0, n_envs, per_env_sources ? 1 : 0, M, epw, walk, et);
    return hipGetLastError();
}

hipError_t launch_state_observe(const MapHeader& h, const BatchPtrs& P, float* out, int normalize, int64_t n_envs, hipStream_t stream) {
    const int64_t total = n_envs * (int64_t)(3 * h.A + h.G);
    if (total == 0) return hipSuccess;
    LLE_NOTE_OBS(OBSK_STATE);
    hipLaunchKernelGGL(state_observe_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, P, out, normalize, n_envs);
    return hipGetLastError();
}

hipError_t launch_env_outputs(const MapHeader& h, const BatchPtrs& P, const EnvOutputs& O, int64_t n_envs, MapSel M, hipStream_t stream) {
    const int64_t total = n_envs * (int64_t)h.A;
    if (total == 0) return hipSuccess;
    LLE_NOTE_OBS(OBSK_ENV_OUTPUTS);
    hipLaunchKernelGGL(env_outputs_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, P, O, n_envs, M);
    return hipGetLastError();
}

hipError_t launch_avail(const MapHeader& h, const BatchPtrs& P, uint8_t* out, int walkable_lasers, int64_t n_envs, bool per_env_sources,
                        MapSel M, hipStream_t stream) {
    const int64_t total = n_envs * (int64_t)h.A;
    if (total == 0) return hipSuccess;
    LLE_NOTE_OBS(OBSK_AVAIL);
    hipLaunchKernelGGL(avail_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, P, out, walkable_lasers, n_envs,
                       per_env_sources ? 1 : 0, M);
    return hipGetLastError();
}

}  // namespace lle
