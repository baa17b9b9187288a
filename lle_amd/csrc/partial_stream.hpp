// partial_stream.hpp -- the partial k x k observation (python/lle/observations.py:312-369) written by the STEP kernel's launch
// (step_kernel MODE 9), from the hand-over records the state machine leaves in LDS: `BatchedLLE(obs_type="partial...")` then
// steps in one launch instead of two.  The writer is the lane-per-(environment, observer) scheme of observers.hip
// partial_lanes_kernel (same bytes: tests/test_gpu_observers.py compares both with the oracle), fed from the wavefront's own
// records instead of the packed state in global memory:
//   record of an environment (step_kernel.hpp): [0 | beam words[L] | ~gem bits | packed position (i | j << 8) of each agent]
// A wavefront holds EPW environments; they are written in batches of E (E x a_pad x S = 64 lanes: S lanes share an observer's
// window), every batch one contiguous block of E rows -- cleared, patched and streamed as a whole.
#pragma once
#include "obs_stream.hpp"

namespace lle {

// The non-empty bitmap of the map -- one bit per cell that holds a wall, a source, an exit, a gem or a laser tile -- with 8 empty
// cells of margin on every side: (H + 16) rows of RW words, cell (i, j) at row i + 8, bit j + 8.  A window row is then k bits of it
// at (i0 + wi + 8, j0 + 8) and no bounds test exists anywhere.  Built once per workgroup; `bm` must be zeroed, a barrier before and
// after (the caller's).
__host__ __device__ inline uint32_t partial_bitmap_row_words(uint32_t W) { return (W + 16u + 31u) / 32u + 1u; }
__host__ __device__ inline uint32_t partial_bitmap_bytes(uint32_t H, uint32_t W) { return ((H + 16u) * partial_bitmap_row_words(W) * 4u + 15u) & ~15u; }
__device__ __forceinline__ void partial_bitmap_fill(uint32_t* bm, const uint64_t* cell_lay, const uint32_t* cell_meta, int H, int W) {
    const uint32_t RW = partial_bitmap_row_words((uint32_t)W);
    for (uint32_t c = threadIdx.x; c < (uint32_t)(H * W); c += blockDim.x) {
        if ((cell_meta[c] & 7u) != K_FLOOR || cell_lay[c] != 0ull) {
            const uint32_t i = c / (uint32_t)W, j = c - i * (uint32_t)W;
            atomicOr(&bm[(i + 8u) * RW + ((j + 8u) >> 5)], 1u << ((j + 8u) & 31u));
        }
    }
}

// rows: E x pitch bytes of LDS private to the wavefront, + 16 bytes behind them that nobody reads (where the writes of a cell that do
// not apply go).  records: the wavefront's hand-over records, scr_stride words apart.  colours: colour byte of every beam word (LDS).
template <bool WT>
__device__ __forceinline__ void write_partial(int A, int L, int W, int k, uint32_t pitch, uint32_t E, uint32_t max_layers, const uint64_t* cell_lay,
                                              const uint32_t* cell_meta, const uint32_t* bm, const uint8_t* colours, int8_t* rows,
                                              const uint32_t* records, uint32_t scr_stride, int8_t* __restrict__ out, int64_t env0,
                                              int64_t n_here_all, uint32_t lane) {
    const uint32_t RW = partial_bitmap_row_words((uint32_t)W);
    const uint32_t logA = A <= 1 ? 0u : (A <= 2 ? 1u : (A <= 4 ? 2u : (A <= 8 ? 3u : 4u)));
    const uint32_t S = 64u / (E << logA);                 // lanes per (env, observer); the launcher keeps E << logA <= 64
    const uint32_t e_slot = lane / (S << logA), a = (lane / S) & ((1u << logA) - 1u), s = lane % S;
    const uint32_t SBL = k <= 8 ? 3u : 4u;                // a window row takes 8 (k <= 8) or 16 bits of the lane's 64-bit set
    const int centre = k / 2;
    const uint32_t kk = (uint32_t)(k * k), layers = (uint32_t)(2 * A + 3), n_chunks = pitch / 16u;
    uint4* rows16 = reinterpret_cast<uint4*>(rows);
    int8_t* dummy = rows + E * pitch;
    const int WALL = A, LASER_0 = A + 1, GEM = 2 * A + 1, EXIT = 2 * A + 2;   // observations.py:318-323
    // layer of the one static byte of a cell, by kind (0xFF: none): FLOOR, WALL, VOID, EXIT | GEM, SOURCE (wall_pos holds the sources too)
    const uint32_t lt_lo = 0xFFu | ((uint32_t)WALL << 8) | (0xFFu << 16) | ((uint32_t)EXIT << 24), lt_hi = (uint32_t)GEM | ((uint32_t)WALL << 8) | 0xFFFF0000u;
    const bool two_layers = max_layers > 1u;

    for (int64_t b0 = 0; b0 < n_here_all; b0 += E) {
        const int64_t left = n_here_all - b0;
        const uint32_t n_here = left < (int64_t)E ? (uint32_t)left : E;
        for (uint32_t c = lane; c < n_here * n_chunks; c += 64) rows16[c] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();  // LDS operations of a wavefront execute in order: everything below lands after the clears
        const bool live = e_slot < n_here && a < (uint32_t)A;
        const uint32_t* rec = records + ((uint32_t)b0 + (live ? e_slot : 0u)) * scr_stride;
        const uint32_t* posw = rec + L + 2;   // packed positions, one word per agent
        const uint32_t pa = posw[live ? a : 0u];
        const int i0 = (int)(pa & 0xFFu) - centre, j0 = (int)((pa >> 8) & 0xFFu) - centre;   // the window's origin on the map
        int8_t* mine = rows + __umul24(live ? e_slot : 0u, pitch) + __umul24(a, layers * kk);   // observer a's block of this env's row
        // ---- other agents (dead ones included: agents_positions): lane s takes agents s, s + S, ...
        if (live)
            for (uint32_t a2 = s; a2 < (uint32_t)A; a2 += S) {
                const uint32_t p2 = posw[a2];
                const uint32_t dy = (uint32_t)((int)(p2 & 0xFFu) - i0), dx = (uint32_t)((int)((p2 >> 8) & 0xFFu) - j0);
                if (dy < (uint32_t)k && dx < (uint32_t)k) mine[__umul24(a2, kk) + __umul24(dy, (uint32_t)k) + dx] = 1;
            }
        // ---- this lane's share of the window's non-empty cells (observers.hip partial_lanes_kernel: the same split, the same sets)
        uint32_t todo2[2] = {0u, 0u};
        const bool diag = k <= 8;
        const uint32_t wi_base = diag ? 0u : s, wi_step = diag ? 1u : S, RH = 32u >> SBL;
        if (live) {
            const uint32_t off = (uint32_t)(j0 + 8);   // >= 1: bit of the window's first column in a bitmap row
            const uint32_t rep = S >= 8 ? 0x01u : (S == 4 ? 0x11u : (S == 2 ? 0x55u : 0xFFu));   // every S-th bit of a row
            const uint32_t kmask = (1u << k) - 1u;
            uint32_t r = 0;
            for (uint32_t wi = wi_base; wi < (uint32_t)k; wi += wi_step, r++) {
                const uint32_t* rowp = bm + __umul24((uint32_t)(i0 + (int)wi + 8), RW) + (off >> 5);
                uint32_t bits = __funnelshift_r(rowp[0], rowp[1], off & 31u) & kmask;   // v_alignbit_b32
                if (diag) bits &= rep << ((s - wi) & (S - 1u));
                const uint32_t sh = (r & (RH - 1u)) << SBL;
                if (r < RH) todo2[0] |= bits << sh;
                else todo2[1] |= bits << sh;
            }
        }
        // One non-empty cell per pass, no branch inside: its static byte (wall / exit / uncollected gem), the two laser layers
        // World.lasers() exposes when lit, the -1 of a source -- each a store whose address is the byte, or `dummy`.
        // (Write order = the reference's, observations.py:347-359; all four commute.)
        const int cell0 = i0 * W + j0;
        const uint32_t not_gems = rec[L + 1];   // ~collected bits
        // Two dependent LDS round trips per cell -- (meta, layers) of the cell, then the beam word / colour bytes they name -- and nothing
        // else for the wavefront to do in between: the loop is software-pipelined, the first round trip of the NEXT cell issued ahead of
        // the second one of the current cell (LDS returns in order: one wait per pass instead of two).
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t todo = todo2[half];
            uint32_t wi_n = 0, wj_n = 0, meta_n = 0;
            uint64_t lay_n = 0;
            bool have = todo != 0u;
            if (have) {
                const uint32_t b = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                const uint32_t r = (b >> SBL) + (half ? RH : 0u);
                wj_n = b & ((1u << SBL) - 1u);
                wi_n = mad24(r, wi_step, wi_base);
                const uint32_t cell = (uint32_t)cell0 + mad24(wi_n, (uint32_t)W, wj_n);
                meta_n = cell_meta[cell];
                lay_n = cell_lay[cell];
            }
            while (have) {
                const uint32_t meta = meta_n, wi = wi_n, wj = wj_n;
                const uint64_t lay = lay_n;
                have = todo != 0u;
                if (have) {   // the next cell's first round trip
                    const uint32_t b = (uint32_t)__builtin_ctz(todo);
                    todo &= todo - 1u;
                    const uint32_t r = (b >> SBL) + (half ? RH : 0u);
                    wj_n = b & ((1u << SBL) - 1u);
                    wi_n = mad24(r, wi_step, wi_base);
                    const uint32_t cell = (uint32_t)cell0 + mad24(wi_n, (uint32_t)W, wj_n);
                    meta_n = cell_meta[cell];
                    lay_n = cell_lay[cell];
                }
                const uint32_t kind = meta & 7u, idx = (meta >> 3) & 31u;
                const uint32_t l0 = (uint32_t)lay & 0xFFFFu;   // World.lasers(): the two outer layers of a cell
                const uint32_t w0 = (l0 >> 1) & 31u, o0 = (l0 >> 6) & 31u;
                const uint32_t src = kind == K_SOURCE ? idx : 0u;   // idx = first beam word of a source cell (gem index otherwise)
                const uint32_t m0 = rec[1u + w0];
                const uint32_t c0 = colours[w0], cs = colours[src];
                const uint32_t lt = ((kind < 4u ? lt_lo : lt_hi) >> ((kind & 3u) * 8u)) & 0xFFu;
                const bool en0 = lt != 0xFFu && !(kind == K_GEM && !((not_gems >> idx) & 1u));
                const bool en1 = (l0 & LAY_VALID) && ((m0 >> o0) & 1u);
                int8_t* cp = mine + mad24(wi, (uint32_t)k, wj);
                if (two_layers) {
                    const uint32_t l1 = (uint32_t)(lay >> 16) & 0xFFFFu, w1 = (l1 >> 1) & 31u, o1 = (l1 >> 6) & 31u;
                    const uint32_t m1 = rec[1u + w1], c1 = colours[w1];
                    const bool en2 = (l1 & LAY_VALID) && ((m1 >> o1) & 1u);
                    *(en2 ? cp + __umul24((uint32_t)LASER_0 + c1, kk) : dummy) = 1;
                }
                *(en0 ? cp + __umul24(lt, kk) : dummy) = 1;
                *(en1 ? cp + __umul24((uint32_t)LASER_0 + c0, kk) : dummy) = 1;
                *(kind == K_SOURCE ? cp + __umul24((uint32_t)LASER_0 + cs, kk) : dummy) = -1;
            }
        }
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + (uint64_t)(env0 + b0) * pitch);
        stream_row<WT>(dst, rows16, 0u, n_here * n_chunks, lane);
        wave_sync();  // the next batch clears the rows: after these reads (in order, same wavefront)
    }
}

}  // namespace lle
