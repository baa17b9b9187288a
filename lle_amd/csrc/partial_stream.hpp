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
        if (meta_kind(cell_meta[c]) != K_FLOOR || cell_lay[c] != 0ull) {
            const uint32_t i = c / (uint32_t)W, j = c - i * (uint32_t)W;
            atomicOr(&bm[(i + 8u) * RW + ((j + 8u) >> 5)], 1u << ((j + 8u) & 31u));
        }
    }
}

// ---- one lane's share of one observer's window, common to the step kernel's writer (write_partial below) and observers.hip partial_lanes_kernel.
// The two differ in where an environment's dynamic state sits; `Rec` hides that:
struct PartialRecStep {    // the step kernel's hand-over record: [0 | beam words[L] | ~gem bits | packed position (i | j << 8) of each agent]
    const uint32_t* rec;
    const uint8_t* colours;   // colour byte of every beam word (LDS, one copy per workgroup)
    uint32_t L;
    __device__ __forceinline__ uint32_t pos(uint32_t a) const { return rec[L + 2u + a]; }
    __device__ __forceinline__ uint32_t beam(uint32_t w) const { return rec[1u + w]; }
    __device__ __forceinline__ uint32_t colour(uint32_t w) const { return colours[w]; }
    __device__ __forceinline__ uint32_t gems_left() const { return rec[L + 1u]; }   // bit g: gem g is still there
};
struct PartialRecPacked {  // partial_lanes_kernel's record: pos u16[As] | gems | beams[L] | colour bytes (the env's own, or the map's)
    const uint8_t* rec8;
    uint32_t half_as, colour_at;   // As / 2 dwords of positions; byte offset of the colour bytes
    __device__ __forceinline__ uint32_t pos(uint32_t a) const { return reinterpret_cast<const uint16_t*>(rec8)[a]; }
    __device__ __forceinline__ uint32_t beam(uint32_t w) const { return reinterpret_cast<const uint32_t*>(rec8)[half_as + 1u + w]; }
    __device__ __forceinline__ uint32_t colour(uint32_t w) const { return rec8[colour_at + w]; }
    __device__ __forceinline__ uint32_t gems_left() const { return ~reinterpret_cast<const uint32_t*>(rec8)[half_as]; }
};
struct PartialGeo {        // uniform over a launch
    int A, W, k;
    uint32_t kk, S, RW;    // k * k; lanes per (environment, observer); words of a bitmap row
    bool two_layers;       // some cell of the map lies under two beams (World.lasers() exposes two layers)
};

// Every byte one non-empty cell gives a window (observations.py:347-359: gem / exit / wall, the laser layers World.lasers() exposes when
// lit, -1 at a source; the writes commute): each a store whose address is the byte, or `dummy` when it does not apply -- no branch.
template <class Rec>
__device__ __forceinline__ void partial_eval_cell(const PartialGeo& G, const Rec& R, uint32_t gems_left, uint32_t meta, uint64_t lay, int8_t* cp, int8_t* dummy) {
    const uint32_t WALL = (uint32_t)G.A, LASER_0 = (uint32_t)G.A + 1u, GEM = 2u * (uint32_t)G.A + 1u, EXIT = 2u * (uint32_t)G.A + 2u;   // observations.py:318-323
    // layer of the one static byte of a cell, by kind (0xFF: none): FLOOR, WALL, VOID, EXIT | GEM, SOURCE (wall_pos holds the sources too)
    const uint32_t lt_lo = 0xFFu | (WALL << 8) | (0xFFu << 16) | (EXIT << 24), lt_hi = GEM | (WALL << 8) | 0xFFFF0000u;
    const uint32_t kind = meta_kind(meta), idx = meta_index(meta);
    const uint32_t l0 = lay_entry(lay, 0);   // World.lasers(): the two outer layers of a cell
    const uint32_t w0 = lay_word(l0), o0 = lay_bit(l0);
    const uint32_t src = kind == K_SOURCE ? idx : 0u;   // idx = first beam word of a source cell (gem index otherwise)
    const uint32_t m0 = R.beam(w0);
    const uint32_t c0 = R.colour(w0), cs = R.colour(src);
    const uint32_t lt = ((kind < 4u ? lt_lo : lt_hi) >> ((kind & 3u) * 8u)) & 0xFFu;
    const bool en0 = lt != 0xFFu && !(kind == K_GEM && !((gems_left >> idx) & 1u));
    const bool en1 = (l0 & LAY_VALID) && ((m0 >> o0) & 1u);
    if (G.two_layers) {
        const uint32_t l1 = lay_entry(lay, 1), w1 = lay_word(l1), o1 = lay_bit(l1);
        const uint32_t m1 = R.beam(w1), c1 = R.colour(w1);
        const bool en2 = (l1 & LAY_VALID) && ((m1 >> o1) & 1u);
        *(en2 ? cp + __umul24(LASER_0 + c1, G.kk) : dummy) = 1;
    }
    *(en0 ? cp + __umul24(lt, G.kk) : dummy) = 1;
    *(en1 ? cp + __umul24(LASER_0 + c0, G.kk) : dummy) = 1;
    *(kind == K_SOURCE ? cp + __umul24(LASER_0 + cs, G.kk) : dummy) = -1;
}

// set bits -> bytes of value 1 at base[bit index]
__device__ __forceinline__ void partial_bits_to_bytes(int8_t* base, uint64_t set) {
    uint32_t lo = (uint32_t)set, hi = (uint32_t)(set >> 32);
    while (lo) { base[__builtin_ctz(lo)] = 1; lo &= lo - 1u; }
    while (hi) { base[32u + (uint32_t)__builtin_ctz(hi)] = 1; hi &= hi - 1u; }
}

// The lane's share of a window in the layout of the window sets (bit wi * k + wj): the S lanes of an observer split it DIAGONALLY -- lane s takes
// the cells with (wi + wj) mod S == s -- so that a row of walls or a beam, the runs maps are made of, is spread over all of them.
__device__ __forceinline__ uint64_t partial_share_mask(uint32_t k, uint32_t S, uint32_t s) {
    const uint32_t rep = S >= 8 ? 0x01u : (S == 4 ? 0x11u : (S == 2 ? 0x55u : 0xFFu)), kmask = (1u << k) - 1u;
    uint64_t m = 0;
    for (uint32_t wi = 0; wi < k; wi++) m |= (uint64_t)((rep << (((s - wi) & (S - 1u)) & 31u)) & kmask) << (wi * k);  // (a share beyond bit 31 repeats an earlier one: harmless, the writes are idempotent)
    return m;
}

// One lane = share s of observer a of one environment: its agents, then its share of the window's non-empty cells.
//   sets != NULL (k = 3, 5, 7): the two window sets of the observer's cell (tables.h), cut to the lane's share -- walls are bits turned into
//     bytes, only gems / exits / laser tiles / sources go through the cell tables;
//   sets == NULL: every non-empty cell of the window, found in the map's non-empty bitmap `bm` row by row, goes through the cell tables
//     (windows up to 8 x 8 split diagonally, larger ones by rows: a lane's rows must fit its 64-bit set, at most four of 16 bits).
template <class Rec>
__device__ __forceinline__ void partial_window(const PartialGeo& G, const Rec& R, bool live, uint32_t a, uint32_t s, int8_t* mine, int8_t* dummy,
                                               const uint64_t* cell_lay, const uint32_t* cell_meta, const uint32_t* bm, const uint64_t* sets, uint64_t share) {
    const uint32_t k = (uint32_t)G.k, kk = G.kk, S = G.S;
    const int centre = G.k / 2, W = G.W;
    const uint32_t pa = R.pos(live ? a : 0u);
    const int i0 = (int)(pa & 0xFFu) - centre, j0 = (int)((pa >> 8) & 0xFFu) - centre;   // the window's origin on the map
    // ---- other agents (dead ones included: agents_positions): lane s takes agents s, s + S, ...
    if (live)
        for (uint32_t a2 = s; a2 < (uint32_t)G.A; a2 += S) {
            const uint32_t p2 = R.pos(a2);
            const uint32_t dy = (uint32_t)((int)(p2 & 0xFFu) - i0), dx = (uint32_t)((int)((p2 >> 8) & 0xFFu) - j0);
            if (dy < k && dx < k) mine[__umul24(a2, kk) + __umul24(dy, k) + dx] = 1;
        }
    const int cell0 = i0 * W + j0;
    const uint32_t gems_left = R.gems_left();
    // `todo2`: the cells that go through the cell tables, as two dwords of a 64-bit set; `decode(b)` -> (map cell, byte offset in a window layer)
    uint32_t todo2[2] = {0u, 0u};
    const uint32_t SBL = k <= 8 ? 3u : 4u;                // bitmap path: a window row takes 8 (k <= 8) or 16 bits of the lane's set
    const bool diag = k <= 8;
    const uint32_t wi_base = diag ? 0u : s, wi_step = diag ? 1u : S, RH = 32u >> SBL;
    if (sets) {
        uint64_t walls = 0, dyn = 0;
        if (live) {
            const uint64_t* e = sets + 2u * ((pa & 0xFFu) * (uint32_t)W + ((pa >> 8) & 0xFFu));
            walls = e[0] & share; dyn = e[1] & share;
        }
        partial_bits_to_bytes(mine + __umul24((uint32_t)G.A, kk), walls);   // WALL layer
        todo2[0] = (uint32_t)dyn; todo2[1] = (uint32_t)(dyn >> 32);
    } else if (live) {
        const uint32_t off = (uint32_t)(j0 + 8);   // >= 1: bit of the window's first column in a bitmap row
        const uint32_t rep = S >= 8 ? 0x01u : (S == 4 ? 0x11u : (S == 2 ? 0x55u : 0xFFu));   // every S-th bit of a row
        const uint32_t kmask = (1u << k) - 1u;
        uint32_t r = 0;
        for (uint32_t wi = wi_base; wi < k; wi += wi_step, r++) {
            const uint32_t* rowp = bm + __umul24((uint32_t)(i0 + (int)wi + 8), G.RW) + (off >> 5);
            uint32_t bits = __funnelshift_r(rowp[0], rowp[1], off & 31u) & kmask;   // v_alignbit_b32
            if (diag) bits &= rep << (((s - wi) & (S - 1u)) & 31u);
            const uint32_t sh = (r & (RH - 1u)) << SBL;
            if (r < RH) todo2[0] |= bits << sh;
            else todo2[1] |= bits << sh;
        }
    }
    // (wi, wj) of bit b of half `half`: sets -- b = wi * k + wj, wi by a multiply-shift exact for b < 64 and k = 3, 5, 7; bitmap -- row-major with
    // 8 / 16 bits per row
    const uint32_t inv_k = k == 3 ? 0x5556u : (k == 5 ? 0x3334u : 0x2493u);   // ceil(2^16 / k)
    auto decode = [&](uint32_t b, int half, uint32_t& cell, uint32_t& cpo) {
        uint32_t wi, wj;
        if (sets) {
            const uint32_t bb = b + (half ? 32u : 0u);
            wi = __umul24(bb, inv_k) >> 16;
            wj = bb - __umul24(wi, k);
            cpo = bb;
        } else {
            const uint32_t r = (b >> SBL) + (half ? RH : 0u);
            wj = b & ((1u << SBL) - 1u);
            wi = mad24(r, wi_step, wi_base);   // (one full-rate v_mad_u32_u24 each: the compiler made quarter-rate 32-bit multiplies of __umul24 here)
            cpo = mad24(wi, k, wj);
        }
        cell = (uint32_t)cell0 + mad24(wi, (uint32_t)W, wj);
    };
    // One cell per pass, no branch inside.  Two dependent LDS round trips per cell -- (meta, layers) of the cell, then the beam word / colour
    // bytes they name -- and nothing else for the wavefront to do in between: the loop is software-pipelined, the first round trip of the NEXT
    // cell issued ahead of the second one of the current cell (LDS returns in order: one wait per pass instead of two).
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t todo = todo2[half];
        uint32_t cpo_n = 0, meta_n = 0;
        uint64_t lay_n = 0;
        bool have = todo != 0u;
        if (have) {
            uint32_t cell;
            decode((uint32_t)__builtin_ctz(todo), half, cell, cpo_n);
            todo &= todo - 1u;
            meta_n = cell_meta[cell];
            lay_n = cell_lay[cell];
        }
        while (have) {
            const uint32_t meta = meta_n, cpo = cpo_n;
            const uint64_t lay = lay_n;
            have = todo != 0u;
            if (have) {   // the next cell's first round trip
                uint32_t cell;
                decode((uint32_t)__builtin_ctz(todo), half, cell, cpo_n);
                todo &= todo - 1u;
                meta_n = cell_meta[cell];
                lay_n = cell_lay[cell];
            }
            partial_eval_cell(G, R, gems_left, meta, lay, mine + cpo, dummy);
        }
    }
}

// rows: E x pitch bytes of LDS private to the wavefront, + 16 bytes behind them that nobody reads (where the writes of a cell that do
// not apply go).  records: the wavefront's hand-over records, scr_stride words apart.  colours: colour byte of every beam word (LDS).
// sets: the window sets of this window size in LDS (tables.h), or NULL: then `bm` is the map's non-empty bitmap.
template <bool WT>
__device__ __forceinline__ void write_partial(int A, int L, int W, int k, uint32_t pitch, uint32_t E, uint32_t max_layers, const uint64_t* cell_lay,
                                              const uint32_t* cell_meta, const uint32_t* bm, const uint8_t* colours, int8_t* rows,
                                              const uint32_t* records, uint32_t scr_stride, int8_t* __restrict__ out, int64_t env0,
                                              int64_t n_here_all, uint32_t lane, const uint64_t* sets = nullptr, uint32_t et = OBS_I8) {
    const uint32_t logA = A <= 1 ? 0u : (A <= 2 ? 1u : (A <= 4 ? 2u : (A <= 8 ? 3u : 4u)));
    const uint32_t S = 64u / (E << logA);                 // lanes per (env, observer); the launcher keeps E << logA <= 64
    const uint32_t e_slot = lane / (S << logA), a = (lane / S) & ((1u << logA) - 1u), s = lane % S;
    const uint32_t kk = (uint32_t)(k * k), layers = (uint32_t)(2 * A + 3), n_chunks = pitch / 16u;
    const PartialGeo G{A, W, k, kk, S, partial_bitmap_row_words((uint32_t)W), max_layers > 1u};
    const uint64_t share = sets ? partial_share_mask((uint32_t)k, S, s) : 0ull;
    uint4* rows16 = reinterpret_cast<uint4*>(rows);
    int8_t* dummy = rows + E * pitch;

    for (int64_t b0 = 0; b0 < n_here_all; b0 += E) {
        const int64_t left = n_here_all - b0;
        const uint32_t n_here = left < (int64_t)E ? (uint32_t)left : E;
        for (uint32_t c = lane; c < n_here * n_chunks; c += 64) rows16[c] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();  // LDS operations of a wavefront execute in order: everything below lands after the clears
        const bool live = e_slot < n_here && a < (uint32_t)A;
        const PartialRecStep R{records + ((uint32_t)b0 + (live ? e_slot : 0u)) * scr_stride, colours, (uint32_t)L};
        int8_t* mine = rows + __umul24(live ? e_slot : 0u, pitch) + __umul24(a, layers * kk);   // observer a's block of this env's row
        partial_window(G, R, live, a, s, mine, dummy, cell_lay, cell_meta, bm, sets, share);
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + (((uint64_t)(env0 + b0) * pitch) << obs_elem_shift(et)));
        if (et != OBS_I8) stream_wide<WT>(dst, rows, n_here * n_chunks, et, lane);  // (the batch's element type: widened at the store, obs_stream.hpp)
        else stream_row<WT>(dst, rows16, 0u, n_here * n_chunks, lane);
        wave_sync();  // the next batch clears the rows: after these reads (in order, same wavefront)
    }
}

}  // namespace lle
