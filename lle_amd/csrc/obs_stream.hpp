// obs_stream.hpp -- device helpers shared by kernels.hip and observers.hip: wavefront-local LDS hand-over, the
// table copy into LDS, and the observation streamer (patch the wave's LDS copy of the static observation, stream it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tables.h"

namespace lle {

// The workgroup IS one wavefront, and a wave's LDS operations execute in issue order, so lanes of the wave may hand
// data to each other through LDS without `s_barrier` and without the `s_waitcnt vmcnt(0)` that `__syncthreads()`
// emits (which would stall every environment of phase 2 on the completion of the previous environment's global
// stores).  wave_sync() only pins the compiler's ordering.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- which block of environments a workgroup serves.
// Workgroups are dispatched round-robin over the eight XCDs (workgroup b runs on XCD b % 8), and each XCD writes through
// its own L2.  Serving blocks in dispatch order interleaves the eight XCDs' rows at workgroup granularity; giving XCD x
// the x-th contiguous eighth of the blocks instead keeps each L2's write stream on one contiguous range of HBM
// addresses.  Measured with a pure row fill (tools/ceiling/hbm_write_ceiling.hip, 1 920-byte rows, buffers larger than the
// Infinity Cache): 5.53 -> 5.79 TB/s at 262 144 rows, 5.56 -> 5.98 at 524 288; no change while the rows fit the cache.
// A bijection of [0, n_blocks) for every n_blocks.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n_blocks) {
    const uint32_t x = b & 7u, q = n_blocks >> 3, r = n_blocks & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
// ... and in which ORDER the launch walks them.  A buffer larger than the 256 MB Infinity Cache that is rewritten launch after
// launch in the same order never finds its lines there: what the previous launch left in the cache is its tail, and this launch
// starts at the head, evicting that tail (to HBM) on its way.  Walked alternately up and down (LAUNCH_REVERSE on every other
// launch: workgroups are dispatched in index order, so mirroring the block index mirrors the order in time), each launch begins
// with the lines the previous one ended with, and that cache-sized share of the rows is rewritten in place.  A pure fill of the
// step kernel's shape (tools/ceiling/pingpong_probe.hip): 480 MB 74.6 -> 69.7 us, 720 MB 129.3 -> 107.4, 1.3 GB 209 -> 198.5,
// 2 GB 324 -> 319; nothing below the cache size.  Results do not depend on it (environments are independent).
__device__ __forceinline__ uint32_t xcd_block_dir(uint32_t b, uint32_t n_blocks, uint32_t flags) {
    const uint32_t blk = xcd_block(b, n_blocks);
    return (flags & LAUNCH_REVERSE) ? n_blocks - 1u - blk : blk;
}

// ... and in which order a WAVEFRONT walks its own rows (round 4, tools/ceiling/alloc_probe3.hip).  Every wavefront owns E consecutive
// rows and writes them in turn; with all of them starting at their row 0, the addresses that the thousands of resident wavefronts
// write at the same moment are a regular lattice with stride E rows (level 6: 16 x 1 920 B = 240 cache lines, a multiple of 16), and
// on buffers whose physical pages are contiguous -- hipDeviceMallocContiguous, and most of what hipMalloc hands out on some boxes --
// that lattice loads the memory channels unevenly: 5.75-5.9 TB/s where a memset reaches 6.6.  Starting wavefront w at its row
// (w mod E) makes the stride E + 1 rows (255 lines: odd): 6.05-6.25 TB/s on the same buffers, nothing lost on the others (a random
// start row does NOT help; rows per wavefront 4 / 2 / 1 reach 6.05 / 6.15 / 6.45 but cost a state machine per fewer environments).
// LAUNCH_ROTATE_ROWS; results do not depend on it.
__device__ __forceinline__ uint32_t row_rotation(uint32_t wave_id, uint32_t rows_per_wave, int64_t n_here, uint32_t flags) {
    const uint32_t r = wave_id & (rows_per_wave - 1u);  // (rows_per_wave is a power of two)
    return ((flags & LAUNCH_ROTATE_ROWS) && (int64_t)r < n_here && n_here == (int64_t)rows_per_wave) ? r : 0u;
}
__device__ __forceinline__ uint32_t rotated(uint32_t k, uint32_t rot, uint32_t n) {
    const uint32_t kk = k + rot;
    return kk >= n ? kk - n : kk;
}

// ---- static tables -> LDS, once per workgroup (section offsets are those of the blob).  The section is a whole
// number of 1 KiB rows; every thread requests all of its rows (up to four) before the first LDS write.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Both copies below request up to EIGHT rows per thread (32 rows per pass of a four-wavefront workgroup) before the first LDS write:
// a pass is a global round trip in front of every workgroup's first instruction of real work.  With four rows per thread the
// per-env-sources section of level 6 (12 + 5 rows) took a second pass for ONE row -- 2.3 us on every step with per-environment
// sources (in-kernel stamps: tables in LDS 3.95 us after entry against 1.61; profiles/r04_pes_tax.md) --, config 5 (33 rows) three.
// (named scalars, not an array: an indexed private array here ended up in scratch memory)
#define LLE_COPY_ROWS(SRC)                                                                                              \
    for (uint32_t r0 = wave_in_wg; r0 < rows; r0 += 8 * waves_per_wg) {                                                \
        const uint32_t r1 = r0 + waves_per_wg, r2 = r1 + waves_per_wg, r3 = r2 + waves_per_wg, r4 = r3 + waves_per_wg, \
                       r5 = r4 + waves_per_wg, r6 = r5 + waves_per_wg, r7 = r6 + waves_per_wg;                          \
        const u32x4 z = {0, 0, 0, 0};                                                                                   \
        u32x4 v0 = SRC(r0), v1 = z, v2 = z, v3 = z, v4 = z, v5 = z, v6 = z, v7 = z;                                      \
        if (r1 < rows) v1 = SRC(r1);                                                                                    \
        if (r2 < rows) v2 = SRC(r2);                                                                                    \
        if (r3 < rows) v3 = SRC(r3);                                                                                    \
        if (r4 < rows) {                                                                                                \
            v4 = SRC(r4);                                                                                               \
            if (r5 < rows) v5 = SRC(r5);                                                                                \
            if (r6 < rows) v6 = SRC(r6);                                                                                \
            if (r7 < rows) v7 = SRC(r7);                                                                                \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0); /* keep the loads together, ahead of the LDS writes */                       \
        dst[r0 * 64] = v0;                                                                                              \
        if (r1 < rows) dst[r1 * 64] = v1;                                                                               \
        if (r2 < rows) dst[r2 * 64] = v2;                                                                               \
        if (r3 < rows) dst[r3 * 64] = v3;                                                                               \
        if (r4 < rows) {                                                                                                \
            dst[r4 * 64] = v4;                                                                                          \
            if (r5 < rows) dst[r5 * 64] = v5;                                                                           \
            if (r6 < rows) dst[r6 * 64] = v6;                                                                           \
            if (r7 < rows) dst[r7 * 64] = v7;                                                                           \
        }                                                                                                               \
    }
__device__ __forceinline__ void copy_tables_to_lds(const uint8_t* __restrict__ tables, uint8_t* lds, uint32_t tab_bytes,
                                                   uint32_t lane, uint32_t wave_in_wg, uint32_t waves_per_wg) {
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(tables) + lane;
    u32x4* dst = reinterpret_cast<u32x4*>(lds) + lane;
    const uint32_t rows = tab_bytes / 1024;
#define LLE_SRC1(r) src[(r) * 64]
    LLE_COPY_ROWS(LLE_SRC1)
#undef LLE_SRC1
}

// Two sections into consecutive LDS ranges (the map tables and, behind them, the per-env-sources section), with the rows of BOTH
// requested before the first LDS write: called one after the other, the second copy's loads would only be issued once the
// first copy's have returned.
__device__ __forceinline__ void copy_tables2_to_lds(const uint8_t* __restrict__ t1, uint32_t bytes1, const uint8_t* __restrict__ t2, uint32_t bytes2,
                                                    uint8_t* lds, uint32_t lane, uint32_t wave_in_wg, uint32_t waves_per_wg) {
    const u32x4* __restrict__ s1 = reinterpret_cast<const u32x4*>(t1) + lane;
    const u32x4* __restrict__ s2 = reinterpret_cast<const u32x4*>(t2) + lane;
    u32x4* dst = reinterpret_cast<u32x4*>(lds) + lane;
    const uint32_t rows1 = bytes1 / 1024, rows = rows1 + bytes2 / 1024;
#define LLE_SRC2(r) (*((r) < rows1 ? s1 + (r) * 64 : s2 + ((r) - rows1) * 64))
    LLE_COPY_ROWS(LLE_SRC2)
#undef LLE_SRC2
}
#undef LLE_COPY_ROWS

// The same LDS image from the PACKED form of the table section (tables.h off_packed), for launches whose workgroups each read their own map's tables
// from memory (batches of thousands of maps): 16-bit cell words widened, the layer words of the few cells under a beam scattered over zeros, dyn table
// and dynamic chunks as they are.  All `T` threads of the workgroup; the first round of every part is requested before the first LDS write.  Contains
// ONE workgroup barrier (the zeros are in place before the scatter): every wavefront of the workgroup calls it.
__device__ __forceinline__ void expand_packed_tables(const uint8_t* __restrict__ pk, uint8_t* lds, uint32_t HW, uint32_t n_lay, uint32_t lds_meta, uint32_t lds_dyn,
                                                     uint32_t tail_bytes, uint32_t tid, uint32_t T) {
    const uint64_t* __restrict__ m4 = reinterpret_cast<const uint64_t*>(pk);  // four cell words each
    const uint16_t* __restrict__ idx = reinterpret_cast<const uint16_t*>(pk + packed_meta_bytes(HW));
    const uint64_t* __restrict__ lay = reinterpret_cast<const uint64_t*>(pk + packed_meta_bytes(HW) + packed_idx_bytes(n_lay));
    const uint4* __restrict__ tail = reinterpret_cast<const uint4*>(pk + packed_meta_bytes(HW) + packed_idx_bytes(n_lay) + ((n_lay * 8u + 15u) & ~15u));
    const uint32_t n4 = (HW + 3u) / 4u, nt = tail_bytes / 16u;
    const uint64_t w0 = tid < n4 ? m4[tid] : 0ull;
    const uint32_t i0 = tid < n_lay ? (uint32_t)idx[tid] : 0u;
    const uint64_t l0 = tid < n_lay ? lay[tid] : 0ull;
    uint4 t0 = {0u, 0u, 0u, 0u};
    if (tid < nt) t0 = tail[tid];
    auto widen = [](uint64_t w) { return uint4{(uint32_t)w & 0xFFFFu, (uint32_t)(w >> 16) & 0xFFFFu, (uint32_t)(w >> 32) & 0xFFFFu, (uint32_t)(w >> 48)}; };
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = tid; i < lds_meta / 16u; i += T) z[i] = uint4{0u, 0u, 0u, 0u};
    uint4* md = reinterpret_cast<uint4*>(lds + lds_meta);
    if (tid < n4) md[tid] = widen(w0);
    for (uint32_t i = tid + T; i < n4; i += T) md[i] = widen(m4[i]);
    uint4* td = reinterpret_cast<uint4*>(lds + lds_dyn);
    if (tid < nt) td[tid] = t0;
    for (uint32_t i = tid + T; i < nt; i += T) td[i] = tail[i];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint64_t* ld = reinterpret_cast<uint64_t*>(lds);
    if (tid < n_lay) ld[i0] = l0;
    for (uint32_t i = tid + T; i < n_lay; i += T) ld[idx[i]] = lay[i];
}

// a * b + c with 24-bit a, b in ONE full-rate instruction (v_mad_u32_u24).  __umul24 is a masked 32-bit product to the compiler, which
// -- in the inner loop of the partial writers -- it turned into v_mul_lo_u32 / v_mad_u64_u32 (quarter rate: three of them per window cell).
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- the small stores of a step (state, events, counters, the fused LLE.step outputs): a few dozen bytes per environment.
// Plain, they stay dirty in the XCD's L2 until the end of the kernel, whose release then writes them back before the next launch
// can start; LLE_SMALL_WT (an A/B build: profiles/r04_pes_tax.md) writes them through as they are issued.
template <typename T>
__device__ __forceinline__ void small_store(T* p, T v) {
#ifdef LLE_SMALL_WT
    if constexpr (sizeof(T) == 1) {
        asm volatile("global_store_byte %0, %1, off sc1" ::"v"(p), "v"((uint32_t)(uint8_t)v) : "memory");
    } else if constexpr (sizeof(T) == 2) {
        asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"((uint32_t)(uint16_t)v) : "memory");
    } else if constexpr (sizeof(T) == 4) {
        uint32_t w;
        __builtin_memcpy(&w, &v, 4);
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    } else {
        static_assert(sizeof(T) == 8, "small_store: 1, 2, 4 or 8 bytes");
        uint64_t w;
        __builtin_memcpy(&w, &v, 8);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    }
#else
    *p = v;
#endif
}

// ---- the stores of an observation stream.
// WT = write-through (`sc1`: agent scope, so the line leaves the XCD's L2 for the Infinity Cache as it is written).
// A plain store leaves the row dirty in L2 and the end of the kernel writes all of it back before the next launch can
// start; measured on level 6 (1 872 B rows): 65 536 envs 22.8 -> 21.3 us per launch, 16 384 envs 11.4 -> 9.7 us.  Once
// the rows of one launch no longer fit the 256 MB Infinity Cache the order flips (262 144 envs: 108 us plain, 166 us
// written through), so the launcher picks the policy from the bytes a launch writes (LAUNCH_WRITE_THROUGH).
// `nt` stores measured 50 % slower than plain ones at every size.
template <bool WT>
__device__ __forceinline__ void stream_store(uint4* p, const uint4& v) {
    if constexpr (WT) {
        const u32x4 w = {v.x, v.y, v.z, v.w};
        // `s_nop 1`: a store of more than 64 bits reads its data VGPRs late, and a VALU write to them within the next two
        // wait states corrupts the stored value (the compiler pads its own stores; it cannot see inside the asm --
        // without the nop the lanes that finish last stored the next store's address instead of the row's bytes)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w));
    } else {
        *p = v;
    }
}

// Chunks [first + lane, n_chunks) of a row from LDS, 1 KiB per wave instruction; four LDS reads are in flight before the
// first store so that a long row is not one LDS round trip per KiB.
template <bool WT>
__device__ __forceinline__ void stream_row(uint4* __restrict__ dst, const uint4* srcv, uint32_t first, uint32_t n_chunks, uint32_t lane) {
    uint32_t c = first + lane;
    for (; c + 192u < n_chunks; c += 256u) {
        const uint4 q0 = srcv[c], q1 = srcv[c + 64u], q2 = srcv[c + 128u], q3 = srcv[c + 192u];
        stream_store<WT>(dst + c, q0);
        stream_store<WT>(dst + c + 64u, q1);
        stream_store<WT>(dst + c + 128u, q2);
        stream_store<WT>(dst + c + 192u, q3);
    }
    for (; c < n_chunks; c += 64u) stream_store<WT>(dst + c, srcv[c]);
}

// A whole row: the first 2 KiB as two unconditional LDS reads ahead of their stores (most maps' rows end there), the
// rest through stream_row.
template <bool WT>
__device__ __forceinline__ void stream_whole_row(uint4* __restrict__ dst, const uint4* srcv, uint32_t n_chunks, uint32_t lane) {
    const uint32_t c0 = lane, c1 = lane + 64u;
    const uint4 v0 = srcv[c0 < n_chunks ? c0 : 0u], v1 = srcv[c1 < n_chunks ? c1 : 0u];
    if (c0 < n_chunks) stream_store<WT>(dst + c0, v0);
    if (c1 < n_chunks) stream_store<WT>(dst + c1, v1);
    if (n_chunks > 128u) stream_row<WT>(dst, srcv, 128u, n_chunks, lane);
}

// The same without the row's HEAD (chunks [head_lo, head_lo + head_n) and, behind them, [head2_lo, head2_lo + head2_n): stored before
// the state machine, see store_heads): tail chunk i is chunk i + (i >= head_lo ? head_n : 0), + head2_n once that reaches head2_lo.
template <bool WT>
__device__ __forceinline__ void stream_row_tail(uint4* __restrict__ dst, const uint4* srcv, uint32_t n_chunks, uint32_t head_lo, uint32_t head_n,
                                                uint32_t lane, uint32_t head2_lo = 0, uint32_t head2_n = 0) {
    const uint32_t n_tail = n_chunks - head_n - head2_n;
    auto chunk_of = [&](uint32_t i) {
        uint32_t c = i + (i >= head_lo ? head_n : 0u);
        if (head2_n) c += c >= head2_lo ? head2_n : 0u;
        return c;
    };
    const uint32_t i0 = lane, i1 = lane + 64u;
    const uint32_t c0 = chunk_of(i0), c1 = chunk_of(i1);
    const uint4 v0 = srcv[i0 < n_tail ? c0 : 0u], v1 = srcv[i1 < n_tail ? c1 : 0u];
    if (i0 < n_tail) stream_store<WT>(dst + c0, v0);
    if (i1 < n_tail) stream_store<WT>(dst + c1, v1);
    for (uint32_t i = lane + 128u; i < n_tail; i += 64u) {
        const uint32_t c = chunk_of(i);
        stream_store<WT>(dst + c, srcv[c]);
    }
}

// Only the DYNAMIC chunks of a row (tables.h off_dyn_chunks; STEP_INCREMENTAL_OBS): chunk tab[i] for i = lane, lane + 64, ...  The other
// lines of the row already hold their bytes.  c0 / c1: this lane's first two chunks (0xFFFF: none), looked up once per wavefront.
template <bool WT>
__device__ __forceinline__ void stream_row_dyn(uint4* __restrict__ dst, const uint4* srcv, const uint16_t* tab, uint32_t n_dyn, uint32_t c0, uint32_t c1,
                                               uint32_t lane) {
    const uint4 v0 = srcv[c0 != 0xFFFFu ? c0 : 0u], v1 = srcv[c1 != 0xFFFFu ? c1 : 0u];
    if (c0 != 0xFFFFu) stream_store<WT>(dst + c0, v0);
    if (c1 != 0xFFFFu) stream_store<WT>(dst + c1, v1);
    for (uint32_t i = lane + 128u; i < n_dyn; i += 64u) {
        const uint32_t c = tab[i];
        stream_store<WT>(dst + c, srcv[c]);
    }
}

// ---- WIDENED rows (tables.h ObsElem): the LDS row stays int8, every 16-byte chunk of the OUTPUT row is built from the 16 >> shift row bytes
// it covers -- 8 of them for fp16 / bf16 (one ds_read_b64 per lane), 4 for fp32 (one ds_read_b32) -- so consecutive lanes still read
// consecutive LDS bytes and write consecutive 16-byte chunks: 1 KiB fully coalesced per wave instruction, as in the int8 stream.
// One chunk at a time, nothing unrolled: these loops run in the registers the int8 stream leaves them (the kernels sit at their register
// caps: a four-deep version put 439 of the 555 kernels into scratch), and sixteen wavefronts per CU hide the LDS round trip of each.
// A row only ever holds -1, 0 and 1 (template, dyn bases, patches): byte b becomes (b & 1 ? ONE : 0) | sign, ONE = 1.0 in the target type.
__device__ __forceinline__ uint4 widen_chunk16(const int8_t* tmpl, uint32_t q, uint32_t one, uint32_t sel_lo, uint32_t sel_hi) {  // fp16 (one = 0x3C00) / bf16 (0x3F80)
    const uint2 w = reinterpret_cast<const uint2*>(tmpl)[q];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t src = k < 2 ? w.x : w.y;
        // bytes 2(k&1), 2(k&1)+1 of `src` into the low bytes of the two halves (v_perm_b32: selector 0x0c = a zero byte)
        const uint32_t x = __builtin_amdgcn_perm(src, src, (k & 1) ? sel_hi : sel_lo);
        o[k] = __umul24(x & 0x00010001u, one) | ((x & 0x00800080u) << 8);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ uint4 widen_chunk32(const int8_t* tmpl, uint32_t q) {
    const uint32_t w = reinterpret_cast<const uint32_t*>(tmpl)[q];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = (w >> (8 * k)) & 0xFFu;
        o[k] = ((b & 1u) ? 0x3F800000u : 0u) | ((b & 0x80u) << 24);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// Everything a widening stream needs besides the row itself is REBUILT per row from two opaque values (a scalar zero and the lane id behind
// empty asm statements): otherwise the compiler hoists the loop-invariant pieces -- the v_perm selectors (a VOP3 operand must sit in a
// register), `one`, the chunk count, the lane's LDS address -- above the loop over the environments and, in the fused rollouts, above the
// loop over the STEPS, where they stay live across the whole state machine (58 rollout kernels at their register caps went to scratch).
struct WideConsts { uint32_t one, sel_lo, sel_hi, sh, lane; };
__device__ __forceinline__ WideConsts wide_consts(uint32_t et, uint32_t lane) {
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    asm volatile("" : "+v"(lane));
    WideConsts c;
    c.one = (et == OBS_F16 ? 0x3C00u : 0x3F80u) | z;
    c.sel_lo = 0x0c010c00u | z;
    c.sel_hi = 0x0c030c02u | z;
    c.sh = obs_elem_shift(et) + z;
    c.lane = lane;
    return c;
}
// output chunks [0, n_chunks << shift) of a row (or of a slice of it: `tmpl` and `dst` then both start at the slice)
template <bool WT>
__device__ __forceinline__ void stream_wide(uint4* __restrict__ dst, const int8_t* tmpl, uint32_t n_chunks, uint32_t et, uint32_t lane_in) {
    const WideConsts c = wide_consts(et, lane_in);
    const uint32_t n_out = n_chunks << c.sh;
    if (c.sh == 2u) {
#pragma clang loop unroll(disable)
        for (uint32_t q = c.lane; q < n_out; q += 64u) stream_store<WT>(dst + q, widen_chunk32(tmpl, q));
    } else {
#pragma clang loop unroll(disable)
        for (uint32_t q = c.lane; q < n_out; q += 64u) stream_store<WT>(dst + q, widen_chunk16(tmpl, q, c.one, c.sel_lo, c.sel_hi));
    }
}
// ... and only the DYNAMIC chunks (STEP_INCREMENTAL_OBS): entries [t_lo, t_hi) of the chunk table, each int8 chunk c = output chunks
// [c << shift, (c + 1) << shift); `c_base`: first chunk of the slice `tmpl` / `dst` start at (split rows; 0 for whole rows)
template <bool WT>
__device__ __forceinline__ void stream_wide_dyn(uint4* __restrict__ dst, const int8_t* tmpl, const uint16_t* tab, uint32_t t_lo, uint32_t t_hi, uint32_t c_base,
                                                uint32_t et, uint32_t lane_in) {
    const WideConsts c = wide_consts(et, lane_in);
#pragma clang loop unroll(disable)
    for (uint32_t i = (t_lo << c.sh) + c.lane; i < (t_hi << c.sh); i += 64u) {
        const uint32_t q = (((uint32_t)tab[i >> c.sh] - c_base) << c.sh) + (i & ((1u << c.sh) - 1u));
        stream_store<WT>(dst + q, c.sh == 2u ? widen_chunk32(tmpl, q) : widen_chunk16(tmpl, q, c.one, c.sel_lo, c.sel_hi));
    }
}

// The heads of the wave's rows: the same `head_n` (<= 64) chunks, straight from the map's pristine template in global
// memory (`v`: this lane's chunk), into every row.  Issued BEFORE the state machine: a launch of the step kernel is
// (ramp) + (state machine, every wavefront at the same time) + (stream), and the lines that no agent, beam or gem can
// touch need not wait for the state machine -- the memory system starts ~2 us earlier (tools/ceiling/head_probe.hip).
template <bool WT>
__device__ __forceinline__ void store_heads(int8_t* __restrict__ obs, uint64_t obs_stride, int64_t env0, int64_t n_here, uint32_t head_lo,
                                            uint32_t head_n, const uint4& v, uint32_t lane, uint32_t rot = 0, uint32_t head2_lo = 0, uint32_t head2_n = 0) {
    // (two runs: lanes [0, head_n) hold the chunks of the first, lanes [head_n, head_n + head2_n) those of the second)
    const uint32_t chunk = lane < head_n ? head_lo + lane : head2_lo + (lane - head_n);
    if (lane < head_n + head2_n)
        for (uint32_t k = 0; k < (uint32_t)n_here; k++)
            stream_store<WT>(reinterpret_cast<uint4*>(obs + (uint64_t)(env0 + rotated(k, rot, (uint32_t)n_here)) * obs_stride) + chunk, v);
}

// Store policy x element width of a launch as compile-time tags: f(std::bool_constant<WT>, std::bool_constant<WIDE>) is called for the one
// combination the launch flags name (CAN_WIDE = false: kernels that never widen -- the ones with row heads -- instantiate two, not four).
template <bool CAN_WIDE, typename F>
__device__ __forceinline__ void dispatch_stream(uint32_t lflags, F&& f) {
    const bool wt = (lflags & LAUNCH_WRITE_THROUGH) != 0;
    if constexpr (CAN_WIDE) {
        if (lflags & LAUNCH_OBS_ELEM_MASK) {
            if (wt) f(std::true_type{}, std::true_type{});
            else f(std::false_type{}, std::true_type{});
            return;
        }
    }
    if (wt) f(std::true_type{}, std::false_type{});
    else f(std::false_type{}, std::false_type{});
}

// ---- phase 2: layered observation of the wave's environments, one environment at a time.
// `scratch` holds one hand-over record per environment: [0 | beam masks | ~gem bits | byte index of each agent].
// `obs_stride` = bytes between the rows of consecutive environments in `obs`.
// HEAD: the rows' heads are already stored (store_heads); stream the rest.
// INCR: only the dynamic chunks (dyn_chunks / n_dyn_chunks: stream_row_dyn).
// WIDE: the row leaves as fp16 / bf16 / fp32 (tables.h ObsElem, `et`), stream_wide.  A template parameter, chosen by the caller OUTSIDE the loop
// over the environments: with the choice inside it, the loop-invariant addresses of both streams were hoisted above the loop together and
// 105 kernels at their register caps went to scratch.
template <bool WT, bool HEAD = false, bool INCR = false, bool WIDE = false>
__device__ __forceinline__ void write_observations(int A, int L, uint32_t D, uint32_t n_chunks, uint64_t obs_stride,
                                                   const uint64_t* dyn, int8_t* tmpl, const uint32_t* scratch,
                                                   uint32_t scr_stride, int8_t* __restrict__ obs, int64_t env0,
                                                   int64_t n_here, uint32_t lane, uint32_t head_lo = 0, uint32_t head_n = 0, uint32_t rot = 0,
                                                   const uint16_t* dyn_chunks = nullptr, uint32_t n_dyn_chunks = 0, uint32_t et = OBS_I8) {
    // (`obs_stride` is the row pitch in BYTES: the map's pitch in elements times the element size)
    static_assert(!(WIDE && HEAD), "the kernels with row heads carry no widening code: the launcher never sends them a widened launch");
    uint32_t dc0 = 0xFFFFu, dc1 = 0xFFFFu;
    if constexpr (INCR) {
        dc0 = lane < n_dyn_chunks ? (uint32_t)dyn_chunks[lane] : 0xFFFFu;
        dc1 = lane + 64u < n_dyn_chunks ? (uint32_t)dyn_chunks[lane + 64u] : 0xFFFFu;
    }
    // Each lane serves the same dyn entry for every environment: decode it once.
    // A laser / gem reference becomes (dword of the hand-over record, bit); an absent one points at the record's
    // zero word, so the per-environment evaluation is branch-free.
    const bool has_d0 = lane < D;
    const uint64_t e0 = has_d0 ? dyn[lane] : 0ull;
    const uint32_t d0_idx = dyn_index(e0);
    const int32_t d0_base = dyn_base(e0);
    const uint32_t d0_refs = dyn_refs(e0), d0_gem = dyn_gem(e0);
    const uint32_t d0_r0 = dyn_ref0(e0), d0_r1 = dyn_ref1(e0);
    const uint32_t d0_w0 = d0_refs >= 1 ? 1u + ref_word(d0_r0) : 0u, d0_s0 = d0_refs >= 1 ? ref_bit(d0_r0) : 0u;
    const uint32_t d0_w1 = d0_refs >= 2 ? 1u + ref_word(d0_r1) : 0u, d0_s1 = d0_refs >= 2 ? ref_bit(d0_r1) : 0u;
    const uint32_t d0_wg = d0_gem != NO_GEM ? (uint32_t)L + 1u : 0u, d0_sg = d0_gem != NO_GEM ? gem_bit(d0_gem) : 0u;
    const bool is_agent_lane = (int)lane < A;
    const uint4* srcv = reinterpret_cast<const uint4*>(tmpl);

    for (uint32_t k0 = 0; k0 < (uint32_t)n_here; k0++) {
        const uint32_t k = rotated(k0, rot, (uint32_t)n_here);  // (row_rotation: the wavefront starts at another one of its rows)
        const uint32_t* sc = scratch + k * scr_stride;
        // (a) bytes that depend on beams / gems
        {
            const uint32_t lit = ((sc[d0_w0] >> d0_s0) | (sc[d0_w1] >> d0_s1) | (sc[d0_wg] >> d0_sg)) & 1u;
            if (has_d0) tmpl[d0_idx] = (int8_t)(lit ? 1 : d0_base);
        }
        for (uint32_t d = lane + 64u; d < D; d += 64) {  // maps with more than 64 dynamic bytes
            const uint64_t e = dyn[d];
            const uint32_t refs = dyn_refs(e), gem = dyn_gem(e);
            const uint32_t r0 = dyn_ref0(e), r1 = dyn_ref1(e);
            const uint32_t w0 = refs >= 1 ? 1u + ref_word(r0) : 0u, w1 = refs >= 2 ? 1u + ref_word(r1) : 0u;
            const uint32_t wg = gem != NO_GEM ? (uint32_t)L + 1u : 0u;
            const uint32_t lit = ((sc[w0] >> (refs >= 1 ? ref_bit(r0) : 0u)) | (sc[w1] >> (refs >= 2 ? ref_bit(r1) : 0u)) |
                                  (sc[wg] >> (gem != NO_GEM ? gem_bit(gem) : 0u))) & 1u;
            tmpl[dyn_index(e)] = (int8_t)(lit ? 1 : dyn_base(e));
        }
        // (b) agents (dead ones included, observations.py:264-265)
        const uint32_t agent_idx = is_agent_lane ? sc[L + 2 + lane] : 0u;
        if (is_agent_lane) tmpl[agent_idx] = 1;
        wave_sync();
        // (c) stream the patched copy as one contiguous row: 16 B per lane, 1 KiB per wave instruction
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(obs + (uint64_t)(env0 + k) * obs_stride);
        if constexpr (WIDE) {
            if constexpr (INCR) stream_wide_dyn<WT>(dst, tmpl, dyn_chunks, 0u, n_dyn_chunks, 0u, et, lane);
            else stream_wide<WT>(dst, tmpl, n_chunks, et, lane);
        } else if constexpr (INCR) stream_row_dyn<WT>(dst, srcv, dyn_chunks, n_dyn_chunks, dc0, dc1, lane);
        else if constexpr (HEAD) stream_row_tail<WT>(dst, srcv, n_chunks, head_lo, head_n, lane);
        else stream_whole_row<WT>(dst, srcv, n_chunks, lane);
        wave_sync();
        // (d) agents off again (their layers are all-zero in the static copy); LDS is in order, so this lands after
        // the reads above and before the next environment's patches
        if (is_agent_lane) tmpl[agent_idx] = 0;
    }
}

// ---- phase 2 for BIG rows (config 5: 20 KB per environment): the row is split over the wavefronts of the workgroup.
// A private copy of the whole row per wavefront limits a CU to one workgroup (4 x 20 KB + tables), i.e. four
// wavefronts per CU to keep the store pipeline full.  Here wavefront w of the workgroup owns the chunks
// [w * cpw, (w + 1) * cpw) of EVERY environment of the workgroup: its private copy is one slice (5 KB), it patches the
// dynamic bytes and agents that fall into its slice and streams the slice of each environment in turn.  The hand-over
// records of all the workgroup's environments sit in one shared LDS area (one workgroup barrier after phase 1).
// `tmpl` = the wave's slice (pristine bytes of chunks [lo, hi)), `lo`, `hi` in 16-byte chunks, `dyn` the whole (sorted
// by byte index) table.  Same bytes as write_observations: every dynamic byte and every agent byte belongs to exactly
// one slice.
// INCR (STEP_INCREMENTAL_OBS): only the slice's DYNAMIC chunks -- entries [t_lo, t_hi) of the ascending table dyn_chunks.
template <bool WT, bool INCR = false, bool WIDE = false>
__device__ __forceinline__ void write_observations_split(int A, int L, uint32_t D, uint32_t lo, uint32_t hi, uint64_t obs_stride,
                                                         const uint64_t* dyn, int8_t* tmpl, const uint32_t* records,
                                                         uint32_t scr_stride, int8_t* __restrict__ obs, int64_t wg_env0,
                                                         int64_t n_wg_here, uint32_t lane, const uint16_t* dyn_chunks = nullptr, uint32_t n_dyn_chunks = 0,
                                                         uint32_t et = OBS_I8) {
    const uint32_t b_lo = lo * 16u, b_hi = hi * 16u;
    uint32_t t_lo = 0, t_hi = 0;
    if constexpr (INCR) {
        for (uint32_t i = lane; i < ((n_dyn_chunks + 63u) & ~63u); i += 64) {
            const uint32_t c = i < n_dyn_chunks ? (uint32_t)dyn_chunks[i] : 0xFFFFFFFFu;
            t_lo += (uint32_t)__popcll(__ballot(c < lo));
            t_hi += (uint32_t)__popcll(__ballot(c < hi));
        }
    }
    // the dyn table is sorted by byte index: this slice's entries are [d_lo, d_hi)
    uint32_t d_lo = 0, d_hi = 0;
    for (uint32_t d = lane; d < ((D + 63u) & ~63u); d += 64) {
        const uint32_t idx = d < D ? dyn_index(dyn[d]) : 0xFFFFFFFFu;
        d_lo += (uint32_t)__popcll(__ballot(idx < b_lo));
        d_hi += (uint32_t)__popcll(__ballot(idx < b_hi));
    }
    const bool has_d0 = d_lo + lane < d_hi;
    const uint64_t e0 = has_d0 ? dyn[d_lo + lane] : 0ull;
    const uint32_t d0_idx = has_d0 ? dyn_index(e0) - b_lo : 0u;
    const int32_t d0_base = dyn_base(e0);
    const uint32_t d0_refs = dyn_refs(e0), d0_gem = dyn_gem(e0);
    const uint32_t d0_r0 = dyn_ref0(e0), d0_r1 = dyn_ref1(e0);
    const uint32_t d0_w0 = d0_refs >= 1 ? 1u + ref_word(d0_r0) : 0u, d0_s0 = d0_refs >= 1 ? ref_bit(d0_r0) : 0u;
    const uint32_t d0_w1 = d0_refs >= 2 ? 1u + ref_word(d0_r1) : 0u, d0_s1 = d0_refs >= 2 ? ref_bit(d0_r1) : 0u;
    const uint32_t d0_wg = d0_gem != NO_GEM ? (uint32_t)L + 1u : 0u, d0_sg = d0_gem != NO_GEM ? gem_bit(d0_gem) : 0u;
    const bool is_agent_lane = (int)lane < A;
    const uint4* srcv = reinterpret_cast<const uint4*>(tmpl);
    const uint32_t n_mine = hi - lo;

    for (int64_t k = 0; k < n_wg_here; k++) {
        const uint32_t* sc = records + (uint32_t)k * scr_stride;
        {
            const uint32_t lit = ((sc[d0_w0] >> d0_s0) | (sc[d0_w1] >> d0_s1) | (sc[d0_wg] >> d0_sg)) & 1u;
            if (has_d0) tmpl[d0_idx] = (int8_t)(lit ? 1 : d0_base);
        }
        for (uint32_t d = d_lo + lane + 64u; d < d_hi; d += 64) {  // slices with more than 64 dynamic bytes
            const uint64_t e = dyn[d];
            const uint32_t refs = dyn_refs(e), gem = dyn_gem(e);
            const uint32_t r0 = dyn_ref0(e), r1 = dyn_ref1(e);
            const uint32_t w0 = refs >= 1 ? 1u + ref_word(r0) : 0u, w1 = refs >= 2 ? 1u + ref_word(r1) : 0u;
            const uint32_t wg = gem != NO_GEM ? (uint32_t)L + 1u : 0u;
            const uint32_t lit = ((sc[w0] >> (refs >= 1 ? ref_bit(r0) : 0u)) | (sc[w1] >> (refs >= 2 ? ref_bit(r1) : 0u)) |
                                  (sc[wg] >> (gem != NO_GEM ? gem_bit(gem) : 0u))) & 1u;
            tmpl[dyn_index(e) - b_lo] = (int8_t)(lit ? 1 : dyn_base(e));
        }
        const uint32_t agent_idx = is_agent_lane ? sc[L + 2 + lane] : 0xFFFFFFFFu;
        const bool agent_here = agent_idx >= b_lo && agent_idx < b_hi;  // (idle lanes: 0xFFFFFFFF is in no slice)
        if (agent_here) tmpl[agent_idx - b_lo] = 1;
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(obs + (uint64_t)(wg_env0 + k) * obs_stride) + ((uint64_t)lo << (WIDE ? obs_elem_shift(et) : 0u));
        if constexpr (WIDE) {
            if constexpr (INCR) stream_wide_dyn<WT>(dst, tmpl, dyn_chunks, t_lo, t_hi, lo, et, lane);
            else stream_wide<WT>(dst, tmpl, n_mine, et, lane);
        } else if constexpr (INCR) {
            for (uint32_t i = t_lo + lane; i < t_hi; i += 64u) {
                const uint32_t c = (uint32_t)dyn_chunks[i] - lo;
                stream_store<WT>(dst + c, srcv[c]);
            }
        } else {
            stream_whole_row<WT>(dst, srcv, n_mine, lane);
        }
        wave_sync();
        if (agent_here) tmpl[agent_idx - b_lo] = 0;
    }
}

// ---- the same, for batches whose environments have their own source colours / enabled flags
// (lle_batch_set_sources).  The layer of a laser byte is LASER_0 + the env's colour of that beam, so it cannot be baked
// into a table: `tmpl` starts as the BARE static observation (walls, voids, exits) and every env writes its elements
// -- -1 at each source, 1 at each exposed laser tile whose beam bit is on, 1 at each uncollected gem, 1 at each agent
// -- streams, and puts the bare bytes back.  All writes of one env commute (equal values wherever two elements share a
// byte; a source and a tile never share a cell), so one pass in any lane order reproduces the reference's write order
// (python/lle/observations.py:216-266), colour aliasing (Q5) included.
// record: [0 | beam masks | ~gem bits | byte index of each agent | colour words (4 colours per dword)]
// `laser_layer` (views: layer of colour c) or NULL (Layered: LASER_0 + c); `gem_layer` = the GEM channel.
__device__ __forceinline__ void elem_eval(uint32_t e, const uint32_t* sc, int A, int L, uint32_t HW, const uint8_t* laser_layer,
                                          uint32_t gem_layer, uint32_t& idx, int32_t& val, bool& on) {
    const uint32_t cell = elem_cell(e), i5 = elem_index(e), off = elem_bit(e), type = elem_type(e);
    const uint32_t colour = (sc[L + 2 + A + (i5 >> 2)] >> ((i5 & 3u) * 8u)) & 0xFFu;
    const bool is_gem = type == ELEM_GEM;
    const uint32_t llayer = laser_layer ? (uint32_t)laser_layer[colour] : (uint32_t)A + colour;
    idx = (is_gem ? gem_layer : llayer) * HW + cell;
    val = type == ELEM_SOURCE ? -1 : 1;
    on = type == ELEM_SOURCE ? true : (is_gem ? ((sc[L + 1] >> i5) & 1u) != 0 : ((sc[1 + i5] >> off) & 1u) != 0);
}
template <bool WT, bool HEAD = false, bool INCR = false, bool WIDE = false>
__device__ __forceinline__ void write_observations_env(int A, int L, uint32_t HW, uint32_t n_elems, uint32_t n_chunks,
                                                       uint64_t obs_stride, const uint32_t* elems, const int8_t* bare,
                                                       int8_t* tmpl, const uint32_t* scratch, uint32_t scr_stride,
                                                       int8_t* __restrict__ obs, int64_t env0, int64_t n_here, uint32_t lane,
                                                       const uint8_t* laser_layer = nullptr, uint32_t gem_layer_in = 0xFFFFFFFFu,
                                                       uint32_t head_lo = 0, uint32_t head_n = 0, uint32_t rot = 0,
                                                       const uint16_t* dyn_chunks = nullptr, uint32_t n_dyn_chunks = 0, uint32_t head2_lo = 0, uint32_t head2_n = 0,
                                                       uint32_t et = OBS_I8) {
    static_assert(!(WIDE && HEAD), "no widening in the kernels with row heads");
    uint32_t dc0 = 0xFFFFu, dc1 = 0xFFFFu;  // INCR: this lane's first two dynamic chunks (stream_row_dyn)
    if constexpr (INCR) {
        dc0 = lane < n_dyn_chunks ? (uint32_t)dyn_chunks[lane] : 0xFFFFu;
        dc1 = lane + 64u < n_dyn_chunks ? (uint32_t)dyn_chunks[lane + 64u] : 0xFFFFu;
    }
    const uint32_t gem_layer = gem_layer_in == 0xFFFFFFFFu ? (uint32_t)(2 * A + 2) : gem_layer_in;
    const bool has_e0 = lane < n_elems;
    const uint32_t e0 = has_e0 ? elems[lane] : 0u;
    // the lane serves the same element for every environment: what does not depend on the environment is decoded once --
    // the cell, the record word and shift that hold its colour, the record word and bit that say whether it shows
    const uint32_t e0_cell = elem_cell(e0), e0_i5 = elem_index(e0), e0_off = elem_bit(e0), e0_type = elem_type(e0);
    const bool e0_gem = e0_type == ELEM_GEM, e0_src = e0_type == ELEM_SOURCE;
    const uint32_t e0_colw = (uint32_t)(L + 2 + A) + (e0_i5 >> 2), e0_colsh = (e0_i5 & 3u) * 8u;
    const uint32_t e0_onw = e0_gem ? (uint32_t)L + 1u : (e0_src ? 0u : 1u + e0_i5);   // gem: ~collected bits; tile: the beam mask; source: always
    const uint32_t e0_onsh = e0_gem ? e0_i5 : e0_off;
    const int8_t e0_val = e0_src ? (int8_t)-1 : (int8_t)1;
    // byte index = e0_base + colour * e0_mul (Layered: layer LASER_0 + colour; a gem: its own layer, whatever the colour word says)
    const uint32_t e0_base = (e0_gem ? gem_layer : (uint32_t)A) * HW + e0_cell, e0_mul = (e0_gem || laser_layer) ? 0u : HW;
    const uint32_t e0_always = (has_e0 && e0_src) ? 1u : 0u, e0_dyn = (has_e0 && !e0_src) ? 1u : 0u;
    const bool is_agent_lane = (int)lane < A;
    const uint4* srcv = reinterpret_cast<const uint4*>(tmpl);
    for (uint32_t k0 = 0; k0 < (uint32_t)n_here; k0++) {
        const uint32_t k = rotated(k0, rot, (uint32_t)n_here);
        const uint32_t* sc = scratch + k * scr_stride;
        const uint32_t colour = (sc[e0_colw] >> e0_colsh) & 0xFFu, onw = sc[e0_onw];
        uint32_t idx0 = e0_base + __umul24(colour, e0_mul);
        if (laser_layer && !e0_gem) idx0 = (uint32_t)laser_layer[colour] * HW + e0_cell;   // views: the layer of colour c is a table
        const bool on0 = ((e0_always | (e0_dyn & (onw >> e0_onsh))) & 1u) != 0u;
        if (on0) tmpl[idx0] = e0_val;
        for (uint32_t d = lane + 64u; d < n_elems; d += 64) {
            uint32_t idx; int32_t val; bool on;
            elem_eval(elems[d], sc, A, L, HW, laser_layer, gem_layer, idx, val, on);
            if (on) tmpl[idx] = (int8_t)val;
        }
        const uint32_t agent_idx = is_agent_lane ? sc[L + 2 + lane] : 0u;
        if (is_agent_lane) tmpl[agent_idx] = 1;
        wave_sync();
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(obs + (uint64_t)(env0 + k) * obs_stride);
        if constexpr (WIDE) {
            if constexpr (INCR) stream_wide_dyn<WT>(dst, tmpl, dyn_chunks, 0u, n_dyn_chunks, 0u, et, lane);
            else stream_wide<WT>(dst, tmpl, n_chunks, et, lane);
        } else if constexpr (INCR) stream_row_dyn<WT>(dst, srcv, dyn_chunks, n_dyn_chunks, dc0, dc1, lane);
        else if constexpr (HEAD) stream_row_tail<WT>(dst, srcv, n_chunks, head_lo, head_n, lane, head2_lo, head2_n);  // (the head is stored already: store_heads)
        else stream_whole_row<WT>(dst, srcv, n_chunks, lane);
        wave_sync();
        // bare bytes back (LDS is in order: after the reads above, before the next environment's writes).  On the agent / laser
        // planes and at a gem the bare byte is zero; only a colour >= n_agents, whose layer aliases WALL / VOID / GEM / EXIT
        // (quirk Q5), has one to read back -- a round trip of the wavefront's per-environment chain that the rest skips
        if (on0) {
            int8_t back = 0;
            if (!e0_gem && colour >= (uint32_t)A) back = bare[idx0];
            tmpl[idx0] = back;
        }
        for (uint32_t d = lane + 64u; d < n_elems; d += 64) {
            uint32_t idx; int32_t val; bool on;
            elem_eval(elems[d], sc, A, L, HW, laser_layer, gem_layer, idx, val, on);
            if (on) tmpl[idx] = bare[idx];
        }
        if (is_agent_lane) tmpl[agent_idx] = 0;
    }
}

}  // namespace lle
