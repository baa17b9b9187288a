// step_kernel.hpp -- World.step with one lane per agent (the hot path) and the dispatch over its instantiations.
// Included by one translation unit per MODE (step_mode0.hip ... step_mode5.hip), so that the ~390 instantiations compile
// in parallel; kernels.hip holds world_kernel and the host-side launch logic.
#pragma once
#include "kernel_common.hpp"
#include "partial_stream.hpp"
#include "step_lanes.hpp"

namespace lle {

// ================================================================================================
// step_kernel<G, LM>: World.step() with one LANE PER AGENT (G = lanes per environment = power of two >= A).
//
// world_kernel<.., MODE_STEP> (kernels.hip) runs one environment per lane: simple, but phase 1 is then a ~4k-instruction
// dependency chain of ONE wave that no amount of occupancy shortens, and every wave runs it at the same time
// (launch cost = chain + observation stream).  Here the agents of an environment sit in G neighbouring lanes, so the
// per-agent loops of move_agents / compute_available_actions / the sampler become lane-parallel and the chain is
// ~G times shorter; a 64-lane wave carries 64/G environments and four times as many waves share each SIMD.
//
// What makes the split legal (same results as the sequential reference, src/core/world.rs:477-505):
//   * leaves of one pass commute (each only turns bits ON, and a leave skipped because an earlier one already lit its
//     bit would have been a no-op): beam |= OR over the group of every lane's suffix;
//   * pre-enters commute (each only clears a suffix): beam &= AND over the group of every lane's prefix;
//   * enter reads the beams (final after leave+pre-enter) and touches only the agent's own flags, its own cell's
//     gem and the occupant slot of its own cell (agents never share a cell), so the enters of one pass are
//     independent; their events are ordered by agent id = lane order (prefix count inside the group);
//   * the three loops stay in the reference's order, and passes repeat while any agent of the environment died.
// Cross-lane traffic is DPP quad permutes (G <= 4) / ds_swizzle (G = 8, 16); nothing goes through memory.

// GEN: the general instantiations -- environments with their own source colours / enabled flags (lle_batch_set_sources)
// and/or batches of several maps (lle_batch_create_multi).  Separate instantiations, so that the default path (one
// map, sources of the map) is compiled without any of it: every `PES` / `tables` / `initp` below folds to a constant.
// ML1: no cell of the map carries more than one laser layer (every level of the reference; no crossing beams): the
// per-layer loops run exactly once and unroll (no variable 64-bit shifts of the layer word).
// MODE 0: one step in place, one map, the map's sources (the default).  MODE 1: + fused rollout (n_steps, trajectory
// rings) and timeline stamps.  MODE 2 (general): + several maps.  MODE 3: + per-env sources (a mode of its own, not a
// run-time flag of MODE 2, so that neither keeps the other's reset state and colour words in registers: both are short
// of them).  MODE 4 / 5: MODE 2 / 3 without the rollout loop, rings and stamps -- the single-step launches of such
// batches (the loop-carried state and the ring pointers are what pushes 2 / 3 into scratch).
// MODE 6: MODE 0 for launches of one to two rounds of workgroups (kernels.hip: row_heads_pay): the rows' static head lines are
// stored before the state machine, and every load of the kernel is issued up front (see HEAD below).  An instantiation
// of its own: a launch without heads runs 0.2-0.4 us slower with that load order (level 1: 5.9 -> 6.1 us at 4 096 envs).
// Maps with at most 8 sources (with the beam masks in LDS, BM below, this lane's share of them is read up front into registers).
// MODE 7: the same for MODE 4 (several maps / the fused LLE.step outputs).
// MODE 8: the same for MODE 5 (per-environment sources: LLE.step with randomize_lasers): the head is the part of the row that no
// colouring can touch (tables.h pes_head_*: WALL / VOID / EXIT lines), stored from the BARE template; every load of the
// prologue -- the env's colours, enabled mask and own reset record among them -- is in flight before the table copy.
// LX >= 0: the exact number of sources, known at compile time (instantiated for the default path of maps with at
// most four sources: the per-beam loops lose their guards and the unused beam registers disappear; 0.4 us on level 6).
// A 64-byte struct at a wave-uniform address that only the host writes, through the scalar cache.
__device__ __forceinline__ EnvOutputs load_uniform(const EnvOutputs* p) {
    static_assert(sizeof(EnvOutputs) == 64, "one s_load_dwordx16");
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    u32x16 w;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p) : "memory");
    EnvOutputs o;
    __builtin_memcpy(&o, &w, sizeof o);
    return o;
}

// Where LaunchArgs.out sits in the kernel-argument segment of step_kernel(BatchPtrs, LaunchArgs): explicit arguments start at offset 0,
// each at its natural alignment.
constexpr size_t KERNARG_ENV_OUT = ((sizeof(BatchPtrs) + alignof(LaunchArgs) - 1) / alignof(LaunchArgs)) * alignof(LaunchArgs) + offsetof(LaunchArgs, out);
__device__ __forceinline__ const EnvOutputs* kernarg_env_out() {
    return reinterpret_cast<const EnvOutputs*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + KERNARG_ENV_OUT);  // (C cast: out of the constant address space)
}

template <int G, int LM, int MODE, bool ML1, int LX = -1>
__global__ void __launch_bounds__(256, (G >= 8 ? 3 : 4)) step_kernel(BatchPtrs P, LaunchArgs K) {
    // MODE 9: MODE 4 whose launch writes the PARTIAL k x k observation (python/lle/observations.py:312-369) from the hand-over records
    // instead of the layered rows (partial_stream.hpp): LLE.step of `obs_type="partial..."` in one launch.  The map's own sources.
    constexpr bool PARTIAL = MODE == 9;
    constexpr bool GEN = (MODE >= 2 && MODE <= 5) || MODE == 7 || MODE == 8 || PARTIAL, ROLL = MODE >= 1 && MODE <= 3;
    constexpr bool PES = MODE == 3 || MODE == 5 || MODE == 8;  // (the launcher picks these exactly when LAUNCH_PER_ENV_SOURCES is set)
    // More than 4 sources: the beam masks live in the env's LDS record instead of LM registers of every lane (step_lanes.hpp BM).
    // 4 agents and 8 sources: state machine 11.5 -> 8.9 us, config 5 (8 agents, 8 sources) 41.6 -> 20.1, 20 sources 121 -> 15.
    // (Round 2 kept the registers in the per-env-sources ROLLOUT mode, MODE 3, where the record form had faulted on the
    // 20-source map; on the round-3 kernels it does not: 8e8 env-steps of fused rollouts against single steps on five
    // many-source maps, tools/soak_mode3.py -- and those instantiations lose 100-700 B of scratch per lane.)
    constexpr bool BM = LM >= 8;
    constexpr int LR = BM ? 1 : LM;  // beam REGISTERS of a lane
    // Big rows (LAUNCH_SPLIT_ROWS, set by the launcher when private whole-row copies would leave one workgroup per CU):
    // the row is split over the wavefronts of the workgroup, see write_observations_split (obs_stream.hpp).  Only the
    // instantiations of maps with more than four agents carry it (rows of 16 KB and more with at most four agents
    // would need maps beyond 36 x 36; those stay on whole-row copies), and not the per-env-sources modes.
    constexpr bool CAN_SPLIT = G >= 8 && !PES && !PARTIAL;
    constexpr bool HEAD = MODE == 6 || MODE == 7 || MODE == 8;  // MODE 0 / 4 / 5 with the static lines of the rows ahead of the state machine (below)
    const bool split = CAN_SPLIT && (K.flags & LAUNCH_SPLIT_ROWS) != 0;
    // The default instantiation is one step in place and nothing else: the fused rollout (n_steps, trajectory rings)
    // and the timeline stamps run on the general one, so that their arguments do not occupy scalar registers here.
    // (-DLLE_STAMP_SINGLE: a diagnostic build whose single-step kernels stamp too -- tools/lle_prof.py stamps --single; never the shipped library)
#ifdef LLE_STAMP_SINGLE
    constexpr bool STAMPED = true;
#else
    constexpr bool STAMPED = ROLL;
#endif
    uint64_t* const stamps = STAMPED ? K.stamps : nullptr;
#undef LLE_STAMP
#define LLE_STAMP(i)                                                                              \
    do {                                                                                          \
        if (STAMPED && stamps && lane == 0) stamps[(uint64_t)wave_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // environments per wavefront: at most 64 / G; fewer (lanes left idle) when the batch is small, so that there
    // are enough wavefronts to spread phase 2 over the chip
    const uint32_t EPW = K.envs_per_wave < (uint32_t)(64 / G) ? K.envs_per_wave : (uint32_t)(64 / G);
    constexpr int NW = (2 * G + 7) / 8;             // 64-bit words of the event list (2 events per agent at most)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, K.flags);  // the block of environments this workgroup serves
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
    const uint8_t* __restrict__ tables = P.tables;
    const InitRecord* __restrict__ initp = P.init;
    uint32_t map_idx = 0u;
    if (GEN) {  // this workgroup's map (its envs never straddle two maps: the launcher sizes workgroups accordingly)
        const uint32_t EPW0 = K.envs_per_wave < (uint32_t)(64 / G) ? K.envs_per_wave : (uint32_t)(64 / G);
        map_idx = map_index_of(K, K.env_base + (int64_t)(blk * waves_per_wg) * EPW0);
        tables += (uint64_t)map_idx * K.table_stride;
        initp += map_idx;
    }
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    const int A = (int)hdr->A, L = LX >= 0 ? LX : (int)hdr->L, W = (int)hdr->W;
    const uint32_t a = lane & (G - 1), grp = lane / G;  // agent id, environment slot in the wave
    const int64_t As = agent_stride_of(A, L);           // env pitch of the per-agent buffers
    const int64_t env0 = K.env_base + (int64_t)wave_id * EPW;
    const int64_t env = env0 + grp;
    const bool env_ok = grp < EPW && env < K.env_limit;
    const bool me = env_ok && (int)a < A;           // this lane carries a real agent
    const bool write_obs = !PARTIAL && hdr->obs_supported && !(K.flags & STEP_NO_OBS);
    // Header fields that are needed late (after the first stores) are read HERE, as scalar loads next to the kernel
    // arguments: read where they are used they become vector loads from global memory with a full wait each, three of
    // them in a row between the state machine and the first observation store.
    const uint32_t h_HW = hdr->HW, h_obs_stride = hdr->obs_stride, h_n_chunks = hdr->n_chunks, h_D = hdr->D;
    // element type of the rows this launch writes (tables.h ObsElem: int8, or widened at the store) and their pitch in BYTES: recomputed from
    // the launch flags where they are used (macros, not variables: nothing new is live across the state machine); the kernels with row heads
    // are never launched widened (kernels.hip launch_step_kernel) and carry none of it
#define LLE_ET() (HEAD ? (uint32_t)OBS_I8 : (K.flags & LAUNCH_OBS_ELEM_MASK) >> LAUNCH_OBS_ELEM_SHIFT)
#define LLE_PITCH() ((uint64_t)h_obs_stride << obs_elem_shift(LLE_ET()))
    uint32_t h_beam_full[LR];
#pragma unroll
    for (int b = 0; b < LR; b++) h_beam_full[b] = (!BM && b < L) ? hdr->beam_full[b] : 0u;
    const uint32_t h_enabled = hdr->enabled_mask;
    const uint32_t h_chain = BM ? hdr->chain_mask : 0u;  // (beams longer than 32 cells: step_lanes.hpp BM walks the chain of words)
    // (with head stores ahead of them, later header reads would be VECTOR loads -- the scalar cache is not coherent with
    // the kernel's own stores, and the compiler cannot tell the header from the rows -- whose wait covers the stores too)
    const uint32_t h_off_cell_meta = hdr->off_cell_meta, h_off_dyn = hdr->off_dyn, h_off_template = hdr->off_template;
    const uint32_t h_max_layers = ML1 ? 1u : hdr->max_layers;
    // STEP_INCREMENTAL_OBS (tables.h): single steps in place with the map's own sources write only the lines dynamic state can change
    // (per-environment sources: the table of the per-env-sources section -- laser planes all dynamic)
    constexpr bool CAN_INCR = !ROLL && !PARTIAL && !HEAD;  // (the launcher sends incremental launches to the kernels without heads: no head to send ahead)
    const uint32_t h_off_dyn_chunks = CAN_INCR ? (PES ? hdr->off_pes_dyn_chunks : hdr->off_dyn_chunks) : 0u;
    const uint32_t h_n_dyn_chunks = CAN_INCR ? (PES ? hdr->n_pes_dyn_chunks : hdr->n_dyn_chunks) : 0u;
    const bool incr = CAN_INCR && (K.flags & STEP_INCREMENTAL_OBS) != 0 && h_n_dyn_chunks < h_n_chunks;
    const uint32_t h_off_recolour = PES ? hdr->off_recolour : 0u, h_off_bare = PES ? hdr->off_bare : 0u;
    const uint32_t h_off_elems = PES ? hdr->off_elems : 0u, h_n_elems = PES ? hdr->n_elems : 0u;
    // (MODE 7: so are the fused LLE.step outputs' descriptor and the header fields of that epilogue)
    constexpr bool ENV_OUT = MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8 || PARTIAL;
    EnvOutputs O_early = {};
    uint32_t h_G = 0, h_H = 0;
    constexpr bool EARLY_OUT = MODE == 7;  // (MODE 4 / 5 / 8 have no scalar registers to park the descriptor in: they spill vector registers for it;
                                           //  they fetch it where it is used through the scalar cache, load_uniform: no vmcnt wait behind the head stores)
    if (EARLY_OUT && K.env_out) O_early = K.out;
    if (ENV_OUT) { h_G = hdr->G; h_H = hdr->H; }
    uint32_t h_init_beams[LR];  // the reset state's beams (shared record; the per-env one is read where it is used)
#pragma unroll
    for (int b = 0; b < LR; b++) h_init_beams[b] = (!BM && b < L) ? initp->beams[b] : 0u;
    const int64_t n_here = (K.env_limit - env0) < (int64_t)EPW ? (K.env_limit - env0) : (int64_t)EPW;
    const uint32_t amask = (1u << A) - 1u;
    // (the row this wavefront starts its stream at, obs_stream.hpp row_rotation: computed where it is used, from values that are live
    // there anyway -- carried across the state machine it cost the tightest instantiations a spilled register)
    // Not in the per-env-sources rollout (MODE 3), whose instantiations sit at the register cap; the other rollouts have had room for it since
    // their late addresses are rebuilt (LLE_ENV_LATE): a 1 GB trajectory ring is written past the Infinity Cache, where the rotation pays.
#define LLE_ROT() ((ROLL && PES) ? 0u : row_rotation(wave_id, EPW, n_here, K.flags))
    LLE_STAMP(0);

    // ---- packed state: own position / availability, and the env-wide words replicated in the group's lanes
    uint32_t pos = 0xFFFF0000u + a, avail = 0, beams[LR];
    uint64_t raw_bits = 0;
    uint32_t gems = 0;
#pragma unroll
    for (int b = 0; b < LR; b++) beams[b] = 0;
    // Per-lane addresses of this env's / agent's records, computed once and kept in vector registers for the stores at
    // the end: the scalar base pointers are then dead during the state machine (scalar registers are the scarce
    // resource of this kernel, vector registers are not).
    const int64_t env_c = env_ok ? env : 0;
    uint64_t* const p_bits = P.bits + env_c;
    uint32_t* const p_gems = P.gems + env_c;
    uint32_t* const p_beams = P.beams + env_c * L;
    uint16_t* const p_pos = P.pos + env_c * As + a;
    uint8_t* const p_avail = P.avail + env_c * As + a;
    uint8_t* const p_err = P.err + env_c;
    uint8_t* const p_evcount = P.evcount + env_c;
    uint8_t* const p_events = P.events + env_c * 2 * As;
    uint8_t* const p_done = P.done + env_c;
    // The general mode is short of VECTOR registers instead (it spilled eight to scratch, and a scratch reload inside
    // the state machine is a memory round trip): there the addresses of the late stores are rebuilt from the scalar
    // bases where they are used, and the per-lane copies above die after the loads.
#define LLE_LATE(field, offset) ((GEN || ROLL) ? P.field + (offset) : p_##field)
    // ... REBUILT, not carried: `env_late` is the env index behind an empty asm, so the compiler cannot fold these addresses with the
    // prologue's (same expressions) and keep them alive across the state machine after all -- which it did, and then split their live
    // ranges with a copy placed AHEAD of the exec-restoring s_or_b64 of a divergent join (hipcc 7.2; the copy ran for the lanes of the
    // region only, the others stored their gems through a stale register: the fault of `step_kernel<4,4,4,false,-1>` on the `nested`
    // map, found with rocgdb's precise memory mode -- profiles/r04_pes_tax.md, tools/isa_exec_copy_scan.py).
#define LLE_ENV_LATE(name)                                   \
    int64_t name = env_c;                                    \
    if ((GEN || ROLL) && !HEAD) asm volatile("" : "+v"(name))   /* (the kernels with row heads have registers to spare and lose 0.3-0.5 us to the late arithmetic) */
    uint32_t init_pos_a = 0xFFFF0000u + a, init_avail_a = 0;  // this agent's reset position / availability
    uint64_t init_bits = 0;
    uint32_t init_gems = 0;
    // per-environment sources: colours (4 per word), enabled mask, and the env's own reset state
    constexpr int CWM = LM / 4;
    const int CW = src_stride_of(L) / 4;
    uint32_t colw[CWM], env_enabled = h_enabled;
    // the env's reset beams, read with the rest of its state in the single-step mode: read where the reset needs them they
    // are a memory round trip between the decision to reset and the state machine, in every wavefront that resets anything
    constexpr bool PRE_BEAMS = PES && !ROLL && LM <= 8;
    // Round 4: where the reset state depends on an env's colours through its beams ONLY (MapHeader.recolour_exact: no cell with more than
    // two laser layers, no chained beam words) the single-step kernels do not read the env's own reset record at all -- five loads per
    // lane ahead of the first store -- but take positions / flags / gems from the map's record and the beams from the table the
    // re-colouring uses (off_recolour: the mask of beam b after World::reset under colour c), by the env's colour and enabled flag.
    const bool shared_init = PES && !ROLL && hdr->recolour_exact != 0u;
    uint32_t env_init_beams[LR];
#pragma unroll
    for (int b = 0; b < LR; b++) env_init_beams[b] = 0u;
#pragma unroll
    for (int q = 0; q < CWM; q++) colw[q] = 0;
    // The caller's action, read with the rest of the state in the single-step modes: a load inside the step would be
    // waited for with a vmcnt(0) that every path executes -- the sampling path right behind its store of the sampled
    // action, i.e. it would wait for that store's acknowledgement before the state machine starts.
    uint32_t act_given = 4u;
    // HEAD && BM: the env's beam masks on their way to its LDS record (a + k * G is this lane's share), read with the rest
    constexpr int BPL = (HEAD && BM) ? (LM + G - 1) / G : 1;
    uint32_t bm_pre[BPL];
#pragma unroll
    for (int k = 0; k < BPL; k++) bm_pre[k] = 0u;
    // The wavefront's counters, likewise (kernel_common.hpp: flush_stats); the default single-step instantiations only --
    // the general ones have no registers to spare, a rollout flushes once per launch.
    constexpr bool PRE_STATS = (MODE == 0 || MODE == 6 || MODE == 7 || MODE == 8) && LM <= 8;  // (16 / 32 beam registers: already spilling; MODE 4 / 5: they spill more with it)
    int64_t stats_old = 0;
    // (a macro, not a lambda: with the beam registers captured by reference the 32-source instantiations kept them in scratch)
#define LLE_LOAD_STATE() \
    do { \
        if (env_ok) { \
            raw_bits = *p_bits; \
            gems = *p_gems; \
        _Pragma("unroll") \
            for (int b = 0; b < LR; b++) \
                if (!BM && b < L) beams[b] = p_beams[b]; \
        } \
        if (me) { \
            pos = (uint32_t)*p_pos; \
            avail = (uint32_t)*p_avail; \
            init_pos_a = (PES && !shared_init) ? (uint32_t)P.init_pos[env * As + a] : (uint32_t)initp->pos[a]; \
            init_avail_a = (PES && !shared_init) ? (uint32_t)P.init_avail[env * As + a] : (uint32_t)initp->avail[a]; \
            if (!ROLL && !(K.flags & STEP_SAMPLE_ACTIONS)) \
                act_given = K.actions_in ? (uint32_t)K.actions_in[env * A + a] : (uint32_t)P.actions[env * As + a]; \
        } \
        if (HEAD && BM && env_ok) { \
        _Pragma("unroll") \
            for (int k = 0; k < BPL; k++) \
                if ((int)a + k * G < L) bm_pre[k] = p_beams[(int)a + k * G]; \
        } \
        init_bits = initp->bits; \
        init_gems = initp->gems; \
        if (PRE_STATS) stats_old = stats_preload(P.stats, wave_id, lane); \
        if (PES && env_ok) { \
            env_enabled = P.src_enabled[env]; \
        _Pragma("unroll") \
            for (int q = 0; q < CWM; q++) \
                if (q < CW) colw[q] = reinterpret_cast<const uint32_t*>(P.src_colour)[env * CW + q]; \
            if ((K.flags & STEP_AUTO_RESET) && !shared_init) { \
                init_bits = P.init_bits[env]; \
                init_gems = P.init_gems[env]; \
        _Pragma("unroll") \
                for (int b = 0; b < LR; b++) \
                    if (PRE_BEAMS && b < L) env_init_beams[b] = P.init_beams[env * L + b]; \
            } \
        } \
    } while (0)

    // HEAD (the default instantiation): the lines of every row that no agent, beam or gem can change (tables.h head_lo /
    // head_n) are stored BEFORE the state machine, from the pristine template in global memory.  Every vector load of the
    // kernel is therefore issued first -- the head chunk, the state, the caller's actions, the table rows -- and has
    // returned before the first store: the vmcnt counter is in order, so a load waited for AFTER the head stores would
    // wait for their acknowledgements (and the compiler does not see the `sc1` stores, which are inline asm).
    const uint32_t head_lo = HEAD ? (PES ? hdr->pes_head_lo : hdr->head_lo) : 0u;
    const uint32_t head_n = (HEAD && write_obs && !split && !incr) ? (PES ? hdr->pes_head_n : hdr->head_n) : 0u;  // (incr: static lines are not written at all)
    // (per-environment sources: a second run of colour-independent lines behind the first, tables.h pes_head2_*)
    const uint32_t head2_lo = (HEAD && PES) ? hdr->pes_head2_lo : 0u, head2_n = (HEAD && PES && head_n) ? hdr->pes_head2_n : 0u;
    uint4 head_v = {0u, 0u, 0u, 0u};
    if (HEAD) {
        if (lane < head_n + head2_n)
            head_v = reinterpret_cast<const uint4*>(tables + (PES ? h_off_bare : h_off_template))[lane < head_n ? head_lo + lane : head2_lo + (lane - head_n)];
        LLE_LOAD_STATE();
    }
    // BM: [length masks | beams of the reset state], one copy per workgroup behind the tables; read here, ahead of the table rows
    uint32_t bt_full = 0u, bt_init = 0u;
    if (BM && (int)threadIdx.x < L) {
        bt_full = hdr->beam_full[threadIdx.x];
        bt_init = initp->beams[threadIdx.x];
    }
    // (split rows: the pristine static observation stays in global memory, every wavefront copies its slice from there)
    // (PARTIAL: no layered row is built, so the pristine template -- the tail of the table section -- stays out of LDS, as with split rows)
    const uint32_t tab_off = hdr->off_cell_lay;
    // (PARTIAL: the cell tables alone, in whole 1-KiB rows -- no dyn table, no template: four workgroups per CU must fit for a one-round launch)
    const uint32_t tab_cells = ((hdr->off_dyn - tab_off) + 1023u) & ~1023u;
    const uint32_t tab_bytes = PARTIAL ? (tab_cells < hdr->lds_table_bytes ? tab_cells : hdr->lds_table_bytes)
                                       : (split ? hdr->lds_split_table_bytes : hdr->lds_table_bytes);
    // (PARTIAL, windows of 3 / 5 / 7: the window sets of the launch's window size come in with the tables, right behind them -- tables.h)
    const bool use_sets = PARTIAL && K.win_sets != nullptr;
    const uint32_t ext_bytes = PES ? hdr->ext_bytes : (use_sets ? win_sets_bytes(hdr->HW) : 0u);
    // split rows: this wavefront's slice of the row, chunks [c_lo, c_hi), is built from the static observation's BITS (tables.h off_tmpl_bits: 2 B
    // per chunk at a fixed offset behind the header) -- requested HERE, ahead of the table rows, with the wavefront's state: the split-row
    // launch is the one whose workgroups each read their own map's tables from memory when a batch has thousands of maps, and every dependent
    // round trip of its prologue then costs a few microseconds in which the workgroup stores nothing (profiles/r05_multi_map.md).
    constexpr int PB = CAN_SPLIT ? 8 : 1;  // chunks per lane that come in early (slices up to 8 KiB; longer ones: the rest after the barrier)
    const uint32_t cpw = (h_n_chunks + waves_per_wg - 1u) / waves_per_wg;  // chunks per slice
    const uint32_t c_lo = wave_in_wg * cpw < h_n_chunks ? wave_in_wg * cpw : h_n_chunks;
    const uint32_t c_hi = c_lo + cpw < h_n_chunks ? c_lo + cpw : h_n_chunks;
    uint32_t pre_bits[PB];
    uint32_t pre_neg = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < PB; q++) pre_bits[q] = 0u;
    if (CAN_SPLIT && split) {
        const uint16_t* __restrict__ bits = reinterpret_cast<const uint16_t*>(tables + sizeof(MapHeader)) + c_lo;
#pragma unroll
        for (int q = 0; q < PB; q++)
            if (lane + 64u * q < c_hi - c_lo) pre_bits[q] = bits[lane + 64u * q];
        if (lane < TMPL_NEG_MAX) pre_neg = reinterpret_cast<const uint32_t*>(tables + sizeof(MapHeader) + tmpl_bits_bytes(h_n_chunks))[lane];
        if (!HEAD) LLE_LOAD_STATE();
    }
    if (PES) copy_tables2_to_lds(tables + tab_off, tab_bytes, tables + h_off_bare, ext_bytes, lds, lane, wave_in_wg, waves_per_wg);
    else if (use_sets) copy_tables2_to_lds(tables + tab_off, tab_bytes, K.win_sets + (uint64_t)map_idx * win_table_bytes(hdr->HW), ext_bytes, lds, lane, wave_in_wg, waves_per_wg);
    else if (CAN_SPLIT && GEN && split && (K.flags & LAUNCH_PACKED_TABLES))  // (many maps, few environments each: obs_stream.hpp expand_packed_tables)
        expand_packed_tables(tables + hdr->off_packed, lds, h_HW, hdr->packed_n_lay, h_off_cell_meta - tab_off, h_off_dyn - tab_off, h_off_template - h_off_dyn,
                             threadIdx.x, blockDim.x);
    else copy_tables_to_lds(tables + tab_off, lds, tab_bytes, lane, wave_in_wg, waves_per_wg);
    constexpr uint32_t bt_bytes = BM ? 2u * LM * 4u : 0u;
    uint32_t* beam_tab = reinterpret_cast<uint32_t*>(lds + tab_bytes + ext_bytes);
    if (BM && (int)threadIdx.x < L) {
        beam_tab[threadIdx.x] = bt_full;
        beam_tab[LM + threadIdx.x] = bt_init;
    }
    // PARTIAL: [non-empty bitmap of the map | colour byte of every beam word] behind the tables, one copy per workgroup
    const uint32_t pm_bytes = PARTIAL ? (use_sets ? 0u : partial_bitmap_bytes(hdr->H, (uint32_t)W)) + 32u : 0u;
    uint32_t* const part_bm = reinterpret_cast<uint32_t*>(lds + tab_bytes + ext_bytes + bt_bytes);
    uint8_t* const part_col = lds + tab_bytes + ext_bytes + bt_bytes + (pm_bytes - 32u);
    if (PARTIAL) {
        for (uint32_t w = threadIdx.x; w < (pm_bytes - 32u) / 4u; w += blockDim.x) part_bm[w] = 0u;
        if ((int)threadIdx.x < L) part_col[threadIdx.x] = hdr->beam_colour[threadIdx.x];
    }
    LLE_STAMP(7);
    if (HEAD) {
        // every load has returned (the table rows were the last ones, and they are in LDS): said with an s_waitcnt the
        // compiler can see, or it puts its own vmcnt(0) where the loaded values are first used -- inside the loop of
        // head stores and in the state machine -- where it would wait for the stores as well
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        const uint32_t hgroup = MODE != 6 ? 1u : ((K.flags & LAUNCH_HEAD_GROUP4) ? 4u : ((K.flags & LAUNCH_HEAD_GROUP2) ? 2u : 1u));  // (the default kernel only)
        if (hgroup == 1u) {
            if (head_n && n_here > 0) {
                if (K.flags & LAUNCH_WRITE_THROUGH) store_heads<true>(P.obs, h_obs_stride, env0, n_here, head_lo, head_n, head_v, lane, LLE_ROT(), head2_lo, head2_n);
                else store_heads<false>(P.obs, h_obs_stride, env0, n_here, head_lo, head_n, head_v, lane, LLE_ROT(), head2_lo, head2_n);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: not the head stores' acknowledgements
        } else {
            // Phase shift inside the workgroup (round 4): one wavefront of every `hgroup` stores the heads of the whole group's rows (they are
            // the same bytes for every row), the others go straight to their state machines -- their tails then stream while the head
            // wavefronts are still in theirs, instead of every wavefront of the CU being in the same phase at the same time.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (ahead of the head stores here: nobody waits for the storing wavefront)
            if (head_n && (wave_in_wg & (hgroup - 1u)) == 0u) {
                for (uint32_t q = 0; q < hgroup && wave_in_wg + q < waves_per_wg; q++) {
                    const int64_t e0q = env0 + (int64_t)q * EPW;
                    int64_t nq = K.env_limit - e0q;
                    nq = nq < 0 ? 0 : (nq > (int64_t)EPW ? (int64_t)EPW : nq);
                    if (nq > 0) {
                        if (K.flags & LAUNCH_WRITE_THROUGH) store_heads<true>(P.obs, h_obs_stride, e0q, nq, head_lo, head_n, head_v, lane);
                        else store_heads<false>(P.obs, h_obs_stride, e0q, nq, head_lo, head_n, head_v, lane);
                    }
                }
            }
        }
    } else {
        __syncthreads();  // the only workgroup barrier
        if (!(CAN_SPLIT && split)) LLE_LOAD_STATE();  // (split rows: requested ahead of the table rows, above)
    }

    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds);
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (h_off_cell_meta - tab_off));
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + (h_off_dyn - tab_off));
    const uint16_t* dyn_chunks = PES ? reinterpret_cast<const uint16_t*>(lds + tab_bytes + (h_off_dyn_chunks - h_off_bare))
                                     : reinterpret_cast<const uint16_t*>(lds + (h_off_dyn_chunks - tab_off));
    const uint32_t scr_stride = (uint32_t)(L + A + 2 + (PES ? CW : 0)) | 1u;
    // a wavefront's private area: [row template | hand-over records]; PARTIAL: [E rows of the partial observation + 16 B | records]
    const uint32_t part_pitch = PARTIAL ? (((uint32_t)(A * (2 * A + 3)) * K.partial_k * K.partial_k + 15u) & ~15u) : 0u;
    const uint32_t row_area = PARTIAL ? K.partial_E * part_pitch + 16u : h_obs_stride;
    const uint32_t priv_bytes = row_area + (PARTIAL ? ((EPW * scr_stride * 4u + 15u) & ~15u) : 64u * scr_stride * 4u) + (PES ? PES_WAVE_EXTRA_BYTES : 0u);  // (PARTIAL: a record per environment of the wavefront, no spare slots)
    int8_t* tmpl = reinterpret_cast<int8_t*>(lds + tab_bytes + ext_bytes + bt_bytes + pm_bytes + wave_in_wg * priv_bytes);
    uint32_t* scratch = reinterpret_cast<uint32_t*>(tmpl + row_area);
    const int8_t* bare = reinterpret_cast<const int8_t*>(lds + tab_bytes);
    const uint32_t* elems = reinterpret_cast<const uint32_t*>(lds + tab_bytes + (h_off_elems - h_off_bare));
    // split rows: [tables | one slice per wavefront | the hand-over records of all the workgroup's environments]
    if (split) {
        tmpl = reinterpret_cast<int8_t*>(lds + tab_bytes + bt_bytes + wave_in_wg * cpw * 16u);
        scratch = reinterpret_cast<uint32_t*>(lds + tab_bytes + bt_bytes + waves_per_wg * cpw * 16u) + wave_in_wg * EPW * scr_stride;
        uint4* mine = reinterpret_cast<uint4*>(tmpl);
        if (hdr->off_tmpl_bits) {
            // Four bits to four bytes: b_i lands on bit 8 i of nibble * (1 + 2^7 + 2^14 + 2^21) (the sixteen partial products fall on distinct
            // bits: no carries).
            auto expand = [](uint32_t b) {
                return uint4{((b & 15u) * 0x00204081u) & 0x01010101u, (((b >> 4) & 15u) * 0x00204081u) & 0x01010101u,
                             (((b >> 8) & 15u) * 0x00204081u) & 0x01010101u, ((b >> 12) * 0x00204081u) & 0x01010101u};
            };
#pragma unroll
            for (int q = 0; q < PB; q++)
                if (lane + 64u * q < c_hi - c_lo) mine[lane + 64u * q] = expand(pre_bits[q]);
            const uint16_t* __restrict__ bits = reinterpret_cast<const uint16_t*>(tables + sizeof(MapHeader)) + c_lo;
            for (uint32_t c = lane + 64u * PB; c < c_hi - c_lo; c += 64) mine[c] = expand(bits[c]);  // (slices of more than 8 KiB)
            // ... and the -1 marks of the sources (LDS serves a wavefront's instructions in order: these bytes land on the chunks written above)
            if (lane < hdr->tmpl_neg_n && pre_neg >= c_lo * 16u && pre_neg < c_hi * 16u) tmpl[pre_neg - c_lo * 16u] = (int8_t)-1;
        } else {
            const uint4* __restrict__ pristine = reinterpret_cast<const uint4*>(tables + h_off_template) + c_lo;
            for (uint32_t c = lane; c < c_hi - c_lo; c += 64) mine[c] = pristine[c];
        }
    } else if (PARTIAL) {
        if (!use_sets) partial_bitmap_fill(part_bm, cell_lay, cell_meta, (int)hdr->H, W);  // (complete behind the barrier in front of the writer, below)
    } else {
        const uint4* pristine = PES ? reinterpret_cast<const uint4*>(bare) : reinterpret_cast<const uint4*>(lds + (h_off_template - tab_off));
        uint4* mine = reinterpret_cast<uint4*>(tmpl);
        for (uint32_t c = lane; c < h_n_chunks; c += 64) mine[c] = pristine[c];
    }
    // BM: the env's beam masks go straight into its hand-over record, which phase 2 reads and the state machine updates in place
    uint32_t* const bm = scratch + grp * scr_stride + 1;
    if (HEAD && BM) {  // (read up front; only the LDS writes are left here, behind the head stores)
        if (env_ok) {
#pragma unroll
            for (int k = 0; k < BPL; k++)
                if ((int)a + k * G < L) bm[(int)a + k * G] = bm_pre[k];
        }
    } else if (BM && env_ok) {
        for (int b = (int)a; b < L; b += G) bm[b] = p_beams[b];
    }
    wave_sync();
    // Every load of the prologue has to be back before the state machine starts anyway.  Saying so with an s_waitcnt the
    // compiler sees keeps it from carrying the loads that only some paths consume (the reset record) as pending: it
    // guards later reuses of their registers with vmcnt(0), and in hardware that also waits for every observation store
    // in flight (the level-1 instantiation had two such waits behind its stream: 14.2 -> 14.8 us at 65 536 envs).
    if (!HEAD) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); HEAD: said before the head stores
    LLE_STAMP(1);

    uint32_t alive = (uint32_t)raw_bits & 0xFFFFu, arrived = (uint32_t)(raw_bits >> 16) & 0xFFFFu, occ = (uint32_t)(raw_bits >> 32) & 0xFFFFu;
    // agents that set_state flagged dead WITHOUT a death event (tables.h GHOST_SHIFT): LLE.compute_done counts events, so
    // they do not end the episode (python/lle/env/env.py:208-217,253-254).  Zero except after such a set_state.
    uint32_t ghost = (uint32_t)(raw_bits >> GHOST_SHIFT);
    const uint32_t enabled = PES ? env_enabled : h_enabled, max_layers = h_max_layers;
    LLE_STAMP(2);

    // ---- n_steps consecutive steps of the wave's environments; the state stays in registers in between.
    // (n_steps = 1 is World.step; more is a fused rollout with on-device action sampling, lle_batch_rollout.)
    StepCounts cnt = {0, 0, 0, 0, 0, 0, 0};
    // Fused rollouts: the seven per-env counters of the launch live in the wavefront's spare record slots (LDS) instead of seven vector
    // registers carried across the steps -- `ds_add_u32` without a return per step, read back once behind the loop.  The per-env-sources
    // rollout (MODE 3) sits at the register cap.
    constexpr bool CNT_LDS = ROLL && PES && G >= 2;   // (tables.h PES_WAVE_EXTRA_BYTES: behind the wavefront's 64 record slots)
    uint32_t* const cnt_lds = scratch + 64u * scr_stride + grp * 8u;   // (8 words per environment; only lane a == 0 touches them)
    if (CNT_LDS && env_ok && a == 0) {
#pragma unroll
        for (int q = 0; q < 7; q++) cnt_lds[q] = 0u;
    }
    const uint32_t n_steps = ROLL ? (K.n_steps ? K.n_steps : 1u) : 1u;
    for (uint32_t it = 0; it < n_steps; it++) {
    const uint64_t t_now = K.t + it;
    // where this step's observation / actions / reward counts go: in place, or slot (ring_pos + step) % ring_slots of
    // the trajectory rings (the launcher passes ring_pos already reduced modulo ring_slots; the slot advances by
    // increment, so the single-step path carries no 64-bit division)
    uint8_t* __restrict__ actions_out = P.actions;
    uint32_t* __restrict__ reward_out = P.reward;
    int8_t* __restrict__ obs_out = P.obs;
    if (ROLL && K.ring_slots) {
        uint32_t slot = (uint32_t)K.ring_pos + it;
        while (slot >= K.ring_slots) slot -= K.ring_slots;
        actions_out = K.ring_actions + (int64_t)slot * K.ring_env_count * As;
        reward_out = K.ring_reward + (int64_t)slot * K.ring_env_count;
        obs_out = K.ring_obs + (int64_t)slot * K.ring_env_count * (int64_t)LLE_PITCH();
    }

    // ---- auto-reset: a finished env restarts from the reset state (identical for every env, see InitRecord)
    uint32_t was_reset = 0;
    if (K.flags & STEP_AUTO_RESET) {
        const bool over = env_ok && ((alive | ghost) != amask || arrived == amask);
        pos = (over && me) ? init_pos_a : pos;
        avail = (over && me) ? init_avail_a : avail;
        alive = over ? ((uint32_t)init_bits & 0xFFFFu) : alive;
        arrived = over ? ((uint32_t)(init_bits >> 16) & 0xFFFFu) : arrived;
        occ = over ? ((uint32_t)(init_bits >> 32) & 0xFFFFu) : occ;
        ghost = over ? 0u : ghost;
        gems = over ? init_gems : gems;
#pragma unroll
        for (int b = 0; b < LR; b++)
            if (!BM && b < L && !(PES && shared_init))
                beams[b] = over ? (PES ? (PRE_BEAMS ? env_init_beams[b] : P.init_beams[env_ok ? env * L + b : 0]) : h_init_beams[b]) : beams[b];
        if (BM && over && !(PES && shared_init))  // the group's lanes share the copy of the record
            for (int b = (int)a; b < L; b += G) bm[b] = PES ? P.init_beams[env * L + b] : beam_tab[LM + b];
        if (PES && shared_init && __ballot(over) != 0ull) {  // the env's beams after World::reset, from its colours and enabled flags
            const uint32_t* rtab0 = reinterpret_cast<const uint32_t*>(lds + tab_bytes + (h_off_recolour - h_off_bare));
            if (!BM) {
#pragma unroll
                for (int b = 0; b < LR; b++)
                    if (b < L) {
                        const uint32_t c = (colw[b >> 2] >> ((b & 3) * 8)) & 0xFFu;
                        const uint32_t m = ((env_enabled >> b) & 1u) ? rtab0[b * (A + 1) + 1 + (int)(c < (uint32_t)A ? c : 0u)] : 0u;
                        beams[b] = over ? m : beams[b];
                    }
            } else if (over) {
                for (int b = (int)a; b < L; b += G) {
                    const uint32_t c = (colw[b >> 2] >> ((b & 3) * 8)) & 0xFFu;
                    bm[b] = ((env_enabled >> b) & 1u) ? rtab0[b * (A + 1) + 1 + (int)(c < (uint32_t)A ? c : 0u)] : 0u;
                }
            }
        }
        was_reset = over ? 1u : 0u;
        // LLE.reset with randomize_lasers (python/lle/env/env.py:189-203): world.reset() -- above, under the colours the env
        // HAD: beams cut at reset stay as they are -- then a fresh colour for every source, uniform over the colours the
        // source may take (MapHeader.colour_ok; every colour on the maps the reference's draw never fails on).  The draw is
        // the action sampler's hash keyed with seed ^ RECOLOUR_SALT and the source id in place of the agent; lane a of
        // the group decides colour a, a group OR hands the pick to all.  The env's reset record changes with the colours
        // in its beams only (tables.h off_recolour).
        if (PES && !ROLL && (K.flags & STEP_RECOLOUR_RESETS) && __ballot(over) != 0ull) {  // (single steps: the rollout modes have no registers for it)
            const uint32_t* rtab = reinterpret_cast<const uint32_t*>(lds + tab_bytes + (h_off_recolour - h_off_bare));
            const uint64_t rkey = action_step_key(K.seed ^ RECOLOUR_SALT, t_now);
#pragma unroll
            for (int b = 0; b < LM; b++) {
                if (b < L) {
                    const uint32_t* row = rtab + b * (A + 1);
                    const uint32_t ok = row[0] & amask;
                    const uint32_t field = action_field(action_hash_pair(rkey, (uint64_t)(K.env_offset + env), (uint32_t)b >> 1), (uint32_t)b);
                    const uint32_t k = (field * (uint32_t)__popc(ok)) >> 16;
                    const bool mine = ((ok >> a) & 1u) != 0u && (uint32_t)__popc(ok & ((1u << a) - 1u)) == k;
                    const uint32_t pick = grp_or<G>(mine ? a + 1u : 0u);  // colour + 1; 0: no colour is allowed, keep
                    const uint32_t sh = (uint32_t)(b & 3) * 8u, old_c = (colw[b >> 2] >> sh) & 0xFFu;
                    const uint32_t c = (over && pick) ? pick - 1u : old_c;
                    colw[b >> 2] = (colw[b >> 2] & ~(0xFFu << sh)) | (c << sh);
                    if (over && a == 0) P.init_beams[env * L + b] = ((env_enabled >> b) & 1u) ? row[1u + (c < (uint32_t)A ? c : 0u)] : 0u;
                }
            }
            if (over && a == 0) {
#pragma unroll
                for (int q = 0; q < CWM; q++)
                    if (q < CW) reinterpret_cast<uint32_t*>(P.src_colour)[env * CW + q] = colw[q];
            }
        }
    }

    // ---- joint action: sampled on the device, or given
    uint32_t act = 4u;
    if (K.flags & STEP_SAMPLE_ACTIONS) {
        const uint32_t hp = action_hash_pair(action_step_key(K.seed, t_now), (uint64_t)(K.env_offset + env), a >> 1);
        act = sample_action(avail, action_field(hp, a));
        if (me) small_store(&actions_out[env * As + a], (uint8_t)act);
    } else if (!ROLL) {
        act = act_given;
        if (me && K.actions_in) P.actions[env * As + a] = (uint8_t)act;
    } else if (K.actions_in) {
        if (me) {
            act = (uint32_t)K.actions_in[env * A + a];  // caller's buffer: contiguous [n][A]
            P.actions[env * As + a] = (uint8_t)act;
        }
    } else if (me) {
        // LLE_BUF_ACTIONS as filled by the caller -- or, in a fused rollout with rings, this step's slot of the ACTION ring:
        // without on-device sampling the ring is the rollout's input (lle_batch_rollout)
        act = (uint32_t)actions_out[env * As + a];
    }

    // ---- World.step for the lanes of this environment: availability check, vertex conflicts, move_agents passes
    // (step_lanes.hpp: the same source runs on the host under sanitizers, tests/hostsim)
    uint64_t evw[NW];
    uint32_t n_ev, meta_step, err;
    bool stepped;
    step_lanes<G, LR, ML1, PES, CWM, true, BM>(cell_lay, cell_meta, A, L, W, max_layers, h_beam_full, a, me, env_ok, enabled, colw, act, pos, avail,
                                               alive, arrived, occ, gems, beams, err, evw, n_ev, meta_step, stepped, nullptr, bm, beam_tab, h_chain);
    LLE_STAMP(3);

    // ---- everything of the step that the observation does not need: availability masks (compute_available_actions,
    // world.rs:343-363), error code, ordered event list, done flag, reward counts, counters.  The observation needs
    // positions, beams and gems only, so a wavefront of the OLDER half of the grid (the one the SIMD serves first, i.e.
    // the one whose first store ends the idle time of the memory system) does this after its stream, a younger one --
    // which waits for memory anyway and would otherwise add it to the end of the launch -- before.
    auto post_step = [&]() {
    LLE_ENV_LATE(env_p);
    if (stepped) avail_lanes<G>(a, me, pos, occ, alive, arrived, meta_step, avail);
#pragma unroll
    for (int k = 0; k < NW; k++) evw[k] = grp_or64<G>(evw[k]);
    if (env_ok && a == 0) {
        small_store(LLE_LATE(err, env_p), (uint8_t)err);
        small_store(LLE_LATE(evcount, env_p), (uint8_t)(n_ev | (was_reset << 7)));
        {
            uint8_t* row = LLE_LATE(events, env_p * 2 * As);  // 2*As bytes per env; this kernel fills the first 2*G
            if (G >= 2) {
                uint32_t* __restrict__ w = reinterpret_cast<uint32_t*>(row);
#pragma unroll
                for (int k = 0; k < G / 2; k++) small_store(&w[k], (uint32_t)(evw[k >> 1] >> ((k & 1) * 32)));
            } else {
                small_store(reinterpret_cast<uint16_t*>(row), (uint16_t)evw[0]);
            }
        }
        small_store(LLE_LATE(done, env_p), (uint8_t)(((alive | ghost) != amask || arrived == amask) ? 1 : 0));
        uint32_t n_died = 0, n_gem = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            n_died += (uint32_t)__popcll(evw[k] & 0x2020202020202020ull);
            n_gem += (uint32_t)__popcll(evw[k] & 0x1010101010101010ull);
        }
        const uint32_t n_exit = n_ev - n_died - n_gem;
        const uint32_t bonus = (err == 0 && arrived == amask) ? 1u : 0u;
        small_store(&reward_out[env], n_gem | (n_exit << 8) | (n_died << 16) | (bonus << 24));
        if (CNT_LDS) {
            const uint32_t add[7] = {1u, n_gem, n_exit, n_died, err != 0 ? 1u : 0u, was_reset, bonus};
#pragma unroll
            for (int q = 0; q < 7; q++) __hip_atomic_fetch_add(&cnt_lds[q], add[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            cnt.steps += 1u; cnt.gems += n_gem; cnt.exits += n_exit; cnt.died += n_died;
            cnt.invalid += err != 0 ? 1u : 0u; cnt.resets += was_reset; cnt.bonus += bonus;
        }
    }
    // ---- LLE.step's other outputs (python/lle/env/env.py:165-187), fused: what lle_batch_env_outputs writes in a launch
    // of its own (4.9 us at 65 536 envs, all of it launch boundary), written here from registers.  walkable_lasers only
    // (the availability without moves into foreign beams needs the neighbours' laser stacks: lle_batch_env_outputs).
    // Only the general single-step instantiations (MODE 4 / 5) carry it -- the launcher routes a launch with outputs there --
    // so that the default path keeps its registers (with the epilogue in MODE 0: 112 -> 127 VGPRs and 21.2 -> 21.6 us
    // for launches that do not even use it).
    if (ENV_OUT && K.env_out) {
        // (read where it is used; through the SCALAR cache by hand: behind the kernel's own stores the compiler would fetch it
        // with vector loads, whose wait also covers every store in flight -- the descriptor is written by the host only)
        const EnvOutputs O = EARLY_OUT ? O_early : load_uniform(kernarg_env_out());
        const int n_gems = (int)h_G, len = 3 * A + n_gems;
        if (me) {
            const int64_t ia = env * A + a;
            if (O.alive) small_store(&O.alive[ia], (uint8_t)((alive >> a) & 1u));
            if (O.arrived) small_store(&O.arrived[ia], (uint8_t)((arrived >> a) & 1u));
            if (O.available) {
                uint8_t* o = O.available + ia * 5;
#pragma unroll
                for (int k = 0; k < 5; k++) small_store(&o[k], (uint8_t)((avail >> k) & 1u));
            }
        }
        // WorldState.as_array: [i0, j0, ..., gems..., alive...] (pyworld_state.rs:79-101); normalised: divided in float64, rounded to
        // float32 on assignment (observations.py:145-175)
#define LLE_STATE_IJ()                                               \
    float fi = (float)(pos & 0xFFu), fj = (float)(pos >> 8);         \
    if (O.normalize_state) {                                         \
        fi = (float)((double)(pos & 0xFFu) / (double)h_H);           \
        fj = (float)((double)(pos >> 8) / (double)W);                \
    }
        // `state` (len floats per env) is contiguous over the wavefront's environments: in the kernels with row heads, where a row is a
        // whole number of 16-byte chunks, it is built in the spare slots of the wavefront's record area (64 slots, EPW in use) and copied
        // out whole -- one 16-byte store per lane instead of three or four partial ones per row (level 6: `state` cost 0.7 us of a step,
        // profiles/r04_pes_tax.md).  Those kernels only: full rows follow the state machine there and hide the copy; in a launch without
        // rows, or with incremental ones, its LDS round trip is exposed (10.9 -> 11.2 us without rows).  Elsewhere, and where the slots do
        // not hold it, every lane stores its own values.
        if (HEAD) {
            const uint32_t st_off = (EPW * scr_stride * 4u + 15u) & ~15u;
            // (16-byte stores: the caller's `state` pointer must be 16-byte aligned for them -- a torch view at a 4-byte offset is not --,
            // otherwise every lane stores its own values as in the kernels without heads)
            const bool staged = !split && O.state && (len & 3) == 0 && (reinterpret_cast<uintptr_t>(O.state) & 15u) == 0u &&
                                st_off + EPW * (uint32_t)len * 4u <= 64u * scr_stride * 4u;
            if (O.state) {
                float* st = staged ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(scratch) + st_off) + grp * (uint32_t)len : O.state + env * len;
                if (me) {
                    LLE_STATE_IJ()
                    st[2 * a] = fi;
                    st[2 * a + 1] = fj;
                    st[2 * A + n_gems + (int)a] = ((alive >> a) & 1u) ? 1.0f : 0.0f;
                }
                if (env_ok)
                    for (int g = (int)a; g < n_gems; g += G) st[2 * A + g] = ((gems >> g) & 1u) ? 1.0f : 0.0f;
                if (staged) {
                    wave_sync();  // (LDS operations of a wavefront execute in order)
                    const float4* src4 = reinterpret_cast<const float4*>(reinterpret_cast<uint8_t*>(scratch) + st_off);
                    float4* dst4 = reinterpret_cast<float4*>(O.state + env0 * len);  // (len * 4 is a multiple of 16: so is env0 * len * 4)
                    const uint32_t n4 = n_here > 0 ? (uint32_t)n_here * (uint32_t)len / 4u : 0u;
                    for (uint32_t c = lane; c < n4; c += 64u) dst4[c] = src4[c];
                    wave_sync();
                }
            }
        } else {
            if (me && O.state) {
                float* st = O.state + env * len;
                LLE_STATE_IJ()
                small_store(&st[2 * a], fi);
                small_store(&st[2 * a + 1], fj);
                small_store(&st[2 * A + n_gems + (int)a], ((alive >> a) & 1u) ? 1.0f : 0.0f);
            }
            if (env_ok && O.state)
                for (int g = (int)a; g < n_gems; g += G) small_store(&O.state[env * len + 2 * A + g], ((gems >> g) & 1u) ? 1.0f : 0.0f);
        }
#undef LLE_STATE_IJ
        if (env_ok && a == 0) {
            if (O.done) small_store(&O.done[env], (uint8_t)(((alive | ghost) != amask || arrived == amask) ? 1 : 0));
            if (O.reward) {
                uint32_t n_died = 0, n_gem = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    n_died += (uint32_t)__popcll(evw[k] & 0x2020202020202020ull);
                    n_gem += (uint32_t)__popcll(evw[k] & 0x1010101010101010ull);
                }
                const float gem = (float)n_gem, died = (float)n_died, ex = (float)(n_ev - n_died - n_gem);
                const float bonus = (err == 0 && arrived == amask) ? 1.0f : 0.0f;
                if (O.reward_kind == 0) {
                    small_store(&O.reward[env], gem + ex - died + bonus);  // reward_strategy.py:58-75
                } else {                                      // reward_strategy.py:90-109: a death zeroes the others
                    const bool dead = n_died > 0;
                    float* o = O.reward + env * 4;
                    small_store(&o[0], dead ? 0.f : gem); small_store(&o[1], dead ? 0.f : ex); small_store(&o[2], -died); small_store(&o[3], dead ? 0.f : bonus);
                }
            }
        }
    }
    };  // post_step

    if (env_ok && a == 0) {
        // hand-over record of this env for phase 2: [0 | beam masks | ~gem bits | ...
        uint32_t* sc = scratch + grp * scr_stride;
        sc[0] = 0u;
#pragma unroll
        for (int b = 0; b < LR; b++)
            if (!BM && b < L) sc[1 + b] = beams[b];  // (BM: the masks are there already)
        sc[L + 1] = ~gems;
        if (PES) {
#pragma unroll
            for (int q = 0; q < CWM; q++)
                if (q < CW) sc[L + 2 + A + q] = colw[q];
        }
    }
    if (me) scratch[grp * scr_stride + L + 2 + a] = PARTIAL ? pos : a * h_HW + cell_of(pos, W);  // ... | byte index of each agent (PARTIAL: its packed position)]
    if (split) {
        // the records of the whole workgroup must be in LDS before any wavefront streams its slice.  LDS only: waiting
        // for vmcnt here (what __syncthreads() does) would hold every step of a fused rollout until the previous step's
        // observation stores have been acknowledged.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        wave_sync();
    }
    LLE_STAMP(4);
    // (deferring it in the single-step launches of MODE 1 / 2 as well measured 1.3-1.5 us SLOWER there: those
    // instantiations already spill, and the deferral lengthens the live ranges)
    const bool post_first = ROLL || PARTIAL || (K.flags & LAUNCH_POST_FIRST) || (!(K.flags & LAUNCH_POST_LAST) && blockIdx.x * 4u >= gridDim.x * 3u);  // (PARTIAL: the writer wants the state machine's registers)
    if (post_first) post_step();

    // (the store policy and the element width of the launch become template arguments here, outside the loops over the environments)
    const uint32_t et = LLE_ET();
    if (split) {
        if (write_obs) {
            const int64_t wg_env0 = K.env_base + (int64_t)(blk * waves_per_wg) * EPW;
            int64_t n_wg = K.env_limit - wg_env0;
            n_wg = n_wg < 0 ? 0 : (n_wg > (int64_t)(waves_per_wg * EPW) ? (int64_t)(waves_per_wg * EPW) : n_wg);
            const uint32_t* records = reinterpret_cast<const uint32_t*>(lds + tab_bytes + bt_bytes + waves_per_wg * cpw * 16u);
            dispatch_stream<!HEAD>(K.flags, [&](auto wt_, auto wide_) {
                constexpr bool WT = decltype(wt_)::value, WIDE = decltype(wide_)::value;
                if (CAN_INCR && incr) write_observations_split<WT, true, WIDE>(A, L, h_D, c_lo, c_hi, LLE_PITCH(), dyn, tmpl, records, scr_stride, obs_out, wg_env0, n_wg, lane, dyn_chunks, h_n_dyn_chunks, et);
                else write_observations_split<WT, false, WIDE>(A, L, h_D, c_lo, c_hi, LLE_PITCH(), dyn, tmpl, records, scr_stride, obs_out, wg_env0, n_wg, lane, nullptr, 0u, et);
            });
        }
        // fused rollout: the next step's records overwrite these
        if (ROLL && n_steps > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (PARTIAL) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the workgroup's bitmap is complete (LDS only: no wait for stores)
        const EnvOutputs O = load_uniform(kernarg_env_out());
        if (O.partial && n_here > 0)
            write_partial<true>(A, L, W, (int)K.partial_k, part_pitch, K.partial_E, h_max_layers, cell_lay, cell_meta, part_bm, part_col, tmpl, scratch,
                                scr_stride, O.partial, env0, n_here, lane, use_sets ? reinterpret_cast<const uint64_t*>(lds + tab_bytes) : nullptr, et);
    } else if (write_obs && n_here > 0) {
        dispatch_stream<!HEAD>(K.flags, [&](auto wt_, auto wide_) {  // see stream_store (obs_stream.hpp)
            constexpr bool WT = decltype(wt_)::value, WIDE = decltype(wide_)::value;
            if (PES && CAN_INCR && incr)
                write_observations_env<WT, false, true, WIDE>(A, L, h_HW, h_n_elems, h_n_chunks, LLE_PITCH(), elems, bare, tmpl, scratch, scr_stride, obs_out, env0, n_here,
                                                              lane, nullptr, 0xFFFFFFFFu, 0u, 0u, LLE_ROT(), dyn_chunks, h_n_dyn_chunks, 0u, 0u, et);
            else if (PES)
                write_observations_env<WT, HEAD, false, WIDE>(A, L, h_HW, h_n_elems, h_n_chunks, LLE_PITCH(), elems, bare, tmpl, scratch, scr_stride, obs_out, env0, n_here,
                                                              lane, nullptr, 0xFFFFFFFFu, head_lo, head_n, LLE_ROT(), nullptr, 0u, head2_lo, head2_n, et);
            else if (CAN_INCR && incr)
                write_observations<WT, false, true, WIDE>(A, L, h_D, h_n_chunks, LLE_PITCH(), dyn, tmpl, scratch, scr_stride, obs_out, env0, n_here, lane, 0u, 0u, LLE_ROT(),
                                                          dyn_chunks, h_n_dyn_chunks, et);
            else
                write_observations<WT, HEAD, false, WIDE>(A, L, h_D, h_n_chunks, LLE_PITCH(), dyn, tmpl, scratch, scr_stride, obs_out, env0, n_here, lane, head_lo, head_n,
                                                          LLE_ROT(), nullptr, 0u, et);
        });
    }
    wave_sync();
    if (!post_first) post_step();
    }  // steps
    LLE_STAMP(5);

    // ---- final state.  Written unconditionally: an env whose action was refused kept its registers unchanged
    // (world.rs:436-453: errors precede any mutation), so this rewrites the same bytes.
    LLE_ENV_LATE(env_f);
    if (me) {
        small_store(LLE_LATE(pos, env_f * As + a), (uint16_t)pos);
        small_store(LLE_LATE(avail, env_f * As + a), (uint8_t)avail);
    }
    if (env_ok && a == 0) {
        small_store(LLE_LATE(bits, env_f), (uint64_t)alive | ((uint64_t)arrived << 16) | ((uint64_t)occ << 32) | ((uint64_t)ghost << GHOST_SHIFT));
        small_store(LLE_LATE(gems, env_f), gems);
        uint32_t* const beams_out = LLE_LATE(beams, env_f * L);
#pragma unroll
        for (int b = 0; b < LR; b++)
            if (!BM && b < L) small_store(&beams_out[b], beams[b]);
    }
    if (BM && env_ok) {
        uint32_t* const beams_out = LLE_LATE(beams, env_f * L);
        for (int b = (int)a; b < L; b += G) small_store(&beams_out[b], bm[b]);
#undef LLE_LATE
#undef LLE_ENV_LATE
#undef LLE_LOAD_STATE
#undef LLE_ROT
#undef LLE_ET
#undef LLE_PITCH
    }
    if (CNT_LDS && env_ok && a == 0) {
        cnt.steps = cnt_lds[0]; cnt.gems = cnt_lds[1]; cnt.exits = cnt_lds[2]; cnt.died = cnt_lds[3];
        cnt.invalid = cnt_lds[4]; cnt.resets = cnt_lds[5]; cnt.bonus = cnt_lds[6];
    }
    flush_stats(P.stats, wave_id, cnt, A, lane, PRE_STATS, stats_old);
    if (STAMPED && stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LLE_STAMP(6);
    }
}

// ---- dispatch of one MODE over G (lanes per environment), LM (beam registers), ML1 and -- for maps with at most four
// sources and no crossing beams -- the exact source count LX
template <int G, int LM, int MODE, bool ML1, int LX = -1>
static hipError_t launch_step_glp(const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    // which instantiations this process has launched (kernels.h debug registry: the coverage test of tests/test_gpu_instantiations.py);
    // LAUNCH_DRY_RUN: the walk of lle_debug_reachable through this very dispatch -- note the instantiation, launch nothing
    static std::atomic<uint32_t> noted{0};
    const uint32_t note_bit = (K.flags & LAUNCH_DRY_RUN) ? 2u : 1u;
    if (!(noted.load(std::memory_order_relaxed) & note_bit)) {
        noted.fetch_or(note_bit, std::memory_order_relaxed);
        debug_note(debug_key(DBG_STEP, G, LM, MODE, ML1, LX), (K.flags & LAUNCH_DRY_RUN) != 0);
    }
    if (K.flags & LAUNCH_DRY_RUN) return hipSuccess;
    dim3 grid((n_waves + wpw - 1) / wpw), block(64 * wpw);
    if (lds > 64 * 1024) {  // gfx950 has 160 KiB of LDS per CU; more than 64 KiB per workgroup is opt-in, per device (kernels.h)
        static LdsGrant granted;
        hipError_t e = granted.ensure(reinterpret_cast<const void*>(&step_kernel<G, LM, MODE, ML1, LX>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((step_kernel<G, LM, MODE, ML1, LX>), grid, block, lds, stream, P, K);
    return hipGetLastError();
}
template <int MODE, int G, int LM>
static hipError_t launch_step_mode_gl(const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    const bool ml1 = (K.flags & LAUNCH_SINGLE_LAYER) != 0;
    if constexpr (LM == 4) {
        // (LM == 4 means at most four beam words, so a single-layer map always takes one of the exact-count instantiations: the generic
        // <G, 4, MODE, true, -1> was compiled until round 4 -- 50 kernels -- and no launch could reach it)
        if (ml1) {
#define LLE_STEP_LX(X) case X: return launch_step_glp<G, 4, MODE, true, X>(P, K, n_waves, wpw, lds, stream);
            switch (K.n_sources) { LLE_STEP_LX(0) LLE_STEP_LX(1) LLE_STEP_LX(2) LLE_STEP_LX(3) LLE_STEP_LX(4) default: return hipErrorInvalidValue; }
#undef LLE_STEP_LX
        }
        return launch_step_glp<G, 4, MODE, false>(P, K, n_waves, wpw, lds, stream);
    } else {
        return ml1 ? launch_step_glp<G, LM, MODE, true>(P, K, n_waves, wpw, lds, stream)
                   : launch_step_glp<G, LM, MODE, false>(P, K, n_waves, wpw, lds, stream);
    }
}
template <int MODE, int G>
static hipError_t launch_step_mode_g(int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    if constexpr (MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {  // (the launcher sends maps with more than 8 sources to MODE 0 / 4 / 5)
        if (lm == 4) return launch_step_mode_gl<MODE, G, 4>(P, K, n_waves, wpw, lds, stream);
        if (lm == 8) return launch_step_mode_gl<MODE, G, 8>(P, K, n_waves, wpw, lds, stream);
        return hipErrorInvalidValue;
    } else {
        switch (lm) {
            case 4: return launch_step_mode_gl<MODE, G, 4>(P, K, n_waves, wpw, lds, stream);
            case 8: return launch_step_mode_gl<MODE, G, 8>(P, K, n_waves, wpw, lds, stream);
            case 16: return launch_step_mode_gl<MODE, G, 16>(P, K, n_waves, wpw, lds, stream);
            default: return launch_step_mode_gl<MODE, G, 32>(P, K, n_waves, wpw, lds, stream);
        }
    }
}
template <int MODE>
static hipError_t launch_step_mode(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    switch (G) {
        case 1: return launch_step_mode_g<MODE, 1>(lm, P, K, n_waves, wpw, lds, stream);
        case 2: return launch_step_mode_g<MODE, 2>(lm, P, K, n_waves, wpw, lds, stream);
        case 4: return launch_step_mode_g<MODE, 4>(lm, P, K, n_waves, wpw, lds, stream);
        case 8: return launch_step_mode_g<MODE, 8>(lm, P, K, n_waves, wpw, lds, stream);
        default: return launch_step_mode_g<MODE, 16>(lm, P, K, n_waves, wpw, lds, stream);
    }
}

}  // namespace lle
