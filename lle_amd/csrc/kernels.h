// kernels.h -- host-visible launch interface of kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "tables.h"

#include <atomic>

namespace lle {

// More than 64 KiB of dynamic LDS per workgroup is an opt-in per (kernel, DEVICE): a code object is loaded once per
// device, and hipFuncSetAttribute acts on the current one.  One of these per kernel instantiation remembers what each
// device has been granted (a process may own a handle on every GPU of the node; lle_hip.h: distinct handles are independent,
// also across threads -- hence the atomics; a lost race costs one redundant hipFuncSetAttribute).
struct LdsGrant {
    static constexpr int MAX_DEVICES = 32;
    std::atomic<uint32_t> bytes[MAX_DEVICES];
    LdsGrant() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
    hipError_t ensure(const void* fn, uint32_t lds) {
        if (lds <= 64u * 1024u) return hipSuccess;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool cached = dev >= 0 && dev < MAX_DEVICES;
        if (cached && lds <= bytes[dev].load(std::memory_order_relaxed)) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && cached) bytes[dev].store(lds, std::memory_order_relaxed);
        return e;
    }
};

// ---- tuning overrides.  The LLE_* environment variables of NOTEBOOK.md section 7 are read ONCE per process into this snapshot
// (first use) -- never on the launch path; lle_tuning_refresh() reads them again (the parity tests and the tuning tools change
// them mid-process).  -1 / 0 = not set.
struct Tuning {
    int step_wpw = 0;          // LLE_STEP_WPW = 1 / 2 / 4
    int step_split = -1;       // LLE_STEP_SPLIT = 0 / 1
    int write_through = -1;    // LLE_WRITE_THROUGH = 0 / 1
    int row_heads = -1;        // LLE_ROW_HEADS = 0 / 1
    int step_epw = 0;          // LLE_STEP_EPW = power of two
    int pingpong = -1;         // LLE_PINGPONG = 0 / 1
    int partial_project = -1;  // LLE_PARTIAL_PROJECT = 0 / 1
    int partial_kernel = 0;    // LLE_PARTIAL_KERNEL: 0 unset, 1 "lanes", 2 "window", 3 "project", 4 anything else ("auto")
    int partial_e = 0, partial_batches = 0, partial_wt = -1, partial_epw = 0;  // LLE_PARTIAL_E / _BATCHES / _WT / _EPW
    int row_rotate = -1;       // LLE_ROW_ROTATE = 0 / 1
    int post_first = -1;       // LLE_POST_FIRST = 0 / 1: every wavefront's small outputs after / before its observation stream (default: by position in the grid)
    int head_group = 0;        // LLE_HEAD_GROUP = 1 / 2 / 4: wavefronts whose row heads ONE wavefront stores (step_kernel.hpp HEAD)
    int packed_tables = -1;    // LLE_PACKED_TABLES = 0 / 1: split-row launches of multi-map batches expand the packed table image (tables.h off_packed)
};
const Tuning& tuning();
void tuning_refresh();

// What a batch has chosen for its own step launches (lle_batch_autotune times the alternatives on the batch's own arena; -1 / 0 =
// the launcher's default rule).  An environment override (Tuning) wins over a batch's choice, a batch's choice over the rule.
struct StepTune {
    int8_t heads = -1;          // row heads ahead of the state machine (MODE 6 / 7 / 8)
    int8_t write_through = -1;  // `sc1` stores of the rows
    int8_t split = -1;          // split rows (big observations)
    int8_t walk = -1;           // alternating walk of outputs larger than the Infinity Cache
    int8_t rotate = -1;         // every wavefront starts its stream at another one of its rows
    int8_t head_group = 0;      // 1 / 2 / 4: wavefronts of a workgroup whose row heads one of them stores (0: the default rule)
    uint8_t epw = 0;            // environments per wavefront
};

// ---- debug registry: which kernel instantiations this process has LAUNCHED, and which ones the dispatch can reach (lle_debug_launched /
// lle_debug_reachable, include/lle_hip.h).  A miscompile of ONE instantiation (round 4: a live-range split ahead of an exec restore in
// step_kernel<4,4,4,false,-1>) is only found by a test that launches that instantiation: the coverage test demands every reachable one.
// Cost on the launch path: one relaxed atomic load per launch (a static flag per instantiation).
enum { DBG_STEP = 0, DBG_WORLD = 1, DBG_OBSERVER = 2 };
enum { OBSK_VIEW = 0, OBSK_PARTIAL_WINDOW, OBSK_PARTIAL_PROJECT, OBSK_PARTIAL_LANES, OBSK_STATE, OBSK_AVAIL, OBSK_ENV_OUTPUTS, OBSK_ROW_FILL_WT,
       OBSK_ROW_FILL_PLAIN, OBSK_CAST_ROWS, OBSK_STATS_SUM, OBSK_COUNT };
constexpr uint32_t debug_key(int kind, int g, int lm, int mode, bool ml1, int lx) {
    return ((uint32_t)kind << 24) | ((uint32_t)g << 16) | ((uint32_t)lm << 8) | ((uint32_t)mode << 4) | ((ml1 ? 1u : 0u) << 3) | (uint32_t)(lx + 1);
}
void debug_note(uint32_t key, bool reachable_walk);
// newline-separated kernel names, sorted; returns the bytes needed (terminator included)
size_t debug_list(bool reachable, char* buf, size_t cap);
void debug_reset_launched();

enum { KMODE_STEP = 0, KMODE_RESET = 1, KMODE_SET_STATE = 2, KMODE_OBSERVE = 3, KMODE_SOURCES = 4, KMODE_ENV_SOURCES = 5 };
constexpr uint32_t MIN_ENVS_PER_WAVE = 4;  // step_kernel<16, .>: 4 environments per wavefront
constexpr uint32_t MIN_STAT_SLOTS = 8192;  // LLE_BUF_STATS never has fewer per-wavefront slots than this (capi.cpp make_layout)

int kernel_variant(int A, int L);
int agent_stride(int A, int L);  // agents per env record in the per-agent buffers (= the variant's agent bound)
const char* kernel_variant_name(int variant);
uint32_t kernel_lds_bytes(const MapHeader& h, uint32_t waves_per_wg, bool pes = false);
uint32_t kernel_waves_per_wg(const MapHeader& h, bool pes = false);
bool step_splits_rows(const MapHeader& h, bool pes, const StepTune& tune = StepTune());  // step_kernel: rows split over the workgroup's wavefronts
bool step_can_split_rows(const MapHeader& h, bool pes);                  // ... whether the instantiation carries it at all
uint32_t split_lds_bytes(const MapHeader& h, uint32_t wpw, uint32_t epw);
// step_kernel instantiations, one translation unit per MODE (step_mode<N>.hip; step_kernel.hpp)
hipError_t launch_step_mode0(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode1(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode2(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode3(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode4(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode5(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode6(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode7(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode8(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
hipError_t launch_step_mode9(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream);
// step_kernel MODE 9 (the partial k x k observation written by the step launch): environments per batch of the writer for this map and
// window, 0 = not served (more than 8 beam words, per-environment sources, or no room in LDS)
uint32_t step_partial_batch(const MapHeader& h, int k, bool pes);
bool write_through_pays(uint64_t bytes, uint32_t row_pitch, int chosen = -1);  // store policy of an observation stream (obs_stream.hpp: stream_store)
hipError_t launch_world_kernel(int mode, const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K, hipStream_t stream);
// World.step with one lane per agent (the default step path)
hipError_t launch_step_kernel(const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K, hipStream_t stream, const StepTune& tune = StepTune());
bool step_has_row_heads(const MapHeader& h, bool pes);  // whether a single-step launch of this map can take the MODE 6 / 7 / 8 kernels
int step_group(int A);
uint32_t step_envs_per_wave(int64_t n, int A, const StepTune& tune = StepTune(), int64_t split_block = 0);
int step_lm(int L);

// observers.hip
hipError_t launch_view_observe(const ViewHeader& v, const BatchPtrs& P, const uint8_t* views_dev, uint32_t n_views, int8_t* out,
                               int64_t row_pitch, int64_t view_pitch, int64_t n_envs, bool pes, uint32_t n_elems, MapSel M,
                               uint32_t views_stride, bool reverse, hipStream_t stream, uint32_t et = OBS_I8 /* element type of the rows: the pitches are in BYTES */);
// do n_views views fit the LDS of one workgroup together?
bool view_kernel_fits(const ViewHeader& v, uint32_t n_views, bool pes, uint32_t n_elems);
uint32_t partial_pitch(int A, int k);
// n_entities: walls (sources included) + exits + gems + exposed laser tiles + sources of the map (the largest of a multi-map batch)
// win_sets: the window tables of k (tables.h; win_table_bytes(HW) apart per map, device memory) or NULL (window sizes other than 3, 5, 7)
hipError_t launch_partial_observe(const MapHeader& h, const BatchPtrs& P, int8_t* out, int k, int64_t n_envs, bool per_env_sources,
                                  MapSel M, uint32_t n_entities, bool reverse, hipStream_t stream, const uint8_t* win_sets = nullptr,
                                  uint32_t force_E = 0 /* environments per batch of the lane kernel; 0: the rule */, uint32_t* rule_E = nullptr,
                                  uint32_t et = OBS_I8 /* element type of the rows (the pitch stays in elements) */);
hipError_t launch_state_observe(const MapHeader& h, const BatchPtrs& P, float* out, int normalize, int64_t n_envs, hipStream_t stream);
hipError_t launch_env_outputs(const MapHeader& h, const BatchPtrs& P, const EnvOutputs& O, int64_t n_envs, MapSel M, hipStream_t stream);
// ceiling probe: n_rows rows of row_bytes (a multiple of 16) filled with the step kernel's store pattern (observers.hip)
hipError_t launch_row_fill_probe(int8_t* out, int64_t n_rows, uint32_t row_bytes, uint32_t rows_per_wave, uint32_t value, bool reverse, bool rotate,
                                 hipStream_t stream);
hipError_t launch_cast_rows(const int8_t* rows, void* out_f16, int64_t bytes, hipStream_t stream);  // bench.py's consumer stand-in (observers.hip)
bool rotate_rows_pays(const StepTune& tune, uint64_t row_bytes_per_launch, uint32_t row_pitch);  // LAUNCH_ROTATE_ROWS (obs_stream.hpp row_rotation)
// out8[k] = sum over the n_blocks per-wavefront slots of stats[slot][k] (one workgroup; lle_batch_stats, lle_batch_stats_allreduce)
hipError_t launch_stats_sum(const int64_t* stats, int64_t n_blocks, int64_t* out8, hipStream_t stream);
hipError_t launch_avail(const MapHeader& h, const BatchPtrs& P, uint8_t* out, int walkable_lasers, int64_t n_envs, bool per_env_sources,
                        MapSel M, hipStream_t stream);

}  // namespace lle
