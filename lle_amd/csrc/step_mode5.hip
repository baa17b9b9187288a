// step_kernel MODE 5: per-env sources (on one or several maps), single step (no rollout loop / rings / stamps).  One translation unit per mode (see step_kernel.hpp).
#include "step_kernel.hpp"

namespace lle {
hipError_t launch_step_mode5(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    return launch_step_mode<5>(G, lm, P, K, n_waves, wpw, lds, stream);
}
}  // namespace lle
