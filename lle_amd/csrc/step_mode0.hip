// step_kernel MODE 0: one step in place, one map, the map's sources (the default).  One translation unit per mode (see step_kernel.hpp).
#include "step_kernel.hpp"

namespace lle {
hipError_t launch_step_mode0(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    return launch_step_mode<0>(G, lm, P, K, n_waves, wpw, lds, stream);
}
}  // namespace lle
