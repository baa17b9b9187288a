// step_kernel MODE 8: MODE 5 (per-env sources, single step) with the colour-independent head lines of the rows stored ahead of the state machine.  One translation unit per mode (see step_kernel.hpp).
#include "step_kernel.hpp"

namespace lle {
hipError_t launch_step_mode8(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    return launch_step_mode<8>(G, lm, P, K, n_waves, wpw, lds, stream);
}
}  // namespace lle
