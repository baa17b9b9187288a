// step_kernel MODE 7: MODE 4 with the rows' static head lines stored ahead of the state machine (kernels.hip: row_heads_pay).  One translation unit per mode (see step_kernel.hpp).
#include "step_kernel.hpp"

namespace lle {
hipError_t launch_step_mode7(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    return launch_step_mode<7>(G, lm, P, K, n_waves, wpw, lds, stream);
}
}  // namespace lle
