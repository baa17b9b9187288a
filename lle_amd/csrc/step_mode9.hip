// step_kernel MODE 9: MODE 4 whose launch writes the partial k x k observation instead of the layered rows (partial_stream.hpp).  One translation unit per mode (see step_kernel.hpp).
#include "step_kernel.hpp"

namespace lle {
hipError_t launch_step_mode9(int G, int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    return launch_step_mode<9>(G, lm, P, K, n_waves, wpw, lds, stream);
}
}  // namespace lle
