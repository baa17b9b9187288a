// How much does the size of the kernel-argument segment cost per launch on MI355X?  Kernels that use every argument
// word (so nothing is dead) and otherwise do nothing; launch-to-launch time over a back-to-back train.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N>
struct Args { uint64_t w[N]; };

template <int N>
__global__ void __launch_bounds__(256) use_args(Args<N> a, uint64_t* out) {
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += a.w[i];
    if (s == 0x123456789ull) out[blockIdx.x] = s;  // never true: keeps the loads alive
}

// same, but the arguments live in device memory behind one pointer (read with scalar loads from a const pointer)
template <int N>
__global__ void __launch_bounds__(256) use_ptr(const Args<N>* __restrict__ a, uint64_t* out) {
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += a->w[i];
    if (s == 0x123456789ull) out[blockIdx.x] = s;
}

// the same with ~19 us of streaming stores behind it: is the argument fetch visible when the GPU is the bottleneck?
template <int N>
__global__ void __launch_bounds__(256) use_args_stream(Args<N> a, uint4* __restrict__ buf, uint32_t rows_per_wave) {
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += a.w[i];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint4 v = make_uint4((uint32_t)s, 2, 3, 4);
    uint4* p = buf + (size_t)wave * rows_per_wave * 64 + lane;
    for (uint32_t r = 0; r < rows_per_wave; r++) p[(size_t)r * 64] = v;
}
template <int N>
__global__ void __launch_bounds__(256) use_ptr_stream(const Args<N>* __restrict__ a, uint4* __restrict__ buf, uint32_t rows_per_wave) {
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += a->w[i];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint4 v = make_uint4((uint32_t)s, 2, 3, 4);
    uint4* p = buf + (size_t)wave * rows_per_wave * 64 + lane;
    for (uint32_t r = 0; r < rows_per_wave; r++) p[(size_t)r * 64] = v;
}
template <int N>
static void run_stream(hipStream_t st, uint4* buf, hipEvent_t e0, hipEvent_t e1) {
    Args<N> a;
    for (int i = 0; i < N; i++) a.w[i] = i + 1;
    Args<N>* dev;
    hipMalloc(&dev, sizeof a);
    hipMemcpy(dev, &a, sizeof a, hipMemcpyHostToDevice);
    const int iters = 300;
    const uint32_t rpw = 29;
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int i = 0; i < iters; i++) {
                if (mode == 0) hipLaunchKernelGGL(use_args_stream<N>, dim3(1024), dim3(256), 0, st, a, buf, rpw);
                else hipLaunchKernelGGL(use_ptr_stream<N>, dim3(1024), dim3(256), 0, st, dev, buf, rpw);
            }
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1)
                printf("stream 122 MB + %4d B of arguments %-28s %6.2f us per launch\n", N * 8, mode == 0 ? "by value (kernarg segment)" : "behind a device pointer", ms / iters * 1e3);
        }
    }
    hipFree(dev);
}

template <int N>
static void run(hipStream_t st, uint64_t* out, hipEvent_t e0, hipEvent_t e1) {
    Args<N> a;
    for (int i = 0; i < N; i++) a.w[i] = i + 1;
    Args<N>* dev;
    hipMalloc(&dev, sizeof a);
    hipMemcpy(dev, &a, sizeof a, hipMemcpyHostToDevice);
    const int iters = 500;
    for (int mode = 0; mode < 2; mode++) {
        for (int i = 0; i < 50; i++) {
            if (mode == 0) hipLaunchKernelGGL(use_args<N>, dim3(1024), dim3(256), 0, st, a, out);
            else hipLaunchKernelGGL(use_ptr<N>, dim3(1024), dim3(256), 0, st, dev, out);
        }
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; i++) {
            if (mode == 0) hipLaunchKernelGGL(use_args<N>, dim3(1024), dim3(256), 0, st, a, out);
            else hipLaunchKernelGGL(use_ptr<N>, dim3(1024), dim3(256), 0, st, dev, out);
        }
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%4d B of arguments %-28s %6.2f us per launch\n", N * 8, mode == 0 ? "by value (kernarg segment)" : "behind a device pointer", ms / iters * 1e3);
    }
    hipFree(dev);
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint64_t* out;
    hipMalloc(&out, 1024 * 8);
    run<2>(st, out, e0, e1);
    run<8>(st, out, e0, e1);
    run<16>(st, out, e0, e1);
    run<32>(st, out, e0, e1);
    run<50>(st, out, e0, e1);
    run<64>(st, out, e0, e1);
    run<100>(st, out, e0, e1);
    uint4* buf;
    hipMalloc(&buf, (size_t)4096 * 29 * 1024 + (1 << 20));
    run_stream<2>(st, buf, e0, e1);
    run_stream<8>(st, buf, e0, e1);
    run_stream<16>(st, buf, e0, e1);
    run_stream<32>(st, buf, e0, e1);
    run_stream<46>(st, buf, e0, e1);
    run_stream<50>(st, buf, e0, e1);
    run_stream<64>(st, buf, e0, e1);
    return 0;
}
