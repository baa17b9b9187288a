// Round 4: WHY does the write rate of a buffer larger than the Infinity Cache depend on the allocation (profiles/r03_hbm_fronts.md §3),
// and can a fast buffer be had on purpose?  The same three writers on buffers obtained in different ways:
//   malloc        hipMalloc, one after the other (round 3's experiment)
//   contiguous    hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous VRAM
//   vmm-1         hipMemAddressReserve + ONE hipMemCreate handle per buffer + hipMemMap
//   vmm-gran      ... one handle per allocation granule (hipMemGetAllocationGranularity, recommended)
//   vmm-64M       ... handles of 64 MiB
//   suballoc      one hipMalloc of all the buffers, cut into pieces
//   pool          hipMallocAsync from the device's default pool
//   ballast       hipMalloc after (and with) an 8 GiB ballast held
// Writers: the step kernel's stream (level 6 x 262 144: 16 rows of 1 920 B per wavefront, XCD-contiguous blocks, sc1), the same in
// dispatch order (one front instead of eight), hipMemsetAsync.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n) {
    const uint32_t x = b & 7u, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
template <bool XCD>
__global__ void __launch_bounds__(256) fill_rows16(uint4* __restrict__ out, uint4 v) {  // grid = rows / 64
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = XCD ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * 16 + k) * 120;
        for (uint32_t c = lane; c < 120; c += 64) st(p + c, w);
    }
}

static hipStream_t s;
static hipEvent_t e0, e1;
static const size_t ROWS = 262144, BYTES = ROWS * 1920;  // 480 MiB
static double timeit(const std::function<void()>& launch, int reps = 20) {
    for (int i = 0; i < 3; i++) launch();
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}
struct Buf { uint4* p; std::function<void()> release; };

static void measure(const char* method, std::vector<Buf>& bufs) {
    uint4 v = {1, 2, 3, 4};
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < bufs.size(); i++) {
            uint4* b = bufs[i].p;
            const double a = timeit([&] { hipLaunchKernelGGL(fill_rows16<true>, dim3(ROWS / 64), dim3(256), 0, s, b, v); });
            const double c = timeit([&] { hipLaunchKernelGGL(fill_rows16<false>, dim3(ROWS / 64), dim3(256), 0, s, b, v); });
            const double m = timeit([&] { (void)hipMemsetAsync(b, 1, BYTES, s); });
            printf("%-10s pass %d buf %2zu va %p (mod 2M %7zu, mod 1G %10zu): rows16/xcd %6.1f us %5.0f GB/s | rows16/dispatch %6.1f us %5.0f GB/s | memset %6.1f us %5.0f GB/s\n",
                   method, pass, i, (void*)b, (size_t)b % (2u << 20), (size_t)b % (1u << 30), a, BYTES / a / 1e3, c, BYTES / c / 1e3, m, BYTES / m / 1e3);
            fflush(stdout);
        }
    for (auto& b : bufs) b.release();
    bufs.clear();
    (void)hipDeviceSynchronize();
}

static bool vmm_alloc(size_t bytes, size_t chunk, Buf& out, size_t gran) {
    const size_t total = (bytes + gran - 1) / gran * gran;
    if (chunk == 0) chunk = total;
    chunk = (chunk + gran - 1) / gran * gran;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) return false;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    for (size_t off = 0; off < total; off += chunk) {
        const size_t sz = off + chunk <= total ? chunk : total - off;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) return false;
        if (hipMemMap((char*)va + off, sz, 0, h, 0) != hipSuccess) return false;
        handles.push_back(h);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) return false;
    out.p = (uint4*)va;
    out.release = [va, total, handles] {
        (void)hipMemUnmap(va, total);
        for (auto h : handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, total);
    };
    return true;
}

int main(int argc, char** argv) {
    const int n_buf = argc > 1 ? atoi(argv[1]) : 6;
    const std::string only = argc > 2 ? argv[2] : "";
    (void)hipStreamCreate(&s);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    printf("device memory: %.1f GiB free of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    size_t gran_min = 0, gran_rec = 0;
    {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        (void)hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum);
        (void)hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
        printf("VMM granularity: minimum %zu, recommended %zu\n", gran_min, gran_rec);
    }
    {   // clocks up
        uint4* w = nullptr; (void)hipMalloc(&w, BYTES);
        uint4 v = {1, 2, 3, 4};
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL(fill_rows16<true>, dim3(ROWS / 64), dim3(256), 0, s, w, v);
        (void)hipStreamSynchronize(s);
        (void)hipFree(w);
    }
    std::vector<Buf> bufs;
    auto want = [&](const char* m) { return only.empty() || only == m; };
    if (want("malloc")) {
        for (int i = 0; i < n_buf; i++) {
            uint4* p = nullptr;
            if (hipMalloc(&p, BYTES + (size_t)(i % 3) * (1 << 20)) != hipSuccess) break;
            bufs.push_back({p, [p] { (void)hipFree(p); }});
        }
        measure("malloc", bufs);
    }
    if (want("contiguous")) {
        for (int i = 0; i < n_buf; i++) {
            void* p = nullptr;
            hipError_t e = hipExtMallocWithFlags(&p, BYTES + (size_t)(i % 3) * (1 << 20), hipDeviceMallocContiguous);
            if (e != hipSuccess) { printf("contiguous: buffer %d refused: %s\n", i, hipGetErrorString(e)); (void)hipGetLastError(); break; }
            bufs.push_back({(uint4*)p, [p] { (void)hipFree(p); }});
        }
        measure("contiguous", bufs);
    }
    const size_t gran = gran_rec ? gran_rec : (gran_min ? gran_min : (2u << 20));
    struct { const char* name; size_t chunk; } vmm[] = {{"vmm-1", 0}, {"vmm-gran", gran}, {"vmm-64M", 64u << 20}};
    for (auto& m : vmm) {
        if (!want(m.name)) continue;
        for (int i = 0; i < n_buf; i++) {
            Buf b;
            if (!vmm_alloc(BYTES, m.chunk, b, gran)) { printf("%s: buffer %d failed: %s\n", m.name, i, hipGetErrorString(hipGetLastError())); break; }
            bufs.push_back(b);
        }
        measure(m.name, bufs);
    }
    if (want("suballoc")) {
        char* big = nullptr;
        if (hipMalloc(&big, (size_t)n_buf * BYTES) == hipSuccess) {
            for (int i = 0; i < n_buf; i++) bufs.push_back({(uint4*)(big + (size_t)i * BYTES), [] {}});
            measure("suballoc", bufs);
            (void)hipFree(big);
        }
    }
    if (want("pool")) {
        for (int i = 0; i < n_buf; i++) {
            void* p = nullptr;
            if (hipMallocAsync(&p, BYTES + (size_t)(i % 3) * (1 << 20), s) != hipSuccess) { (void)hipGetLastError(); break; }
            bufs.push_back({(uint4*)p, [p] { (void)hipFreeAsync(p, s); }});
        }
        (void)hipStreamSynchronize(s);
        measure("pool", bufs);
    }
    if (want("ballast")) {
        void* ballast = nullptr;
        if (hipMalloc(&ballast, (size_t)8 << 30) == hipSuccess) {
            for (int i = 0; i < n_buf; i++) {
                uint4* p = nullptr;
                if (hipMalloc(&p, BYTES + (size_t)(i % 3) * (1 << 20)) != hipSuccess) break;
                bufs.push_back({p, [p] { (void)hipFree(p); }});
            }
            measure("ballast", bufs);
            (void)hipFree(ballast);
        }
    }
    if (want("malloc2")) {  // again, at the end: does the order of a process's allocations matter, or the time?
        for (int i = 0; i < n_buf; i++) {
            uint4* p = nullptr;
            if (hipMalloc(&p, BYTES + (size_t)(i % 3) * (1 << 20)) != hipSuccess) break;
            bufs.push_back({p, [p] { (void)hipFree(p); }});
        }
        measure("malloc2", bufs);
    }
    return 0;
}
