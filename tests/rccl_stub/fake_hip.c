/* LD_PRELOAD shim for tests/test_comm_stub.py (CPU): the handful of HIP runtime calls the communicator code of lle_amd/csrc/comm.cpp makes,
 * answered for a pretend node of FAKE_HIP_DEVICES GPUs with host memory.  FAKE_HIP_FAIL_MALLOC = k: the k-th hipMalloc fails. */
#include <stdlib.h>
static int g_dev = 0, g_mallocs = 0, g_live = 0;
int fake_hip_live_allocations(void) { return g_live; }
int fake_hip_current_device(void) { return g_dev; }
int hipGetDeviceCount(int* n) { const char* e = getenv("FAKE_HIP_DEVICES"); *n = e ? atoi(e) : 2; return 0; }
int hipGetDevice(int* d) { *d = g_dev; return 0; }
int hipSetDevice(int d) { g_dev = d; return 0; }
int hipMalloc(void** p, size_t n) {
    const char* e = getenv("FAKE_HIP_FAIL_MALLOC");
    if (e && atoi(e) == ++g_mallocs) return 2;  /* hipErrorOutOfMemory */
    *p = calloc(1, n);
    g_live++;
    return 0;
}
int hipFree(void* p) { free(p); g_live--; return 0; }
