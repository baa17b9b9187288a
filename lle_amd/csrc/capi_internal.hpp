// capi_internal.hpp -- what the other translation units of the C ABI (comm.cpp) need from capi.cpp.
#pragma once
#include <cstdint>
#include <string>

struct lle_batch;

namespace lle {
int capi_fail(int status, const std::string& msg);  // sets lle_last_status / lle_last_error of this thread; returns status
int capi_ok();                                      // LLE_OK, and says so in lle_last_status
int capi_batch_device(const lle_batch* b);
// this batch's eight counters (lle_batch_stats) summed over its per-wavefront slots INTO device memory `out8_dev`, enqueued
// on `stream`; reset_counters: the slots are zeroed behind the sum
int capi_batch_stats_to_device(lle_batch* b, int64_t* out8_dev, int reset_counters, void* stream);
int capi_batch_reset_counters(lle_batch* b, void* stream);  // the slots back to zero, enqueued on `stream`
}  // namespace lle
