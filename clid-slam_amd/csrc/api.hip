// Error plumbing and ABI version of libclid_native.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.hpp"

static thread_local char g_err[512] = "";

extern "C" void clid_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* clid_last_error(void) { return g_err; }
extern "C" int clid_abi_version(void) { return 8; }

// Small device -> host read-back through a pinned landing buffer (a pageable destination makes the runtime stage the
// copy and block for ~100 us): the data-dependent counts of the map-maintenance calls.  Synchronises `stream`.
extern "C" int clid_read_back(const void* device_src, int32_t bytes, void* host_dst, void* stream) {
  static thread_local void* pinned = nullptr;
  if (bytes < 0 || bytes > 256 || (bytes && (!device_src || !host_dst))) {
    clid_set_error("clid_read_back: bad argument (at most 256 bytes)");
    return CLID_E_ARG;
  }
  if (bytes == 0) return CLID_OK;
  if (!pinned && hipHostMalloc(&pinned, 256, hipHostMallocDefault) != hipSuccess) {
    pinned = nullptr;
    clid_set_error("clid_read_back: cannot allocate the pinned buffer");
    return CLID_E_HIP;
  }
  hipStream_t s = (hipStream_t)stream;
  // the polled event belongs to a device: one per (host thread, device), created on the device that is current now (a
  // thread that reads back on another GPU's stream would otherwise record a foreign-device event)
  static thread_local hipEvent_t evs[16] = {};
  int dev = 0;
  bool ok = hipGetDevice(&dev) == hipSuccess && dev >= 0;
  hipEvent_t* evp = (ok && dev < 16) ? &evs[dev] : nullptr;
  ok = ok && hipMemcpyAsync(pinned, device_src, (size_t)bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
  if (ok) {
    // the host is about to size the next launches with these numbers: poll an event instead of hipStreamSynchronize
    // (20 us per frame of eight read-backs on the sequence workload); anything unexpected falls back to the synchronise
    bool polled = false;
    if (evp && (*evp || hipEventCreateWithFlags(evp, hipEventDisableTiming) == hipSuccess) && hipEventRecord(*evp, s) == hipSuccess) {
      hipError_t q;
      while ((q = hipEventQuery(*evp)) == hipErrorNotReady) {
      }
      polled = q == hipSuccess;
    }
    if (!polled) {
      (void)hipGetLastError();
      ok = hipStreamSynchronize(s) == hipSuccess;
    }
  }
  if (!ok) {
    clid_set_error("clid_read_back: copy failed: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  memcpy(host_dst, pinned, (size_t)bytes);
  return CLID_OK;
}
