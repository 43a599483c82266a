// Compact probe table: the exact, cache-resident equivalent of `buffer_pt_index[hash]` ->
// travel-distance filter -> `global2local` (model/neural_points.py:984-1009, 595-598).
//
// The reference probes a 5e7-slot int64 table (400 MB) 81 times per query; almost every probe
// returns -1 or an id that is dropped again by the time filter / the local-window mapping.  Which
// ids survive does not depend on the query, so they are resolved ONCE per map state here and stored
// in an open-addressing table keyed by the *same slot number* the reference computes.  A probe
// of slot s returns id j iff the reference's chain would have produced local id j for slot s, so
// hash collisions of the big table (a foreign id returned and rejected by the distance test, a
// point shadowed by a later insert) are reproduced exactly.
#include "common.hpp"

namespace clid {

__global__ void k_table_build(const int64_t* __restrict__ ids, int n, const float* __restrict__ pts,
                              const int64_t* __restrict__ big, int B, float res,
                              const int* __restrict__ ts_create, const float* __restrict__ travel,
                              int cur_ts, int time_filtering, float diff_travel, int4* tab,
                              float4* tab_pos, int log2cap, float4* pos4, unsigned* filter, int log2filter) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long long gi = ids ? ids[j] : (long long)j;
  const float x = pts[gi * 3 + 0], y = pts[gi * 3 + 1], z = pts[gi * 3 + 2];
  pos4[j] = make_float4(x, y, z, 0.f);
  const int slot = base_slot(x, y, z, res, B);
  if (big[slot] != gi) return;  // shadowed by a later insert into the same slot: unreachable
  if (time_filtering) {         // np.py:1003-1009
    const float gap = fabsf(fsub(travel[cur_ts], travel[ts_create[gi]]));
    if (!(gap < diff_travel)) return;
  }
  if (filter) {  // probe prefilter: one bit per key (csrc/train.hip, search-only kernel)
    const unsigned b = filter_bit(slot, log2filter);
    atomicOr(&filter[b >> 5], 1u << (b & 31));
  }
  // 4-key buckets, keys claimed in order (so "key3 taken" == bucket full); overflow walks to the next bucket
  const unsigned mask = (1u << log2cap) - 1u;
  unsigned pos = tab_home(slot, log2cap);
  int* cells = reinterpret_cast<int*>(tab);
  for (;;) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (atomicCAS(&cells[(size_t)pos * 4 + m], -1, slot) == -1) {
        tab_pos[(size_t)pos * 4 + m] = make_float4(x, y, z, __int_as_float(j));
        return;
      }
    }
    pos = (pos + 1) & mask;
  }
}

}  // namespace clid

extern "C" int clid_table_build(const int64_t* ids, int32_t n, const float* neural_points,
                                const int64_t* buffer_pt_index, int64_t buffer_size, float resolution,
                                const int32_t* point_ts_create, const float* travel_dist,
                                int32_t cur_ts, int32_t time_filtering, float diff_travel,
                                int32_t* tab_out, float* tab_pos_out, int32_t log2cap, float* pos4_out,
                                uint32_t* filter_out, int32_t log2filter, void* stream) {
  if (n < 0 || log2cap < 5 || log2cap > 30 || buffer_size <= 0 || buffer_size >= (1LL << 30)) {
    clid_set_error("clid_table_build: bad argument (n=%d log2cap=%d buffer_size=%lld)", n, log2cap,
                   (long long)buffer_size);
    return CLID_E_ARG;
  }
  if ((4LL << log2cap) < 2LL * n || !tab_out || !tab_pos_out || !pos4_out) {
    clid_set_error("clid_table_build: table capacity 4*2^%d < 2*n (n=%d) or null output", log2cap, n);
    return CLID_E_ARG;
  }
  if (filter_out && (log2filter < 10 || log2filter > 30)) {
    clid_set_error("clid_table_build: log2filter=%d out of range", log2filter);
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(tab_out, 0xFF, sizeof(int32_t) * 4 * ((size_t)1 << log2cap), s) != hipSuccess ||
      (filter_out && hipMemsetAsync(filter_out, 0, sizeof(uint32_t) * (((size_t)1 << log2filter) / 32), s) != hipSuccess)) {
    clid_set_error("clid_table_build: memset failed");
    return CLID_E_HIP;
  }
  if (n == 0) return CLID_OK;
  hipLaunchKernelGGL(clid::k_table_build, dim3((n + 255) / 256), dim3(256), 0, s, ids, n, neural_points,
                     buffer_pt_index, (int)buffer_size, resolution, point_ts_create, travel_dist, cur_ts,
                     time_filtering, diff_travel, reinterpret_cast<int4*>(tab_out), reinterpret_cast<float4*>(tab_pos_out), log2cap,
                     reinterpret_cast<float4*>(pos4_out), filter_out, log2filter);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
