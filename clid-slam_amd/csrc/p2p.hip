// Gradient exchange over peer-mapped buffers (new -- the reference is single-GPU, slam.py:11).
//
// The per-iteration all-reduce of the sharded mapping loop moves 0.1 - 2.5 MB (the compact buffer [848 | 9 x touched
// rows]).  At that size a ring all-reduce over 8 GPUs is bound by its 2 (P - 1) = 14 latency steps,
// not by bytes.  xGMI is all-to-all, so every rank can read every peer directly: ONE launch per rank,
//
//   barrier A | stage 1: rank r reads slice r of every rank's buffer (remote loads), adds them in rank order, writes the
//               sum over slice r of its OWN buffer
//   barrier B | stage 2: rank r copies slice s (s != r) from rank s's buffer into its own
//
// two xGMI hops instead of fourteen.  Every element is summed by exactly one rank in the order 0 .. P-1 and then copied,
// so all replicas receive bit-identical sums.  The barriers are per block (block b of every rank runs the same launch
// geometry and owns the same offsets inside every slice): P flag words per block in uncached memory, written by the
// peers with system-scope stores; a monotonically increasing epoch instead of a reset.  Two exchange buffers are used
// alternately (buffer `which` = iteration parity): a rank may pack iteration t + 1 while a slower peer still copies
// iteration t's result out of the other buffer, and nobody can be two exchanges ahead (barrier A).
//
// Memory is exported / imported with HIP IPC handles; the host distributes the handle blobs with its own mechanism
// (torch.distributed all_gather), exactly as it distributes the RCCL id.  A flag wait gives up after the object's timeout
// (default 600 s like an RCCL collective's watchdog; clid_p2p_set_timeout) and raises the object's error word instead of
// hanging the GPU for ever.  The error word is LOCAL: a caller must agree on it across the ranks (clid_mapping_run_dist does,
// through the RCCL communicator, and returns CLID_E_P2P_TIMEOUT on EVERY rank) before it decides anything, and the sums
// of a timed-out exchange are garbage -- the host restores its pre-call state and repeats the call over RCCL
// (Mapper.mapping).  clid_p2p_selftest runs exchanges of exactly representable values before the object is trusted (the
// caller takes the MIN over the ranks and keeps RCCL otherwise).
#include <string.h>
#include <unistd.h>

#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kP2pMaxWorld = 8, kP2pBlocks = 64, kP2pThreads = 256;
constexpr size_t kP2pFlagBytes = (size_t)kP2pBlocks * kP2pMaxWorld * sizeof(unsigned) * 2;  // + room for the error word
constexpr double kP2pTicksPerSecond = 1.0e8;  // wall_clock64 runs at 100 MHz
constexpr double kP2pDefaultTimeoutS = 600.0;  // a benign skew between ranks (one of them saves a mesh, evaluates, ...) is
                                               // minutes at most: the default matches the RCCL watchdog's order of magnitude

struct P2pBlob {  // what one rank exports
  hipIpcMemHandle_t data, flags;
  int64_t capacity;
  int32_t rank, pid;
};

struct P2pPtrs {
  f32x4* data[kP2pMaxWorld];
  unsigned* flags[kP2pMaxWorld];
};

__device__ __forceinline__ void p2p_barrier(const P2pPtrs& p, int rank, int world, unsigned epoch, int* err,
                                            long long timeout_ticks) {
  // every thread publishes its own stores first (write-back to memory, system scope), then the block meets, then one
  // thread per peer raises this block's flag at the peer and waits for the peer's flag here
  __scoped_atomic_thread_fence(__ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const int peer = threadIdx.x;
    __scoped_atomic_store_n(&p.flags[peer][blockIdx.x * kP2pMaxWorld + rank], epoch, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = &p.flags[rank][blockIdx.x * kP2pMaxWorld + peer];
    const long long t0 = wall_clock64();
    while ((int)(__scoped_atomic_load_n(mine, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > timeout_ticks || __scoped_atomic_load_n(err, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM)) {  // (one timeout fails the object: no further waits)
        atomicExch(err, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __scoped_atomic_thread_fence(__ATOMIC_ACQUIRE, __MEMORY_SCOPE_SYSTEM);  // nothing read from a peer before this point is reused
}

// OR_BITS: the 16-byte words are combined with a bitwise OR instead of four float additions (0 / 1 flag bytes: OR == MAX)
template <bool OR_BITS>
__global__ void __launch_bounds__(kP2pThreads) k_p2p_allreduce(P2pPtrs p, int rank, int world, long long n4, unsigned epoch,
                                                               int* err, long long timeout_ticks) {
  const long long per = (n4 + world - 1) / world;  // float4 per slice
  const long long stride = (long long)gridDim.x * kP2pThreads;
  const long long t = (long long)blockIdx.x * kP2pThreads + threadIdx.x;
  p2p_barrier(p, rank, world, epoch + 1, err, timeout_ticks);
  {
    const long long off = per * rank;
    const long long len = (off + per <= n4 ? per : (n4 > off ? n4 - off : 0));
    for (long long i = t; i < len; i += stride) {
      f32x4 acc = __builtin_nontemporal_load(&p.data[0][off + i]);
      for (int r = 1; r < world; ++r) {
        const f32x4 v = __builtin_nontemporal_load(&p.data[r][off + i]);
        if (OR_BITS) {
          acc.x = __uint_as_float(__float_as_uint(acc.x) | __float_as_uint(v.x));
          acc.y = __uint_as_float(__float_as_uint(acc.y) | __float_as_uint(v.y));
          acc.z = __uint_as_float(__float_as_uint(acc.z) | __float_as_uint(v.z));
          acc.w = __uint_as_float(__float_as_uint(acc.w) | __float_as_uint(v.w));
        } else {
          acc.x = __fadd_rn(acc.x, v.x);
          acc.y = __fadd_rn(acc.y, v.y);
          acc.z = __fadd_rn(acc.z, v.z);
          acc.w = __fadd_rn(acc.w, v.w);
        }
      }
      p.data[rank][off + i] = acc;
    }
  }
  p2p_barrier(p, rank, world, epoch + 2, err, timeout_ticks);
  for (int s = 0; s < world; ++s) {
    if (s == rank) continue;
    const long long off = per * s;
    const long long len = (off + per <= n4 ? per : (n4 > off ? n4 - off : 0));
    for (long long i = t; i < len; i += stride) p.data[rank][off + i] = __builtin_nontemporal_load(&p.data[s][off + i]);
  }
}

// self-test patterns: small integers (sums exact in fp32 in any order)
__device__ __forceinline__ float p2p_pattern(long long i, int rank, int salt) {
  return (float)((i * 7 + rank * 13 + salt * 5) % 251);
}
__global__ void k_p2p_fill(float* buf, long long n, int rank, int salt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    buf[i] = p2p_pattern(i, rank, salt);
}
__global__ void k_p2p_verify(const float* buf, long long n, int world, int salt, int* bad) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float want = 0.f;
    for (int r = 0; r < world; ++r) want += p2p_pattern(i, r, salt);
    if (buf[i] != want) atomicAdd(bad, 1);
  }
}

}  // namespace

struct clid_p2p {
  int rank, world, device;
  int64_t capacity;       // bytes of ONE exchange buffer
  char* data;             // [2][capacity]
  unsigned* flags;        // [kP2pBlocks][kP2pMaxWorld] + error word + self-test counter
  void* peer_data[kP2pMaxWorld];
  void* peer_flags[kP2pMaxWorld];
  bool connected;
  unsigned epoch;
  int cur;                // the buffer the next exchange runs on
  long long timeout_ticks;
  int* agree;             // one int32 in ordinary device memory: the error word as the communicator's MAX sees it
};

static int* p2p_err_word(clid_p2p* p) { return reinterpret_cast<int*>(p->flags + kP2pBlocks * kP2pMaxWorld); }

extern "C" int64_t clid_p2p_blob_bytes(void) { return (int64_t)sizeof(P2pBlob); }

extern "C" int clid_p2p_create(int32_t rank, int32_t world, int64_t capacity_bytes, clid_p2p** out, uint8_t* blob_out_host) {
  if (!out || !blob_out_host || world < 1 || world > kP2pMaxWorld || rank < 0 || rank >= world || capacity_bytes < 16) {
    clid_set_error("clid_p2p_create: bad argument (rank %d of %d, at most %d ranks)", rank, world, kP2pMaxWorld);
    return CLID_E_ARG;
  }
  clid_p2p* p = new clid_p2p();
  p->rank = rank;
  p->world = world;
  p->capacity = (capacity_bytes + 255) & ~(int64_t)255;
  p->connected = false;
  p->epoch = 0;
  p->cur = 0;
  p->timeout_ticks = (long long)(kP2pDefaultTimeoutS * kP2pTicksPerSecond);
  p->agree = nullptr;
  for (int r = 0; r < kP2pMaxWorld; ++r) p->peer_data[r] = p->peer_flags[r] = nullptr;
  void* d = nullptr;
  void* f = nullptr;
  if (hipGetDevice(&p->device) != hipSuccess || hipMalloc(&d, (size_t)p->capacity * 2) != hipSuccess) {
    clid_set_error("clid_p2p_create: cannot allocate 2 x %lld bytes", (long long)p->capacity);
    delete p;
    return CLID_E_HIP;
  }
  // flag words live in uncached memory (every access goes to memory: they are written by other devices)
  if (hipExtMallocWithFlags(&f, kP2pFlagBytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc(&f, kP2pFlagBytes) != hipSuccess) {
      clid_set_error("clid_p2p_create: cannot allocate the flag words");
      (void)hipFree(d);
      delete p;
      return CLID_E_HIP;
    }
  }
  void* ag = nullptr;
  if (hipMalloc(&ag, 256) != hipSuccess || hipMemset(ag, 0, 256) != hipSuccess) {
    clid_set_error("clid_p2p_create: cannot allocate the agreement word");
    (void)hipFree(d);
    (void)hipFree(f);
    delete p;
    return CLID_E_HIP;
  }
  p->agree = static_cast<int*>(ag);
  if (hipMemset(f, 0, kP2pFlagBytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    clid_set_error("clid_p2p_create: flag reset failed");
    (void)hipFree(d);
    (void)hipFree(f);
    delete p;
    return CLID_E_HIP;
  }
  p->data = static_cast<char*>(d);
  p->flags = static_cast<unsigned*>(f);
  P2pBlob blob;
  memset(&blob, 0, sizeof(blob));
  blob.capacity = p->capacity;
  blob.rank = rank;
  blob.pid = (int32_t)getpid();
  if (world > 1 && (hipIpcGetMemHandle(&blob.data, d) != hipSuccess || hipIpcGetMemHandle(&blob.flags, f) != hipSuccess)) {
    clid_set_error("clid_p2p_create: hipIpcGetMemHandle failed: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d);
    (void)hipFree(f);
    delete p;
    return CLID_E_HIP;
  }
  memcpy(blob_out_host, &blob, sizeof(blob));
  p->peer_data[rank] = d;
  p->peer_flags[rank] = f;
  if (world == 1) p->connected = true;
  *out = p;
  return CLID_OK;
}

extern "C" int clid_p2p_connect(clid_p2p* p, const uint8_t* blobs_host) {
  if (!p || !blobs_host) {
    clid_set_error("clid_p2p_connect: null argument");
    return CLID_E_ARG;
  }
  if (p->connected) return CLID_OK;
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank) continue;
    P2pBlob b;
    memcpy(&b, blobs_host + (size_t)r * sizeof(P2pBlob), sizeof(b));
    if (b.rank != r || b.capacity != p->capacity) {
      clid_set_error("clid_p2p_connect: blob %d names rank %d / capacity %lld (expected %lld)", r, b.rank, (long long)b.capacity,
                     (long long)p->capacity);
      return CLID_E_ARG;
    }
    if (hipIpcOpenMemHandle(&p->peer_data[r], b.data, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
        hipIpcOpenMemHandle(&p->peer_flags[r], b.flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      clid_set_error("clid_p2p_connect: hipIpcOpenMemHandle for rank %d failed: %s", r, hipGetErrorString(hipGetLastError()));
      return CLID_E_HIP;
    }
  }
  p->connected = true;
  return CLID_OK;
}

extern "C" int32_t clid_p2p_world(const clid_p2p* p) { return p ? p->world : CLID_E_ARG; }
extern "C" int64_t clid_p2p_capacity(const clid_p2p* p) { return p ? p->capacity : CLID_E_ARG; }

// the buffer the NEXT clid_p2p_allreduce works on (the two buffers alternate: fill it, exchange, read the sums from it)
extern "C" void* clid_p2p_buffer(clid_p2p* p) {
  if (!p) return nullptr;
  return p->data + (size_t)p->cur * p->capacity;
}

static int p2p_exchange(clid_p2p* p, int64_t count_floats, bool or_bits, void* stream) {
  if (!p || count_floats < 0 || count_floats * 4 > p->capacity) {
    clid_set_error("clid_p2p_allreduce: bad argument (%lld floats, capacity %lld bytes)", (long long)count_floats,
                   p ? (long long)p->capacity : 0LL);
    return CLID_E_ARG;
  }
  if (!p->connected) {
    clid_set_error("clid_p2p_allreduce: not connected");
    return CLID_E_ARG;
  }
  const int which = p->cur;
  p->cur ^= 1;
  if (count_floats == 0 || p->world == 1) return CLID_OK;
  P2pPtrs ptrs;
  for (int r = 0; r < kP2pMaxWorld; ++r) {
    ptrs.data[r] = r < p->world ? reinterpret_cast<f32x4*>(static_cast<char*>(p->peer_data[r]) + (size_t)which * p->capacity) : nullptr;
    ptrs.flags[r] = r < p->world ? static_cast<unsigned*>(p->peer_flags[r]) : nullptr;
  }
  const long long n4 = (count_floats + 3) / 4;
  if (or_bits)
    hipLaunchKernelGGL(k_p2p_allreduce<true>, dim3(kP2pBlocks), dim3(kP2pThreads), 0, (hipStream_t)stream, ptrs, p->rank, p->world,
                       n4, p->epoch, p2p_err_word(p), p->timeout_ticks);
  else
    hipLaunchKernelGGL(k_p2p_allreduce<false>, dim3(kP2pBlocks), dim3(kP2pThreads), 0, (hipStream_t)stream, ptrs, p->rank, p->world,
                       n4, p->epoch, p2p_err_word(p), p->timeout_ticks);
  p->epoch += 2;
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_p2p_allreduce(clid_p2p* p, int64_t count_floats, void* stream) {
  return p2p_exchange(p, count_floats, false, stream);
}

// bitwise OR over the ranks of `bytes` bytes at `buf` (any device memory: staged through the current exchange buffer), in
// place -- the MAX of 0 / 1 flag bytes (the touched-row flags of a chunk, clid_mapping_run_dist)
extern "C" int clid_p2p_allreduce_or(clid_p2p* p, void* buf, int64_t bytes, void* stream) {
  if (!p || !buf || bytes < 0 || bytes + 16 > p->capacity) {
    clid_set_error("clid_p2p_allreduce_or: bad argument (%lld bytes, capacity %lld)", (long long)bytes, p ? (long long)p->capacity : 0LL);
    return CLID_E_ARG;
  }
  if (bytes == 0 || p->world == 1) return CLID_OK;
  hipStream_t s = (hipStream_t)stream;
  char* x = p->data + (size_t)p->cur * p->capacity;
  const int64_t padded = (bytes + 15) & ~(int64_t)15;
  if (hipMemcpyAsync(x, buf, (size_t)bytes, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      (padded > bytes && hipMemsetAsync(x + bytes, 0, (size_t)(padded - bytes), s) != hipSuccess)) {
    clid_set_error("clid_p2p_allreduce_or: staging failed");
    return CLID_E_HIP;
  }
  if (int e = p2p_exchange(p, padded / 4, true, stream)) return e;
  if (hipMemcpyAsync(buf, x, (size_t)bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    clid_set_error("clid_p2p_allreduce_or: copy back failed");
    return CLID_E_HIP;
  }
  return CLID_OK;
}

// flag waits of later exchanges give up after `seconds` of wall clock (same value on every rank, please)
extern "C" int clid_p2p_set_timeout(clid_p2p* p, double seconds) {
  if (!p || !(seconds > 0.0) || seconds > 86400.0) {
    clid_set_error("clid_p2p_set_timeout: bad argument");
    return CLID_E_ARG;
  }
  p->timeout_ticks = (long long)(seconds * kP2pTicksPerSecond);
  return CLID_OK;
}

// 0 = no flag wait has timed out ON THIS RANK since the object was created (synchronises `stream`); CLID_E_P2P_TIMEOUT
// otherwise.  Local knowledge: agree on it across the ranks before acting (clid_p2p_agree).
extern "C" int clid_p2p_status(clid_p2p* p, void* stream) {
  if (!p) return CLID_E_ARG;
  int32_t e = 0;
  if (int rc = clid_read_back(p2p_err_word(p), 4, &e, stream)) return rc;
  if (e) {
    clid_set_error("clid_p2p: a peer did not arrive at an exchange within %.0f s (rank %d of %d)",
                   (double)p->timeout_ticks / kP2pTicksPerSecond, p->rank, p->world);
    return CLID_E_P2P_TIMEOUT;
  }
  return CLID_OK;
}

// Collective over `comm` (the RCCL communicator of the same ranks): the MAX of the ranks' error words, so EVERY rank
// returns CLID_E_P2P_TIMEOUT when ANY rank saw a flag wait give up, and CLID_OK on all of them otherwise.  Synchronises
// `stream`.
extern "C" int clid_p2p_agree(clid_p2p* p, clid_comm* comm, void* stream) {
  if (!p || !comm) {
    clid_set_error("clid_p2p_agree: null argument");
    return CLID_E_ARG;
  }
  if (hipMemcpyAsync(p->agree, p2p_err_word(p), 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    clid_set_error("clid_p2p_agree: copy failed");
    return CLID_E_HIP;
  }
  if (int e = clid_comm_allreduce(comm, p->agree, 1, 1, 1, stream)) return e;
  int32_t e = 0;
  if (int rc = clid_read_back(p->agree, 4, &e, stream)) return rc;
  if (e) {
    clid_set_error("clid_p2p: a rank of the group gave up waiting at an exchange (timeout %.0f s; seen from rank %d of %d): "
                   "the sums of this call are invalid on every rank",
                   (double)p->timeout_ticks / kP2pTicksPerSecond, p->rank, p->world);
    return CLID_E_P2P_TIMEOUT;
  }
  return CLID_OK;
}

// test aid: raise this rank's error word as a timed-out flag wait would (the fallback path of the hosts is tested with it)
extern "C" int clid_debug_p2p_fail(clid_p2p* p, void* stream) {
  if (!p) return CLID_E_ARG;
  const int32_t one = 1;
  if (hipMemcpyAsync(p2p_err_word(p), &one, 4, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    clid_set_error("clid_debug_p2p_fail: copy failed");
    return CLID_E_HIP;
  }
  return CLID_OK;
}

// Collective.  Exchanges of exactly representable patterns over both buffers and three sizes, verified on the device;
// returns 0 when every element of every exchange is right and no wait timed out on THIS rank (the caller agrees on the
// result across the ranks before using the object).
extern "C" int clid_p2p_selftest(clid_p2p* p, void* stream) {
  if (!p || !p->connected) {
    clid_set_error("clid_p2p_selftest: not connected");
    return CLID_E_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  int* bad = p2p_err_word(p) + 1;
  if (hipMemsetAsync(bad, 0, 4, s) != hipSuccess) return CLID_E_HIP;
  const long long cap = p->capacity / 4;
  const long long sizes[3] = {1, cap < 100003 ? cap : 100003, cap};
  for (int salt = 0; salt < 12; ++salt) {  // every size on both buffers, twice
    float* buf = static_cast<float*>(clid_p2p_buffer(p));
    const long long n = sizes[(salt >> 1) % 3];
    hipLaunchKernelGGL(k_p2p_fill, dim3(256), dim3(256), 0, s, buf, n, p->rank, salt);
    if (int e = clid_p2p_allreduce(p, n, stream)) return e;
    hipLaunchKernelGGL(k_p2p_verify, dim3(256), dim3(256), 0, s, buf, n, p->world, salt, bad);
  }
  CLID_CHECK_LAUNCH();
  int32_t got[2] = {0, 0};
  if (int rc = clid_read_back(p2p_err_word(p), 8, got, stream)) return rc;
  if (got[0] || got[1]) {
    clid_set_error("clid_p2p_selftest: %d wrong elements, timeout flag %d (rank %d of %d)", got[1], got[0], p->rank, p->world);
    return CLID_E_HIP;
  }
  return CLID_OK;
}

// test aid: device-to-device copy on `stream` (tests fill / read the exchange buffers, which are not torch tensors)
extern "C" int clid_debug_copy(void* dst, const void* src, int64_t bytes, void* stream) {
  if (!dst || !src || bytes < 0) return CLID_E_ARG;
  if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    clid_set_error("clid_debug_copy: %s", hipGetErrorString(hipGetLastError()));
    return CLID_E_HIP;
  }
  return CLID_OK;
}

extern "C" int clid_p2p_destroy(clid_p2p* p) {
  if (!p) return CLID_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank) continue;
    if (p->peer_data[r]) (void)hipIpcCloseMemHandle(p->peer_data[r]);
    if (p->peer_flags[r]) (void)hipIpcCloseMemHandle(p->peer_flags[r]);
  }
  (void)hipFree(p->data);
  (void)hipFree(p->flags);
  (void)hipFree(p->agree);
  delete p;
  return CLID_OK;
}
