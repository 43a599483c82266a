// RCCL inside the C ABI (new -- the reference is single-GPU, slam.py:11).
// One communicator per process (one process per GPU); the gradient all-reduce of the sharded mapping loop is
// issued on the SAME stream as the kernels around it, so a whole Mapper.mapping call stays one host call with no
// Python and no stream hand-off between decode -> all-reduce -> Adam (clid_mapping_run_dist, train.hip).
//
// RCCL is resolved at run time (dlopen) so that (a) the library loads on boxes without RCCL and (b) the process
// shares the RCCL instance PyTorch-ROCm already mapped (same soname) instead of linking a second copy.
#include <dlfcn.h>

#include "common.hpp"

namespace {

typedef struct { char internal[128]; } nccl_uid;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
typedef void* nccl_comm;
// rccl.h: ncclSum = 0, ncclMax = 2; ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclUint8 = 1, kNcclInt32 = 2, kNcclFloat32 = 7;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*CommCount)(nccl_comm, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.GetErrorString;
  return r;
}

int fail(const char* what, int rc) {
  clid_set_error("%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
  return CLID_E_HIP;
}

}  // namespace

struct clid_comm {
  nccl_comm comm;
  int rank, world;
};

extern "C" int clid_comm_unique_id(uint8_t* id_out_host) {
  if (!id_out_host) {
    clid_set_error("clid_comm_unique_id: null argument");
    return CLID_E_ARG;
  }
  if (!rccl().ok) {
    clid_set_error("clid_comm_unique_id: librccl.so could not be loaded");
    return CLID_E_HIP;
  }
  nccl_uid id;
  if (int rc = rccl().GetUniqueId(&id)) return fail("ncclGetUniqueId", rc);
  for (int i = 0; i < 128; ++i) id_out_host[i] = (uint8_t)id.internal[i];
  return CLID_OK;
}

extern "C" int clid_comm_init(const uint8_t* id_host, int32_t rank, int32_t world, clid_comm** comm_out) {
  if (!id_host || !comm_out || world < 1 || rank < 0 || rank >= world) {
    clid_set_error("clid_comm_init: bad argument (rank %d of %d)", rank, world);
    return CLID_E_ARG;
  }
  if (!rccl().ok) {
    clid_set_error("clid_comm_init: librccl.so could not be loaded");
    return CLID_E_HIP;
  }
  nccl_uid id;
  for (int i = 0; i < 128; ++i) id.internal[i] = (char)id_host[i];
  nccl_comm c = nullptr;
  if (int rc = rccl().CommInitRank(&c, world, id, rank)) return fail("ncclCommInitRank", rc);
  *comm_out = new clid_comm{c, rank, world};
  return CLID_OK;
}

extern "C" int clid_comm_size(const clid_comm* comm) {
  if (!comm) return CLID_E_ARG;
  int n = 0;
  if (int rc = rccl().CommCount(comm->comm, &n)) return fail("ncclCommCount", rc);
  return n;
}

extern "C" int clid_comm_allreduce(clid_comm* comm, void* buf, int64_t count, int32_t dtype, int32_t op_max,
                                   void* stream) {
  if (!comm || !buf || count < 0 || dtype < 0 || dtype > 2) {
    clid_set_error("clid_comm_allreduce: bad argument");
    return CLID_E_ARG;
  }
  if (count == 0) return CLID_OK;
  const int nd = dtype == 0 ? kNcclFloat32 : (dtype == 1 ? kNcclInt32 : kNcclUint8);
  if (int rc = rccl().AllReduce(buf, buf, (size_t)count, nd, op_max ? kNcclMax : kNcclSum, comm->comm, (hipStream_t)stream))
    return fail("ncclAllReduce", rc);
  return CLID_OK;
}

// 1 when librccl resolves in this process (dlopen + every symbol): lets the ranks agree BEFORE anyone enters the collective
// ncclCommInitRank, where a rank that could not load RCCL would leave the others waiting forever
extern "C" int clid_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int clid_comm_destroy(clid_comm* comm) {
  if (!comm) return CLID_OK;
  const int rc = rccl().CommDestroy(comm->comm);
  delete comm;
  return rc ? fail("ncclCommDestroy", rc) : CLID_OK;
}
