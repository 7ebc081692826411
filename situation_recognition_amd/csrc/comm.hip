// Data-parallel gradient exchange through the C ABI: RCCL all-reduce (sum) of a device buffer over xGMI.
//
// Replaces what the reference's only parallelism, nn.DataParallel (reference sr.py:467-470), does inside backward
// (sr.py:79): reduce_add_coalesced of the replicas' gradients onto GPU 0.  Here every rank is one process on one GPU, the
// trainable gradients live in one flat fp32 buffer per rank, and the exchange is one in-place ncclAllReduce(sum) per bucket
// on a stream of the caller's choice (it overlaps the rest of the backward).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that never calls sr_comm_* never loads it, and a host that
// already carries RCCL (PyTorch-ROCm does) shares that instance.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "../../include/srhip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.CommCount;
  });
  return r;
}

static_assert(sizeof(ncclUniqueId) == SR_COMM_ID_BYTES, "include/srhip.h: SR_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

}  // namespace

extern "C" int sr_comm_unique_id(void* id_out) {
  if (!id_out) return SR_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return SR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return SR_ERR_LAUNCH;
  std::memcpy(id_out, &id, sizeof(id));
  return SR_OK;
}

extern "C" int sr_comm_init(const void* id_in, int rank, int world, void** comm_out) {
  if (!id_in || !comm_out || world < 1 || rank < 0 || rank >= world) return SR_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return SR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof(id));
  ncclComm_t comm = nullptr;
  if (r.CommInitRank(&comm, world, id, rank) != ncclSuccess || !comm) return SR_ERR_LAUNCH;
  *comm_out = comm;
  return SR_OK;
}

extern "C" int sr_comm_world(void* comm) {
  if (!comm) return SR_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return SR_ERR_UNSUPPORTED;
  int n = 0;
  if (r.CommCount(static_cast<ncclComm_t>(comm), &n) != ncclSuccess) return SR_ERR_LAUNCH;
  return n;
}

extern "C" int sr_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream) {
  if (!comm || !buf || count <= 0) return SR_ERR_ARG;
  if (dtype != SR_F32 && dtype != SR_BF16) return SR_ERR_DTYPE;
  const Rccl& r = rccl();
  if (!r.ok) return SR_ERR_UNSUPPORTED;
  const ncclResult_t rc = r.AllReduce(buf, buf, (size_t)count, dtype == SR_F32 ? ncclFloat32 : ncclBfloat16, ncclSum,
                                      static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
  return rc == ncclSuccess ? SR_OK : SR_ERR_LAUNCH;
}

extern "C" int sr_comm_destroy(void* comm) {
  if (!comm) return SR_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return SR_ERR_UNSUPPORTED;
  return r.CommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? SR_OK : SR_ERR_LAUNCH;
}
