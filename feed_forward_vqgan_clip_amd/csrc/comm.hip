// comm.hip — the gradient exchange behind the C ABI (`ffvc_allreduce_bucket`, persistent state behind an opaque
// handle with explicit create / destroy).  Reference: hvd.DistributedOptimizer's averaged-gradient all-reduce, main.py:626-629.
//
// One RCCL communicator per process (one process per GPU), created from a unique id that rank 0 hands to the other ranks
// through whatever side channel the host already has (torch.distributed's store / a broadcast of 128 bytes).  The bucket
// all-reduce is a plain in-place ncclAllReduce(sum) on the stream the caller passes: the host side (distributed.py) gives it
// a dedicated exchange stream fenced by events, so neither the dgrad chain nor the weight-gradient stream ever waits on it.
//
// RCCL is NOT a link-time dependency of libffvc_hip.so: the process already holds one (PyTorch's librccl.so), and a second
// copy in the same address space would own a second set of IPC handles and proxy threads.  The entry points below resolve
// the five nccl* symbols at run time from the RCCL image that is already loaded (dlopen(..., RTLD_NOLOAD)), or from an
// explicit path (ffvc_rccl_load).
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};
RcclApi g_api;
std::mutex g_mu;

bool bind(void* lib) {
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
  a.GetVersion = (decltype(a.GetVersion))dlsym(lib, "ncclGetVersion");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString) return false;
  g_api = a;
  return true;
}

bool ensure_api() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_api.lib) return true;
  // the RCCL that is already in the process (PyTorch's), by soname; never load a second one implicitly
  for (const char* name : {"librccl.so", "librccl.so.1"}) {
    void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h && bind(h)) return true;
  }
  return false;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

int fail(const char* who, ncclResult_t r) {
  ffvc_set_error("%s: %s", who, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
  return -2000 - (int)r;
}

}  // namespace

extern "C" int ffvc_rccl_load(const char* path) {
  FFVC_CHECK_ARG(path && *path, "ffvc_rccl_load: empty path");
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_api.lib) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h || !bind(h)) {
    ffvc_set_error("ffvc_rccl_load: %s does not provide the nccl* entry points (%s)", path, h ? "symbols missing" : dlerror());
    return FFVC_E_BADARG;
  }
  return 0;
}

extern "C" int ffvc_rccl_available(void) {
  if (!ensure_api()) return 0;
  int v = 0;
  if (g_api.GetVersion && g_api.GetVersion(&v) == ncclSuccess) return v > 0 ? v : 1;
  return 1;
}

extern "C" int ffvc_rccl_unique_id(void* out128) {
  FFVC_CHECK_ARG(out128, "ffvc_rccl_unique_id: null buffer");
  if (!ensure_api()) {
    ffvc_set_error("ffvc_rccl_unique_id: no RCCL image in the process (import torch first, or ffvc_rccl_load(path))");
    return FFVC_E_BADARG;
  }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = g_api.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  memcpy(out128, &id, sizeof(id));
  return 0;
}

extern "C" int ffvc_rccl_comm_create(const void* id128, int rank, int world, void** handle) {
  FFVC_CHECK_ARG(id128 && handle && world >= 1 && rank >= 0 && rank < world, "ffvc_rccl_comm_create: bad arguments (rank %d of %d)", rank, world);
  if (!ensure_api()) {
    ffvc_set_error("ffvc_rccl_comm_create: no RCCL image in the process");
    return FFVC_E_BADARG;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);     // collective: every rank calls it with the same id
  if (r != ncclSuccess) {
    delete c;
    return fail("ncclCommInitRank", r);
  }
  *handle = c;
  return 0;
}

// In-place sum over the ranks of buf[0 .. count) (fp32 gradients, or their 16-bit wire copy), enqueued on `stream`.
extern "C" int ffvc_allreduce_bucket(void* handle, void* buf, int64_t count, int dtype, void* stream) {
  FFVC_CHECK_ARG(handle && buf && count > 0, "ffvc_allreduce_bucket: bad arguments");
  FFVC_CHECK_ARG(dtype == FFVC_F32 || dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_allreduce_bucket: dtype %d", dtype);
  Comm* c = (Comm*)handle;
  const ncclDataType_t dt = dtype == FFVC_F32 ? ncclFloat32 : dtype == FFVC_F16 ? ncclFloat16 : ncclBfloat16;
  const ncclResult_t r = g_api.AllReduce(buf, buf, (size_t)count, dt, ncclSum, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllReduce", r);
  return 0;
}

extern "C" int ffvc_rccl_comm_destroy(void* handle) {
  if (!handle) return 0;
  Comm* c = (Comm*)handle;
  ncclResult_t r = ncclSuccess;
  if (c->comm && g_api.CommDestroy) r = g_api.CommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) return fail("ncclCommDestroy", r);
  return 0;
}
