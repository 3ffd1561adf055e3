// Multi-GPU side of the C ABI (include/fsdp.h, fsdp_comm_*): one process per GPU, RCCL over xGMI.
//
// Frames shard embarrassingly: there is NO data-path collective.  RCCL carries only what north_star names —
// the start-up broadcast of constant tables (skidpad track table, 92 576 B; parameter block; the consistency check of the
// previous-path table) — plus the bench's timing barrier and max-reduction.  Every collective runs on the context's own
// HIP stream (no extra stream: each stream of the process takes one of the runtime's hardware queues, fsdp.h
// fsdp_set_overlap).  librccl is opened with dlopen on first use, so the library itself loads on hosts without RCCL.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>

namespace fsdp_comm {

struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

inline Api& api() {
  static Api a;
  return a;
}

// dlopen librccl once; returns false (api().error set) if it cannot be loaded
inline bool load() {
  Api& a = api();
  if (a.handle) return true;
  const char* names[] = {getenv("FSDP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (a.handle) break;
  }
  if (!a.handle) {
    a.error = std::string("cannot dlopen librccl (set FSDP_RCCL_LIB): ") + (dlerror() ? dlerror() : "");
    return false;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(a.handle, name);
    if (!p) {
      ok = false;
      a.error = std::string("librccl lacks ") + name;
    }
    return p;
  };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
  a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
  a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  if (!ok) {
    dlclose(a.handle);
    a.handle = nullptr;
  }
  return ok;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* d_buf = nullptr;  // device staging of the host-buffer collectives
  size_t cap = 0;
};

}  // namespace fsdp_comm
