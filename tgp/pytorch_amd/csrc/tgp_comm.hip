// tgp_comm.hip -- the collective of the data-parallel step behind the C ABI (`tgp_allreduce`): ONE
// in-place sum all-reduce of the flat [gradients | ELBO, ELL, KL] buffer per step, on the caller's stream -- so a host that
// is not PyTorch can run the multi-GPU step, and a PyTorch host can keep the collective on the compute stream (inside the
// captured step) instead of on torch.distributed's side stream.  One process per GPU, one communicator per process.
//
// RCCL is bound at run time (dlopen) on purpose: libtgp_hip.so must load -- and every single-GPU entry point work -- on a
// host whose RCCL is somewhere else or absent; nothing here is linked.  The handful of declarations below restate the
// public NCCL/RCCL C API (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy, ncclGetErrorString).
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

namespace {
struct UniqueId { char internal[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                        // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat64 = 8;   // ncclFloat64
constexpr int kSum = 0;       // ncclSum

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn get_error_string = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

// the table as it stands (a copy taken under the lock: comm_load may be filling it on another host thread)
Rccl rccl() {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_rccl;
}

int comm_error(const Rccl& r, const char* what, int rc) {
  set_error_text("%s: %s (%d)", what, r.get_error_string ? r.get_error_string(rc) : "RCCL error", rc);   // (thread-local text)
  return TGP_E_COMM;
}
}  // namespace

// path == NULL: "librccl.so" by the loader's rules (a process that has imported torch already holds torch/lib/librccl.so)
int comm_load(const char* path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl.handle != nullptr) return 0;
  void* h = dlopen(path != nullptr ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) {
    set_error_text("dlopen(%s): %s", path != nullptr ? path : "librccl.so", dlerror());
    return TGP_E_COMM;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
  r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
  r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  r.get_error_string = reinterpret_cast<GetErrorStringFn>(dlsym(h, "ncclGetErrorString"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    set_error_text("%s does not export the NCCL API", path != nullptr ? path : "librccl.so");
    dlclose(h);
    return TGP_E_COMM;
  }
  g_rccl = r;
  return 0;
}

int comm_unique_id(void* id128) {
  const Rccl r = rccl();
  if (r.handle == nullptr) return TGP_E_COMM;
  UniqueId id;
  if (int rc = r.get_unique_id(&id)) return comm_error(r, "ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int comm_init(const void* id128, int nranks, int rank, void** comm) {
  const Rccl r = rccl();
  if (r.handle == nullptr) return TGP_E_COMM;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  if (int rc = r.comm_init_rank(&c, nranks, id, rank)) return comm_error(r, "ncclCommInitRank", rc);
  *comm = c;
  return 0;
}

int comm_allreduce(void* comm, double* buf, int64_t n, hipStream_t st) {
  const Rccl r = rccl();
  if (r.handle == nullptr) return TGP_E_COMM;
  if (int rc = r.all_reduce(buf, buf, (size_t)n, kFloat64, kSum, comm, st)) return comm_error(r, "ncclAllReduce", rc);
  return 0;
}

int comm_destroy(void* comm) {
  const Rccl r = rccl();
  if (r.handle == nullptr) return TGP_E_COMM;
  if (int rc = r.comm_destroy(comm)) return comm_error(r, "ncclCommDestroy", rc);
  return 0;
}

}  // namespace tgp
