// The path's one exchange step as a C-ABI call: all-reduce(sum) of the flat fp32 gradient bucket over RCCL / xGMI with a
// communicator this library owns (SURVEY 8(b) `flat_allreduce`).  The reference has no distributed code at all -- its
// optimiser (Fitting/FittingFC/declare_fitter.py:58-61) is single-process -- so there is no reference interface to mirror;
// the bucket layout is what gh_adam_step consumes (grad_scale = 1 / world averages the summed gradients in the same pass).
//
// librccl is loaded on FIRST USE with dlopen: nothing else in libget_hip.so depends on it, a host that never calls
// gh_comm_* never loads it, and a process that already holds a copy (torch.distributed's "nccl" backend) gets that same
// copy (RTLD_NOLOAD is tried first) instead of a second RCCL in the address space.
#include <dlfcn.h>

#include <mutex>

#include "../../include/get_hip.h"
#include "common.h"

namespace {

struct NcclId { char bytes[128]; };                 // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
typedef int (*get_id_fn)(NcclId*);
typedef int (*init_rank_fn)(void**, int, NcclId, int);
typedef int (*destroy_fn)(void*);
typedef int (*count_fn)(void*, int*);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*errstr_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_id_fn get_id = nullptr;
  init_rank_fn init_rank = nullptr;
  destroy_fn destroy = nullptr;
  count_fn count = nullptr, user_rank = nullptr;
  allreduce_fn allreduce = nullptr;
  bcast_fn bcast = nullptr;
  errstr_fn errstr = nullptr;
  char path[256] = {0};
};

Rccl g_rccl;
std::mutex g_rccl_mu;

constexpr int kNcclFloat32 = 7;     // ncclDataType_t: ncclFloat32
constexpr int kNcclSum = 0;         // ncclRedOp_t: ncclSum

// 0 = loaded; the error text names every candidate tried
int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle) return 0;
  const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  const char* used = nullptr;
  for (const char* c : cands) {                      // a copy the process already holds wins
    h = dlopen(c, RTLD_NOW | RTLD_NOLOAD);
    if (h) { used = c; break; }
  }
  if (!h) {
    const char* env = getenv("GET_AMD_RCCL");        // explicit override for hosts with a non-standard install
    if (env && *env) { h = dlopen(env, RTLD_NOW | RTLD_LOCAL); if (h) used = env; }
  }
  for (int i = 0; !h && i < 4; ++i) {
    h = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
    if (h) used = cands[i];
  }
  GH_REQUIRE(h != nullptr, "gh_comm: librccl not found (tried librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so[.1], $GET_AMD_RCCL): %s",
             dlerror());
  Rccl r;
  r.handle = h;
  r.get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (init_rank_fn)dlsym(h, "ncclCommInitRank");
  r.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
  r.count = (count_fn)dlsym(h, "ncclCommCount");
  r.user_rank = (count_fn)dlsym(h, "ncclCommUserRank");
  r.allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
  r.bcast = (bcast_fn)dlsym(h, "ncclBroadcast");
  r.errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
  GH_REQUIRE(r.get_id && r.init_rank && r.destroy && r.count && r.user_rank && r.allreduce && r.bcast && r.errstr,
             "gh_comm: %s lacks one of the ncclGetUniqueId / CommInitRank / CommDestroy / CommCount / CommUserRank / AllReduce / "
             "Broadcast / GetErrorString symbols", used);
  snprintf(r.path, sizeof(r.path), "%s", used);
  g_rccl = r;
  return 0;
}

#define GH_CHECK_NCCL(expr)                                                                                   \
  do {                                                                                                        \
    const int _r = (expr);                                                                                    \
    if (_r != 0) {                                                                                            \
      gh::set_error("%s:%d: %s -> RCCL error %d (%s)", __FILE__, __LINE__, #expr, _r, g_rccl.errstr(_r));     \
      return 1;                                                                                               \
    }                                                                                                         \
  } while (0)

}  // namespace

extern "C" int gh_comm_unique_id(void* id128) {
  GH_REQUIRE(id128 != nullptr, "gh_comm_unique_id: NULL output");
  if (int e = load_rccl()) return e;
  NcclId id;
  GH_CHECK_NCCL(g_rccl.get_id(&id));
  memcpy(id128, id.bytes, sizeof(id.bytes));
  return 0;
}

extern "C" int gh_comm_init(const void* id128, int rank, int world, void** comm) {
  GH_REQUIRE(id128 && comm, "gh_comm_init: NULL id / output");
  GH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "gh_comm_init: rank %d not in [0, %d)", rank, world);
  if (int e = load_rccl()) return e;
  NcclId id;
  memcpy(id.bytes, id128, sizeof(id.bytes));
  void* c = nullptr;
  GH_CHECK_NCCL(g_rccl.init_rank(&c, world, id, rank));      // binds to the calling thread's current HIP device
  *comm = c;
  return 0;
}

extern "C" int gh_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (int e = load_rccl()) return e;
  GH_CHECK_NCCL(g_rccl.destroy(comm));
  return 0;
}

extern "C" int gh_comm_info(void* comm, int* rank, int* world) {
  GH_REQUIRE(comm != nullptr, "gh_comm_info: NULL communicator");
  if (int e = load_rccl()) return e;
  if (world) GH_CHECK_NCCL(g_rccl.count(comm, world));
  if (rank) GH_CHECK_NCCL(g_rccl.user_rank(comm, rank));
  return 0;
}

extern "C" const char* gh_comm_library(void) {
  if (load_rccl()) return nullptr;
  return g_rccl.path;
}

extern "C" int gh_flat_allreduce(void* comm, float* buf, int64_t count, gh_stream_t stream) {
  GH_REQUIRE(comm != nullptr, "gh_flat_allreduce: NULL communicator (gh_comm_init first)");
  GH_REQUIRE(count >= 0 && (count == 0 || buf != nullptr), "gh_flat_allreduce: bad buffer / count");
  if (count == 0) return 0;
  if (int e = load_rccl()) return e;
  GH_CHECK_NCCL(g_rccl.allreduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream));   // in place
  return 0;
}

extern "C" int gh_flat_broadcast(void* comm, float* buf, int64_t count, int root, gh_stream_t stream) {
  GH_REQUIRE(comm != nullptr, "gh_flat_broadcast: NULL communicator (gh_comm_init first)");
  GH_REQUIRE(count >= 0 && (count == 0 || buf != nullptr), "gh_flat_broadcast: bad buffer / count");
  if (count == 0) return 0;
  if (int e = load_rccl()) return e;
  GH_CHECK_NCCL(g_rccl.bcast(buf, buf, (size_t)count, kNcclFloat32, root, comm, (hipStream_t)stream));
  return 0;
}
