// Gradient exchange of the data-parallel train step behind the C ABI (include/igcn.h, igcn_comm_*): one RCCL
// communicator per process (one process per GPU), ONE ncclAllReduce(sum) over the flat fp32 gradient bucket per
// step, enqueued on the caller's stream — so it sits on the launch stream between the kernels that pack the
// gradients and the Adam kernel, and can be captured into the step's hipGraph.
//
// RCCL is resolved at run time (dlsym on the symbols already in the process — PyTorch-ROCm loads its own librccl —
// then dlopen("librccl.so.1")): libigcn.so has no link-time dependency on it, builds without it, and uses the same
// RCCL build as torch.distributed in the same process.  Types restated from rccl.h (ncclUniqueId = 128 opaque
// bytes, ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0).
#include <dlfcn.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_comm_destroy)(Comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_get_error_string error_string = nullptr;
  bool tried = false, ok = false;
};

Rccl g_rccl;

void* find_symbol(void* lib, const char* name) {
  void* p = dlsym(RTLD_DEFAULT, name);
  if (!p && lib) p = dlsym(lib, name);
  return p;
}

bool load_rccl() {
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  void* lib = nullptr;
  if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
  }
  g_rccl.get_unique_id = (fn_get_unique_id)find_symbol(lib, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)find_symbol(lib, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)find_symbol(lib, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)find_symbol(lib, "ncclAllReduce");
  g_rccl.error_string = (fn_get_error_string)find_symbol(lib, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
  return g_rccl.ok;
}

const char* rccl_err(int rc) { return g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error"; }

#define IGCN_RCCL_OR_FAIL(what)                                                         \
  do {                                                                                  \
    if (!load_rccl()) {                                                                 \
      igcn_set_error("%s: RCCL (librccl.so) could not be found in this process", what); \
      return IGCN_ERR_UNSUPPORTED;                                                      \
    }                                                                                   \
  } while (0)

}  // namespace

extern "C" int igcn_comm_unique_id_bytes(void) { return (int)sizeof(UniqueId); }

extern "C" int igcn_comm_get_unique_id(void* out_bytes) {
  IGCN_REQUIRE(out_bytes != nullptr, "comm_get_unique_id: NULL buffer");
  IGCN_RCCL_OR_FAIL("comm_get_unique_id");
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) {
    igcn_set_error("comm_get_unique_id: %s", rccl_err(rc));
    return IGCN_ERR_LAUNCH;
  }
  memcpy(out_bytes, &id, sizeof(id));
  return IGCN_OK;
}

extern "C" int igcn_comm_init(int world_size, int rank, const void* unique_id, void** comm_out) {
  IGCN_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size && unique_id != nullptr && comm_out != nullptr,
               "comm_init: bad arguments (world_size=%d rank=%d)", world_size, rank);
  IGCN_RCCL_OR_FAIL("comm_init");
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  Comm c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, world_size, id, rank);
  if (rc != 0 || c == nullptr) {
    igcn_set_error("comm_init: ncclCommInitRank failed: %s", rccl_err(rc));
    return IGCN_ERR_LAUNCH;
  }
  *comm_out = c;
  return IGCN_OK;
}

extern "C" int igcn_comm_allreduce(void* comm, float* buf, int64_t n, void* stream) {
  IGCN_REQUIRE(comm != nullptr && (buf != nullptr || n == 0) && n >= 0, "comm_allreduce: bad arguments");
  IGCN_RCCL_OR_FAIL("comm_allreduce");
  if (n == 0) return IGCN_OK;
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, (Comm)comm,
                                   (hipStream_t)stream);
  if (rc != 0) {
    igcn_set_error("comm_allreduce: ncclAllReduce failed: %s", rccl_err(rc));
    return IGCN_ERR_LAUNCH;
  }
  return IGCN_OK;
}

extern "C" int igcn_comm_destroy(void* comm) {
  if (comm == nullptr) return IGCN_OK;
  IGCN_RCCL_OR_FAIL("comm_destroy");
  const int rc = g_rccl.comm_destroy((Comm)comm);
  if (rc != 0) {
    igcn_set_error("comm_destroy: %s", rccl_err(rc));
    return IGCN_ERR_LAUNCH;
  }
  return IGCN_OK;
}
