// comm.hip — the multi-GPU side of libsvgp_mi355x: data-parallel shards of the expectation term
// sum_i E_q[log p(y_i | f_i)] (reference src/SparseVariationalApproximationModule.jl:355-359 is a plain sum over
// points) combined by ONE RCCL all-reduce over xGMI, issued by the library on the context's stream on the
// device-resident result vector: no host hop between the strip kernels and the collective.
//
// Two ways to get a communicator (include/svgp_mi355x.h):
//   * one process per GPU: svgp_comm_unique_id (rank 0) + svgp_ctx_attach_comm (every rank) -> ncclCommInitRank;
//   * one process, several GPUs (a Julia host): svgp_group_create -> ncclCommInitAll over the member contexts.
// RCCL is loaded with dlopen on first use: a single-GPU host needs no librccl at all, and a process that already
// holds one (torch ships its own copy, built against its own HIP runtime) keeps using that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "ctx.hpp"

namespace svgp {
namespace {

struct Rccl {
  void* handle = nullptr;
  std::string err;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char* off = getenv("SVGP_DISABLE_RCCL"); off && off[0] == '1') {   // test knob: behave as if librccl were absent
      r.err = "RCCL disabled by SVGP_DISABLE_RCCL";
      return;
    }
    const char* env = getenv("SVGP_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // a copy that is already part of the process first (RTLD_NOLOAD), then the loader's search path
    for (int pass = 0; pass < 2 && !r.handle; ++pass)
      for (const char* n : names) {
        if (!n) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (r.handle) break;
      }
    if (!r.handle) {
      const char* e = dlerror();
      r.err = std::string("librccl could not be loaded: ") + (e ? e : "not found");
      return;
    }
#define SVGP_SYM(f)                                                                  \
  r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.handle, "nccl" #f));                 \
  if (!r.f && r.err.empty()) r.err = "librccl lacks the symbol nccl" #f
    SVGP_SYM(GetUniqueId); SVGP_SYM(CommInitRank); SVGP_SYM(CommInitAll); SVGP_SYM(CommDestroy); SVGP_SYM(CommAbort);
    SVGP_SYM(AllReduce); SVGP_SYM(GroupStart); SVGP_SYM(GroupEnd); SVGP_SYM(GetErrorString);
#undef SVGP_SYM
  });
  return r;
}

int nccl_fail(svgp_ctx* ctx, const char* what, ncclResult_t rc) {
  const Rccl& r = rccl();
  return fail(ctx, SVGP_RCCL_ERROR, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

}  // namespace

int comm_allreduce(svgp_ctx* ctx, void* buf, size_t count, int dtype) {
  if (!ctx->comm) return SVGP_OK;
  const Rccl& r = rccl();
  const ncclResult_t rc = r.AllReduce(buf, buf, count, dtype == SVGP_F64 ? ncclDouble : ncclFloat, ncclSum,
                                      static_cast<ncclComm_t>(ctx->comm), ctx->stream);
  return rc == ncclSuccess ? SVGP_OK : nccl_fail(ctx, "ncclAllReduce", rc);
}

int comm_group_start(svgp_ctx* ctx) {
  const Rccl& r = rccl();
  if (!r.GroupStart) return fail(ctx, SVGP_RCCL_ERROR, r.err);
  const ncclResult_t rc = r.GroupStart();
  return rc == ncclSuccess ? SVGP_OK : nccl_fail(ctx, "ncclGroupStart", rc);
}

int comm_group_end(svgp_ctx* ctx) {
  const Rccl& r = rccl();
  if (!r.GroupEnd) return fail(ctx, SVGP_RCCL_ERROR, r.err);
  const ncclResult_t rc = r.GroupEnd();
  return rc == ncclSuccess ? SVGP_OK : nccl_fail(ctx, "ncclGroupEnd", rc);
}

void comm_abort(svgp_ctx* ctx) {
  if (!ctx || !ctx->comm) return;
  const Rccl& r = rccl();
  if (r.CommAbort) (void)r.CommAbort(static_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->world = 1;
  ctx->rank = 0;
}

}  // namespace svgp

using namespace svgp;

extern "C" {

int32_t svgp_comm_unique_id(void* id_out) {
  if (!id_out) return SVGP_INVALID_ARG;
  static_assert(sizeof(ncclUniqueId) == SVGP_COMM_ID_BYTES, "SVGP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  const Rccl& r = rccl();
  if (!r.err.empty()) return SVGP_RCCL_ERROR;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return SVGP_RCCL_ERROR;
  memcpy(id_out, &id, sizeof id);
  return SVGP_OK;
}

int32_t svgp_ctx_attach_comm(svgp_ctx* ctx, const void* id, int32_t world_size, int32_t rank) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(ctx, SVGP_INVALID_ARG, "bad communicator arguments");
  if (ctx->comm) return fail(ctx, SVGP_INVALID_ARG, "the context already has a communicator");
  const Rccl& r = rccl();
  if (!r.err.empty()) return fail(ctx, SVGP_RCCL_ERROR, r.err);
  HIPC(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = r.CommInitRank(&comm, world_size, uid, rank);
  if (rc != ncclSuccess) return nccl_fail(ctx, "ncclCommInitRank", rc);
  ctx->comm = comm;
  ctx->world = world_size;
  ctx->rank = rank;
  ctx->comm_owned_by_group = false;
  return SVGP_OK;
}

int32_t svgp_ctx_detach_comm(svgp_ctx* ctx) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!ctx->comm) return SVGP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  const Rccl& r = rccl();
  const ncclResult_t rc = r.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->world = 1;
  ctx->rank = 0;
  return rc == ncclSuccess ? SVGP_OK : nccl_fail(ctx, "ncclCommDestroy", rc);
}

int32_t svgp_ctx_comm_info(const svgp_ctx* ctx, int32_t* world_size, int32_t* rank) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (world_size) *world_size = ctx->world;
  if (rank) *rank = ctx->rank;
  return SVGP_OK;
}

}  // extern "C"

// ---- one process, several GPUs -------------------------------------------------------------------------------
extern "C" {

int32_t svgp_group_create(int32_t n_devices, const int32_t* device_ids, svgp_group** out) {
  if (!out) return SVGP_INVALID_ARG;
  *out = nullptr;
  if (n_devices < 1 || !device_ids) return SVGP_INVALID_ARG;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j)
      if (device_ids[i] == device_ids[j]) return SVGP_INVALID_ARG;   // RCCL refuses a device twice in one communicator
  const Rccl& r = rccl();
  if (!r.err.empty()) return SVGP_RCCL_ERROR;
  svgp_group* g = new (std::nothrow) svgp_group();
  if (!g) return SVGP_OOM;
  int rc = SVGP_OK;
  for (int i = 0; i < n_devices && rc == SVGP_OK; ++i) {
    svgp_ctx* c = nullptr;
    rc = svgp_ctx_create(device_ids[i], nullptr, &c);
    if (rc == SVGP_OK) g->ctxs.push_back(c);
  }
  std::vector<ncclComm_t> comms(size_t(n_devices), nullptr);
  if (rc == SVGP_OK) {
    std::vector<int> devs(device_ids, device_ids + n_devices);
    if (r.CommInitAll(comms.data(), n_devices, devs.data()) != ncclSuccess) rc = SVGP_RCCL_ERROR;
  }
  if (rc != SVGP_OK) {
    for (svgp_ctx* c : g->ctxs) svgp_ctx_destroy(c);
    delete g;
    return rc;
  }
  for (int i = 0; i < n_devices; ++i) {
    g->ctxs[size_t(i)]->comm = comms[size_t(i)];
    g->ctxs[size_t(i)]->world = n_devices;
    g->ctxs[size_t(i)]->rank = i;
    g->ctxs[size_t(i)]->comm_owned_by_group = true;
  }
  *out = g;
  return SVGP_OK;
}

int32_t svgp_group_size(const svgp_group* g) { return g ? int32_t(g->ctxs.size()) : 0; }

svgp_ctx* svgp_group_ctx(svgp_group* g, int32_t i) {
  return (g && i >= 0 && size_t(i) < g->ctxs.size()) ? g->ctxs[size_t(i)] : nullptr;
}

int32_t svgp_group_destroy(svgp_group* g) {
  if (!g) return SVGP_OK;
  for (svgp_ctx* c : g->ctxs) {
    svgp_ctx_detach_comm(c);
    svgp_ctx_destroy(c);
  }
  delete g;
  return SVGP_OK;
}

}  // extern "C"
