// RCCL at the C-ABI level (pmd_comm_init(ctx, rccl_unique_id, rank, world)): a caller that is not
// built on torch.distributed initialises one communicator per context and runs the collectives of the sharded path
// (all-reduce of the partial Gram matrices / background projections, all-gather of per-tile results) on the context's
// HIP stream.  RCCL is resolved at run time (dlopen): the library has no link-time dependency on it, and a process that has
// already loaded an RCCL (PyTorch ships one) shares that copy.  localmd_amd/parallel.py keeps using torch.distributed
// (backend "nccl" = the same RCCL); these entry points are the equivalent for the reference-side binding.
#include "pmd_internal.h"
#include <dlfcn.h>

namespace {

// the part of the NCCL / RCCL API that is used (rccl.h): opaque communicator, 128-byte unique id, result code
struct pmd_nccl_id { char internal[128]; };
typedef void* pmd_nccl_comm;
typedef int (*fn_get_unique_id)(pmd_nccl_id*);
typedef int (*fn_comm_init_rank)(pmd_nccl_comm*, int, pmd_nccl_id, int);
typedef int (*fn_comm_destroy)(pmd_nccl_comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, pmd_nccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, pmd_nccl_comm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_INT8 = 0, NCCL_SUM = 0;   // ncclDataType_t / ncclRedOp_t values of nccl.h

struct rccl_api {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
};

rccl_api* load_rccl() {
  static rccl_api api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (api.handle) {
      api.get_unique_id = (fn_get_unique_id)dlsym(api.handle, "ncclGetUniqueId");
      api.comm_init_rank = (fn_comm_init_rank)dlsym(api.handle, "ncclCommInitRank");
      api.comm_destroy = (fn_comm_destroy)dlsym(api.handle, "ncclCommDestroy");
      api.all_reduce = (fn_all_reduce)dlsym(api.handle, "ncclAllReduce");
      api.all_gather = (fn_all_gather)dlsym(api.handle, "ncclAllGather");
      api.error_string = (fn_error_string)dlsym(api.handle, "ncclGetErrorString");
    }
  }
  const bool ok = api.handle && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce && api.all_gather;
  return ok ? &api : nullptr;
}

int rccl_fail(pmd_ctx* ctx, rccl_api* api, const char* what, int rc) {
  return pmd_fail(ctx, PMD_ERR_BLAS, what, (api && api->error_string) ? api->error_string(rc) : "RCCL error");
}

}  // namespace

int pmd_comm_unique_id_impl(void* out128) {
  rccl_api* api = load_rccl();
  if (!api || !out128) return PMD_ERR_UNSUPPORTED;
  pmd_nccl_id id;
  if (api->get_unique_id(&id) != 0) return PMD_ERR_BLAS;
  memcpy(out128, &id, sizeof(id));
  return PMD_OK;
}

int pmd_comm_init_impl(pmd_ctx* ctx, const void* unique_id128, int rank, int world) {
  rccl_api* api = load_rccl();
  if (!api) return pmd_fail(ctx, PMD_ERR_UNSUPPORTED, "pmd_comm_init", "librccl could not be loaded");
  if (!unique_id128 || world < 1 || rank < 0 || rank >= world) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_init", "bad argument");
  if (ctx->comm) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_init", "the context already has a communicator");
  pmd_nccl_id id;
  memcpy(&id, unique_id128, sizeof(id));
  PMD_HIP(ctx, hipSetDevice(ctx->device));
  pmd_nccl_comm comm = nullptr;
  const int rc = api->comm_init_rank(&comm, world, id, rank);
  if (rc != 0) return rccl_fail(ctx, api, "ncclCommInitRank", rc);
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return PMD_OK;
}

int pmd_comm_destroy_impl(pmd_ctx* ctx) {
  rccl_api* api = load_rccl();
  if (ctx->comm && api) {
    PMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    api->comm_destroy((pmd_nccl_comm)ctx->comm);
  }
  ctx->comm = nullptr;
  ctx->comm_world = 0;
  return PMD_OK;
}

// in-place sum over the ranks, enqueued on the context's stream
int pmd_comm_all_reduce_f32_impl(pmd_ctx* ctx, float* buf, size_t count) {
  rccl_api* api = load_rccl();
  if (!api || !ctx->comm) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_all_reduce_f32", "no communicator (pmd_comm_init)");
  if (count == 0) return PMD_OK;
  const int rc = api->all_reduce(buf, buf, count, NCCL_FLOAT32, NCCL_SUM, (pmd_nccl_comm)ctx->comm, ctx->stream);
  return rc == 0 ? PMD_OK : rccl_fail(ctx, api, "ncclAllReduce", rc);
}

// recv[r * bytes_per_rank ...] = rank r's send block, on every rank
int pmd_comm_all_gather_impl(pmd_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank) {
  rccl_api* api = load_rccl();
  if (!api || !ctx->comm) return pmd_fail(ctx, PMD_ERR_ARG, "pmd_comm_all_gather", "no communicator (pmd_comm_init)");
  if (bytes_per_rank == 0) return PMD_OK;
  const int rc = api->all_gather(send, recv, bytes_per_rank, NCCL_INT8, (pmd_nccl_comm)ctx->comm, ctx->stream);
  return rc == 0 ? PMD_OK : rccl_fail(ctx, api, "ncclAllGather", rc);
}
