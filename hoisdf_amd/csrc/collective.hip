// (e) multi-GPU: thin C-ABI wrapper over RCCL for hosts that do not go through torch.distributed
// (the Python engine uses torch.distributed's "nccl" backend, which is the same RCCL).
// Reference: the gradient reduce-add of nn.DataParallel, common/base.py:103 / main/train.py:113,138.
// Built into a separate library (libhoisdf_rccl.so) so that libhoisdf_hip.so carries no RCCL dependency and a
// process that already loaded PyTorch's bundled librccl never sees two copies.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hoisdf_collective.h"

static thread_local char g_err[256] = "";
static int fail(const char* what, const char* detail) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, detail);
  return HOISDF_COLL_ERR;
}
#define RCCL_TRY(call, what)                                  \
  do {                                                        \
    ncclResult_t r_ = (call);                                 \
    if (r_ != ncclSuccess) return fail(what, ncclGetErrorString(r_)); \
  } while (0)

extern "C" const char* hoisdf_coll_last_error(void) { return g_err; }

extern "C" int hoisdf_coll_unique_id(hoisdf_coll_id* id) {
  if (!id) return fail("coll_unique_id", "null pointer");
  static_assert(sizeof(ncclUniqueId) <= sizeof(hoisdf_coll_id), "id buffer too small");
  ncclUniqueId u;
  RCCL_TRY(ncclGetUniqueId(&u), "ncclGetUniqueId");
  memset(id, 0, sizeof(*id));
  memcpy(id->bytes, &u, sizeof(u));
  return HOISDF_COLL_OK;
}

extern "C" int hoisdf_coll_init(void** comm, int world, int rank, const hoisdf_coll_id* id) {
  if (!comm || !id || world <= 0 || rank < 0 || rank >= world) return fail("coll_init", "bad arguments");
  ncclUniqueId u;
  memcpy(&u, id->bytes, sizeof(u));
  ncclComm_t c;
  RCCL_TRY(ncclCommInitRank(&c, world, u, rank), "ncclCommInitRank");
  *comm = c;
  return HOISDF_COLL_OK;
}

// in-place sum over ranks of `count` floats, asynchronous on `stream`; the caller divides by the world size
// (gradient = mean over ranks) or folds 1/world into the optimizer step.
extern "C" int hoisdf_allreduce(void* comm, float* buf, long count, void* stream) {
  if (!comm || (!buf && count > 0) || count < 0) return fail("allreduce", "bad arguments");
  if (count == 0) return HOISDF_COLL_OK;
  RCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
  return HOISDF_COLL_OK;
}

extern "C" int hoisdf_coll_destroy(void* comm) {
  if (!comm) return HOISDF_COLL_OK;
  RCCL_TRY(ncclCommDestroy((ncclComm_t)comm), "ncclCommDestroy");
  return HOISDF_COLL_OK;
}
