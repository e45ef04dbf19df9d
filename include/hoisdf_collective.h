/* hoisdf_collective.h - C ABI of libhoisdf_rccl.so: the one collective of the hot path.
 *
 * Replaces: the gradient reduce-add that nn.DataParallel performs after backward (reference common/base.py:103,
 * main/train.py:113,138).  One process per GPU; every rank calls hoisdf_allreduce once per gradient bucket
 * (52.1 M fp32 = 208 MB per step in 64 MB buckets) on its own stream; RCCL moves the data over xGMI.
 * The Python engine (hoisdf_amd/ddp.py) issues the same RCCL all-reduce through torch.distributed; this library
 * is for hosts without PyTorch.  Same conventions as hoisdf.h: caller-allocated device buffers, stream-taking,
 * asynchronous, int status, thread-local message. */
#ifndef HOISDF_COLLECTIVE_H
#define HOISDF_COLLECTIVE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { HOISDF_COLL_OK = 0, HOISDF_COLL_ERR = -1 };

/* opaque rendezvous token created on rank 0 and broadcast out-of-band (file, socket, MPI) to the other ranks */
typedef struct { char bytes[256]; } hoisdf_coll_id;

const char* hoisdf_coll_last_error(void);
int hoisdf_coll_unique_id(hoisdf_coll_id* id);
/* hipSetDevice(local GPU) before calling; blocks until all `world` ranks have joined */
int hoisdf_coll_init(void** comm, int world, int rank, const hoisdf_coll_id* id);
/* in-place SUM over ranks of count floats in device memory, asynchronous on stream */
int hoisdf_allreduce(void* comm, float* buf, long count, void* stream);
int hoisdf_coll_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
