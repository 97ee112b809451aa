/* libvxm_comm.so -- the data-parallel exchange of the VxmDense training step as a C ABI:
 * one RCCL communicator per process (one process per MI355X), used from one thread, and exactly one collective on the
 * data path: the SUM all-reduce of the flat fp32 gradient bucket (327,331 elements = 1.31 MB) over xGMI, plus the one-off
 * parameter broadcast at start-up.  Replaces torch.nn.DataParallel's per-step broadcast + gather + reduce-to-GPU-0
 * (reference scripts/torch/train.py:151-154).  It is a separate library so that libvxm_hip.so carries no RCCL dependency;
 * like that library it links its runtime by SONAME only and must be loaded after `import torch`.
 * Every function returns 0 on success, a non-zero status otherwise (vxm_comm_last_error_string() has the text). */
#ifndef VXM_COMM_H
#define VXM_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VXM_COMM_UNIQUE_ID_BYTES 128

const char* vxm_comm_last_error_string(void);
/* rank 0 creates the rendezvous token (ncclGetUniqueId) and hands its 128 bytes to the other ranks out of band
 * (the torchrun store, a file, MPI ...) */
int vxm_comm_unique_id(void* out /* VXM_COMM_UNIQUE_ID_BYTES */);
/* collective over all ranks: binds the communicator to the CURRENT HIP device of the calling process.  Bounded: if the other ranks
 * do not join within VXM_COMM_INIT_TIMEOUT_S seconds (default 180) it returns status 5 instead of blocking for ever (a communicator
 * that completes after the caller gave up is aborted by the helper thread that built it: no leak, no half-member). */
int vxm_comm_init(int rank, int world, const void* unique_id);
int vxm_comm_world(void);                      /* ranks of the communicator (ncclCommCount at init); 0 before init */
int vxm_comm_rccl_version(void);               /* ncclGetVersion code of the RCCL the library resolved to (e.g. 22606); 0 on failure */
/* in-place SUM all-reduce / broadcast of n floats on `stream` (asynchronous, ordered with the kernels on that stream) */
int vxm_allreduce_sum_f32(float* buf, int64_t n, void* stream);
int vxm_broadcast_f32(float* buf, int64_t n, int root, void* stream);
int vxm_comm_destroy(void);
/* ncclCommAbort: give up a communicator whose collective can no longer complete (a peer failed before entering it); frees it */
int vxm_comm_abort(void);

#ifdef __cplusplus
}
#endif
#endif
