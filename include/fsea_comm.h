/*
 * fsea_comm.h -- C ABI of libfsea_rccl.so: the one exchange step of the multi-GPU sweep, a gather of
 * finished output tiles (or spectrum rows) to the root GPU of a node.
 *
 * Host model: ONE process, one host thread per GPU ("member"), each with its own fsea_plan and stream
 * (include/fsea.h); member 0 is the root.  This is what the reference's sweep becomes on an 8-GPU node:
 * the per-frequency loop of c/fft-batch-broad.c:176-206 is cut into one contiguous frequency range per
 * GPU, and the tiles c/fft-stitch-broad.c:62-87 reads back from PNG files travel over xGMI instead.
 *
 * Backends:
 *   "rccl"  ncclCommInitAll over the members' devices; a gather is ncclSend on every non-root member and
 *           one ncclGroupStart/ncclGroupEnd of ncclRecv on the root, so that every peer -> root transfer
 *           uses its own xGMI link concurrently (/opt/rocm/include/rccl/rccl.h: ncclSend :700,
 *           ncclRecv :722).  Needs distinct devices.
 *   "copy"  event-ordered hipMemcpyPeerAsync issued by the root; used when a device appears more than
 *           once in the member list (several members sharing one GPU: how the multi-member control path
 *           is exercised on a single-GPU box) or when FSEA_COMM_BACKEND=copy is set.
 * Plain C: no HIP / RCCL types in the signatures.  0 on success, negative on failure
 * (fsea_comm_last_error()).  Not linked by libfsea_hip.so or libfsea_nrf.so; only the multi-GPU tool
 * (fsea-fft-sweep) uses it, so a host process never holds two RCCL copies (torch ships its own).
 */
#ifndef FSEA_COMM_H
#define FSEA_COMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsea_comm fsea_comm;

/* devices[m] = HIP device of member m; call once, before the member threads start. */
int fsea_comm_create(fsea_comm **comm, int n_members, const int *devices);
int fsea_comm_destroy(fsea_comm *comm);
int fsea_comm_size(const fsea_comm *comm);
const char *fsea_comm_backend(const fsea_comm *comm); /* "rccl" or "copy" */

/* A stream on `device` for a member's launches and transfers (hipStream_t as void*). */
int fsea_comm_stream_create(int device, void **stream);
int fsea_comm_stream_destroy(int device, void *stream);

/* Collective over the members: EVERY member thread calls it once per gather, with identical `bytes`
 * and `offsets` arrays (n_members entries).  Member m contributes bytes[m] bytes at d_src (may be 0);
 * the root receives them at d_dst_root + offsets[m]; the root's own part is a device copy.
 * Asynchronous: ordered behind the work already queued on the calling member's `stream`; the root's
 * data is complete once the root's stream has passed the call.  d_src must stay unchanged until then
 * (fsea_comm_barrier).  d_dst_root is ignored on non-root members. */
int fsea_comm_gather(fsea_comm *comm, int member, const void *d_src, const size_t *bytes, const size_t *offsets,
                     void *d_dst_root, void *stream);

/* Host barrier of the member threads that also drains every member's `stream`: after it, all gathers
 * issued before it are complete on the root and all source buffers may be reused. */
int fsea_comm_barrier(fsea_comm *comm, int member, void *stream);

/* First-contact check of the RCCL backend on a box with ONE GPU, where fsea_comm_create has nothing to gather over:
 * a one-rank communicator on `device` (ncclCommInitAll), then the gather's own call pattern -- ncclGroupStart,
 * ncclSend + ncclRecv of `bytes` bytes of a known pattern with peer 0 (the rank itself), ncclGroupEnd -- on a
 * non-blocking stream, and a byte-for-byte comparison of what arrived.  0 = RCCL loaded, initialised, moved the bytes
 * and they are right; non-zero with fsea_comm_last_error() otherwise.  Says nothing about xGMI. */
int fsea_comm_selftest_rccl(int device, size_t bytes);

const char *fsea_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
