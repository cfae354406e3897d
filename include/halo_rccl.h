/* Optional: a native all-gather for the sharded entry points -- libhalo_rccl.so (csrc/rccl_gather.hip, links librccl).
 *
 * include/halo_accumulation.h keeps collectives outside the library: halo_pcdl_open_sharded / halo_pcdl_check_sharded call the
 * caller's halo_allgather_fn.  A host that has no collective layer of its own (the reference's Rust crate: SURVEY.md 8e,
 * BASELINE.json's "RCCL reduce of partial group sums over xGMI") links this library next to libhalo_hip.so and passes
 * halo_allgather_rccl with a halo_rccl handle as `user`.  RCCL has no reduction over an elliptic-curve group, so the exchange is
 * an all-gather of the ranks' records (96-byte partial points, 264-byte round records) and every host adds them in rank
 * order (halo_point_sum) -- the same bytes on every rank, hence the same sum.  The core library does not depend on this one.
 */
#ifndef HALO_RCCL_H
#define HALO_RCCL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct halo_rccl halo_rccl;

#define HALO_RCCL_ID_BYTES 128
/* Rank 0 makes the communicator's id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever channel started
 * them (environment, file, the launcher's store). */
int halo_rccl_unique_id(uint8_t id[HALO_RCCL_ID_BYTES]);
/* One communicator per rank (ncclCommInitRank) on `device`, with its own stream and staging buffers.  Collective: every rank of
 * `world` must call it with the same id. */
int halo_rccl_create(const uint8_t id[HALO_RCCL_ID_BYTES], int rank, int world, int device, halo_rccl **out);
/* The same around a communicator and stream the host already owns (ncclComm_t, hipStream_t passed as void *): nothing is
 * created or destroyed but the staging buffers. */
int halo_rccl_wrap(void *nccl_comm, void *hip_stream, int world, int device, halo_rccl **out);
void halo_rccl_destroy(halo_rccl *g);
/* halo_allgather_fn (include/halo_accumulation.h): user = the halo_rccl handle.  `words` u64 from `send` (host memory) of every
 * rank arrive in `recv` (host memory, world x words) in rank order: pinned staging -> ncclAllGather on the handle's stream -> back.
 * Returns 0, or non-zero when RCCL or HIP reports an error (halo_rccl_last_error()). */
int halo_allgather_rccl(void *user, const uint64_t *send, size_t words, uint64_t *recv);
/* number of all-gathers this handle has carried, and the ranks of its communicator */
size_t halo_rccl_calls(const halo_rccl *g);
int halo_rccl_world(const halo_rccl *g);
const char *halo_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
