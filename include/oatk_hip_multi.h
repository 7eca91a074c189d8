/*
 * include/oatk_hip_multi.h -- reads sharded by record over several MI355X, from C (the exchange steps of the hot path).
 *
 * One handle (oatk_hip_ctx) per GPU holds a contiguous range of the reads: rank r scans reads [first_r, first_r + n_r) with sid0 = first_r
 * and counts them on its own.  Two collective calls then make the ranks agree:
 *
 *   oatk_hip_merge_counts   the count-table merge before graph construction: all-gather of every rank's distinct k-mer hashes (with their
 *                           s-mers), the same sorted global key array on every rank, a local -> global id map, and an ALL-REDUCE (sum) of
 *                           the dense coverage vector -- collect_syncmer_from_reads (syncmer.c:1397) for the union of the shards.
 *   oatk_hip_ec_sharded     read_error_correction (syncerr.c:819) with sharded reads: the graph of ALL reads from the all-gathered adjacent
 *                           pairs (replicated: it is small next to the reads), every rank corrects its own reads in global ids; k-mers of
 *                           good syncmers a shard never saw travel once; the refreshed coverage is all-reduced (update_syncmer_db :769).
 *
 * The collectives run over RCCL (xGMI): an oatk_comm wraps an ncclComm_t created from a 128-byte id that rank 0 makes and hands to the others
 * by whatever means the host program has (a file, a socket, MPI, a launcher's key-value store).  librccl is loaded when the first
 * communicator is made, not before.  Ranks that live in ONE process -- one thread and one handle per GPU, or, for tests on a single GPU,
 * several handles on it -- can use a local communicator group instead (rendezvous in host memory, device-to-device copies).
 *
 * Results are bit-identical to one handle holding all the reads (tests/test_gpu_multi_c.py).  Every rank must make the same calls in the same
 * order; a call returns only when the rank's part is complete.
 */
#ifndef OATK_HIP_MULTI_H
#define OATK_HIP_MULTI_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oatk_comm oatk_comm;

/* RCCL: rank 0 calls oatk_comm_unique_id and distributes the 128 bytes; every rank then calls oatk_comm_create with its own device */
int oatk_comm_unique_id(uint8_t id[128]);
oatk_comm *oatk_comm_create(const uint8_t id[128], int rank, int n_ranks, int device);
/* ranks inside one process: make the group once, then one communicator per rank (each used by its own thread) */
typedef struct oatk_comm_group oatk_comm_group;
oatk_comm_group *oatk_comm_group_create(int n_ranks);
oatk_comm *oatk_comm_group_rank(oatk_comm_group *g, int rank);
void oatk_comm_group_destroy(oatk_comm_group *g);
void oatk_comm_destroy(oatk_comm *c);
int oatk_comm_rank(const oatk_comm *c);
int oatk_comm_size(const oatk_comm *c);
const char *oatk_comm_backend(const oatk_comm *c);          /* "rccl" or "local" */

/* after oatk_hip_count on every rank.  Resident afterwards (oatk_hip_buffer): MG_H u64[n_global] hashes ascending, MG_S u64[n_global] s-mers,
 * MG_COV u32[n_global] coverage over all shards, MG_L2G u32[n_local] global id of every local syncmer.  OATK_E_SPLIT when this shard's table
 * holds one hash twice (a split collision: ranks by hash are ambiguous), OATK_E_SMER when one hash carries different s-mers on two shards. */
int oatk_hip_merge_counts(oatk_hip_ctx *ctx, oatk_comm *comm, uint64_t *n_global);

/* the whole EC round for sharded reads (merge included if it has not run).  Afterwards the handle's corrected chains (OATK_BUF_EC_*) are in
 * GLOBAL syncmer ids, and MG_EC_COV u32[n_global] / MG_EC_DEL u8[n_global] hold update_syncmer_db's table over all shards; stats12 (may be
 * NULL) receives the block statistics summed over the ranks, *n_imported (may be NULL) the k-mers this rank had to be sent. */
int oatk_hip_ec_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, double max_edist, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c,
                        double max_arc_f, uint64_t *stats12, uint64_t *n_imported);

enum { OATK_BUF_MG_H = 220, OATK_BUF_MG_S, OATK_BUF_MG_COV, OATK_BUF_MG_L2G, OATK_BUF_MG_EC_COV, OATK_BUF_MG_EC_DEL };

#ifdef __cplusplus
}
#endif
#endif
