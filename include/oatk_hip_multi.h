/*
 * include/oatk_hip_multi.h -- reads sharded by record over several MI355X, from C (the exchange steps of the hot path).
 *
 * One handle (oatk_hip_ctx) per GPU holds a contiguous range of the reads: rank r scans reads [first_r, first_r + n_r) with sid0 = first_r
 * and counts them on its own.  Two collective calls then make the ranks agree:
 *
 *   oatk_hip_merge_counts   the count-table merge before graph construction -- collect_syncmer_from_reads (syncmer.c:1397) for the union of
 *                           the shards.  The merged table is PARTITIONED BY HASH RANGE (SURVEY.md 8e option 1): rank r owns the hashes h
 *                           with floor(h N / 2^64) = r.  A shard's table is sorted by hash, so what it owes each owner is a contiguous
 *                           range of it; one personalised exchange (grouped ncclSend / ncclRecv) takes (hash, s-mer, count) to the owners,
 *                           each owner sorts and sums ITS range, and the ranges in rank order are the global table in hash order: a
 *                           global id = the owner's first id + the rank inside its range (what the reference's ids are, :1419-1438).
 *                           The reply brings every shard the global id and the coverage over all shards of each of ITS syncmers.  Per
 *                           rank, traffic and sort are the size of one shard's table whatever the number of shards.
 *   oatk_hip_ec_sharded     read_error_correction (syncerr.c:819) with sharded reads: the graph of ALL reads, replicated on every rank but
 *                           only as far as the correction can use it (include/oatk_hip_ec.h: the light graph) -- owners announce their
 *                           candidates (syncmers seen >= err_mer_c times: thousands), shards exchange their pairs between candidates as
 *                           run-length compressed segments; every rank corrects its own reads in global ids; k-mers of good syncmers a
 *                           shard never saw travel once; the refreshed table (update_syncmer_db :769) ends up partitioned like the merged
 *                           one (candidates by an all-reduce over their list, the rest along the routes of the merge).
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

/* after oatk_hip_count on every rank.  Resident afterwards (oatk_hip_buffer): this rank's RANGE of the global table -- MG_H u64[n_owned] hashes
 * ascending, MG_S u64[n_owned] s-mers, MG_COV u32[n_owned] coverage over all shards (global ids first_id .. first_id + n_owned, oatk_hip_multi_range;
 * the ranges of all ranks in rank order are the whole table) -- and for its own syncmers MG_L2G u32[n_local] global id, MG_LCOV u32[n_local]
 * coverage over all shards.  Entries that share a hash -- one per shard for a true syncmer; or different k-mers, a 64-bit collision, within a
 * shard or across shards -- are compared k-mer by k-mer on the owner (their shards send the k-mers) and clustered in first-seen order like
 * process_kmer_cluster (syncmer.c:1293-1335) does for one database: the ids are the reference's, collisions included.  OATK_E_SMER when
 * identical k-mers carry different s-mers (fatal in the reference, :1370). */
int oatk_hip_merge_counts(oatk_hip_ctx *ctx, oatk_comm *comm, uint64_t *n_global);
int oatk_hip_multi_range(oatk_hip_ctx *ctx, uint64_t *first_id, uint64_t *n_owned, uint64_t *n_global);      /* any pointer may be NULL */

/* the whole EC round for sharded reads (merge included if it has not run).  Afterwards the handle's corrected chains (OATK_BUF_EC_*) are in
 * GLOBAL syncmer ids, and MG_EC_COV u32[n_owned] / MG_EC_DEL u8[n_owned] hold update_syncmer_db's table over all shards for this rank's range;
 * stats12 (may be NULL) receives the block statistics summed over the ranks, *n_imported (may be NULL) the k-mers this rank had to be sent.
 * With err_arc_c < err_mer_c (never from syncasm, run_syncasm.c:124) the light graph cannot serve: every pair of every shard travels and the
 * whole table is gathered on every rank -- correct, and as expensive as the number of shards makes it. */
int oatk_hip_ec_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, double max_edist, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c,
                        double max_arc_f, uint64_t *stats12, uint64_t *n_imported);

enum { OATK_BUF_MG_H = 220, OATK_BUF_MG_S, OATK_BUF_MG_COV, OATK_BUF_MG_L2G, OATK_BUF_MG_EC_COV, OATK_BUF_MG_EC_DEL, OATK_BUF_MG_LCOV };

#ifdef __cplusplus
}
#endif
#endif
