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
 *
 * STATUS OF THE RCCL BACKEND: validated against the real librccl with ONE rank (every collective call on the stream of a real communicator),
 * with 2 - 8 ranks over the local group on one GPU, and with 2 - 4 ranks over a mock of the RCCL entry points whose ranks are threads
 * (tests/mock_rccl_run.py) -- not yet on two physical GPUs: no multi-GPU node has been available to its builders.  Treat it as experimental
 * until tests/test_gpu_multi_c.py has passed on such a node.  A communicator is bound to the device it was created for: a call with a handle of
 * another device is refused (OATK_E_ARG), and a rank that fails between collectives aborts the communicator / poisons the group so that its
 * peers return an error instead of waiting for it.
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
/* What this rank has put into the collectives since the communicator was made (or since the last call with reset != 0): out[0] small all-gathers
 * (counts, verdicts), out[2] all-gathers of arrays, out[4] all-reduces, out[6] personalised exchanges -- calls; out[1], out[3], out[5], out[7] the
 * bytes of this rank's contribution to them (an exchange counts what goes to every rank, its own share included).  What a scaling estimate needs:
 * the number of latencies and the bytes per link (bench.py: scale_model). */
void oatk_comm_traffic(oatk_comm *c, uint64_t out[8], int reset);

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

/* ---- up to the graph hand-off -----------------------------------------------------------------------------------------------------------------
 * What syncasm() does after the count and after the correction consumes properties of ALL reads: one syncmer_db_t whose syncmers carry their
 * occurrences in (sid, idx) order (syncmer.c:1353-1360; rebuilt by update_syncmer_db, syncerr.c:796-805), the graph of run_syncasm.c:138, the
 * consensus sums and distance tables behind scg_consensus (syncasm.c:716-823), the statistics of sr_db_stat (syncmer.c:867).  The calls below
 * produce them from sharded reads; every rank makes the same call; results are bit-identical to one handle holding all the reads
 * (tests/test_gpu_multi_tail.py).  Per rank, the traffic of each is bounded by what the result itself weighs -- none grows with the number
 * of ranks the way an all-gather of every shard's chains would.
 *
 * oatk_hip_gather_table   the table as it stands -- after oatk_hip_merge_counts: what collect_syncmer_from_reads returns; after
 *                         oatk_hip_ec_sharded: what update_syncmer_db leaves -- assembled on rank `root` (ids for oatk_hip_buffer, valid there):
 *                           MG_G_H, MG_G_S u64[n_global]   MG_G_COV u32[n_global]   MG_G_DEL u8[n_global]
 *                           MG_G_OCC_OFF u64[n_global + 1]   MG_G_OCC u64[sum of coverage]   sid << 32 | idx << 1 | rev, per syncmer in (sid, idx) order
 *                         The owners' ranges are concatenated; every rank sends its occurrence words with their global ids once (12 bytes per
 *                         occurrence) and root sorts them stably by id -- the ranks' parts arrive in rank order, which is sid order.  Before the
 *                         correction every rank also keeps MG_POS_GKID u64[n_occ]: sr_t.k_mer of its own reads as the count leaves it (global
 *                         id << 1, syncmer.c:1378), slot order of OATK_BUF_SCM_OFF. */
int oatk_hip_gather_table(oatk_hip_ctx *ctx, oatk_comm *comm, int root);
/* make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) + asmg_finalize (include/oatk_hip_graph.h) of all reads, after oatk_hip_ec_sharded: the
 * refreshed coverage and deletion marks are all-gathered from their owners (5 bytes per syncmer); each rank sorts and run-length encodes the
 * canonical keys of ITS adjacent pairs between two surviving vertices and the (key, count) lists are all-gathered -- a true arc is one entry per
 * rank however many reads cross it; the same graph is then built on every rank: OATK_BUF_AG_* in global ids. */
int oatk_hip_asm_graph_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, uint32_t min_k_cov, double min_a_cov_f, uint64_t *n_vtx, uint64_t *n_arc);
/* oatk_hip_consensus (include/oatk_hip_cons.h) of all reads, after oatk_hip_ec_sharded: every rank adds up the run lengths of its own occurrences of
 * the selected syncmers (not deleted, coverage over all shards >= min_cov); totals and counts are all-reduced, the first uncorrected occurrence is
 * the minimum over the ranks (sid in the top bits).  OATK_BUF_CONS_* identical on every rank, CONS_FIRST naming a read of whichever rank. */
int oatk_hip_consensus_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, uint32_t min_cov);
/* oatk_hip_overlap_hist of all reads, after oatk_hip_ec_sharded, for the pairs both of whose members are not deleted and seen >= min_cov times (the
 * only pairs scg_consensus asks about are neighbours in the graph; min_cov = 0: every pair).  Each rank's pairs travel as weighted segments (key,
 * distance, calls) -- its list sorted by key, stably, so a segment is a run of consecutive add_ovl_count calls (syncasm.c:477-582) -- and the
 * segments of all ranks in rank order are the calls of one database in read order.  OATK_BUF_OVL_* identical on every rank. */
int oatk_hip_overlap_hist_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, uint32_t min_cov, uint64_t *n_pairs, uint64_t *n_entries);
/* oatk_hip_stat (include/oatk_hip_stat.h) of all reads, right after the scans (run_syncasm.c:88) or after oatk_hip_ec_sharded (:131).  The key space
 * is cut into one range per rank; every rank sorts and run-length encodes its own keys and sends each range's (key, count) entries to its owner,
 * which merges them and histograms the multiplicities; the 1001-bin histograms and the distinct counts are all-reduced.  `out` identical on every rank. */
struct oatk_stat_raw_s;
int oatk_hip_stat_sharded(oatk_hip_ctx *ctx, oatk_comm *comm, struct oatk_stat_raw_s *out);

enum { OATK_BUF_MG_G_H = 230, OATK_BUF_MG_G_S, OATK_BUF_MG_G_COV, OATK_BUF_MG_G_DEL, OATK_BUF_MG_G_OCC_OFF, OATK_BUF_MG_G_OCC, OATK_BUF_MG_POS_GKID };

#ifdef __cplusplus
}
#endif
#endif
