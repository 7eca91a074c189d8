/*
 * include/oatk_hip_stat.h -- C ABI of the scan statistics on the device (sr_db_stat, syncmer.c:867-1028).
 *
 * sr_db_stat sorts all syncmer occurrences twice (by s-mer, by k-mer) to tabulate how often every s-mer / k-mer occurs, histograms
 * those multiplicities and looks for the coverage peaks (ha_analyze_count :768-864); `syncasm -c 0` derives min_k_cov from the k-mer
 * peak (run_syncasm.c:89-92).  The two sorts and the tabulation run on the device; what is left for the host is arithmetic on two
 * 1001-entry histograms (liboatk_host.so: oatk_sr_db_stat).
 */
#ifndef OATK_HIP_STAT_H
#define OATK_HIP_STAT_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OATK_STAT_MAX_DEPTH 1000          /* MAX_DEPTH, syncmer.c:753 */

typedef struct oatk_stat_raw_s {
    uint64_t n_reads, n_syncmers;                    /* n, m of syncmer.c:889-904                                                       */
    int64_t sum_dist;                                /* sum of (p1 - p0 - k) over syncmers adjacent on a read, both with a position      */
    uint64_t n_dist;                                 /* number of such pairs                                                             */
    uint64_t smer_unique, kmer_unique;               /* distinct s-mers / k-mers                                                         */
    int64_t smer_cnt[OATK_STAT_MAX_DEPTH + 1];       /* kh_ctab_cnt (:648-667): [c] = s-mers seen c times, [MAX_DEPTH] = MAX_DEPTH or more */
    int64_t kmer_cnt[OATK_STAT_MAX_DEPTH + 1];
    /* what the reference reports as "singletons" when NO s-mer / k-mer occurs exactly once: kh_ctab_stat (:637-643) then leaves the
     * value of the last bucket it walked in its variable.  Reproduced by replaying the table (khashl, kh_hash_uint32); 0 otherwise. */
    int64_t smer_no_singleton, kmer_no_singleton;
} oatk_stat_raw_t;

/* On the resident batch, at whichever stage it is: after the scan (k-mers are hashes), after the count (ids), after the error
 * correction (corrected chains; corrected entries carry no position and are left out of the distances, as in the reference). */
int oatk_hip_stat(oatk_hip_ctx *ctx, oatk_stat_raw_t *out);

/* The same in two steps, for reads sharded over GPUs (after oatk_hip_ec_correct; k-mer keys are then global ids): every shard lists the s-mer code
 * and the k-mer key of each of its chain entries (DEVICE pointers, n_keys entries each) and its additive figures add4 = {reads, syncmers,
 * sum of distances, number of distances}; the caller all-gathers the two lists (any order: they are sorted here) and sums add4; any shard
 * then tabulates the whole. */
int oatk_hip_stat_keys(oatk_hip_ctx *ctx, const void **d_smer, const void **d_kkey, uint64_t *n_keys, int64_t *add4);
int oatk_hip_stat_from_keys(oatk_hip_ctx *ctx, const uint64_t *d_smer, const uint64_t *d_kkey, uint64_t n, const int64_t *add4, oatk_stat_raw_t *out);

#ifdef __cplusplus
}
#endif
#endif
