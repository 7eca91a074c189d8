/*
 * include/oatk_hip_cons.h -- C ABI of the base-space syncmer consensus on the device (scg_syncmer_consensus, syncasm.c:888-1003).
 *
 * Call after oatk_hip_count (reads as scanned) or after oatk_hip_ec (corrected chains: corrected entries are skipped, :958-959).
 * Reads sharded over GPUs are not supported yet: the totals of a syncmer are sums over all shards.
 */
#ifndef OATK_HIP_CONS_H
#define OATK_HIP_CONS_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* For every syncmer that is not deleted and has coverage >= min_cov (and >= 1): the rounded mean run length at each of the k hoco
 * positions of its FORWARD k-mer over its uncorrected occurrences -- the `b` of syncasm.c:994 -- their number, and the first of them
 * (whose bases the reference prints, :912-936).  The reverse orientation is the mirror image.  Results stay resident:
 *   CONS_SEL   u32[n_sel]       the syncmer ids, ascending          CONS_SLOT  u32[n_scm]   slot of a syncmer in CONS_SEL, ~0 = not selected
 *   CONS_RL    u32[n_sel * k]   lround(total run length / CONS_MSEQ) per forward position
 *   CONS_MSEQ  u32[n_sel]       occurrences that took part (0: every occurrence was corrected away, the reference prints N's)
 *   CONS_FIRST u64[n_sel]       sid << 32 | idx << 1 | rev of the first of them, ~0 if none */
int oatk_hip_consensus(oatk_hip_ctx *ctx, uint32_t min_cov);

/* The same for a caller-chosen list of syncmer ids (DEVICE pointer, ascending).  With reads sharded over GPUs (oatk_hip_ec_set_global ...
 * oatk_hip_ec_correct done) the ids are global and the results cover THIS shard's occurrences only: add CONS_TOT (u64[n * k], the totals
 * before the division) and CONS_MSEQ over the shards, then round(total / count) is the reference's value; the bases come from the shard
 * of lowest rank whose CONS_FIRST is not ~0 (oatk_amd/multi.py: ShardedEc.consensus). */
int oatk_hip_consensus_ids(oatk_hip_ctx *ctx, const uint32_t *d_ids, uint64_t n);

/* What calc_syncmer_overlap (syncasm.c:477-582) tabulates, for EVERY pair of syncmers adjacent on a read, in one pass over the resident
 * chains (after oatk_hip_ec: the corrected ones, pairs with a corrected member left out as :499 / :511 do).  Per pair, under its canonical
 * key ((v << 32 | w) with v <= w, or the complementary pair's; v, w = syncmer id << 1 | strand): the distinct distances between the two
 * syncmers in the order the reference's walk meets them first, their counts, and whether the walk's last add_ovl_count call was a repeat
 * (a khashl table grows at the call AFTER the insert that filled it, so this decides the final bucket order, khashl.h:199).  Replaying
 * kh_put for the distinct distances in this order -- plus one more put of any of them when the flag is set -- leaves the reference's own
 * table in exactly the state its walk would, so the mode and its tie-break (:558-571) come out identical.
 *   OVL_KEY  u64[n_pairs]  ascending      OVL_OFF u64[n_pairs + 1]      OVL_DIST i32[n_entries]   OVL_CNT u32[n_entries]   OVL_TAIL u8[n_pairs]
 * Returns OATK_E_SPLIT when a pair has more than 64 distinct distances. */
int oatk_hip_overlap_hist(oatk_hip_ctx *ctx, uint64_t *n_pairs, uint64_t *n_entries);
/* With reads sharded over GPUs (after oatk_hip_ec_correct, global ids) the table of a pair is made of all shards' adjacencies: every shard lists its
 * pairs (canonical key -- ~0 for fillers -- and distance, in (read, slot) order; DEVICE pointers), the caller all-gathers both lists in shard
 * order, which is the order of one context holding all reads, and any shard builds the same tables from the whole. */
int oatk_hip_overlap_pairs(oatk_hip_ctx *ctx, const void **d_keys, const void **d_dist, uint64_t *n);
int oatk_hip_overlap_hist_from_pairs(oatk_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_dist, uint64_t n, uint64_t *n_pairs, uint64_t *n_entries);

enum { OATK_BUF_CONS_SEL = 140, OATK_BUF_CONS_SLOT, OATK_BUF_CONS_RL, OATK_BUF_CONS_MSEQ, OATK_BUF_CONS_FIRST, OATK_BUF_CONS_TOT };
enum { OATK_BUF_OVL_KEY = 150, OATK_BUF_OVL_OFF, OATK_BUF_OVL_DIST, OATK_BUF_OVL_CNT, OATK_BUF_OVL_TAIL };

#ifdef __cplusplus
}
#endif
#endif
