/*
 * include/oatk_hip_align.h -- C ABI of the read -> unitig alignment on the device: scg_read_alignment (alignment.c:596-691) with its
 * per-read routine scg_ra_analysis_thread (alignment.c:180-594).  No base-level work: a read is its chain of syncmers; every syncmer
 * is looked up in the unitigs (scg->idx_u), hits are sorted, colinear fragments are collected per unitig, fragments are chained across
 * arcs without gaps or clipping, and ALL chains of maximal score are reported.
 *
 * The reads' chains are the resident ones (after oatk_hip_ec: the corrected chains).  The unitig graph lives on the host and changes
 * between calls (unitigging, cleaning, multiplexing), so it is passed in, flattened.
 */
#ifndef OATK_HIP_ALIGN_H
#define OATK_HIP_ALIGN_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HOST pointers.  n_scm must equal the number of syncmers of the resident count -- with reads sharded over GPUs (oatk_hip_ec_set_global ...
 * oatk_hip_ec_correct done) the size of the GLOBAL table: reads align independently, so every shard aligns its own against the same graph
 * in global ids and the results are simply concatenated in shard order (RA_ALN_SID counts from the shard's first read). */
typedef struct {
    uint64_t n_scm, n_utg, n_arc;
    const uint64_t *su_off;    /* [n_scm + 1] scg->idx_u as offsets into the two arrays below (syncasm.c:116-181)                 */
    const uint64_t *su_uid;    /* [su_off[n_scm]] unitig << 1 | strand of the syncmer on it (scm_utg_uid, scm_utg_rev)             */
    const uint32_t *su_pos;    /*                 position on the unitig (scm_utg_pos)                                             */
    const uint32_t *utg_n;     /* [n_utg] syncmers per unitig (asmg_vtx_t.n)                                                       */
    const uint64_t *idx_p;     /* [2 n_utg] asmg_t.idx_p / idx_n                                                                   */
    const uint64_t *idx_n;
    const uint64_t *arc_w;     /* [n_arc] asmg_arc_t.w, .ln, .del in array order                                                   */
    const uint64_t *arc_ln;
    const uint8_t *arc_del;
} oatk_ra_graph_t;

/* old_ra: [n_reads] the filter of alignment.c:610-634 (bit 0: align this read; >> 1: the score an alignment must reach), HOST pointer,
 * NULL = every read, no threshold.  stats3: [0] reads with at least one alignment, [1] with exactly one (alignment.c:581-582),
 * [2] reads that exceeded the per-read working limits (listed in RA_SKIPPED; the caller aligns those itself).
 *
 * Resident result (ids for oatk_hip_buffer), alignments in read order, per read in the reference's backtrace order:
 *   RA_ALN_SID  u32[n_aln]      read index (sid - sid0)          RA_ALN_OFF u64[n_aln + 1]  fragments of alignment i
 *   RA_ALN_S    f64[n_aln]      scg_ra_t.s = 1 / (alignments of the read) + score
 *   RA_FRG_UID  u64[n_frg]      ra_frg_t.uid                      RA_FRG_UBEG / RA_FRG_UEND u32[n_frg]   RA_FRG_SBEG / RA_FRG_SEND u32[n_frg]
 *   RA_SKIPPED  u32[stats3[2]]  read indices */
int oatk_hip_read_alignment(oatk_hip_ctx *ctx, const oatk_ra_graph_t *g, const int64_t *old_ra, uint64_t *n_aln, uint64_t *n_frg, uint64_t *stats3);

/* Test hook: 1 = count, scan, run the routine a second time writing in place (the path taken when the output pool of the normal,
 * single-run path is too small), 0 = default.  Results never depend on it. */
int oatk_hip_debug_align_two_pass(oatk_hip_ctx *ctx, int on);

enum {
    OATK_BUF_RA_ALN_SID = 200, OATK_BUF_RA_ALN_OFF, OATK_BUF_RA_ALN_S, OATK_BUF_RA_FRG_UID, OATK_BUF_RA_FRG_UBEG, OATK_BUF_RA_FRG_UEND,
    OATK_BUF_RA_FRG_SBEG, OATK_BUF_RA_FRG_SEND, OATK_BUF_RA_SKIPPED
};

#ifdef __cplusplus
}
#endif
#endif
