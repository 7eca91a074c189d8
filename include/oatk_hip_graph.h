/*
 * include/oatk_hip_graph.h -- C ABI of the assembly graph built on the device: the second make_syncmer_graph call of
 * syncasm(), make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (run_syncasm.c:138, syncasm.c:203-299), with
 * everything asmg_finalize(g, 1) does to it (graph.c:148-203 cleanup, :70-113 sort + index, :205-233 symmetry flags,
 * :126-146 link ids).
 *
 * Call order: oatk_hip_scan -> oatk_hip_count [-> oatk_hip_ec] -> oatk_hip_asm_graph.  The chains, coverages and deletion marks
 * it reads are the resident ones: after oatk_hip_ec the corrected chains and the refreshed table (update_syncmer_db,
 * syncerr.c:769), otherwise the fresh count (cov = occurrences, nothing deleted).
 *
 * What the reference does, and so does this:
 *   syncmer i is dropped when it is deleted or cov < min_k_cov (syncasm.c:226-233); scm.del is updated (AG_SCM_DEL);
 *   every pair of syncmers adjacent on a read is counted under its canonical oriented key (:242-261);
 *   a key becomes an arc and (unless it is its own) the complementary arc when neither end is dropped and
 *   count >= min_a_cov_f * MIN(cov(v), cov(w)) in double arithmetic (:264-282);
 *   dropped vertices are squeezed out and the arcs renumbered (asmg_cleanup), arcs sorted by (v, w) and indexed,
 *   a self-complementary arc ends with comp = 1, and arc + complement share a link id numbered in arc order.
 * Every vertex of the result is one syncmer: vtx.n = 1, vtx.a[0] = AG_VTX_SCM << 1, vtx.cov = AG_VTX_COV, seq = NULL, ln = ls = 0.
 */
#ifndef OATK_HIP_GRAPH_H
#define OATK_HIP_GRAPH_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* n_vtx / n_arc: dimensions of the result (may be NULL).  Returns OATK_E_SPLIT for the corner the reference leaves
 * unspecified (duplicate (v, w) arcs, graph.c:252 "TODO fix multi-arc"). */
int oatk_hip_asm_graph(oatk_hip_ctx *ctx, uint32_t min_k_cov, double min_a_cov_f, uint64_t *n_vtx, uint64_t *n_arc);

/* Reads sharded by record: the graph is a property of ALL reads.  Each shard lists the canonical keys of its
 * adjacent pairs (global ids after oatk_hip_ec_set_global; entries ~0 are fillers), the caller all-gathers them and sums the
 * per-shard EC_SCM_COV / EC_SCM_FWD (del = fwd == 0), then any shard builds the same graph from the whole.  DEVICE pointers. */
int oatk_hip_asm_pairs(oatk_hip_ctx *ctx, const void **d_keys, uint64_t *n_pairs);
int oatk_hip_asm_graph_from_pairs(oatk_hip_ctx *ctx, const uint64_t *d_keys, uint64_t n_pairs, uint64_t n_scm, const uint32_t *d_cov,
                                  const uint8_t *d_del, uint32_t min_k_cov, double min_a_cov_f, uint64_t *n_vtx, uint64_t *n_arc);

/* Resident result (ids for oatk_hip_buffer); oriented vertex = vertex id << 1 | strand:
 *   AG_SCM_DEL  u8[n_scm]      syncmer_t.del after the coverage filter
 *   AG_VTX_SCM  u32[n_vtx]     the syncmer a vertex stands for (ascending)      AG_VTX_COV u32[n_vtx]
 *   AG_IDX_P    u64[2 n_vtx]   (valid where AG_IDX_N > 0)                        AG_IDX_N   u32[2 n_vtx]
 *   AG_ARC_V / AG_ARC_W u64[n_arc]   AG_ARC_COV u32[n_arc]   AG_ARC_COMP u8[n_arc]   AG_ARC_LINK u64[n_arc] */
enum {
    OATK_BUF_AG_SCM_DEL = 180, OATK_BUF_AG_VTX_SCM, OATK_BUF_AG_VTX_COV, OATK_BUF_AG_IDX_P, OATK_BUF_AG_IDX_N,
    OATK_BUF_AG_ARC_V, OATK_BUF_AG_ARC_W, OATK_BUF_AG_ARC_COV, OATK_BUF_AG_ARC_COMP, OATK_BUF_AG_ARC_LINK
};

#ifdef __cplusplus
}
#endif
#endif
