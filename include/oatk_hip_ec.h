/*
 * include/oatk_hip_ec.h -- C ABI of the device error correction (replaces read_error_correction, syncerr.c:819).
 *
 * Call order: oatk_hip_scan[_host] -> oatk_hip_count -> oatk_hip_ec.  The per-read syncmer chains, the hoco strings and
 * the syncmer table of the batch are already resident in the handle; the caller supplies the EC graph that the
 * reference builds on the host (make_syncmer_graph(sr_db, scm_db, 0, 0.) + scg_consensus(hoco), run_syncasm.c:109-117),
 * flattened in arc-array order.  Vertex sequences are NOT needed: in this graph vertex i is syncmer i and its hoco
 * consensus is the oriented k-mer of the syncmer's first occurrence (syncasm.c:910-940), which the device reads in place.
 */
#ifndef OATK_HIP_EC_H
#define OATK_HIP_EC_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* asmg_t (graph.h:39-63) flattened; HOST pointers.  n_vtx must equal the number of syncmers of the resident count. */
typedef struct {
    uint64_t n_vtx, n_arc;
    const uint64_t *idx_p;     /* [2 n_vtx] asmg_t.idx_p: first arc of oriented vertex v            */
    const uint64_t *idx_n;     /* [2 n_vtx] asmg_t.idx_n: number of arcs                             */
    const uint64_t *arc_v;     /* [n_arc]   asmg_arc_t.v                                             */
    const uint64_t *arc_w;     /* [n_arc]   asmg_arc_t.w                                             */
    const uint64_t *arc_ls;    /* [n_arc]   asmg_arc_t.ls (overlap in hoco bases)                    */
    const uint32_t *arc_cov;   /* [n_arc]   asmg_arc_t.cov                                           */
    const uint8_t *arc_del;    /* [n_arc]   asmg_arc_t.del                                           */
} oatk_ec_graph_t;

/* read_error_correction(sr_db, g, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, ...) (syncerr.c:819):
 * marks error syncmers (find_error_syncmers :679), corrects every read's chain (:339-612), refreshes the syncmer table
 * (update_syncmer_db :769).  Results stay resident; see OATK_BUF_EC_*. */
int oatk_hip_ec(oatk_hip_ctx *ctx, const oatk_ec_graph_t *g, double max_edist, uint32_t err_mer_c, uint32_t max_err_c,
                uint32_t err_arc_c, double max_arc_f);

/* the reference's stats[11] (syncerr.c:76): [0] tail blocks, [1..4] by status FAILURE/SUCCESS/AMBISNQ/AMBISEQ,
 * [5] middle blocks, [6..9] by status, [10] blocks shorter than 10 bases; plus [11] blocks re-run with large slabs */
int oatk_hip_ec_stats(oatk_hip_ctx *ctx, uint64_t *stats12);

/* Test hook: the longest block (hoco bases) the first and the second solver tier accept; longer blocks fall through to the
 * next tier (the last one keeps its scratch in HBM and takes anything).  0 = defaults (3072, 16384).  Results never depend on it. */
int oatk_hip_debug_ec_tiers(oatk_hip_ctx *ctx, int cap_t0, int cap_t1);

/* The device edit distance on its own: wf_ed_core (levdist.c:265-312) in extension mode without traceback -- the routine every error block
 * is solved with (ec_wave.hpp: ecw_step), one wavefront per job, no reads and no graph involved.  Job j aligns the target
 * t_codes[t_off[j], t_off[j + 1]) against growing prefixes of the query q_codes[q_off[j], q_off[j + 1]): for every s in
 * [step_off[j], step_off[j + 1]) the same wavefront is advanced with query length step_ql[s] (ascending; a step of a wf_config_t that is
 * RESUMED, syncerr.c:165-195; one step with the whole query = wf_ed) and out3[3 s ..] receives (score, t_end, q_end) exactly as
 * wf_config_t carries them after the call (ends are one past the last aligned base).  bw[j] < 0: no band.  Codes are one base per byte,
 * values 0..3 -- the alphabet of sr_t.hoco_s, which is all the correction ever aligns.  All pointers are HOST memory. */
int oatk_hip_debug_wf_ed(oatk_hip_ctx *ctx, uint64_t n_jobs, const uint8_t *t_codes, const uint64_t *t_off, const uint8_t *q_codes, const uint64_t *q_off,
                         const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3);
/* The same jobs through the step of the workgroup solver (round 5: the blocks that are not small, one workgroup per block with the wavefront in
 * registers, R = 1, 2 or 6 diagonals per lane; R = 16: the second stage's step, sixteen waves and four steps per barrier, 2 bw + 3 <= 896).  A job needs 2 bw + 3
 * (no band: tl + ql + 3) <= 256 R diagonals; otherwise OATK_E_ARG. */
int oatk_hip_debug_wf_ed_wg(oatk_hip_ctx *ctx, int R, uint64_t n_jobs, const uint8_t *t_codes, const uint64_t *t_off, const uint8_t *q_codes, const uint64_t *q_off,
                            const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3);
/* The two tables by which a long arc of the error-block search is known to die by score without a step (oatk_amd/csrc/ec_tables.hpp: ecb_table): for job j,
 * target t_codes[t_off[j], t_off[j + 1]) of tl bases and string s_codes[s_off[j], s_off[j + 1]) of at most 1024, out[out_off[j] + u], u = 0 .. tl, receives the least edit
 * cost of fitting the WHOLE string into the target from position u on (any end), and out[out_off[j] + tl + 1 + u] that of some PREFIX of the string against the target from
 * u to its end.  out_off[j + 1] - out_off[j] >= 2 (tl + 1).  Host pointers.  (OATK_DEBUG_TABLES_SEG="<waves>:<cut>": as the solver builds them -- by that many waves, table 0
 * in stretches that are exact up to `cut`, table 1 only near the target's end; entries that are not written come back as -1.) */
int oatk_hip_debug_tables(oatk_hip_ctx *ctx, uint64_t n_jobs, const uint8_t *t_codes, const uint64_t *t_off, const uint8_t *s_codes, const uint64_t *s_off, int32_t *out, const uint64_t *out_off);

/* Resident results of oatk_hip_ec (ids for oatk_hip_buffer):
 *   EC_N_SCM   u32[n_reads]      sr_t.n after correction
 *   EC_SCM_OFF u64[n_reads+1]    slots of the corrected chains
 *   EC_KMER    u64[n_occ']       sr_t.k_mer: id << 1 | corrected        EC_MPOS u32  sr_t.m_pos   EC_SMER u64  sr_t.s_mer
 *   EC_SCM_COV u32[n_scm], EC_SCM_DEL u8[n_scm], EC_SCM_OCC_OFF u64[n_scm+1], EC_SCM_OCC u64[n_occ']   (update_syncmer_db)
 *   EC_ERR_DEL u8[n_scm]         syncmer_t.del right after find_error_syncmers (before the refresh)
 */
enum {
    OATK_BUF_EC_N_SCM = 100, OATK_BUF_EC_SCM_OFF, OATK_BUF_EC_KMER, OATK_BUF_EC_MPOS, OATK_BUF_EC_SMER,
    OATK_BUF_EC_SCM_COV, OATK_BUF_EC_SCM_DEL, OATK_BUF_EC_SCM_OCC_OFF, OATK_BUF_EC_SCM_OCC, OATK_BUF_EC_ERR_DEL,
    OATK_BUF_EC_SCM_FWD,        /* u32[n_scm]  forward-strand occurrences per syncmer after correction (del = !fwd, syncerr.c:803-812) */
    OATK_BUF_EC_VTX_SRC,        /* u64[n_scm]  after oatk_hip_ec_mark: byte offset of the hoco string holding the vertex's k-mer, ~0 = none here */
    /* what the search cost, block by block (measurement aid; the layout may change): 12 u32 per error block -- beg_utg lo/hi, end_utg lo/hi, read,
     * beg_pos, length l, r, then 4 words of launch data -- and 12 u32 per block -- status, path entries, path offset lo/hi, flags, short, arcs tried
     * (DFS steps), dead ends counted (n_path, syncerr.c:147), wavefront steps, diagonals covered / 64, time on the wave that finished it (10 ns units), kernel variant */
    OATK_BUF_EC_BLOCK_WORK, OATK_BUF_EC_BLOCK_OUT
};

/* oatk_hip_ec in two steps, for callers that need to act in between (sharded reads, below):
 *   oatk_hip_ec_mark     find_error_syncmers (syncerr.c:679) on the resident graph; EC_ERR_DEL and EC_VTX_SRC become readable
 *   oatk_hip_ec_correct  everything else of read_error_correction */
int oatk_hip_ec_mark(oatk_hip_ctx *ctx, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f);
int oatk_hip_ec_correct(oatk_hip_ctx *ctx, double max_edist);

/* Builds the EC graph on the device from the resident scan + count instead of taking it from the host:
 * make_syncmer_graph(sr_db, scm_db, 0, 0.) (syncasm.c:203-299: one vertex per syncmer, one arc + its complement per pair of
 * syncmers adjacent on a read, arc.cov = number of such adjacencies, arcs in (v, w) order, graph.c:70-113) and the arc
 * overlaps scg_consensus(hoco) assigns (syncasm.c:793-812 with calc_syncmer_overlap :477-582, khashl tie order included).
 * After it, oatk_hip_ec(ctx, NULL, ...) corrects against the resident graph and nothing of the EC round touches the host.
 * Returns OATK_E_SPLIT for the corner the reference leaves unspecified (duplicate (v, w) arcs, graph.c:252).
 *
 * Resident graph (ids for oatk_hip_buffer), oriented vertex = syncmer id << 1 | strand:
 *   EG_IDX_P u64[2 n_scm] (valid where EG_IDX_N > 0)   EG_IDX_N u32[2 n_scm]
 *   EG_ARC_V u64[n_arc]  EG_ARC_W u64[n_arc]  EG_ARC_LS u32[n_arc]  EG_ARC_COV u32[n_arc]  EG_ARC_COMP u8[n_arc] */
int oatk_hip_ec_graph(oatk_hip_ctx *ctx);
/* The LIGHT graph: all that read_error_correction (syncerr.c:819) asks of the graph when err_arc_c >= err_mer_c, as at its one call site
 * (run_syncasm.c:124 passes min_k_cov for both).  find_error_syncmers (:679-757) deletes every syncmer seen fewer than err_mer_c times and
 * every arc that touches one; before that it asks of such an arc only that it exists (:699-706) -- it cannot be "good", an arc being seen at
 * most as often as its rarer end.  So only pairs between two syncmers with coverage >= err_mer_c are sorted into arcs, and the rest leave
 * one flag per oriented vertex.  Marks, corrected reads and refreshed table are identical to those from the full graph; the arc arrays
 * (OATK_BUF_EG_*) hold the kept arcs only.  oatk_hip_ec_mark / oatk_hip_ec refuse (OATK_E_ARG) thresholds a light graph cannot serve. */
int oatk_hip_ec_graph_light(oatk_hip_ctx *ctx, uint32_t err_mer_c);

/* ---- reads sharded by record over several GPUs (one context per GPU) --------------------------------------
 * The EC graph is a property of ALL reads, so every shard builds the same graph from the adjacent pairs of all shards and
 * corrects its own reads against it, in GLOBAL syncmer ids (ranks in the merged, sorted hash table -- what the reference's ids
 * are, syncmer.c:1419-1438).  The exchange steps between the calls belong to the caller: include/oatk_hip_multi.h does them over RCCL from
 * C (oatk_hip_ec_sharded: the light graph from weighted segments, the table partitioned by hash range), oatk_amd/multi.py over
 * torch.distributed in the simplest form, which is the one listed here:
 *
 *   oatk_hip_ec_set_global(n_global, l2g, cov, s)   after the count tables were merged: local id -> global id, global coverage
 *                                                   and s-mer codes (DEVICE pointers; copied).  cov / s need only be right for the ids
 *                                                   the shard looks at: its own syncmers and those seen >= err_mer_c times anywhere
 *   oatk_hip_ec_pairs(&keys, &dist, &n)             this shard's adjacent pairs: canonical key (global ids) and distance, in
 *                                                   (read, slot) order; entries with key ~0 (first slot of a read) are fillers
 *        -- all-gather keys and dist in shard order --
 *   oatk_hip_ec_graph_from_pairs(keys, dist, n)     the graph of all reads, resident
 *   oatk_hip_ec_mark(...)                           identical marks on every shard
 *        -- a live vertex this shard never saw (EC_ERR_DEL == 0, EC_VTX_SRC == ~0) needs its k-mer from a shard that did:
 *           oatk_hip_ec_export_kmers there, oatk_hip_ec_import_kmers here --
 *   oatk_hip_ec_correct(max_edist)                  chains (EC_KMER in global ids), EC_SCM_COV / EC_SCM_FWD of THIS shard's
 *                                                   reads over global ids (sum them over the shards; del = !fwd)
 * All pointers below are DEVICE pointers.  stride: bytes per exported k-mer, a multiple of 16, >= (k + 3) / 4 + 8. */
int oatk_hip_ec_set_global(oatk_hip_ctx *ctx, uint64_t n_global, const uint32_t *d_l2g, const uint32_t *d_cov, const uint64_t *d_s);
int oatk_hip_ec_pairs(oatk_hip_ctx *ctx, const void **d_keys, const void **d_dist, uint64_t *n_pairs);
int oatk_hip_ec_graph_from_pairs(oatk_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_dist, uint64_t n_pairs);
/* the same from run-length compressed lists: entry i stands for (d_val[i] >> 32) consecutive pairs of key d_keys[i] at distance (uint32) d_val[i].
 * What a shard sends instead of its pairs: sorted by key (stably) its list collapses to a few segments per arc, and the table that
 * calc_syncmer_overlap (syncasm.c:477-582) fills -- khashl<int,int>, whose bucket order breaks the ties of the mode (:558-571) -- depends on
 * the calls only through their order, which the segments of all shards in shard order preserve. */
int oatk_hip_ec_graph_from_segments(oatk_hip_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_val, uint64_t n_seg);
int oatk_hip_ec_export_kmers(oatk_hip_ctx *ctx, const uint32_t *d_ids, uint64_t n, uint8_t *d_out, uint32_t stride, uint8_t *d_rev);
int oatk_hip_ec_import_kmers(oatk_hip_ctx *ctx, const uint32_t *d_ids, const uint8_t *d_rev, const uint8_t *d_kmers, uint64_t n, uint32_t stride);
/* room kept behind the hoco strings for imported k-mers (default 1 MiB); call before the scan */
int oatk_hip_ec_reserve_import(oatk_hip_ctx *ctx, uint64_t bytes);

enum {
    OATK_BUF_EG_IDX_P = 120, OATK_BUF_EG_IDX_N, OATK_BUF_EG_ARC_V, OATK_BUF_EG_ARC_W, OATK_BUF_EG_ARC_LS, OATK_BUF_EG_ARC_COV,
    OATK_BUF_EG_ARC_COMP
};

#ifdef __cplusplus
}
#endif
#endif
