/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the algorithm on oatk's syncasm hot path:
 * homopolymer compression + closed-syncmer scan + k-mer hash, syncmer count /
 * ID assignment, scan statistics, wavefront edit distance and per-read
 * syncmer-chain error correction.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).
 *
 * Parity status: PINNED.  oracle/_ref (the reference compiled from its own
 * sources by oracle/Makefile) is run side by side with these functions in
 * tests/test_oracle_vs_ref.py (this container), and the committed vectors in
 * tests/golden/ were produced by that compiled reference with
 * tests/make_golden.py.
 *
 * Nothing in the product path (oatk_amd/, include/, bench.py's timed region)
 * may include, link or call anything in this directory.
 */
#ifndef OATK_ORACLE_H
#define OATK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NO_HASH UINT64_MAX

/* flat result of scanning a batch of reads; arrays are concatenated in read order */
typedef struct {
    uint64_t n_reads;
    uint64_t tot_hoco, tot_bytes, tot_scm, tot_lrl, tot_nn;
    uint32_t *hoco_l;   /* [n_reads]   sr_t.hoco_l                       */
    uint32_t *n_scm;    /* [n_reads]   sr_t.n                            */
    uint32_t *n_lrl;    /* [n_reads]   entries of ho_l_rl per read       */
    uint32_t *n_nn;     /* [n_reads]   entries of n_nucl per read        */
    uint8_t  *hoco_s;   /* [tot_bytes] ceil(hoco_l/4) bytes per read     */
    uint8_t  *ho_rl;    /* [tot_hoco]                                    */
    uint32_t *ho_l_rl;  /* [tot_lrl]                                     */
    uint32_t *n_nucl;   /* [tot_nn]                                      */
    uint32_t *m_pos;    /* [tot_scm]   pos << 1 | rev                    */
    uint64_t *s_mer;    /* [tot_scm]   canonical s-mer << 1 | strand-ish */
    uint64_t *k_mer;    /* [tot_scm]   MurmurHash64A of oriented k-mer   */
} orc_scan_t;

/* mode 0: streaming state machine (syncmer.c:243-421 restated step by step)
 * mode 1: stateless window predicates (the formulation the HIP kernel uses) */
orc_scan_t *orc_scan_batch(const uint8_t *seq, const uint64_t *off, uint64_t n_reads, int K, int S, int mode);
void orc_scan_free(orc_scan_t *r);

uint8_t  orc_nt4(uint8_t ch);
uint64_t orc_hash64(uint64_t key, uint64_t mask);
uint64_t orc_murmur64a(const void *key, uint32_t len, uint64_t seed);
uint64_t orc_kmer_hash(const uint8_t *hoco_s, uint32_t pos, uint32_t rev, int K);
/* MSB-first 2-bit packing of the oriented k-mer into out[(K-1)/4+1] */
void orc_kmer_pack(const uint8_t *hoco_s, uint32_t pos, uint32_t rev, int K, uint8_t *out);

/* syncmer database produced by the count step */
typedef struct {
    uint64_t n_scm, tot_occ;
    uint64_t *h;       /* [n_scm] k-mer hash                                 */
    uint64_t *s;       /* [n_scm] s-mer code                                 */
    uint32_t *cov;     /* [n_scm]                                            */
    uint64_t *occ_off; /* [n_scm+1] CSR offsets into occ                     */
    uint64_t *occ;     /* [tot_occ] sid << 32 | idx << 1 | rev, (sid,idx) asc */
    uint64_t *k_id;    /* [tot_occ of input order] rewritten sr->k_mer = id << 1 */
    int      err;      /* 1 = identical k-mers with different s-mers (fatal in the reference) */
} orc_count_t;

orc_count_t *orc_count(const orc_scan_t *sc, int K);
void orc_count_free(orc_count_t *c);

/* edit distance (levdist.c); state is resumable across growing query prefixes */
typedef struct orc_wf orc_wf_t;
orc_wf_t *orc_wf_new(const char *ts, int32_t tl, int32_t bw);
void orc_wf_step(orc_wf_t *w, const char *qs, int32_t ql, int32_t *out3);
void orc_wf_free(orc_wf_t *w);
orc_wf_t *orc_wf_clone(const orc_wf_t *w);
void orc_wf_ed(int32_t tl, const char *ts, int32_t ql, const char *qs, int32_t bw, int32_t *out3);
/* closed form by full DP: min over last row / last column, ties -> smallest diagonal */
void orc_ed_bruteforce(int32_t tl, const char *ts, int32_t ql, const char *qs, int32_t *out3);
/* (score, t_end, q_end) currently held by the state */
void orc_wf_state(const orc_wf_t *w, int32_t *out3);

/* error correction (oracle/ec.c).  The reference's asmg_t flattened in arc-array order. */
typedef struct {
    uint64_t n_vtx, n_arc;
    const uint64_t *vtx_len;      /* [n_vtx] consensus length                       */
    uint8_t *vtx_del;             /* [n_vtx] updated by orc_find_error_syncmers      */
    const uint64_t *vtx_seq_off;  /* [n_vtx] offset of the hoco consensus in seq     */
    const char *seq;
    const uint64_t *arc_w;        /* [n_arc] target oriented vertex                  */
    const uint64_t *arc_ls;       /* [n_arc] overlap length                          */
    const uint32_t *arc_cov;      /* [n_arc]                                         */
    uint8_t *arc_del;             /* [n_arc] updated by orc_find_error_syncmers      */
    const uint64_t *idx_p, *idx_n; /* [2 n_vtx] first arc / arc count per oriented vertex */
} orc_graph_t;

/* the graph reads are corrected against (oracle/ecgraph.c): make_syncmer_graph(…, 0, 0.) + arc overlaps */
typedef struct {
    uint64_t n_scm;
    const uint64_t *occ_off;      /* [n_scm + 1] */
    const uint64_t *occ;          /* sid << 32 | idx << 1 | rev, (sid, idx) ascending per syncmer */
} orc_count_view_t;

typedef struct {
    uint64_t n_vtx, n_arc;
    uint64_t *arc_v, *arc_w, *arc_ls;
    uint32_t *arc_cov;
    uint8_t *arc_comp;
    uint64_t *idx_p, *idx_n;
    int multi_arc;                /* duplicate (v, w): the reference's order is then unspecified (graph.c:252 "TODO fix multi-arc") */
} orc_ecgraph_t;

orc_ecgraph_t *orc_ecgraph_build(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos,
                                 const orc_count_view_t *c, int K);
void orc_ecgraph_free(orc_ecgraph_t *g);

/* the assembly graph of the (corrected) reads (oracle/asmgraph.c): make_syncmer_graph(…, min_k_cov, min_a_cov_f) + asmg_finalize(g, 1);
 * vertices are the surviving syncmers in order, scm_del is updated in place like syncasm.c:228 does */
typedef struct {
    uint64_t n_vtx, n_arc;
    uint32_t *vtx_scm, *vtx_cov;
    uint64_t *arc_v, *arc_w, *arc_link;
    uint32_t *arc_cov;
    uint8_t *arc_comp;
    uint64_t *idx_p, *idx_n;
    int multi_arc;
} orc_asmgraph_t;

orc_asmgraph_t *orc_asmgraph_build(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos,
                                   uint64_t n_syncmers, const uint32_t *scm_cov, uint8_t *scm_del, uint32_t min_k_cov, double min_a_cov_f);
void orc_asmgraph_free(orc_asmgraph_t *g);

/* read -> unitig alignment (oracle/align.c): what scg_read_alignment reads of scg_t, flattened, and what it leaves in scg_ra_v */
typedef struct {
    uint64_t n_scm, n_utg, n_arc;
    const uint64_t *su_off, *su_uid;      /* [n_scm + 1] into su_uid / su_pos; utg << 1 | strand */
    const uint32_t *su_pos, *utg_n;
    const uint64_t *idx_p, *idx_n, *arc_w, *arc_ln;
    const uint8_t *arc_del;
} orc_ra_graph_t;
typedef struct {
    uint64_t n_aln, n_frg, m_aln, m_frg, n_mapped, n_unique;
    uint64_t *sid; uint32_t *n; double *s;                                  /* per alignment */
    uint64_t *uid, *u_beg, *u_end; uint32_t *s_beg, *s_end;                 /* per fragment */
} orc_ra_out_t;
orc_ra_out_t *orc_read_alignment(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos, const orc_ra_graph_t *g,
                                 const int64_t *old_ra);
void orc_ra_out_free(orc_ra_out_t *o);

/* base-space consensus of a syncmer (oracle/consensus.c): flat view of the reads (per-read arrays concatenated in read order) */
typedef struct {
    uint64_t sid0;
    const uint64_t *scm_off;      /* [n_reads + 1] slots of the per-read chains                    */
    const uint64_t *k_mer;        /* id << 1 | corrected                                            */
    const uint32_t *m_pos;
    const uint64_t *hs_off;       /* [n_reads] byte offset of the read's hoco_s                    */
    const uint8_t *hoco_s;
    const uint64_t *rl_off;       /* [n_reads] offset of the read's ho_rl                          */
    const uint8_t *ho_rl;
    const uint64_t *lrl_off;      /* [n_reads] offset of the read's ho_l_rl                        */
    const uint32_t *ho_l_rl;
} orc_reads_view_t;

void orc_consensus_rl(const orc_reads_view_t *v, uint64_t n_occ, const uint64_t *occ, int K, uint64_t *tot_rl, uint32_t *m_seq, uint64_t *first_occ);
/* `out` must hold |beg<0| + (K - beg) * (1 + max run length) characters; returns the length the reference returns */
int64_t orc_consensus_string(const orc_reads_view_t *v, const uint64_t *tot_rl, uint32_t m_seq, uint64_t first_occ, int K, int rev, int64_t beg,
                             int hoco_seq, char *out);

int64_t orc_find_error_syncmers(const orc_graph_t *g, const uint32_t *scm_cov, uint8_t *scm_del, uint32_t err_mer_c,
                                uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f);

typedef struct {
    uint64_t tot, updated_reads;
    uint32_t *n_scm;              /* [n_reads] syncmers per read after correction    */
    uint64_t *k_mer;              /* [tot] id << 1 | corrected                       */
    uint32_t *m_pos;              /* [tot]                                           */
    uint64_t *s_mer;              /* [tot]                                           */
    long stats[11];               /* syncerr.c:76: tail {n, fail, ok, ambiseq?...}, middle {...}, overlapped */
} orc_ec_out_t;

void orc_ec_reads(const orc_graph_t *g, const uint8_t *scm_del, const uint64_t *scm_s, int K, double max_edist, uint64_t n_reads,
                  const uint32_t *hoco_l, const uint8_t *hoco_s, const uint64_t *hoco_byte_off, const uint32_t *n_scm,
                  const uint64_t *k_mer, const uint32_t *m_pos, const uint64_t *s_mer, orc_ec_out_t *out);
void orc_ec_out_free(orc_ec_out_t *o);
void orc_update_db(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t n_syncmers,
                   uint32_t *cov, uint8_t *del, uint64_t *occ_off, uint64_t *occ);

#ifdef __cplusplus
}
#endif
#endif
