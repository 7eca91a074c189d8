/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A flat, ctypes-friendly window onto the *compiled reference* (c-zhou/oatk,
 * sources read where they lie under /root/reference; nothing is copied into
 * this repo).  oracle/Makefile compiles this file together with the
 * reference's own .c files into oracle/_ref/liboatk_ref.so.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Every function here is a thin adaptor: it calls a public reference symbol
 * (cited) and copies struct members into caller-provided flat arrays so that
 * Python/numpy can compare them with the HIP path and with the oracle C files.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>

#include "sstream.h"
#include "syncmer.h"
#include "syncasm.h"
#include "graph.h"
#include "levdist.h"
#include "misc.h"

/* each reference main() defines this global (run_syncasm.c:327); the library has none */
int VERBOSE = 0;

/* declared at run_syncasm.c:53, defined syncerr.c:819 */
void read_error_correction(sr_db_t *sr_db, scg_t *g, double max_edist, uint32_t err_mer_c, uint32_t max_err_c,
        uint32_t err_arc_c, double max_arc_f, int threads, FILE *fo, int verbose);
/* defined syncerr.c:679 */
int64_t find_error_syncmers(scg_t *g, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f, int del_err);
/* defined run_syncasm.c:56 */
int syncasm(char **file_in, int n_file, size_t m_data, int k, int s, int bubble_size, int tip_size, int min_k_cov,
        double min_a_cov_f, double weak_cross, int do_ec, int do_unzip, int n_threads, char *out, scg_meta_t *meta, int VERBOSE);

void refx_set_verbose(int v) { VERBOSE = v; }

/* ---- scan: sstream_open (sstream.c:70) + sr_read (syncmer.c:487) ---- */
sr_db_t *refx_scan(char **files, int n_files, int k, int s, int n_threads)
{
    sstream_t *rdr = sstream_open(files, n_files);
    if (!rdr) return 0;
    sr_db_t *db = (sr_db_t *) malloc(sizeof(sr_db_t));
    sr_db_init(db, k, s);
    sr_read(rdr, db, 0, n_threads);
    sstream_close(rdr);
    return db;
}

/* the same with sr_read's data cap (run_syncasm.c:81: m_data) */
sr_db_t *refx_scan_cap(char **files, int n_files, int k, int s, int n_threads, size_t m_data)
{
    sstream_t *rdr = sstream_open(files, n_files);
    if (!rdr) return 0;
    sr_db_t *db = (sr_db_t *) malloc(sizeof(sr_db_t));
    sr_db_init(db, k, s);
    sr_read(rdr, db, m_data, n_threads);
    sstream_close(rdr);
    return db;
}

void refx_srdb_destroy(sr_db_t *db) { sr_db_destroy(db); }
uint64_t refx_srdb_n(sr_db_t *db) { return db->n; }
int refx_srdb_validate(sr_db_t *db) { return sr_db_validate(db); }

/* per-read scalar members: hoco_l, n (sr_t, syncmer.h:48-70) */
void refx_srdb_lengths(sr_db_t *db, uint32_t *hoco_l, uint32_t *n_scm, uint64_t *sid)
{
    size_t i;
    for (i = 0; i < db->n; ++i) {
        hoco_l[i] = db->a[i].hoco_l;
        n_scm[i] = db->a[i].n;
        sid[i] = db->a[i].sid;
    }
}

/* concatenate per-read arrays in read order; any destination may be NULL */
void refx_srdb_flatten(sr_db_t *db, uint8_t *hoco_s, uint8_t *ho_rl, uint32_t *ho_l_rl, uint64_t *n_l_rl,
        uint32_t *m_pos, uint64_t *s_mer, uint64_t *k_mer)
{
    size_t i, j, ob = 0, orl = 0, ol = 0, os = 0;
    for (i = 0; i < db->n; ++i) {
        sr_t *r = &db->a[i];
        size_t nb = ((size_t) r->hoco_l + 3) / 4, nl = 0;
        if (hoco_s && nb) memcpy(hoco_s + ob, r->hoco_s, nb);
        if (ho_rl && r->hoco_l) memcpy(ho_rl + orl, r->ho_rl, r->hoco_l);
        for (j = 0; j < r->hoco_l; ++j)
            if (r->ho_rl[j] == 255) {
                if (ho_l_rl) ho_l_rl[ol + nl] = r->ho_l_rl[nl];
                ++nl;
            }
        if (m_pos && r->n) memcpy(m_pos + os, r->m_pos, sizeof(uint32_t) * r->n);
        if (s_mer && r->n) memcpy(s_mer + os, r->s_mer, sizeof(uint64_t) * r->n);
        if (k_mer && r->n) memcpy(k_mer + os, r->k_mer, sizeof(uint64_t) * r->n);
        ob += nb, orl += r->hoco_l, ol += nl, os += r->n;
    }
    if (n_l_rl) *n_l_rl = ol;
}

/* sr_t does not store how many entries n_nucl holds (syncmer.c:321); the caller knows */
const char *refx_srdb_name(sr_db_t *db, uint64_t i) { return db->a[i].sname; }
void refx_srdb_nnucl(sr_db_t *db, uint64_t i, uint32_t cnt, uint32_t *out)
{
    if (cnt) memcpy(out, db->a[i].n_nucl, sizeof(uint32_t) * cnt);
}

void refx_srdb_stat(sr_db_t *db, int32_t *out8, double *outd)
{
    FILE *fo = fopen("/dev/null", "w");
    sr_db_stat(db, fo, 0); /* syncmer.c:867 */
    fclose(fo);
    sr_stat_t *st = db->stats;
    if (!st) return;
    out8[0] = st->smer_unique, out8[1] = st->smer_singleton, out8[2] = st->smer_peak_hom, out8[3] = st->smer_peak_het;
    out8[4] = st->kmer_unique, out8[5] = st->kmer_singleton, out8[6] = st->kmer_peak_hom, out8[7] = st->kmer_peak_het;
    outd[0] = (double) st->syncmer_n, outd[1] = st->syncmer_per_read, outd[2] = st->syncmer_avg_dist;
    outd[3] = st->smer_avg_cnt, outd[4] = st->kmer_avg_cnt;
}

/* ---- count: collect_syncmer_from_reads (syncmer.c:1397) ---- */
syncmer_db_t *refx_collect(sr_db_t *db) { return collect_syncmer_from_reads(db); }
void refx_scmdb_destroy(syncmer_db_t *s) { syncmer_db_destroy(s); }
uint64_t refx_scmdb_n(syncmer_db_t *s) { return s? s->n : 0; }
uint64_t refx_scmdb_total_cov(syncmer_db_t *s)
{
    uint64_t t = 0; size_t i;
    for (i = 0; i < s->n; ++i) t += s->a[i].cov;
    return t;
}
void refx_scmdb_flatten(syncmer_db_t *s, uint64_t *h, uint64_t *smer, uint32_t *cov, uint8_t *del, uint64_t *m_pos)
{
    size_t i, o = 0;
    for (i = 0; i < s->n; ++i) {
        syncmer_t *a = &s->a[i];
        if (h) h[i] = a->h;
        if (smer) smer[i] = a->s;
        if (cov) cov[i] = a->cov;
        if (del) del[i] = a->del;
        if (m_pos && a->cov) memcpy(m_pos + o, a->m_pos, sizeof(uint64_t) * a->cov);
        o += a->cov;
    }
}

/* ---- EC graph: make_syncmer_graph (syncasm.c:203), scg_consensus hoco (syncasm.c:716) ---- */
scg_t *refx_make_graph(sr_db_t *db, syncmer_db_t *s, uint32_t min_k_cov, double min_a_cov_f)
{
    return make_syncmer_graph(db, s, min_k_cov, min_a_cov_f);
}
void refx_consensus(sr_db_t *db, scg_t *g, int hoco, int save) { scg_consensus(db, g, hoco, save, 0); }
void refx_scg_destroy(scg_t *g) { scg_destroy(g); }
int64_t refx_find_error_syncmers(scg_t *g, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f, int del_err)
{
    return find_error_syncmers(g, err_mer_c, max_err_c, err_arc_c, max_arc_f, del_err);
}
void refx_graph_dims(scg_t *g, uint64_t *n_vtx, uint64_t *n_arc, uint64_t *seq_bytes)
{
    asmg_t *a = g->utg_asmg;
    uint64_t i, sb = 0;
    *n_vtx = a->n_vtx, *n_arc = a->n_arc;
    for (i = 0; i < a->n_vtx; ++i) if (a->vtx[i].seq) sb += a->vtx[i].len;
    *seq_bytes = sb;
}
/* asmg_t (graph.h:39-63) flattened: vertices (len, del, seq offset), arcs in array order, CSR index */
void refx_graph_flatten(scg_t *g, uint64_t *vtx_len, uint8_t *vtx_del, uint32_t *vtx_cov, uint64_t *vtx_seq_off, char *seq,
        uint64_t *arc_v, uint64_t *arc_w, uint64_t *arc_ls, uint32_t *arc_cov, uint8_t *arc_del, uint8_t *arc_comp,
        uint64_t *idx_p, uint64_t *idx_n)
{
    asmg_t *a = g->utg_asmg;
    uint64_t i, so = 0;
    for (i = 0; i < a->n_vtx; ++i) {
        if (vtx_len) vtx_len[i] = a->vtx[i].len;
        if (vtx_del) vtx_del[i] = a->vtx[i].del;
        if (vtx_cov) vtx_cov[i] = a->vtx[i].cov;
        if (vtx_seq_off) vtx_seq_off[i] = so;
        if (a->vtx[i].seq) {
            if (seq) memcpy(seq + so, a->vtx[i].seq, a->vtx[i].len);
            so += a->vtx[i].len;
        }
    }
    for (i = 0; i < a->n_arc; ++i) {
        if (arc_v) arc_v[i] = a->arc[i].v;
        if (arc_w) arc_w[i] = a->arc[i].w;
        if (arc_ls) arc_ls[i] = a->arc[i].ls;
        if (arc_cov) arc_cov[i] = a->arc[i].cov;
        if (arc_del) arc_del[i] = a->arc[i].del;
        if (arc_comp) arc_comp[i] = a->arc[i].comp;
    }
    if (idx_p) memcpy(idx_p, a->idx_p, sizeof(uint64_t) * a->n_vtx * 2);
    if (idx_n) memcpy(idx_n, a->idx_n, sizeof(uint64_t) * a->n_vtx * 2);
}

/* the rest of asmg_t: per vertex the syncmer count and first syncmer (vtx.n, vtx.a[0]), per arc ln and link_id */
void refx_graph_flatten2(scg_t *g, uint64_t *vtx_n, uint64_t *vtx_a0, uint64_t *arc_ln, uint64_t *arc_link)
{
    asmg_t *a = g->utg_asmg;
    uint64_t i;
    for (i = 0; i < a->n_vtx; ++i) vtx_n[i] = a->vtx[i].n, vtx_a0[i] = a->vtx[i].n? a->vtx[i].a[0] : UINT64_MAX;
    for (i = 0; i < a->n_arc; ++i) arc_ln[i] = a->arc[i].ln, arc_link[i] = a->arc[i].link_id;
}

/* ---- EC: read_error_correction (syncerr.c:819) ---- */
void refx_ec(sr_db_t *db, scg_t *g, double max_edist, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c,
        double max_arc_f, int n_threads)
{
    read_error_correction(db, g, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, n_threads, 0, 0);
}

/* ---- edit distance: wf_ed (levdist.c:312), resumable wf_ed_core (levdist.c:265) ---- */
void refx_wf_ed(int32_t tl, const char *ts, int32_t ql, const char *qs, int32_t is_ext, int32_t bw, int32_t *out3)
{
    int32_t score = 0, t_end = 0, q_end = 0;
    wf_ed(tl, ts, ql, qs, is_ext, bw, &score, &t_end, &q_end, 0);
    out3[0] = score, out3[1] = t_end, out3[2] = q_end;
}

/* the state initialisation mirrors the caller at syncerr.c:465-482 */
wf_config_t *refx_wf_new(int32_t tl, const char *ts, int32_t bw)
{
    wf_config_t *c = (wf_config_t *) calloc(1, sizeof(wf_config_t));
    c->ts = (char *) malloc(tl + 8);
    memcpy(c->ts, ts, tl);
    c->tl = tl;
    c->is_ext = 1;
    c->bw = bw;
    c->wf_diag = (wf_diag_t *) calloc(1, sizeof(wf_diag_t));
    c->wf_diag->m = (size_t) tl * 4 + 16;
    c->wf_diag->a = (wf_diag1_t *) malloc(sizeof(wf_diag1_t) * c->wf_diag->m);
    c->wf_diag->n = 1;
    c->wf_diag->a[0].d = 0;
    c->wf_diag->a[0].k = -1;
    return c;
}
/* advance the same state with a longer query (qs must extend the previous one) */
void refx_wf_step(wf_config_t *c, int32_t ql, char *qs, int32_t *out3)
{
    c->qs = qs, c->ql = ql;
    wf_ed_core(c);
    out3[0] = c->score, out3[1] = c->t_end, out3[2] = c->q_end;
}
void refx_wf_free(wf_config_t *c) { c->qs = 0; wf_config_destroy(c, 1); }

/* ---- whole program: syncasm() (run_syncasm.c:56) ---- */
int refx_syncasm(char **files, int n_files, int k, int s, int min_k_cov, double min_a_cov_f, int do_ec, int do_unzip,
        int n_threads, char *out)
{
    /* remaining defaults from run_syncasm.c:356-367 */
    return syncasm(files, n_files, 0, k, s, 100000, 10000, min_k_cov, min_a_cov_f, 0.3, do_ec, do_unzip, n_threads, out, 0, 0);
}

/* ---- the rest of syncasm() after the scan and the count (run_syncasm.c:107-319), on caller-provided databases ----
 * Lets a test feed sr_db / scm_db built by ANOTHER implementation (this repo's device path) into the reference's own
 * graph construction, error correction, cleaning, unzipping and GFA output, and compare the GFA bytes.  Only calls
 * public reference functions, in the order run_syncasm.c does. */
static int syncasm_tail(sr_db_t *sr_db, syncmer_db_t *scm_db, asmg_t *given_asmg, int k, int bubble_size, int tip_size, int min_k_cov,
        double min_a_cov_f, double weak_cross, int do_ec, int do_unzip, int n_threads, const char *out);
/* who aligns the reads in the tail: the reference's scg_read_alignment, or a routine of the same signature supplied by a test */
typedef void (*refx_aligner_t)(sr_db_t *, scg_ra_v *, scg_t *, int, int);
static refx_aligner_t tail_aligner = 0;
void refx_set_aligner(refx_aligner_t f) { tail_aligner = f; }
#define TAIL_ALIGN(db, v, g, t, u) do { if (tail_aligner) tail_aligner(db, v, g, t, u); else scg_read_alignment(db, v, g, t, u); } while (0)
int refx_syncasm_tail(sr_db_t *sr_db, syncmer_db_t *scm_db, int k, int bubble_size, int tip_size, int min_k_cov,
        double min_a_cov_f, double weak_cross, int do_ec, int do_unzip, int n_threads, const char *out)
{
    return syncasm_tail(sr_db, scm_db, 0, k, bubble_size, tip_size, min_k_cov, min_a_cov_f, weak_cross, do_ec, do_unzip, n_threads, out);
}
/* the same from run_syncasm.c:160 on, with the assembly graph of :138 (scg->utg_asmg) supplied by the caller; takes ownership of it */
int refx_syncasm_tail_graph(sr_db_t *sr_db, syncmer_db_t *scm_db, asmg_t *asmg, int k, int bubble_size, int tip_size, int min_k_cov,
        double min_a_cov_f, double weak_cross, int do_unzip, int n_threads, const char *out)
{
    return syncasm_tail(sr_db, scm_db, asmg, k, bubble_size, tip_size, min_k_cov, min_a_cov_f, weak_cross, 0, do_unzip, n_threads, out);
}
static int syncasm_tail(sr_db_t *sr_db, syncmer_db_t *scm_db, asmg_t *given_asmg, int k, int bubble_size, int tip_size, int min_k_cov,
        double min_a_cov_f, double weak_cross, int do_ec, int do_unzip, int n_threads, const char *out)
{
    scg_t *scg = 0;
    scg_ra_v *ra_db = 0;
    FILE *fo, *nul = fopen("/dev/null", "w");
    char path[4096];
    int ret = 0;
    if (do_ec) {                                                   /* run_syncasm.c:107-133 */
        scg = make_syncmer_graph(sr_db, scm_db, 0, 0.);
        scg_consensus(sr_db, scg, 1, 1, 0);
        read_error_correction(sr_db, scg, 0.02, min_k_cov, min_k_cov * 10, min_k_cov, min_a_cov_f, n_threads, 0, 0);
        sr_db_stat(sr_db, nul, 0);
        scg_destroy(scg); scg = 0;
    }
    if (given_asmg) {                                              /* the assembly graph was built elsewhere: wrap it the way syncasm.c:208-296 does */
        scg = (scg_t *) calloc(1, sizeof(scg_t));
        scg->scm_db = scm_db, scg->utg_asmg = given_asmg;          /* the syncmer index is rebuilt by process_mergeable_unitigs below */
    } else scg = make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f);  /* :138 */
    if (!scg || scg_is_empty(scg)) { ret = 1; goto done; }
    process_mergeable_unitigs(scg);                                /* :161 */
    snprintf(path, sizeof(path), "%s.utg.gfa", out);
    fo = fopen(path, "w"); scg_consensus(sr_db, scg, 0, 0, fo); fclose(fo);
    {
        uint64_t cleaned = 1;                                      /* :183-193 */
        while (cleaned) {
            cleaned = 0;
            if (do_unzip <= 0) {
                cleaned += asmg_pop_bubble(scg->utg_asmg, bubble_size, 0, 0, 1, 0, 0);
                cleaned += asmg_remove_weak_crosslink(scg->utg_asmg, weak_cross, 10, 0, 0);
            }
            cleaned += asmg_drop_tip(scg->utg_asmg, INT32_MAX, tip_size, 1, 0, 0);
        }
        process_mergeable_unitigs(scg);
    }
    ra_db = (scg_ra_v *) calloc(1, sizeof(scg_ra_v));
    if (do_unzip > 0) {                                            /* :209-292 */
        int round = 0, updated = 1;
        uint32_t max_n_scm = (uint32_t) ceil(30000.0 / k);
        while (updated != 0 && round < do_unzip) {
            ++round;
            TAIL_ALIGN(sr_db, ra_db, scg, n_threads, 1);
            scg_update_utg_cov(scg);
            updated = scg_multiplex(scg, ra_db, max_n_scm, 10, .3);
        }
        TAIL_ALIGN(sr_db, ra_db, scg, n_threads, 1);
        scg_ra_arc_coverage(scg, sr_db, ra_db, 0, 0);
        asmg_remove_weak_crosslink(scg->utg_asmg, weak_cross, 10, 0, 0);
        scg_demultiplex(scg);
        TAIL_ALIGN(sr_db, ra_db, scg, n_threads, 0);
        scg_ra_utg_coverage(scg, sr_db, ra_db, 0);
        scg_ra_arc_coverage(scg, sr_db, ra_db, 1, 0);
        scg_consensus(sr_db, scg, 0, 0, 0);
        {
            uint64_t cleaned = 1;
            while (cleaned) {
                cleaned = 0;
                cleaned += asmg_pop_bubble(scg->utg_asmg, bubble_size, 0, 0, 1, 0, 0);
                cleaned += asmg_remove_weak_crosslink(scg->utg_asmg, weak_cross, 10, 0, 0);
                cleaned += asmg_drop_tip(scg->utg_asmg, INT32_MAX, tip_size, 1, 0, 0);
            }
        }
        process_mergeable_unitigs(scg);
    }
    TAIL_ALIGN(sr_db, ra_db, scg, n_threads, 0);           /* :295-303 */
    scg_ra_utg_coverage(scg, sr_db, ra_db, 0);
    scg_ra_arc_coverage(scg, sr_db, ra_db, 1, 0);
    snprintf(path, sizeof(path), "%s.utg.final.gfa", out);
    fo = fopen(path, "w"); scg_consensus(sr_db, scg, 0, 0, fo); fclose(fo);
done:
    fclose(nul);
    scg_destroy(scg);
    scg_ra_v_destroy(ra_db);
    return ret;
}

/* ---- unitigs and their consensus (syncasm.c:1048-1061, :1004-1046) ---- */
void refx_process_unitigs(scg_t *g) { process_mergeable_unitigs(g); }
/* per vertex of the (unitig) graph: number of syncmers, deleted flag; `a` receives the oriented syncmer lists back to back (may be NULL) */
uint64_t refx_utg_lists(scg_t *g, uint64_t *n, uint8_t *del, uint64_t *a)
{
    asmg_t *u = g->utg_asmg;
    uint64_t i, tot = 0;
    for (i = 0; i < u->n_vtx; ++i) {
        if (n) n[i] = u->vtx[i].n;
        if (del) del[i] = u->vtx[i].del;
        if (a) memcpy(a + tot, u->vtx[i].a, sizeof(uint64_t) * u->vtx[i].n);
        tot += u->vtx[i].n;
    }
    return tot;
}
/* scg_unitig_consensus of the syncmer list v[0..n): returns the length, the string in out (up to cap bytes) */
int64_t refx_unitig_consensus(sr_db_t *db, scg_t *g, uint64_t *v, uint64_t n, int hoco_seq, char *out, int64_t cap)
{
    kstring_t c = {0, 0, 0};
    int64_t l = scg_unitig_consensus(db, v, n, scg_a_scm(g), &c, hoco_seq);
    if (out && l > 0) memcpy(out, c.s, l < cap? l : cap);
    free(c.s);
    return l;
}

/* ---- read -> unitig alignment (alignment.c:596-691) ---- */
scg_ra_v *refx_ra_new(void) { return (scg_ra_v *) calloc(1, sizeof(scg_ra_v)); }
void refx_ra_destroy(scg_ra_v *v) { scg_ra_v_destroy(v); }
void refx_read_alignment(sr_db_t *db, scg_ra_v *v, scg_t *g, int n_threads, int for_unzip) { scg_read_alignment(db, v, g, n_threads, for_unzip); }
void refx_ra_dims(scg_ra_v *v, uint64_t *n_aln, uint64_t *n_frg)
{
    uint64_t i, f = 0;
    for (i = 0; i < v->n; ++i) f += v->a[i].n;
    *n_aln = v->n, *n_frg = f;
}
void refx_ra_flatten(scg_ra_v *v, uint64_t *sid, uint32_t *n, double *s, uint64_t *uid, uint64_t *u_beg, uint64_t *u_end, uint32_t *s_beg, uint32_t *s_end)
{
    uint64_t i, j, f = 0;
    for (i = 0; i < v->n; ++i) {
        sid[i] = v->a[i].sid, n[i] = v->a[i].n, s[i] = v->a[i].s;
        for (j = 0; j < v->a[i].n; ++j, ++f)
            uid[f] = v->a[i].a[j].uid, u_beg[f] = v->a[i].a[j].u_beg, u_end[f] = v->a[i].a[j].u_end, s_beg[f] = v->a[i].a[j].s_beg, s_end[f] = v->a[i].a[j].s_end;
    }
}
/* the inverse of refx_ra_flatten */
scg_ra_v *refx_ra_build(uint64_t n_aln, uint64_t *sid, uint32_t *n, double *s, uint64_t *uid, uint64_t *u_beg, uint64_t *u_end, uint32_t *s_beg, uint32_t *s_end)
{
    scg_ra_v *v = (scg_ra_v *) calloc(1, sizeof(scg_ra_v));
    uint64_t i, j, f = 0;
    v->n = v->m = n_aln;
    v->a = (scg_ra_t *) calloc(n_aln? n_aln : 1, sizeof(scg_ra_t));
    for (i = 0; i < n_aln; ++i) {
        v->a[i].sid = sid[i], v->a[i].n = n[i], v->a[i].s = s[i];
        v->a[i].a = (ra_frg_t *) calloc(n[i]? n[i] : 1, sizeof(ra_frg_t));
        for (j = 0; j < n[i]; ++j, ++f)
            v->a[i].a[j].uid = uid[f], v->a[i].a[j].u_beg = u_beg[f], v->a[i].a[j].u_end = u_end[f], v->a[i].a[j].s_beg = s_beg[f], v->a[i].a[j].s_end = s_end[f];
    }
    return v;
}
/* what scg_ra_analysis_thread reads of scg_t, as flat arrays (tests feed them to the oracle and to the device) */
void refx_ra_graph_dims(scg_t *g, uint64_t *n_scm, uint64_t *n_su, uint64_t *n_utg, uint64_t *n_arc)
{
    *n_scm = scg_n_scm(g), *n_su = (uint64_t) (g->idx_u[scg_n_scm(g)] - g->idx_u[0]), *n_utg = g->utg_asmg->n_vtx, *n_arc = g->utg_asmg->n_arc;
}
void refx_ra_graph_flatten(scg_t *g, uint64_t *su_off, uint64_t *su_uid, uint32_t *su_pos, uint32_t *utg_n, uint64_t *idx_p, uint64_t *idx_n,
        uint64_t *arc_w, uint64_t *arc_ln, uint8_t *arc_del)
{
    asmg_t *u = g->utg_asmg;
    uint64_t i, ns = scg_n_scm(g), nsu = (uint64_t) (g->idx_u[ns] - g->idx_u[0]);
    for (i = 0; i <= ns; ++i) su_off[i] = (uint64_t) (g->idx_u[i] - g->idx_u[0]);
    for (i = 0; i < nsu; ++i) {
        uint128_t x = g->idx_u[0][i];
        su_uid[i] = scm_utg_uid(x) << 1 | scm_utg_rev(x), su_pos[i] = (uint32_t) scm_utg_pos(x);
    }
    for (i = 0; i < u->n_vtx; ++i) utg_n[i] = (uint32_t) u->vtx[i].n;
    memcpy(idx_p, u->idx_p, 8 * 2 * u->n_vtx); memcpy(idx_n, u->idx_n, 8 * 2 * u->n_vtx);
    for (i = 0; i < u->n_arc; ++i) arc_w[i] = u->arc[i].w, arc_ln[i] = u->arc[i].ln, arc_del[i] = u->arc[i].del;
}
/* an scg_t holding just what scg_read_alignment reads, from flat arrays (the inverse of refx_ra_graph_flatten): lets a test align against
 * graphs no assembler would build.  Free with refx_scg_flat_destroy. */
scg_t *refx_scg_from_flat(syncmer_db_t *scm_db, uint64_t n_utg, uint64_t n_arc, const uint64_t *su_off, const uint64_t *su_uid, const uint32_t *su_pos,
        const uint32_t *utg_n, const uint64_t *idx_p, const uint64_t *idx_n, const uint64_t *arc_v, const uint64_t *arc_w, const uint64_t *arc_ln, const uint8_t *arc_del)
{
    scg_t *g = (scg_t *) calloc(1, sizeof(scg_t));
    asmg_t *u = (asmg_t *) calloc(1, sizeof(asmg_t));
    uint64_t i, s, ns = scm_db->n, nsu = su_off[ns];
    g->scm_db = scm_db, g->utg_asmg = u;
    u->n_vtx = u->m_vtx = n_utg, u->n_arc = u->m_arc = n_arc;
    u->vtx = (asmg_vtx_t *) calloc(n_utg? n_utg : 1, sizeof(asmg_vtx_t));
    u->arc = (asmg_arc_t *) calloc(n_arc? n_arc : 1, sizeof(asmg_arc_t));
    u->idx_p = (uint64_t *) calloc(2 * n_utg + 1, 8), u->idx_n = (uint64_t *) calloc(2 * n_utg + 1, 8);
    for (i = 0; i < n_utg; ++i) u->vtx[i].n = utg_n[i];
    for (i = 0; i < n_arc; ++i) u->arc[i].v = arc_v[i], u->arc[i].w = arc_w[i], u->arc[i].ln = arc_ln[i], u->arc[i].del = arc_del[i];
    memcpy(u->idx_p, idx_p, 8 * 2 * n_utg); memcpy(u->idx_n, idx_n, 8 * 2 * n_utg);
    g->scm_u = (uint128_t *) calloc(nsu? nsu : 1, sizeof(uint128_t));
    g->idx_u = (uint128_t **) calloc(ns + 1, sizeof(uint128_t *));
    for (s = 0; s <= ns; ++s) g->idx_u[s] = g->scm_u + su_off[s];
    for (s = 0; s < ns; ++s)
        for (i = su_off[s]; i < su_off[s + 1]; ++i)       /* syncasm.c:142: scm << 78 | utg << 36 | pos, scm = id << 1 | rev */
            g->scm_u[i] = ((uint128_t) (s << 1 | (su_uid[i] & 1)) << 78) | ((uint128_t) (su_uid[i] >> 1) << 36) | su_pos[i];
    return g;
}
void refx_scg_flat_destroy(scg_t *g)
{
    free(g->utg_asmg->vtx); free(g->utg_asmg->arc); free(g->utg_asmg->idx_p); free(g->utg_asmg->idx_n); free(g->utg_asmg);
    free(g->scm_u); free(g->idx_u); free(g);
}
/* graph surgery between alignment rounds, as run_syncasm.c:209-232 does it */
void refx_update_utg_cov(scg_t *g) { scg_update_utg_cov(g); }
int refx_multiplex(scg_t *g, scg_ra_v *v, uint32_t max_n_scm, double min_n_r, double min_d_f) { return scg_multiplex(g, v, max_n_scm, min_n_r, min_d_f); }

/* ---- base-space consensus of one syncmer (syncasm.c:888-1003) ---- */
int64_t refx_syncmer_consensus(sr_db_t *db, syncmer_db_t *s, uint64_t id, int rev, int64_t beg, int hoco_seq, char *out, int64_t cap)
{
    kstring_t ks = {0, 0, 0};
    int64_t l = scg_syncmer_consensus(db, &s->a[id], rev, beg, &ks, hoco_seq);
    int64_t n = (int64_t) ks.l < cap? (int64_t) ks.l : cap;
    if (n > 0) memcpy(out, ks.s, (size_t) n);
    free(ks.s);
    return l;
}

/* ---- databases made by hand: just the members make_syncmer_graph reads (sr_t.n / k_mer / m_pos, syncmer_t.cov / del), so that a test can put
 * chains in front of the reference that no genome would produce.  Free with refx_fake_dbs_free. ---- */
sr_db_t *refx_fake_srdb(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos)
{
    sr_db_t *db = (sr_db_t *) calloc(1, sizeof(sr_db_t));
    uint64_t i, o = 0;
    db->n = db->m = n_reads;
    db->a = (sr_t *) calloc(n_reads? n_reads : 1, sizeof(sr_t));
    for (i = 0; i < n_reads; o += n_scm[i], ++i) {
        sr_t *r = &db->a[i];
        r->sid = i, r->n = n_scm[i];
        r->k_mer = (uint64_t *) malloc(8 * (n_scm[i] + 1)); memcpy(r->k_mer, k_mer + o, 8 * n_scm[i]);
        r->m_pos = (uint32_t *) malloc(4 * (n_scm[i] + 1)); memcpy(r->m_pos, m_pos + o, 4 * n_scm[i]);
    }
    return db;
}
/* the same with s-mers attached: what sr_db_stat (syncmer.c:867) reads */
void refx_fake_srdb_smer(sr_db_t *db, const uint64_t *s_mer)
{
    uint64_t i, o = 0;
    for (i = 0; i < db->n; o += db->a[i].n, ++i) {
        sr_t *r = &db->a[i];
        r->s_mer = (uint64_t *) malloc(8 * (r->n + 1)); memcpy(r->s_mer, s_mer + o, 8 * r->n);
    }
}
syncmer_db_t *refx_fake_scmdb(uint64_t n, const uint32_t *cov, const uint8_t *del)
{
    syncmer_db_t *s = (syncmer_db_t *) calloc(1, sizeof(syncmer_db_t));
    uint64_t i;
    s->n = s->m = n;
    s->a = (syncmer_t *) calloc(n? n : 1, sizeof(syncmer_t));
    for (i = 0; i < n; ++i) s->a[i].cov = cov[i], s->a[i].del = del[i];
    return s;
}
void refx_fake_scmdb_del(syncmer_db_t *s, uint8_t *del) { uint64_t i; for (i = 0; i < s->n; ++i) del[i] = s->a[i].del; }
void refx_fake_dbs_free(sr_db_t *db, syncmer_db_t *s)
{
    uint64_t i;
    if (db) { for (i = 0; i < db->n; ++i) { free(db->a[i].k_mer); free(db->a[i].m_pos); free(db->a[i].s_mer); } free(db->stats); free(db->a); free(db); }
    if (s) { free(s->a); free(s); }
}

/* ---- the reference's own hash table (khashl.h, instantiated exactly like syncasm.c:63 does) fed a raw sequence of add_ovl_count keys: what
 * calc_syncmer_overlap's table looks like after a walk, and the distance it then picks (the selection loop of syncasm.c:558-571 in five
 * lines).  `h` persists across calls like the table scg_unitig_consensus hands down (cleared, size kept); NULL = a fresh one. ---- */
#include "khashl.h"
KHASHL_MAP_INIT(KH_LOCAL, refx_kh_t, refx_kh, int, int, kh_hash_dummy, kh_eq_generic)
void *refx_kh_new(void) { return refx_kh_init(); }
void refx_kh_free(void *h) { refx_kh_destroy((refx_kh_t *) h); }
int refx_kh_mode(void *hm, const int *seq, uint64_t n)
{
    refx_kh_t *h = hm? (refx_kh_t *) hm : refx_kh_init();
    uint64_t i;
    int absent, movl = 0, mcnt = 0;
    khint_t k;
    refx_kh_m_clear(h);
    for (i = 0; i < n; ++i) {
        k = refx_kh_put(h, seq[i], &absent);
        if (absent) kh_val(h, k) = 1; else ++kh_val(h, k);
    }
    for (k = 0; k < kh_end(h); ++k) if (kh_exist(h, k) && kh_val(h, k) > mcnt) mcnt = kh_val(h, k), movl = kh_key(h, k);
    if (!hm) refx_kh_destroy(h);
    return movl;
}

#ifdef REFX_HOOKED
/* ---- the hooked build (make ref_hooked, oracle/ref_hooks.h): the glue a maintainer would write around the host library's adaptors ---- */
#include "ref_hooks.h"
int64_t (*refx_hook_cons)(void *, void *, int, int64_t, void *, int) = 0;
int (*refx_hook_ovl)(void *, uint64_t, void *, uint64_t, const int32_t **, const uint32_t **, int *) = 0;

typedef int64_t (*cons_fn_t)(const void *cs, const void *sr_db, uint64_t scm_id, int rev, int64_t beg, void *c_seq, int hoco_seq);   /* oatk_scg_syncmer_consensus */
typedef int (*ovl_fn_t)(const void *ovl, uint64_t v, uint64_t w, const int32_t **dist, const uint32_t **cnt, int *tail);              /* oatk_overlap_lookup */
static cons_fn_t g_cons_fn; static const void *g_cons; static ovl_fn_t g_ovl_fn; static const void *g_ovl; static syncmer_t *g_base;
static uint64_t g_served[4];                         /* consensus served / declined, overlaps served / declined */

static int64_t cons_trampoline(void *sr_db, void *scm, int rev, int64_t beg, void *c_seq, int hoco_seq)
{
    int64_t l = g_cons_fn(g_cons, sr_db, (uint64_t) ((syncmer_t *) scm - g_base), rev, beg, c_seq, hoco_seq);
    ++g_served[l >= 0? 0 : 1];
    return l;
}
static int ovl_trampoline(void *m1, uint64_t rc1, void *m2, uint64_t rc2, const int32_t **dist, const uint32_t **cnt, int *tail)
{
    int n = g_ovl_fn(g_ovl, (uint64_t) ((syncmer_t *) m1 - g_base) << 1 | rc1, (uint64_t) ((syncmer_t *) m2 - g_base) << 1 | rc2, dist, cnt, tail);
    ++g_served[2];
    return n;
}
/* scm_db: the table whose entries the hooks will be asked about; NULL function pointers take a hook out again */
void refx_hooks_install(syncmer_db_t *scm_db, void *cons_fn, const void *cons, void *ovl_fn, const void *ovl)
{
    g_base = scm_db? scm_db->a : 0;
    g_cons_fn = (cons_fn_t) cons_fn, g_cons = cons, g_ovl_fn = (ovl_fn_t) ovl_fn, g_ovl = ovl;
    refx_hook_cons = cons_fn? cons_trampoline : 0;
    refx_hook_ovl = ovl_fn? ovl_trampoline : 0;
    memset(g_served, 0, sizeof(g_served));
}
void refx_hooks_served(uint64_t *out4) { memcpy(out4, g_served, sizeof(g_served)); }
#endif
