/*
 * oatk_amd/csrc/host/multi_host.c -- the reference's hot-path entry points over several MI355X (include/oatk_multi.h).
 *
 * One device handle per GPU, the reads spread over them by position in the input.  Each entry point is the one-handle adaptor of this directory
 * with its device step replaced by the collective of include/oatk_hip_multi.h -- run on one host thread per handle, because every rank has to be
 * inside the collective at the same time -- and its second half (device arrays -> the reference's structs, host_internal.h) unchanged: after the
 * table merge and after the correction the whole syncmer table is gathered on handle 0 (oatk_hip_gather_table), graph, consensus sums and distance
 * tables are built identically on every handle and read from handle 0, and what is per read (ids, corrected chains, alignments) comes from the
 * handle that holds the read, in handle order = read order.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_cons.h"
#include "oatk_hip_ec.h"
#include "oatk_hip_graph.h"
#include "oatk_multi.h"
#include "host_internal.h"

struct oatk_multi {
    int n;
    oatk_hip_ctx *ctx[64];
    oatk_comm *comm[64];
    oatk_comm_group *grp;          /* ranks that share a device talk through the in-process group */
    uint64_t first[65];            /* handle r holds reads [first[r], first[r + 1]) */
    int have_reads;
    char err[512];
};

/* ---- one host thread per handle ---- */
typedef struct { pthread_mutex_t mu; pthread_cond_t cv; int open, go; } rank_gate_t;
typedef struct { const uint8_t *id; int rank, n, dev; oatk_comm *c; rank_gate_t *gate; } mk_comm_t;

/* (test hook: OATK_DEBUG_FAIL_THREAD=r makes the r-th thread of the next fan-out fail to start, as pthread_create does when the process is out of threads) */
static int start_thread(pthread_t *th, void *(*fn)(void *), void *arg, int r)
{
    const char *ev = getenv("OATK_DEBUG_FAIL_THREAD");
    if (ev && *ev && atoi(ev) == r) return -1;
    return pthread_create(th, 0, fn, arg);
}

static int gate_wait(rank_gate_t *g)
{
    pthread_mutex_lock(&g->mu);
    while (!g->open) pthread_cond_wait(&g->cv, &g->mu);
    const int go = g->go;
    pthread_mutex_unlock(&g->mu);
    return go;
}

static void gate_open(rank_gate_t *g, int go)
{
    pthread_mutex_lock(&g->mu);
    g->open = 1, g->go = go;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}

static void *mk_comm_entry(void *p)
{
    mk_comm_t *k = (mk_comm_t *) p;
    /* ncclCommInitRank returns when every rank has called it: nobody calls it before every rank has a thread to call it on */
    if (gate_wait(k->gate)) k->c = oatk_comm_create(k->id, k->rank, k->n, k->dev);
    return 0;
}

static void *xmalloc(size_t n)
{
    void *p = malloc(n? n : 1);
    if (!p) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    return p;
}

typedef int (*rank_fn)(oatk_multi *m, int rank, void *arg);
typedef struct { oatk_multi *m; int rank; rank_fn fn; void *arg; int rc; rank_gate_t *gate; } rank_job_t;

static void *rank_entry(void *p)
{
    rank_job_t *j = (rank_job_t *) p;
    /* nobody enters a collective before every rank has a thread to enter it on */
    j->rc = gate_wait(j->gate)? j->fn(j->m, j->rank, j->arg) : OATK_E_STATE;
    return 0;
}

/* fn on every rank at once; the first failing rank's code and message (a rank that fails inside a collective releases the others: they come back
 * with an error of their own, include/oatk_hip_multi.h) */
static int run_ranks(oatk_multi *m, rank_fn fn, void *arg)
{
    pthread_t th[64];
    rank_job_t job[64];
    rank_gate_t gate;
    int r, rc = OATK_OK, started[64], all = 1;
    pthread_mutex_init(&gate.mu, 0), pthread_cond_init(&gate.cv, 0), gate.open = 0, gate.go = 0;
    for (r = 0; r < m->n; ++r) {
        job[r].m = m, job[r].rank = r, job[r].fn = fn, job[r].arg = arg, job[r].rc = OATK_OK, job[r].gate = &gate;
        started[r] = r > 0 && start_thread(&th[r], rank_entry, &job[r], r) == 0;
        if (r > 0 && !started[r]) all = 0;
    }
    /* (a rank without a thread cannot be run after the others: they would wait for it inside the collective.  The threads that did start are
     * sent home and the call fails as a whole) */
    gate_open(&gate, all);
    rank_entry(&job[0]);
    for (r = 1; r < m->n; ++r) if (started[r]) pthread_join(th[r], 0);
    pthread_mutex_destroy(&gate.mu), pthread_cond_destroy(&gate.cv);
    if (!all) {
        snprintf(m->err, sizeof(m->err), "could not start one host thread per handle (%d handles)", m->n);
        return OATK_E_STATE;
    }
    for (r = 0; r < m->n; ++r)
        if (job[r].rc != OATK_OK && (rc == OATK_OK || (rc == OATK_E_STATE && job[r].rc != OATK_E_STATE))) {
            rc = job[r].rc;                              /* prefer the cause over the peers' "somebody failed" */
            snprintf(m->err, sizeof(m->err), "handle %d: %s", r, oatk_hip_last_error(m->ctx[r]));
        }
    return rc;
}

oatk_multi *oatk_multi_create(const int *devices, int n)
{
    int r, q, distinct = 1;
    if (!devices || n < 1 || n > 64) return 0;
    for (r = 0; r < n; ++r) for (q = 0; q < r; ++q) if (devices[q] == devices[r]) distinct = 0;
    oatk_multi *m = (oatk_multi *) calloc(1, sizeof(oatk_multi));
    if (!m) return 0;
    m->n = n;
    for (r = 0; r < n; ++r) if (!(m->ctx[r] = oatk_hip_create(devices[r]))) { oatk_multi_destroy(m); return 0; }
    if (distinct && n > 1) {
        uint8_t id[128];
        if (oatk_comm_unique_id(id) != OATK_OK) { oatk_multi_destroy(m); return 0; }
        /* ncclCommInitRank blocks until every rank has called it: one thread per rank */
        mk_comm_t mk[64];
        pthread_t th[64];
        rank_gate_t gate;
        int started[64], all = 1;
        pthread_mutex_init(&gate.mu, 0), pthread_cond_init(&gate.cv, 0), gate.open = 0, gate.go = 0;
        for (r = 0; r < n; ++r) { mk[r].id = id, mk[r].rank = r, mk[r].n = n, mk[r].dev = devices[r], mk[r].c = 0, mk[r].gate = &gate; }
        for (r = 1; r < n; ++r) { started[r] = start_thread(&th[r], mk_comm_entry, &mk[r], r) == 0; if (!started[r]) all = 0; }
        /* (a rank without a thread: the threads that did start are sent home before any of them is inside ncclCommInitRank, joined, and the call fails as a whole) */
        gate_open(&gate, all);
        mk_comm_entry(&mk[0]);
        for (r = 1; r < n; ++r) if (started[r]) pthread_join(th[r], 0);
        pthread_mutex_destroy(&gate.mu), pthread_cond_destroy(&gate.cv);
        for (r = 0; r < n; ++r) m->comm[r] = mk[r].c;
        for (r = 0; r < n; ++r) if (!m->comm[r]) { oatk_multi_destroy(m); return 0; }
    } else {
        if (!(m->grp = oatk_comm_group_create(n))) { oatk_multi_destroy(m); return 0; }
        for (r = 0; r < n; ++r) if (!(m->comm[r] = oatk_comm_group_rank(m->grp, r))) { oatk_multi_destroy(m); return 0; }
    }
    return m;
}

void oatk_multi_destroy(oatk_multi *m)
{
    int r;
    if (!m) return;
    for (r = 0; r < m->n; ++r) if (m->comm[r]) oatk_comm_destroy(m->comm[r]);
    if (m->grp) oatk_comm_group_destroy(m->grp);
    for (r = 0; r < m->n; ++r) if (m->ctx[r]) oatk_hip_destroy(m->ctx[r]);
    free(m);
}

int oatk_multi_size(const oatk_multi *m) { return m? m->n : 0; }
oatk_hip_ctx *oatk_multi_ctx(oatk_multi *m, int rank) { return m && rank >= 0 && rank < m->n? m->ctx[rank] : 0; }
const char *oatk_multi_backend(const oatk_multi *m) { return m && m->n? oatk_comm_backend(m->comm[0]) : ""; }
const char *oatk_multi_last_error(oatk_multi *m) { return m? m->err : "no handles"; }
void oatk_multi_range(const oatk_multi *m, int rank, uint64_t *first, uint64_t *n)
{
    if (first) *first = m->first[rank];
    if (n) *n = m->first[rank + 1] - m->first[rank];
}

static void note_err(oatk_multi *m, int rank, int rc)
{
    if (rc) snprintf(m->err, sizeof(m->err), "handle %d: %s", rank, oatk_hip_last_error(m->ctx[rank]));
}

/* one resident buffer of one handle into host memory at dst (bytes checked against what the caller expects when want != ~0) */
static int fetch_into(oatk_hip_ctx *ctx, int which, void *dst, uint64_t want)
{
    const void *d = 0;
    uint64_t b = 0;
    int rc = oatk_hip_buffer(ctx, which, &d, &b);
    if (rc) return rc;
    if (want != UINT64_MAX && b != want) return OATK_E_STATE;
    return b? oatk_hip_d2h(ctx, dst, d, b) : OATK_OK;
}
static void *fetch(oatk_hip_ctx *ctx, int which, uint64_t *bytes, int *rc)
{
    const void *d = 0;
    *bytes = 0;
    *rc = oatk_hip_buffer(ctx, which, &d, bytes);
    if (*rc) return 0;
    void *h = xmalloc(*bytes);
    if (*bytes) *rc = oatk_hip_d2h(ctx, h, d, *bytes);
    if (*rc) { free(h); return 0; }
    return h;
}

/* ---------------------------------------------------------------- sr_read ---------------------------------------------------------------- */

int oatk_multi_sr_read_files(oatk_multi *m, oatk_sr_db_t *sr_db, char **files, int n_files) { return oatk_multi_sr_read_files_capped(m, sr_db, files, n_files, 0); }

int oatk_multi_sr_read_files_capped(oatk_multi *m, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t m_data)
{
    m->have_reads = 0;
    const int rc = oatk_host_sr_read_files_n(m->ctx, m->n, sr_db, files, n_files, m->first, m_data);
    if (rc) { int r; for (r = 0; r < m->n; ++r) if (oatk_hip_last_error(m->ctx[r])[0]) { note_err(m, r, rc); break; } }
    else m->have_reads = 1;
    return rc;
}

/* --------------------------------------------------------------- sr_db_stat -------------------------------------------------------------- */

static int stat_rank(oatk_multi *m, int rank, void *arg)
{
    oatk_stat_raw_t *raw = (oatk_stat_raw_t *) arg;
    return oatk_hip_stat_sharded(m->ctx[rank], m->comm[rank], &raw[rank]);
}

int oatk_multi_sr_db_stat(oatk_multi *m, oatk_sr_db_t *sr_db, FILE *fo, int verbose)
{
    oatk_stat_raw_t *raw = (oatk_stat_raw_t *) calloc((size_t) m->n, sizeof(oatk_stat_raw_t));
    (void) verbose;
    int rc = run_ranks(m, stat_rank, raw);
    if (!rc) rc = oatk_host_stat_report(sr_db, &raw[0], fo);       /* (the same on every rank) */
    free(raw);
    return rc;
}

/* ------------------------------------------------------- collect_syncmer_from_reads ------------------------------------------------------ */

static int count_rank(oatk_multi *m, int rank, void *arg) { (void) arg; return oatk_hip_count(m->ctx[rank]); }
static int merge_rank(oatk_multi *m, int rank, void *arg)
{
    (void) arg;
    int rc = oatk_hip_merge_counts(m->ctx[rank], m->comm[rank], 0);
    if (!rc) rc = oatk_hip_gather_table(m->ctx[rank], m->comm[rank], 0);
    return rc;
}

/* the table that sits on handle 0 after oatk_hip_gather_table */
typedef struct { uint64_t n_scm, n_occ; uint64_t *h, *s, *occ_off, *occ; uint32_t *cov; uint8_t *del; } table_t;

static void table_free(table_t *t) { free(t->h); free(t->s); free(t->occ_off); free(t->occ); free(t->cov); free(t->del); memset(t, 0, sizeof(*t)); }

static int table_fetch(oatk_multi *m, table_t *t, int with_hs)
{
    uint64_t b;
    int rc = 0;
    memset(t, 0, sizeof(*t));
    t->cov = (uint32_t *) fetch(m->ctx[0], OATK_BUF_MG_G_COV, &b, &rc); if (rc) goto fail;
    t->n_scm = b / 4;
    t->del = (uint8_t *) fetch(m->ctx[0], OATK_BUF_MG_G_DEL, &b, &rc); if (rc) goto fail;
    t->occ_off = (uint64_t *) fetch(m->ctx[0], OATK_BUF_MG_G_OCC_OFF, &b, &rc); if (rc) goto fail;
    t->occ = (uint64_t *) fetch(m->ctx[0], OATK_BUF_MG_G_OCC, &b, &rc); if (rc) goto fail;
    t->n_occ = b / 8;
    if (with_hs) {
        t->h = (uint64_t *) fetch(m->ctx[0], OATK_BUF_MG_G_H, &b, &rc); if (rc) goto fail;
        t->s = (uint64_t *) fetch(m->ctx[0], OATK_BUF_MG_G_S, &b, &rc); if (rc) goto fail;
    }
    return OATK_OK;
fail:
    note_err(m, 0, rc);
    table_free(t);
    return rc;
}

oatk_syncmer_db_t *oatk_multi_collect_syncmer_from_reads(oatk_multi *m, oatk_sr_db_t *sr_db, int *rc_out)
{
    int rc, r;
    oatk_syncmer_db_t *db = 0;
    uint64_t *kid = 0;
    table_t t;
    memset(&t, 0, sizeof(t));
    if (!m->have_reads || m->first[m->n] != sr_db->n) { snprintf(m->err, sizeof(m->err), "the handles do not hold this database's reads"); rc = OATK_E_STATE; goto done; }
    /* the local counts first, all of them: a refusal of one (an oversized hash group) must not leave the others inside the merge */
    rc = run_ranks(m, count_rank, 0);
    if (rc == OATK_E_SMER) {                               /* fatal in the reference, syncmer.c:1370-1375 */
        fprintf(stderr, "[E::%s] identical kmers have different smers\n", "collect_syncmer_from_reads");
        exit(EXIT_FAILURE);
    }
    if (rc) goto done;
    {
        uint64_t n_occ = 0;
        for (r = 0; r < m->n; ++r) { oatk_hip_info_t inf; oatk_hip_info(m->ctx[r], &inf); n_occ += inf.n_occ; }
        if (n_occ == 0) { rc = OATK_OK; goto done; }       /* syncmer.c:1414-1417: NULL, no error */
    }
    rc = run_ranks(m, merge_rank, 0);
    if (rc == OATK_E_SMER) {
        fprintf(stderr, "[E::%s] identical kmers have different smers\n", "collect_syncmer_from_reads");
        exit(EXIT_FAILURE);
    }
    if (rc) goto done;
    if ((rc = table_fetch(m, &t, 1)) != OATK_OK) goto done;
    /* every read's ids from the handle that holds it: handle order is read order */
    kid = (uint64_t *) xmalloc(8 * (t.n_occ + 1));
    {
        uint64_t at = 0;
        for (r = 0; r < m->n && !rc; ++r) {
            oatk_hip_info_t inf;
            oatk_hip_info(m->ctx[r], &inf);
            if (inf.n_occ) rc = fetch_into(m->ctx[r], OATK_BUF_MG_POS_GKID, kid + at, inf.n_occ * 8);
            note_err(m, r, rc);
            at += inf.n_occ;
        }
        if (!rc && at != t.n_occ) { snprintf(m->err, sizeof(m->err), "the gathered table and the handles' chains disagree"); rc = OATK_E_STATE; }
    }
    if (!rc) db = oatk_host_build_syncmer_db(sr_db, t.n_scm, t.n_occ, t.h, t.s, t.cov, t.occ_off, &t.occ, kid);
done:
    free(kid);
    table_free(&t);
    if (rc_out) *rc_out = rc;
    return db;
}

/* --------------------------------------------------------- read_error_correction --------------------------------------------------------- */

typedef struct { double max_edist, max_arc_f; uint32_t err_mer_c, max_err_c, err_arc_c; uint64_t st[64][12]; } ec_arg_t;

static int ec_rank(oatk_multi *m, int rank, void *arg)
{
    ec_arg_t *a = (ec_arg_t *) arg;
    int rc = oatk_hip_ec_sharded(m->ctx[rank], m->comm[rank], a->max_edist, a->err_mer_c, a->max_err_c, a->err_arc_c, a->max_arc_f, a->st[rank], 0);
    if (!rc) rc = oatk_hip_gather_table(m->ctx[rank], m->comm[rank], 0);
    return rc;
}

int oatk_multi_read_error_correction(oatk_multi *m, oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, double max_edist, uint32_t err_mer_c, uint32_t max_err_c,
                                     uint32_t err_arc_c, double max_arc_f, uint64_t *stats12)
{
    ec_arg_t *a = (ec_arg_t *) calloc(1, sizeof(ec_arg_t));
    int rc, r;
    table_t t;
    uint32_t *new_n = 0, *new_m = 0;
    uint64_t *new_k = 0, *new_s = 0;
    memset(&t, 0, sizeof(t));
    a->max_edist = max_edist, a->max_arc_f = max_arc_f, a->err_mer_c = err_mer_c, a->max_err_c = max_err_c, a->err_arc_c = err_arc_c;
    if (!m->have_reads || m->first[m->n] != sr_db->n) { snprintf(m->err, sizeof(m->err), "the handles do not hold this database's reads"); rc = OATK_E_STATE; goto done; }
    rc = run_ranks(m, ec_rank, a);
    if (rc) goto done;
    if (stats12) memcpy(stats12, a->st[0], sizeof(a->st[0]));       /* summed over the ranks, the same on each */
    /* everything is fetched before anything is rewritten: a failure here leaves the reads and the table as they were */
    if ((rc = table_fetch(m, &t, 0)) != OATK_OK) goto done;
    if (t.n_scm != scm_db->n) { snprintf(m->err, sizeof(m->err), "the table is not the handles'"); rc = OATK_E_STATE; goto done; }
    new_n = (uint32_t *) xmalloc(4 * (sr_db->n + 1));
    new_k = (uint64_t *) xmalloc(8 * (t.n_occ + 1)), new_s = (uint64_t *) xmalloc(8 * (t.n_occ + 1)), new_m = (uint32_t *) xmalloc(4 * (t.n_occ + 1));
    {
        uint64_t at = 0;
        for (r = 0; r < m->n && !rc; ++r) {
            const uint64_t nr = m->first[r + 1] - m->first[r];
            uint64_t b = 0, tot = 0;
            const void *d = 0;
            if (nr == 0) continue;
            rc = fetch_into(m->ctx[r], OATK_BUF_EC_N_SCM, new_n + m->first[r], nr * 4);
            if (!rc) rc = oatk_hip_buffer(m->ctx[r], OATK_BUF_EC_KMER, &d, &b);
            tot = b / 8;
            if (!rc && at + tot > t.n_occ) rc = OATK_E_STATE;
            if (!rc) rc = fetch_into(m->ctx[r], OATK_BUF_EC_KMER, new_k + at, tot * 8);
            if (!rc) rc = fetch_into(m->ctx[r], OATK_BUF_EC_SMER, new_s + at, tot * 8);
            if (!rc) rc = fetch_into(m->ctx[r], OATK_BUF_EC_MPOS, new_m + at, tot * 4);
            note_err(m, r, rc);
            at += tot;
        }
        if (!rc && at != t.n_occ) { snprintf(m->err, sizeof(m->err), "the refreshed table and the corrected chains disagree"); rc = OATK_E_STATE; }
    }
    if (!rc) oatk_host_ec_write_back(sr_db, scm_db, new_n, &new_k, &new_m, &new_s, t.cov, t.del, t.occ_off, &t.occ);
done:
    free(new_n); free(new_k); free(new_m); free(new_s);
    table_free(&t);
    free(a);
    return rc;
}

/* ------------------------------------------------- the graph, the consensus sums, the tables ------------------------------------------------ */

typedef struct { uint32_t c; double f; uint64_t nv[64], na[64]; } ag_arg_t;

static int ag_rank(oatk_multi *m, int rank, void *arg)
{
    ag_arg_t *a = (ag_arg_t *) arg;
    return oatk_hip_asm_graph_sharded(m->ctx[rank], m->comm[rank], a->c, a->f, &a->nv[rank], &a->na[rank]);
}

oatk_asmg_t *oatk_multi_make_syncmer_asmg(oatk_multi *m, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f, int *rc)
{
    int r0 = 0;
    if (!rc) rc = &r0;
    *rc = 0;
    if (!scm_db || scm_db->n == 0) return 0;                                   /* syncasm.c:205 */
    ag_arg_t *a = (ag_arg_t *) calloc(1, sizeof(ag_arg_t));
    a->c = min_k_cov, a->f = min_a_cov_f;
    *rc = run_ranks(m, ag_rank, a);
    oatk_asmg_t *g = *rc? 0 : oatk_host_asmg_from_resident(m->ctx[0], scm_db, a->nv[0], a->na[0], rc);
    if (*rc && !g) note_err(m, 0, *rc);
    free(a);
    return g;
}

static int cons_rank(oatk_multi *m, int rank, void *arg) { return oatk_hip_consensus_sharded(m->ctx[rank], m->comm[rank], *(uint32_t *) arg); }

oatk_consensus_t *oatk_multi_consensus_fetch(oatk_multi *m, uint32_t min_cov, int k, int *rc)
{
    *rc = run_ranks(m, cons_rank, &min_cov);
    return *rc? 0 : oatk_host_consensus_from_resident(m->ctx[0], k, rc);
}

typedef struct { uint32_t c; uint64_t np[64], ne[64]; } ovl_arg_t;
static int ovl_rank(oatk_multi *m, int rank, void *arg)
{
    ovl_arg_t *a = (ovl_arg_t *) arg;
    return oatk_hip_overlap_hist_sharded(m->ctx[rank], m->comm[rank], a->c, &a->np[rank], &a->ne[rank]);
}

oatk_overlap_t *oatk_multi_overlap_fetch(oatk_multi *m, uint32_t min_cov, int *rc)
{
    int r0 = 0;
    if (!rc) rc = &r0;
    ovl_arg_t *a = (ovl_arg_t *) calloc(1, sizeof(ovl_arg_t));
    a->c = min_cov;
    *rc = run_ranks(m, ovl_rank, a);
    oatk_overlap_t *o = *rc? 0 : oatk_host_overlap_from_resident(m->ctx[0], a->np[0], a->ne[0], rc);
    free(a);
    return o;
}

/* ---------------------------------------------------------- scg_read_alignment ----------------------------------------------------------- */

int oatk_multi_scg_read_alignment(oatk_multi *m, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip, uint64_t *n_skipped)
{
    if (!m->have_reads || m->first[m->n] != sr_db->n) { snprintf(m->err, sizeof(m->err), "the handles do not hold this database's reads"); return OATK_E_STATE; }
    const int rc = oatk_host_read_alignment_n(m->ctx, m->first, m->n, sr_db, ra_v, g, for_unzip, n_skipped, 0);
    if (rc && rc != OATK_E_SPLIT) { int r; for (r = 0; r < m->n; ++r) if (oatk_hip_last_error(m->ctx[r])[0]) { note_err(m, r, rc); break; } }
    return rc;
}
