/*
 * oatk_amd/csrc/host/ec_host.c -- host side of the drop-in boundary for read_error_correction (syncerr.c:819).
 *
 * Flattens the reference's asmg_t (array of arc structs + CSR index, graph.h:39-63) into the plain arrays
 * oatk_hip_ec takes, runs the correction on the MI355X on the batch that is still resident from scan + count, and
 * writes the results back into the reference's structs the way the reference itself would:
 *   - every read: k_mer / m_pos / s_mer / n                      (syncerr.c:600-612)
 *   - syncmer table: cov, del, m_pos; c and h released           (update_syncmer_db, syncerr.c:769-814)
 *   - graph: deleted error syncmers and their arcs              (find_error_syncmers with del_err = 1, syncerr.c:748-752)
 * With asmg == NULL the EC graph itself (run_syncasm.c:109-117) is built on the device as well and there is nothing to flatten.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_ec.h"
#include "oatk_syncasm.h"
#include "host_internal.h"

static void *xmalloc(size_t n)
{
    void *p = malloc(n? n : 1);
    if (!p) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    return p;
}

static void *fetch(oatk_hip_ctx *ctx, int which, uint64_t *bytes, int *rc)
{
    const void *d = 0;
    *bytes = 0;
    *rc = oatk_hip_buffer(ctx, which, &d, bytes);
    if (*rc) return 0;
    void *h = xmalloc(*bytes);
    *rc = oatk_hip_d2h(ctx, h, d, *bytes);
    if (*rc) { free(h); return 0; }
    return h;
}

typedef struct {
    oatk_sr_db_t *sr_db;
    oatk_syncmer_db_t *scm_db;
    const uint32_t *new_n;
    const uint64_t *new_k;
    const uint32_t *new_m;
    const uint64_t *new_s;
    const uint32_t *cov;
    const uint8_t *del;
    const uint64_t *occ_off, *occ;
    uint64_t *new_off;
    int adopt;                     /* arenas in use (include/oatk_syncasm.h): point into the fetched arrays instead of copying out of them */
} ecw_job_t;

/* what the reference's own threads leave per read (syncerr.c:600-612) and update_syncmer_db per syncmer (:769-814) */
static void ecw_worker(void *arg, int tid, int n_threads)
{
    const ecw_job_t *j = (const ecw_job_t *) arg;
    uint64_t i;
    const uint64_t nr = j->sr_db->n, ra = nr * (uint64_t) tid / (uint64_t) n_threads, rb = nr * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    for (i = ra; i < rb; ++i) {
        oatk_sr_t *r = &j->sr_db->a[i];
        const uint32_t n = j->new_n[i];
        const uint64_t o = j->new_off[i];
        oatk_sr_member_free(r->k_mer); oatk_sr_member_free(r->m_pos); oatk_sr_member_free(r->s_mer);
        if (j->adopt) {             /* arenas in use: the arrays fetched from the device ARE the reads' storage from now on */
            r->k_mer = n? (uint64_t *) j->new_k + o : 0, r->s_mer = n? (uint64_t *) j->new_s + o : 0, r->m_pos = n? (uint32_t *) j->new_m + o : 0;
        } else {
            r->k_mer = (uint64_t *) memcpy(xmalloc(8 * (size_t) n), j->new_k + o, 8 * (size_t) n);
            r->m_pos = (uint32_t *) memcpy(xmalloc(4 * (size_t) n), j->new_m + o, 4 * (size_t) n);
            r->s_mer = (uint64_t *) memcpy(xmalloc(8 * (size_t) n), j->new_s + o, 8 * (size_t) n);
        }
        r->n = n;
    }
    const uint64_t ns = j->scm_db->n, sa = ns * (uint64_t) tid / (uint64_t) n_threads, sb = ns * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    for (i = sa; i < sb; ++i) {
        oatk_syncmer_t *m = &j->scm_db->a[i];
        oatk_sr_member_free(m->m_pos);
        m->cov = j->cov[i], m->del = j->del[i];
        m->m_pos = j->adopt? (j->cov[i]? (uint64_t *) j->occ + j->occ_off[i] : 0)         /* (an empty list: NULL, which free() and realloc() take) */
                           : (uint64_t *) memcpy(xmalloc(8 * (size_t) j->cov[i]), j->occ + j->occ_off[i], 8 * (size_t) j->cov[i]);
    }
}

void oatk_host_ec_write_back(oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, const uint32_t *new_n, uint64_t **new_k, uint32_t **new_m, uint64_t **new_s,
                             const uint32_t *cov, const uint8_t *del, const uint64_t *occ_off, uint64_t **occ)
{
    uint64_t i;
    ecw_job_t job = {sr_db, scm_db, new_n, *new_k, *new_m, *new_s, cov, del, occ_off, *occ, 0, 0};
    job.new_off = (uint64_t *) xmalloc(8 * (sr_db->n + 1));
    for (i = 0, job.new_off[0] = 0; i < sr_db->n; ++i) job.new_off[i + 1] = job.new_off[i] + new_n[i];
    job.adopt = oatk_host_arena();
    free(scm_db->c); scm_db->c = 0;
    free(scm_db->h); scm_db->h = 0;
    oatk_par_run(ecw_worker, &job);
    if (job.adopt) {
        /* what the reads and the table pointed into before (their fill-time arenas, the arrays adopted by the count) stays until the
         * databases are cleaned: a few arrays of the size of the chains, against a free() per read and per syncmer */
        const size_t tot = (size_t) job.new_off[sr_db->n];
        oatk_host_arena_adopt(*new_k, 8 * tot, sr_db), *new_k = 0;
        oatk_host_arena_adopt(*new_m, 4 * tot, sr_db), *new_m = 0;
        oatk_host_arena_adopt(*new_s, 8 * tot, sr_db), *new_s = 0;
        oatk_host_arena_adopt(*occ, 8 * tot, scm_db), *occ = 0;
    }
    free(job.new_off);
}

int oatk_read_error_correction(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, oatk_asmg_t *asmg, double max_edist,
                               uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f, uint64_t *stats12)
{
    uint64_t i, b;
    int rc;
    uint64_t *arc_v = 0, *arc_w = 0;
    uint64_t nv = 0, na = 0;
    if (!asmg) {
        /* no host graph at all: make_syncmer_graph(sr_db, scm_db, 0, 0.) + the hoco arc overlaps are built on the device too -- of them only
         * what the correction can use when the thresholds allow it (include/oatk_hip_ec.h: the light graph; run_syncasm.c:124 always does) */
        rc = err_mer_c > 0 && err_arc_c >= err_mer_c? oatk_hip_ec_graph_light(ctx, err_mer_c) : oatk_hip_ec_graph(ctx);
        if (rc) return rc;
        rc = oatk_hip_ec(ctx, 0, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f);
        if (rc) return rc;
    } else {
    nv = asmg->n_vtx, na = asmg->n_arc;
    /* asmg_t -> flat arrays */
    uint64_t *idx_n = (uint64_t *) xmalloc(8 * 2 * nv);
    arc_v = (uint64_t *) xmalloc(8 * na), arc_w = (uint64_t *) xmalloc(8 * na);
    uint64_t *arc_ls = (uint64_t *) xmalloc(8 * na);
    uint32_t *arc_cov = (uint32_t *) xmalloc(4 * na);
    uint8_t *arc_del = (uint8_t *) xmalloc(na);
    memcpy(idx_n, asmg->idx_n, 8 * 2 * nv);
    for (i = 0; i < na; ++i) {
        const oatk_asmg_arc_t *a = &asmg->arc[i];
        arc_v[i] = a->v, arc_w[i] = a->w, arc_ls[i] = a->ls, arc_cov[i] = a->cov, arc_del[i] = (uint8_t) a->del;
    }
    oatk_ec_graph_t g;
    g.n_vtx = nv, g.n_arc = na, g.idx_p = asmg->idx_p, g.idx_n = idx_n, g.arc_v = arc_v, g.arc_w = arc_w, g.arc_ls = arc_ls;
    g.arc_cov = arc_cov, g.arc_del = arc_del;
    rc = oatk_hip_ec(ctx, &g, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f);
    free(idx_n); free(arc_ls); free(arc_cov); free(arc_del);
    if (rc) { free(arc_v); free(arc_w); return rc; }
    }
    if (stats12) oatk_hip_ec_stats(ctx, stats12);

    /* everything is fetched before anything is rewritten: a failure here leaves the reads and the table as they were */
    uint32_t *new_n = 0, *new_m = 0, *cov = 0;
    uint64_t *new_k = 0, *new_s = 0, *occ_off = 0, *occ = 0;
    uint8_t *del = 0, *err_del = 0;
    new_n = (uint32_t *) fetch(ctx, OATK_BUF_EC_N_SCM, &b, &rc); if (rc) goto done;
    new_k = (uint64_t *) fetch(ctx, OATK_BUF_EC_KMER, &b, &rc); if (rc) goto done;
    new_m = (uint32_t *) fetch(ctx, OATK_BUF_EC_MPOS, &b, &rc); if (rc) goto done;
    new_s = (uint64_t *) fetch(ctx, OATK_BUF_EC_SMER, &b, &rc); if (rc) goto done;
    cov = (uint32_t *) fetch(ctx, OATK_BUF_EC_SCM_COV, &b, &rc); if (rc) goto done;
    del = (uint8_t *) fetch(ctx, OATK_BUF_EC_SCM_DEL, &b, &rc); if (rc) goto done;
    err_del = (uint8_t *) fetch(ctx, OATK_BUF_EC_ERR_DEL, &b, &rc); if (rc) goto done;
    occ_off = (uint64_t *) fetch(ctx, OATK_BUF_EC_SCM_OCC_OFF, &b, &rc); if (rc) goto done;
    occ = (uint64_t *) fetch(ctx, OATK_BUF_EC_SCM_OCC, &b, &rc); if (rc) goto done;

    /* graph: what find_error_syncmers(..., del_err = 1) leaves behind -- every arc touching a marked syncmer */
    if (asmg) {
        for (i = 0; i < nv; ++i) if (err_del[i]) asmg->vtx[i].del = 1;
        for (i = 0; i < na; ++i) if (err_del[arc_v[i] >> 1] || err_del[arc_w[i] >> 1]) asmg->arc[i].del = 1;
    }

    /* reads and syncmer table, on the host threads */
    oatk_host_ec_write_back(sr_db, scm_db, new_n, &new_k, &new_m, &new_s, cov, del, occ_off, &occ);
done:
    free(arc_v); free(arc_w);
    free(new_n); free(new_k); free(new_m); free(new_s); free(cov); free(del); free(err_del); free(occ_off); free(occ);
    return rc;
}
