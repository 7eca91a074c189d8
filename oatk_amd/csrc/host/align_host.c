/*
 * oatk_amd/csrc/host/align_host.c -- host side of the drop-in boundary for scg_read_alignment (alignment.c:596-691).
 *
 * Flattens what the per-read routine reads from the reference's scg_t -- the syncmer -> unitig index (scg->idx_u over scg->scm_u,
 * syncasm.c:116-181, entries scm_id[49] | rev[1] | utg[42] | pos[36]), the unitig sizes and the arcs with their overlaps in syncmers
 * (graph.h:39-63) -- builds the old_ra filter exactly like :610-634, runs the alignment on the MI355X against the chains still resident
 * in the context, and rebuilds scg_ra_v the way the reference leaves it: alignments in read order, one malloc'ed fragment array each
 * (scg_ra_v_destroy frees it), score s = 1 / n + max.  Reads whose working set exceeds the device routine's limits are reported
 * back (n_skipped > 0): the caller runs the original routine for those.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_align.h"
#include "oatk_syncasm.h"
#include "host_internal.h"
#include <pthread.h>

typedef unsigned __int128 u128_t;

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

int oatk_scg_read_alignment(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip, uint64_t *n_skipped,
                            uint32_t **skipped)
{
    const uint64_t first[2] = {0, sr_db->n};
    return oatk_host_read_alignment_n(&ctx, first, 1, sr_db, ra_v, g, for_unzip, n_skipped, skipped);
}

/* one handle's share: its reads against the same graph (reads align independently, include/oatk_hip_align.h) */
typedef struct {
    oatk_hip_ctx *ctx;
    const oatk_ra_graph_t *fg;
    const int64_t *old_ra;
    int rc;
    uint64_t n_aln, n_frg, st[3];
    uint32_t *a_sid, *f_ub, *f_ue, *f_sb, *f_se, *sk;
    uint64_t *a_off, *f_uid;
    double *a_s;
} ra_part_t;

static void *ra_part_run(void *arg)
{
    ra_part_t *p = (ra_part_t *) arg;
    uint64_t b;
    int rc = oatk_hip_read_alignment(p->ctx, p->fg, p->old_ra, &p->n_aln, &p->n_frg, p->st);
    if (!rc) p->a_sid = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_ALN_SID, &b, &rc);
    if (!rc) p->a_off = (uint64_t *) fetch(p->ctx, OATK_BUF_RA_ALN_OFF, &b, &rc);
    if (!rc) p->a_s = (double *) fetch(p->ctx, OATK_BUF_RA_ALN_S, &b, &rc);
    if (!rc) p->f_uid = (uint64_t *) fetch(p->ctx, OATK_BUF_RA_FRG_UID, &b, &rc);
    if (!rc) p->f_ub = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_FRG_UBEG, &b, &rc);
    if (!rc) p->f_ue = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_FRG_UEND, &b, &rc);
    if (!rc) p->f_sb = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_FRG_SBEG, &b, &rc);
    if (!rc) p->f_se = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_FRG_SEND, &b, &rc);
    if (!rc && p->st[2]) p->sk = (uint32_t *) fetch(p->ctx, OATK_BUF_RA_SKIPPED, &b, &rc);
    p->rc = rc;
    return 0;
}

static void ra_part_free(ra_part_t *p)
{
    free(p->a_sid); free(p->a_off); free(p->a_s); free(p->f_uid); free(p->f_ub); free(p->f_ue); free(p->f_sb); free(p->f_se); free(p->sk);
}

int oatk_host_read_alignment_n(oatk_hip_ctx **ctx, const uint64_t *first, int n, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip,
                               uint64_t *n_skipped, uint32_t **skipped)
{
    uint64_t i, j;
    int rc = 0, r;
    if (n_skipped) *n_skipped = 0;
    if (skipped) *skipped = 0;
    oatk_asmg_t *ug = g->utg_asmg;
    uint64_t n_live = 0;
    for (i = 0; i < ug->n_vtx; ++i) n_live += !ug->vtx[i].del;                      /* asmg_vtx_n1, graph.h:138 */
    if (sr_db->n == 0 || !n_live) return 0;                                        /* alignment.c:598 */

    /* old_ra, alignment.c:610-634 */
    int64_t *old_ra = (int64_t *) calloc(sr_db->n, sizeof(int64_t));
    const uint64_t sid0 = sr_db->a[0].sid;
    if (for_unzip && ra_v->n > 0) {
        double fractpart, intpart;
        for (j = 0; j < ra_v->n; ++j) {
            oatk_scg_ra_t *ra = &ra_v->a[j];
            const uint64_t sid = ra->sid - sid0;
            if (ra->n > 2 && (old_ra[sid] & 1) == 0) {
                fractpart = modf(ra->s, &intpart);
                if (fractpart < DBL_EPSILON) intpart -= 1;
                old_ra[sid] = (uint64_t) intpart << 1 | 1;
            }
        }
    } else for (j = 0; j < sr_db->n; ++j) old_ra[j] = 1;

    /* scg_t -> flat arrays */
    const uint64_t ns = g->scm_db->n, nu = ug->n_vtx, na = ug->n_arc;
    const u128_t *su0 = (const u128_t *) g->idx_u[0];
    const uint64_t nsu = (uint64_t) ((const u128_t *) g->idx_u[ns] - su0);
    uint64_t *su_off = (uint64_t *) xmalloc(8 * (ns + 1)), *su_uid = (uint64_t *) xmalloc(8 * nsu);
    uint32_t *su_pos = (uint32_t *) xmalloc(4 * nsu), *utg_n = (uint32_t *) xmalloc(4 * nu);
    for (i = 0; i <= ns; ++i) su_off[i] = (uint64_t) ((const u128_t *) g->idx_u[i] - su0);
    for (i = 0; i < nsu; ++i) {
        const u128_t x = su0[i];
        su_uid[i] = (uint64_t) ((x >> 36) & 0x3FFFFFFFFFFULL) << 1 | (uint64_t) ((x >> 78) & 1);      /* scm_utg_uid, scm_utg_rev (syncasm.h:44-47) */
        su_pos[i] = (uint32_t) (x & 0xFFFFFFFFFULL);                                                  /* scm_utg_pos */
    }
    for (i = 0; i < nu; ++i) utg_n[i] = (uint32_t) ug->vtx[i].n;
    uint64_t *arc_w = (uint64_t *) xmalloc(8 * na), *arc_ln = (uint64_t *) xmalloc(8 * na);
    uint8_t *arc_del = (uint8_t *) xmalloc(na);
    for (i = 0; i < na; ++i) arc_w[i] = ug->arc[i].w, arc_ln[i] = ug->arc[i].ln, arc_del[i] = ug->arc[i].del;
    oatk_ra_graph_t fg = {ns, nu, na, su_off, su_uid, su_pos, utg_n, ug->idx_p, ug->idx_n, arc_w, arc_ln, arc_del};

    /* every handle aligns its own reads (the handles of several GPUs side by side: one host thread each) */
    ra_part_t *part = (ra_part_t *) calloc((size_t) n, sizeof(ra_part_t));
    pthread_t *th = (pthread_t *) calloc((size_t) n, sizeof(pthread_t));
    for (r = 0; r < n; ++r) part[r].ctx = ctx[r], part[r].fg = &fg, part[r].old_ra = old_ra + first[r];
    for (r = 1; r < n; ++r) if (pthread_create(&th[r], 0, ra_part_run, &part[r]) != 0) { ra_part_run(&part[r]); th[r] = 0; }
    ra_part_run(&part[0]);
    for (r = 1; r < n; ++r) if (th[r]) pthread_join(th[r], 0);
    free(th);
    free(su_off); free(su_uid); free(su_pos); free(utg_n); free(arc_w); free(arc_ln); free(arc_del); free(old_ra);
    uint64_t n_aln = 0, st[3] = {0, 0, 0};
    for (r = 0; r < n; ++r) {
        if (part[r].rc && !rc) rc = part[r].rc;
        n_aln += part[r].n_aln, st[0] += part[r].st[0], st[1] += part[r].st[1], st[2] += part[r].st[2];
    }
    if (rc) goto done;
    if (st[2] && !skipped) {                                                       /* a caller that cannot finish the skipped reads itself gets */
        if (n_skipped) *n_skipped = st[2];                                         /* all or nothing: ra_v is still the previous round's        */
        rc = OATK_E_SPLIT;
        goto done;
    }

    /* scg_ra_v_clean (alignment.c:47-54), then the new alignments (:650-664): the handles' results in handle order are read order */
    for (i = 0; i < ra_v->n; ++i) free(ra_v->a[i].a);
    free(ra_v->a);
    ra_v->n = ra_v->m = n_aln;
    ra_v->a = (oatk_scg_ra_t *) xmalloc(sizeof(oatk_scg_ra_t) * n_aln);
    {
        uint64_t at = 0;
        for (r = 0; r < n; ++r) {
            const ra_part_t *p = &part[r];
            for (i = 0; i < p->n_aln; ++i, ++at) {
                oatk_scg_ra_t *ra = &ra_v->a[at];
                const uint64_t o = p->a_off[i], m = p->a_off[i + 1] - o;
                ra->sid = sid0 + first[r] + p->a_sid[i], ra->n = (uint32_t) m, ra->s = p->a_s[i];
                ra->a = (oatk_ra_frg_t *) xmalloc(sizeof(oatk_ra_frg_t) * m);
                for (j = 0; j < m; ++j) {
                    oatk_ra_frg_t *f = &ra->a[j];
                    f->uid = p->f_uid[o + j], f->u_beg = p->f_ub[o + j], f->u_end = p->f_ue[o + j], f->s_beg = p->f_sb[o + j], f->s_end = p->f_se[o + j];
                }
            }
        }
    }
    {
        uint64_t n_r = 0;
        for (i = 0; i < sr_db->n; ++i) n_r += sr_db->a[i].n > 0;
        fprintf(stderr, "[M::%s] %lu mappable reads, %lu mapped (%lu unique mapping)\n", "scg_read_alignment", n_r, st[0], st[1]);      /* :685 */
    }
    if (st[2]) {
        uint32_t *sk = skipped? (uint32_t *) xmalloc(4 * st[2]) : 0;
        uint64_t at = 0;
        for (r = 0; r < n && sk; ++r) for (i = 0; i < part[r].st[2]; ++i) sk[at++] = (uint32_t) (first[r] + part[r].sk[i]);
        if (n_skipped) *n_skipped = st[2];
        if (skipped) *skipped = sk;
    }
done:
    for (r = 0; r < n; ++r) ra_part_free(&part[r]);
    free(part);
    return rc;
}
