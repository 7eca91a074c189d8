/*
 * oatk_amd/csrc/host/graph_host.c -- host side of the drop-in boundary for the assembly graph,
 * make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299, called at run_syncasm.c:138).
 *
 * The graph is built on the MI355X from the batch still resident in the context (oatk_hip_asm_graph: the corrected chains after
 * oatk_hip_ec, the counted ones otherwise) and returned as the reference's own asmg_t (graph.h:39-63): an array of vertex structs
 * (one syncmer each), an array of arc structs in (v, w) order and the per-vertex index -- allocated the way the reference
 * allocates them (each vtx.a its own block), so asmg_destroy (graph.c) frees it.  scm_db->a[i].del is updated like syncasm.c:228.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_graph.h"
#include "oatk_syncasm.h"
#include "host_internal.h"

static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n? n : 1, sz);
    if (!p) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    return p;
}

static void *fetch(oatk_hip_ctx *ctx, int which, uint64_t *bytes, int *rc)
{
    const void *d = 0;
    *bytes = 0;
    *rc = oatk_hip_buffer(ctx, which, &d, bytes);
    if (*rc) return 0;
    void *h = xcalloc(*bytes, 1);
    *rc = oatk_hip_d2h(ctx, h, d, *bytes);
    if (*rc) { free(h); return 0; }
    return h;
}

oatk_asmg_t *oatk_make_syncmer_asmg(oatk_hip_ctx *ctx, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f, int *rc)
{
    uint64_t nv = 0, na = 0, i, b;
    int r = 0;
    if (!rc) rc = &r;
    *rc = 0;
    if (!scm_db || scm_db->n == 0) return 0;                                   /* syncasm.c:205 */
    *rc = oatk_hip_asm_graph(ctx, min_k_cov, min_a_cov_f, &nv, &na);
    if (*rc) return 0;
    (void) i, (void) b;
    return oatk_host_asmg_from_resident(ctx, scm_db, nv, na, rc);
}

oatk_asmg_t *oatk_host_asmg_from_resident(oatk_hip_ctx *ctx, oatk_syncmer_db_t *scm_db, uint64_t nv, uint64_t na, int *rc)
{
    uint64_t i, b;
    uint8_t *del = (uint8_t *) fetch(ctx, OATK_BUF_AG_SCM_DEL, &b, rc); if (*rc) return 0;
    if (b != scm_db->n) { free(del); *rc = OATK_E_STATE; return 0; }           /* the table is not the resident batch's */
    uint32_t *vscm = (uint32_t *) fetch(ctx, OATK_BUF_AG_VTX_SCM, &b, rc); if (*rc) return 0;
    uint32_t *vcov = (uint32_t *) fetch(ctx, OATK_BUF_AG_VTX_COV, &b, rc); if (*rc) return 0;
    uint64_t *idx_p = (uint64_t *) fetch(ctx, OATK_BUF_AG_IDX_P, &b, rc); if (*rc) return 0;
    uint32_t *idx_n = (uint32_t *) fetch(ctx, OATK_BUF_AG_IDX_N, &b, rc); if (*rc) return 0;
    uint64_t *av = (uint64_t *) fetch(ctx, OATK_BUF_AG_ARC_V, &b, rc); if (*rc) return 0;
    uint64_t *aw = (uint64_t *) fetch(ctx, OATK_BUF_AG_ARC_W, &b, rc); if (*rc) return 0;
    uint32_t *acov = (uint32_t *) fetch(ctx, OATK_BUF_AG_ARC_COV, &b, rc); if (*rc) return 0;
    uint8_t *acomp = (uint8_t *) fetch(ctx, OATK_BUF_AG_ARC_COMP, &b, rc); if (*rc) return 0;
    uint64_t *alink = (uint64_t *) fetch(ctx, OATK_BUF_AG_ARC_LINK, &b, rc); if (*rc) return 0;

    for (i = 0; i < scm_db->n; ++i) scm_db->a[i].del = del[i];
    oatk_asmg_t *g = (oatk_asmg_t *) xcalloc(1, sizeof(oatk_asmg_t));
    g->n_vtx = g->m_vtx = nv, g->n_arc = g->m_arc = na;
    g->vtx = (oatk_asmg_vtx_t *) xcalloc(nv, sizeof(oatk_asmg_vtx_t));
    g->arc = (oatk_asmg_arc_t *) xcalloc(na, sizeof(oatk_asmg_arc_t));
    g->idx_p = (uint64_t *) xcalloc(2 * nv, 8), g->idx_n = (uint64_t *) xcalloc(2 * nv, 8);
    for (i = 0; i < nv; ++i) {
        oatk_asmg_vtx_t *v = &g->vtx[i];
        v->n = 1, v->a = (uint64_t *) xcalloc(1, 8), v->a[0] = (uint64_t) vscm[i] << 1, v->cov = vcov[i];
    }
    for (i = 0; i < na; ++i) {
        oatk_asmg_arc_t *a = &g->arc[i];
        a->v = av[i], a->w = aw[i], a->cov = acov[i], a->comp = acomp[i], a->link_id = alink[i];
    }
    for (i = 0; i < 2 * nv; ++i) g->idx_n[i] = idx_n[i], g->idx_p[i] = idx_n[i]? idx_p[i] : 0;
    free(del); free(vscm); free(vcov); free(idx_p); free(idx_n); free(av); free(aw); free(acov); free(acomp); free(alink);
    return g;
}

void oatk_asmg_destroy(oatk_asmg_t *g)
{
    uint64_t i;
    if (!g) return;
    for (i = 0; i < g->n_vtx; ++i) { free(g->vtx[i].a); free(g->vtx[i].seq); }
    free(g->vtx); free(g->arc); free(g->idx_p); free(g->idx_n);
    free(g);
}
