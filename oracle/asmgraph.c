/*
 * oracle/asmgraph.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU restatement of the assembly graph the reference
 * builds from the (corrected) reads, make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299) followed by
 * asmg_finalize(g, 1) (graph.c:250-263).  Pinned against the compiled reference in tests/test_oracle_asmgraph.py.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static int u64_cmp(const void *a, const void *b) { uint64_t x = *(const uint64_t *) a, y = *(const uint64_t *) b; return (x > y) - (x < y); }
typedef struct { uint64_t k; uint32_t cov, comp; } arc_t;
static int arc_cmp(const void *a, const void *b) { uint64_t x = ((const arc_t *) a)->k, y = ((const arc_t *) b)->k; return (x > y) - (x < y); }

orc_asmgraph_t *orc_asmgraph_build(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos,
                                   uint64_t n_syncmers, const uint32_t *scm_cov, uint8_t *scm_del, uint32_t min_k_cov, double min_a_cov_f)
{
    orc_asmgraph_t *g = (orc_asmgraph_t *) calloc(1, sizeof(orc_asmgraph_t));
    uint64_t i, j, tot = 0, np = 0, o = 0, nv = 0, na = 0;
    /* syncasm.c:226-233: the coverage filter also marks the syncmer table */
    uint64_t *vidx = (uint64_t *) malloc(8 * (n_syncmers + 1));
    for (i = 0; i < n_syncmers; ++i) {
        scm_del[i] |= scm_cov[i] < min_k_cov;
        vidx[i] = scm_del[i]? UINT64_MAX : nv++;                   /* asmg_cleanup, graph.c:153-173 */
    }
    g->n_vtx = nv;
    g->vtx_scm = (uint32_t *) malloc(4 * (nv + 1)); g->vtx_cov = (uint32_t *) malloc(4 * (nv + 1));
    for (i = 0; i < n_syncmers; ++i) if (!scm_del[i]) g->vtx_scm[vidx[i]] = (uint32_t) i, g->vtx_cov[vidx[i]] = scm_cov[i] & 0x3FFFFFFFu;
    /* canonical keys of adjacent pairs, syncasm.c:242-261 */
    for (i = 0; i < n_reads; ++i) tot += n_scm[i];
    uint64_t *keys = (uint64_t *) malloc(8 * (tot + 1));
    for (i = 0; i < n_reads; ++i) {
        for (j = 1; j < n_scm[i]; ++j) {
            uint64_t v0 = (k_mer[o + j - 1] >> 1) << 1 | (m_pos[o + j - 1] & 1), v1 = (k_mer[o + j] >> 1) << 1 | (m_pos[o + j] & 1);
            keys[np++] = v0 <= v1? v0 << 32 | v1 : (v1 ^ 1) << 32 | (v0 ^ 1);
        }
        o += n_scm[i];
    }
    qsort(keys, np, 8, u64_cmp);
    /* arcs + complements that pass the filter, syncasm.c:264-282, renumbered (graph.c:175-200) */
    arc_t *arc = (arc_t *) malloc(sizeof(arc_t) * (2 * np + 2));
    for (i = 0; i < np; i = j) {
        for (j = i; j < np && keys[j] == keys[i]; ++j) {}
        uint64_t v0 = keys[i] >> 32, v1 = keys[i] & 0xFFFFFFFFULL;
        uint32_t v_v = (uint32_t) (j - i), c0 = scm_cov[v0 >> 1], c1 = scm_cov[v1 >> 1];
        if (v_v < min_a_cov_f * (c0 < c1? c0 : c1) || scm_del[v0 >> 1] || scm_del[v1 >> 1]) continue;
        uint64_t n0 = vidx[v0 >> 1] << 1 | (v0 & 1), n1 = vidx[v1 >> 1] << 1 | (v1 & 1);
        arc[na].k = n0 << 32 | n1, arc[na].cov = v_v & 0x3FFFFFFFu, arc[na].comp = 0, ++na;
        if ((v1 ^ 1) != v0) arc[na].k = (n1 ^ 1) << 32 | (n0 ^ 1), arc[na].cov = v_v & 0x3FFFFFFFu, arc[na].comp = 1, ++na;
    }
    qsort(arc, na, sizeof(arc_t), arc_cmp);                        /* graph.c:70-83 */
    g->n_arc = na;
    g->arc_v = (uint64_t *) malloc(8 * (na + 1)); g->arc_w = (uint64_t *) malloc(8 * (na + 1)); g->arc_link = (uint64_t *) malloc(8 * (na + 1));
    g->arc_cov = (uint32_t *) malloc(4 * (na + 1)); g->arc_comp = (uint8_t *) malloc(na + 1);
    g->idx_p = (uint64_t *) calloc(2 * nv + 1, 8); g->idx_n = (uint64_t *) calloc(2 * nv + 1, 8);
    for (i = 0; i < na; ++i) {
        g->arc_v[i] = arc[i].k >> 32, g->arc_w[i] = arc[i].k & 0xFFFFFFFFULL, g->arc_cov[i] = arc[i].cov, g->arc_comp[i] = (uint8_t) arc[i].comp;
        if (i && arc[i].k == arc[i - 1].k) g->multi_arc = 1;
        if (g->idx_n[g->arc_v[i]]++ == 0) g->idx_p[g->arc_v[i]] = i;                  /* graph.c:85-113 */
        g->arc_link[i] = UINT64_MAX;
    }
    /* asmg_arc_fix_symm, graph.c:205-233: every arc writes !comp into its complement, in arc order */
    for (i = 0; i < na; ++i) {
        uint64_t cv = g->arc_w[i] ^ 1, cw = g->arc_v[i] ^ 1, p = g->idx_p[cv], n = g->idx_n[cv], t;
        for (t = 0; t < n; ++t) if (g->arc_w[p + t] == cw) { g->arc_comp[p + t] = g->arc_comp[i] ^ 1; break; }
    }
    /* asmg_shrink_link_id, graph.c:126-146 */
    uint64_t link = 0;
    for (i = 0; i < na; ++i) {
        if (g->arc_link[i] != UINT64_MAX) continue;
        uint64_t cv = g->arc_w[i] ^ 1, cw = g->arc_v[i] ^ 1, p = g->idx_p[cv], n = g->idx_n[cv], t;
        g->arc_link[i] = link;
        for (t = 0; t < n; ++t) if (g->arc_w[p + t] == cw) { g->arc_link[p + t] = link; break; }
        ++link;
    }
    free(vidx); free(keys); free(arc);
    return g;
}

void orc_asmgraph_free(orc_asmgraph_t *g)
{
    if (!g) return;
    free(g->vtx_scm); free(g->vtx_cov); free(g->arc_v); free(g->arc_w); free(g->arc_link); free(g->arc_cov); free(g->arc_comp); free(g->idx_p); free(g->idx_n);
    free(g);
}
