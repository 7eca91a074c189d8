/*
 * oracle/ecgraph.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates how syncasm builds the graph it corrects reads against (run_syncasm.c:109-117):
 *   make_syncmer_graph(sr_db, scm_db, 0, 0.)   syncasm.c:203-299   one vertex per syncmer; an arc per pair of syncmers
 *                                                                  adjacent on some read, keyed canonically, with its
 *                                                                  complement; sorted by (v, w) (graph.c:70-83)
 *   scg_consensus(hoco = 1), arc part          syncasm.c:793-812   arc.ls = K - (most frequent distance between the two
 *                                                                  syncmers on the reads), calc_syncmer_overlap :477-582
 * including the reference's tie rule for "most frequent": the first key in khashl BUCKET ORDER that reaches the maximum
 * count (syncasm.c:558-571), with khashl's identity hash, Fibonacci bucket mapping, linear probing and its growth /
 * kick-out rehash (khashl.h:82, :150-192, :194-218) -- restated in khl_* below.
 *
 * Only the case the error correction needs is covered: nothing is deleted yet (fresh databases, min_k_cov = 0,
 * min_a_cov_f = 0), so asmg_cleanup is the identity.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- khashl<int,int> with kh_hash_dummy, as instantiated at syncasm.c:63 ---- */
typedef struct { uint32_t bits, count, cap; uint8_t *used; int32_t *keys, *vals; } khl_t;

static uint32_t khl_h2b(uint32_t hash, uint32_t bits) { return (hash * 2654435769U) >> (32 - bits); }

static void khl_resize(khl_t *h, uint32_t want)               /* khashl.h:150-192 */
{
    uint32_t j = 0, x = want, n_old = h->keys? 1U << h->bits : 0U, nb, bits;
    while ((x >>= 1) != 0) ++j;
    if (want & (want - 1)) ++j;
    bits = j > 2? j : 2;
    nb = 1U << bits;
    uint8_t *nused = (uint8_t *) calloc(nb, 1);
    h->keys = (int32_t *) realloc(h->keys, sizeof(int32_t) * nb);
    h->vals = (int32_t *) realloc(h->vals, sizeof(int32_t) * nb);
    for (j = 0; j != n_old; ++j) {
        if (!h->used[j]) continue;
        int32_t key = h->keys[j], val = h->vals[j];
        h->used[j] = 0;
        for (;;) {                                             /* kick-out process */
            uint32_t i = khl_h2b((uint32_t) key, bits);
            while (nused[i]) i = (i + 1) & (nb - 1);
            nused[i] = 1;
            if (i < n_old && h->used[i]) {
                int32_t tk = h->keys[i], tv = h->vals[i];
                h->keys[i] = key, h->vals[i] = val, key = tk, val = tv;
                h->used[i] = 0;
            } else {
                h->keys[i] = key, h->vals[i] = val;
                break;
            }
        }
    }
    free(h->used);
    h->used = nused, h->bits = bits;
}

static void khl_add1(khl_t *h, int32_t key)                   /* add_ovl_count, syncasm.c:465-474 */
{
    uint32_t nb = h->keys? 1U << h->bits : 0U;
    if (h->count >= (nb >> 1) + (nb >> 2)) { khl_resize(h, nb + 1U); nb = 1U << h->bits; }
    uint32_t i = khl_h2b((uint32_t) key, h->bits), last = i;
    while (h->used[i] && h->keys[i] != key) { i = (i + 1U) & (nb - 1); if (i == last) break; }
    if (!h->used[i]) h->keys[i] = key, h->vals[i] = 1, h->used[i] = 1, ++h->count;
    else ++h->vals[i];
}

typedef struct { uint64_t key; } pair_t;
static int u64_cmp(const void *a, const void *b) { uint64_t x = *(const uint64_t *) a, y = *(const uint64_t *) b; return (x > y) - (x < y); }

/* calc_syncmer_overlap (syncasm.c:477-582) for the arc v -> w; returns the winning distance (0 when no read supports it) */
static int32_t overlap_mode(const orc_count_view_t *c, const uint64_t *scm_off, const uint32_t *m_pos, uint64_t v, uint64_t w)
{
    const uint64_t m1 = v >> 1, m2 = w >> 1, rc1 = v & 1, rc2 = w & 1;
    const uint64_t *pos1 = c->occ + c->occ_off[m1], *pos2 = c->occ + c->occ_off[m2];
    const uint64_t n1 = c->occ_off[m1 + 1] - c->occ_off[m1], n2 = c->occ_off[m2 + 1] - c->occ_off[m2];
    khl_t h;
    memset(&h, 0, sizeof(h));
    uint64_t p1, p2 = 0, i;
    for (p1 = 0; p1 < n1; ++p1) {
        const uint64_t r1 = pos1[p1] >> 32, i1 = (pos1[p1] >> 1) & 0x7FFFFFFFULL, c1 = pos1[p1] & 1;
        const int64_t l1 = m_pos[scm_off[r1] + i1] >> 1;
        while (p2 < n2 && (pos2[p2] >> 32) < r1) ++p2;
        for (i = p2; i < n2 && (pos2[i] >> 32) == r1; ++i) {
            const uint64_t i2 = (pos2[i] >> 1) & 0x7FFFFFFFULL, c2 = pos2[i] & 1;
            const int64_t l2 = m_pos[scm_off[r1] + i2] >> 1;
            if (i1 == i2 + 1 && c1 != rc1 && c2 != rc2) khl_add1(&h, (int32_t) (l1 - l2));
            else if (i1 + 1 == i2 && c1 == rc1 && c2 == rc2) khl_add1(&h, (int32_t) (l2 - l1));
        }
    }
    int32_t movl = 0, mcnt = 0;
    uint32_t k, nb = h.keys? 1U << h.bits : 0U;
    for (k = 0; k < nb; ++k) if (h.used[k] && h.vals[k] > mcnt) mcnt = h.vals[k], movl = h.keys[k];
    free(h.used); free(h.keys); free(h.vals);
    return movl;
}

orc_ecgraph_t *orc_ecgraph_build(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos,
                                 const orc_count_view_t *c, int K)
{
    orc_ecgraph_t *g = (orc_ecgraph_t *) calloc(1, sizeof(orc_ecgraph_t));
    uint64_t i, j, tot = 0, np = 0, o = 0;
    uint64_t *scm_off = (uint64_t *) malloc(sizeof(uint64_t) * (n_reads + 1));
    scm_off[0] = 0;
    for (i = 0; i < n_reads; ++i) scm_off[i + 1] = scm_off[i] + n_scm[i], tot += n_scm[i];
    /* canonical keys of adjacent pairs, syncasm.c:242-261 */
    uint64_t *keys = (uint64_t *) malloc(sizeof(uint64_t) * (tot + 1));
    for (i = 0; i < n_reads; ++i) {
        for (j = 1; j < n_scm[i]; ++j) {
            uint64_t v0 = (k_mer[o + j - 1] >> 1) << 1 | (m_pos[o + j - 1] & 1), v1 = (k_mer[o + j] >> 1) << 1 | (m_pos[o + j] & 1);
            keys[np++] = v0 <= v1? v0 << 32 | v1 : (v1 ^ 1) << 32 | (v0 ^ 1);
        }
        o += n_scm[i];
    }
    qsort(keys, np, sizeof(uint64_t), u64_cmp);
    /* arcs + complements, syncasm.c:264-282 (no filter applies: min_a_cov_f = 0, nothing deleted) */
    uint64_t na = 0, cap = 2 * np + 2;
    uint64_t *av = (uint64_t *) malloc(8 * cap), *aw = (uint64_t *) malloc(8 * cap);
    uint32_t *ac = (uint32_t *) malloc(4 * cap);
    uint8_t *acomp = (uint8_t *) malloc(cap);
    for (i = 0; i < np; ) {
        for (j = i; j < np && keys[j] == keys[i]; ++j) {}
        uint64_t v0 = keys[i] >> 32, v1 = keys[i] & 0xFFFFFFFFULL;
        av[na] = v0, aw[na] = v1, ac[na] = (uint32_t) (j - i), acomp[na] = 0, ++na;
        if ((v1 ^ 1) != v0) av[na] = v1 ^ 1, aw[na] = v0 ^ 1, ac[na] = (uint32_t) (j - i), acomp[na] = 1, ++na;
        i = j;
    }
    /* sort by (v, w), graph.c:70-83; (v, w) pairs are unique except in the multi-arc corner, which is flagged */
    uint64_t *ord = (uint64_t *) malloc(8 * (na + 1));
    for (i = 0; i < na; ++i) ord[i] = (av[i] << 32 | aw[i]);
    /* sort indices by key: pack key and index (na < 2^31 in tests) */
    typedef struct { uint64_t k; uint64_t idx; } ki_t;
    ki_t *ki = (ki_t *) malloc(sizeof(ki_t) * (na + 1));
    for (i = 0; i < na; ++i) ki[i].k = ord[i], ki[i].idx = i;
    qsort(ki, na, sizeof(ki_t), u64_cmp);              /* key is the first member */
    g->n_vtx = c->n_scm, g->n_arc = na;
    g->arc_v = (uint64_t *) malloc(8 * (na + 1)); g->arc_w = (uint64_t *) malloc(8 * (na + 1)); g->arc_ls = (uint64_t *) calloc(na + 1, 8);
    g->arc_cov = (uint32_t *) malloc(4 * (na + 1)); g->arc_comp = (uint8_t *) malloc(na + 1);
    g->idx_p = (uint64_t *) calloc(2 * c->n_scm + 1, 8); g->idx_n = (uint64_t *) calloc(2 * c->n_scm + 1, 8);
    for (i = 0; i < na; ++i) {
        uint64_t s = ki[i].idx;
        g->arc_v[i] = av[s], g->arc_w[i] = aw[s], g->arc_cov[i] = ac[s], g->arc_comp[i] = acomp[s];
        if (i && ki[i].k == ki[i - 1].k) g->multi_arc = 1;
    }
    for (i = 0; i < na; ++i) { if (g->idx_n[g->arc_v[i]]++ == 0) g->idx_p[g->arc_v[i]] = i; }   /* asmg_arc_index, graph.c:85-113 */
    /* asmg_arc_fix_symm (graph.c:205-233) flips the flag of an arc that is its own complement (v -> v^1) */
    for (i = 0; i < na; ++i) if ((g->arc_w[i] ^ 1) == g->arc_v[i]) g->arc_comp[i] ^= 1;
    /* overlaps, syncasm.c:793-812: only for non-complement arcs; the complement gets the same value */
    for (i = 0; i < na; ++i) {
        if (g->arc_comp[i]) continue;
        int64_t l = overlap_mode(c, scm_off, m_pos, g->arc_v[i], g->arc_w[i]);
        if (l < K) l = l < 0? K : K - l;               /* scg_syncmer_consensus(beg = l) returns |beg<0| + K - max(beg,0); then MIN with len = K */
        else l = 0;
        g->arc_ls[i] = (uint64_t) l;
        uint64_t cv = g->arc_w[i] ^ 1, cw = g->arc_v[i] ^ 1, p = g->idx_p[cv], n = g->idx_n[cv], t;
        for (t = 0; t < n; ++t) if (g->arc_w[p + t] == cw) { g->arc_ls[p + t] = (uint64_t) l; break; }    /* asmg_arc(): first match */
    }
    free(scm_off); free(keys); free(av); free(aw); free(ac); free(acomp); free(ord); free(ki);
    return g;
}

void orc_ecgraph_free(orc_ecgraph_t *g)
{
    if (!g) return;
    free(g->arc_v); free(g->arc_w); free(g->arc_ls); free(g->arc_cov); free(g->arc_comp); free(g->idx_p); free(g->idx_n);
    free(g);
}
