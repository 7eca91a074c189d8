/*
 * oracle/align.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU restatement of the read -> unitig alignment,
 * scg_ra_analysis_thread (alignment.c:180-594) as driven by scg_read_alignment (:596-691), over flat arrays.  Plain dynamic arrays, a
 * stable merge sort where the reference relies on glibc's, recursion for the backtrace like the reference's.  Pinned against the
 * compiled reference in tests/test_oracle_align.py and through tests/golden/align_*.npz.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct { uint64_t uid, next; uint32_t u_pos, s_pos; } hit_t;
typedef struct { uint64_t uid; uint32_t u_beg, u_end, s_beg, s_end, s_cnt; int64_t score0, score; uint32_t np, mp; uint32_t *prev; } frg_t;

static int hit_cmp(const void *a, const void *b)                          /* sr_scm_cmpfunc, alignment.c:93-107 */
{
    const hit_t *x = (const hit_t *) a, *y = (const hit_t *) b;
    if (x->uid != y->uid) return (x->uid > y->uid) - (x->uid < y->uid);
    if (x->s_pos != y->s_pos) return (x->s_pos > y->s_pos) - (x->s_pos < y->s_pos);
    return (x->u_pos > y->u_pos) - (x->u_pos < y->u_pos);
}
static int frg_le(const frg_t *x, const frg_t *y)                          /* sr_frg_cmpfunc, :109-119: x <= y */
{
    if (x->s_beg != y->s_beg) return x->s_beg < y->s_beg;
    return x->s_end <= y->s_end;
}
static void frg_msort(frg_t *a, frg_t *tmp, size_t n)                      /* stable, like glibc 2.35's qsort (msort.c) */
{
    if (n < 2) return;
    size_t h = n / 2, i = 0, j = h, k = 0;
    frg_msort(a, tmp, h); frg_msort(a + h, tmp, n - h);
    while (i < h && j < n) tmp[k++] = frg_le(&a[i], &a[j])? a[i++] : a[j++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(frg_t) * n);
}

typedef struct { orc_ra_out_t *o; const frg_t *f; uint32_t *stack; uint64_t n_syn, read; size_t first_aln; int64_t max_score; } bt_t;

static void out_reserve(orc_ra_out_t *o, uint64_t na, uint64_t nf)
{
    if (o->n_aln + na > o->m_aln) {
        o->m_aln = (o->n_aln + na) * 2 + 16;
        o->sid = (uint64_t *) realloc(o->sid, 8 * o->m_aln); o->n = (uint32_t *) realloc(o->n, 4 * o->m_aln); o->s = (double *) realloc(o->s, 8 * o->m_aln);
    }
    if (o->n_frg + nf > o->m_frg) {
        o->m_frg = (o->n_frg + nf) * 2 + 64;
        o->uid = (uint64_t *) realloc(o->uid, 8 * o->m_frg); o->u_beg = (uint64_t *) realloc(o->u_beg, 8 * o->m_frg); o->u_end = (uint64_t *) realloc(o->u_end, 8 * o->m_frg);
        o->s_beg = (uint32_t *) realloc(o->s_beg, 4 * o->m_frg); o->s_end = (uint32_t *) realloc(o->s_end, 4 * o->m_frg);
    }
}
static void backtrace(bt_t *b, uint32_t node, uint32_t len)                /* aln_frg_backtrace, :132-157, with the 90 % test of :541-548 */
{
    b->stack[len++] = node;
    const frg_t *f = &b->f[node];
    if (f->np == 0) {
        uint64_t s = 0;
        uint32_t t;
        for (t = 0; t < len; ++t) s += b->f[b->stack[t]].s_cnt;
        if ((double) s / b->n_syn < .9) return;
        orc_ra_out_t *o = b->o;
        out_reserve(o, 1, len);
        o->sid[o->n_aln] = b->read, o->n[o->n_aln] = len, o->s[o->n_aln] = 0;
        for (t = len; t-- > 0; ) {                                         /* the path is collected from its end: earliest fragment first */
            const frg_t *q = &b->f[b->stack[t]];
            o->uid[o->n_frg] = q->uid, o->u_beg[o->n_frg] = q->u_beg, o->u_end[o->n_frg] = q->u_end, o->s_beg[o->n_frg] = q->s_beg, o->s_end[o->n_frg] = q->s_end;
            ++o->n_frg;
        }
        ++o->n_aln;
    } else {
        uint32_t i;
        for (i = 0; i < f->np; ++i) backtrace(b, f->prev[i], len);
    }
}

static int64_t arc_ln(const orc_ra_graph_t *g, uint64_t v, uint64_t w)     /* asmg_arc1, graph.h:193-205 */
{
    uint64_t i, p = g->idx_p[v], n = g->idx_n[v];
    for (i = 0; i < n; ++i) if (g->arc_w[p + i] == w && !g->arc_del[p + i]) return (int64_t) g->arc_ln[p + i];
    return -1;
}

orc_ra_out_t *orc_read_alignment(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos, const orc_ra_graph_t *g,
                                 const int64_t *old_ra)
{
    orc_ra_out_t *o = (orc_ra_out_t *) calloc(1, sizeof(orc_ra_out_t));
    uint64_t r, co = 0;
    size_t m_hit = 0, m_frg = 0;
    hit_t *H = 0;
    frg_t *F = 0, *T = 0;
    uint32_t *stack = 0;
    for (r = 0; r < n_reads; co += n_scm[r], ++r) {
        const int64_t old = old_ra? old_ra[r] : 1;
        const uint64_t n = n_scm[r];
        uint64_t j, k;
        if ((old & 1) == 0 || n == 0) continue;
        size_t nh = 0, nf = 0;
        for (j = 0; j < n; ++j) {                                          /* :233-251 */
            const uint64_t s = k_mer[co + j] >> 1;
            for (k = g->su_off[s]; k < g->su_off[s + 1]; ++k) {
                if (nh == m_hit) m_hit = m_hit * 2 + 64, H = (hit_t *) realloc(H, sizeof(hit_t) * m_hit);
                const uint64_t u = g->su_uid[k] >> 1, t = (g->su_uid[k] & 1) ^ (m_pos[co + j] & 1);
                H[nh].uid = u << 1 | t, H[nh].u_pos = t? g->utg_n[u] - g->su_pos[k] - 1 : g->su_pos[k], H[nh].s_pos = (uint32_t) j, H[nh].next = 0xFFFFFFFFFFFFFFFEULL;
                ++nh;
            }
        }
        if (nh == 0) continue;
        qsort(H, nh, sizeof(hit_t), hit_cmp);                              /* total order: any sort gives the same array */
        for (j = 0; j < nh; ) {                                            /* :259-342 */
            const uint64_t u = H[j].uid;
            size_t p = j, g0, g1, g2;
            while (++p < nh && H[p].uid == u) {}
            for (g0 = j, g1 = j; g1 < p && H[g1].s_pos == H[g0].s_pos; ++g1) {}
            while (g1 < p) {
                for (g2 = g1; g2 < p && H[g2].s_pos == H[g1].s_pos; ++g2) {}
                size_t s1 = g0, t1 = g1;
                for (; s1 < g1; ++s1) {
                    while (t1 < g2 && H[t1].u_pos <= H[s1].u_pos) ++t1;
                    if (t1 < g2 && H[t1].u_pos > H[s1].u_pos) H[s1].next = (uint64_t) t1 << 1;
                }
                g0 = g1, g1 = g2;
            }
            for (k = j; k < p; ++k) {
                size_t s = k;
                if (H[s].next & 1) continue;
                const uint32_t u_beg = H[s].u_pos, s_beg = H[s].s_pos;
                uint32_t s_cnt = 1;
                int64_t u_gap = 0, s_gap = 0;
                for (;;) {
                    const uint64_t t = H[s].next >> 1;
                    if (t == 0x7FFFFFFFFFFFFFFFULL) break;
                    u_gap += llabs((int64_t) H[t].u_pos - (int64_t) H[s].u_pos) - 1, s_gap += llabs((int64_t) H[t].s_pos - (int64_t) H[s].s_pos) - 1;
                    H[s].next |= 1;
                    ++s_cnt;
                    s = t;
                }
                if (s_cnt == 1) continue;
                H[s].next |= 1;
                if (s_gap > u_gap) u_gap = s_gap;
                if (u_gap < 0) u_gap = 0;
                const int64_t score = (int64_t) s_cnt - u_gap;
                if (score < 0) continue;
                if (nf == m_frg) m_frg = m_frg * 2 + 64, F = (frg_t *) realloc(F, sizeof(frg_t) * m_frg), T = (frg_t *) realloc(T, sizeof(frg_t) * m_frg);
                frg_t f = {u, u_beg, H[s].u_pos, s_beg, H[s].s_pos, s_cnt, score, score, 0, 0, 0};
                F[nf++] = f;
            }
            for (k = j; k < p; ++k) {
                if (H[k].next != 0xFFFFFFFFFFFFFFFEULL) continue;
                if (nf == m_frg) m_frg = m_frg * 2 + 64, F = (frg_t *) realloc(F, sizeof(frg_t) * m_frg), T = (frg_t *) realloc(T, sizeof(frg_t) * m_frg);
                frg_t f = {u, H[k].u_pos, H[k].u_pos, H[k].s_pos, H[k].s_pos, 1, 1, 1, 0, 0, 0};
                F[nf++] = f;
            }
            j = p;
        }
        if (nf == 0) continue;
        frg_msort(F, T, nf);                                               /* :431 */
        for (j = 0; j < nf; ++j) {                                         /* :434-476 */
            const frg_t *f = &F[j];
            const int64_t p = f->s_end, score = f->score;
            if ((int64_t) g->utg_n[f->uid >> 1] - (int64_t) f->u_end - 1 > 0) continue;
            for (k = j + 1; k < nf; ++k) {
                frg_t *f1 = &F[k];
                if (f1->u_beg > 0) continue;
                const int64_t ln = arc_ln(g, f->uid, f1->uid);
                if (ln < 0) continue;
                const int64_t u_ovl = ln < p + 1? ln : p + 1, p1 = f1->s_beg;
                if (p1 > p + 1) break;
                if (p1 + u_ovl != p + 1) continue;
                const int64_t score1 = score + f1->score0 - u_ovl;
                if (score1 <= score || score1 < f1->score || (score1 == f1->score && f1->np == 0)) continue;
                if (score1 > f1->score) f1->score = score1, f1->np = 0;
                if (f1->np == f1->mp) f1->mp = f1->mp * 2 + 4, f1->prev = (uint32_t *) realloc(f1->prev, 4 * f1->mp);
                f1->prev[f1->np++] = (uint32_t) j;
            }
        }
        int64_t max_score = 0;
        for (j = 0; j < nf; ++j) if (F[j].score > max_score) max_score = F[j].score;
        if (max_score >= (old >> 1)) {                                     /* :505-513 */
            stack = (uint32_t *) realloc(stack, 4 * (nf + 1));
            bt_t b = {o, F, stack, n, r, o->n_aln, max_score};
            for (j = 0; j < nf; ++j) if (F[j].score >= max_score) backtrace(&b, (uint32_t) j, 0);
            const uint64_t n_a = o->n_aln - b.first_aln;
            for (j = b.first_aln; j < o->n_aln; ++j) o->s[j] = 1.0 / n_a + max_score;                 /* :575-576 */
            o->n_mapped += n_a > 0, o->n_unique += n_a == 1;
        }
        for (j = 0; j < nf; ++j) free(F[j].prev);
    }
    free(H); free(F); free(T); free(stack);
    return o;
}

void orc_ra_out_free(orc_ra_out_t *o)
{
    if (!o) return;
    free(o->sid); free(o->n); free(o->s); free(o->uid); free(o->u_beg); free(o->u_end); free(o->s_beg); free(o->s_end);
    free(o);
}
