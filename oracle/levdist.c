/*
 * oracle/levdist.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates the extension-mode diagonal-transition (Landau-Vishkin) edit
 * distance of oatk's syncasm error correction (reference levdist.c:265-310
 * `wf_ed_core`, :156-224 `wf_step_basic`, :75-96 `wf_extend`, :99-113
 * `wf_prune_bw`), including the property the DFS relies on: the wavefront is
 * RESUMABLE -- after a call with query prefix qs[0,ql) it can be advanced with
 * a longer query that extends the previous one (syncerr.c:165-171,192-195).
 *
 * Wavefront entry (d, k): diagonal d = query index - target index, k = index
 * of the last target character matched on that diagonal (-1 before any).
 *
 * Also: orc_ed_bruteforce(), the closed form the wavefront result equals
 * whenever it is <= bw (full DP; min over last row and last column including
 * the empty-prefix corners; ties -> smallest diagonal).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

struct orc_wf {
    char *ts;
    int32_t tl, bw;
    int32_t score, t_end, q_end;
    int32_t n, cap;
    int32_t *d, *k;       /* current wavefront, ascending d */
    int32_t *nd, *nk;     /* scratch for the next one */
};

static void wf_reserve(orc_wf_t *w, int32_t need)
{
    if (need <= w->cap) return;
    w->cap = need * 2 + 16;
    w->d = (int32_t *) realloc(w->d, sizeof(int32_t) * w->cap);
    w->k = (int32_t *) realloc(w->k, sizeof(int32_t) * w->cap);
    w->nd = (int32_t *) realloc(w->nd, sizeof(int32_t) * w->cap);
    w->nk = (int32_t *) realloc(w->nk, sizeof(int32_t) * w->cap);
}

/* initial state as set up by the caller, syncerr.c:465-482: one diagonal d=0, k=-1, score 0 */
orc_wf_t *orc_wf_new(const char *ts, int32_t tl, int32_t bw)
{
    orc_wf_t *w = (orc_wf_t *) calloc(1, sizeof(orc_wf_t));
    w->ts = (char *) malloc((size_t) tl + 1);
    memcpy(w->ts, ts, tl);
    w->tl = tl, w->bw = bw;
    wf_reserve(w, 8);
    w->n = 1, w->d[0] = 0, w->k[0] = -1;
    return w;
}

void orc_wf_free(orc_wf_t *w)
{
    if (!w) return;
    free(w->ts); free(w->d); free(w->k); free(w->nd); free(w->nk); free(w);
}

/* one wavefront step; returns 1 when an end was reached (levdist.c:156-224) */
static int wf_step(orc_wf_t *w, const char *qs, int32_t ql)
{
    const char *ts = w->ts;
    int32_t tl = w->tl, n = w->n, j;
    w->t_end = w->q_end = -1;
    /* extend along exact matches, lowest diagonal first; stop at the FIRST diagonal touching either end :163-175 */
    for (j = 0; j < n; ++j) {
        int32_t k = w->k[j], d = w->d[j];
        if (k >= tl || k + d >= ql) continue;
        int32_t lim = (ql - d < tl? ql - d : tl) - 1;
        while (k < lim && ts[k + 1] == qs[k + d + 1]) ++k;
        if (k + d == ql - 1 || k == tl - 1) {      /* extension mode: either end suffices */
            w->t_end = k, w->q_end = k + d;
            return 1;                              /* note: this diagonal's k is NOT stored (:171 precedes :174) */
        }
        w->k[j] = k;
    }
    /* next wavefront: diagonal d gets max(k[d-1], k[d]+1, k[d+1]+1) :178-205 */
    wf_reserve(w, n + 4);
    int32_t *a = w->k, *b = w->nk, *bd = w->nd, *ad = w->d;
    bd[0] = ad[0] - 1, b[0] = a[0] + 1;
    bd[1] = ad[0], b[1] = ((n == 1 || a[0] > a[1])? a[0] : a[1]) + 1;
    for (j = 1; j < n - 1; ++j) {
        int32_t k = a[j - 1];
        if (a[j] + 1 > k) k = a[j] + 1;
        if (a[j + 1] + 1 > k) k = a[j + 1] + 1;
        bd[j + 1] = ad[j], b[j + 1] = k;
    }
    if (n >= 2) bd[n] = ad[n - 1], b[n] = a[n - 2] > a[n - 1] + 1? a[n - 2] : a[n - 1] + 1;
    bd[n + 1] = ad[n - 1] + 1, b[n + 1] = a[n - 1];
    /* trimming :207-210 and wf_prune_bw :99-113 (extension mode) */
    int32_t st = 0, en = n + 2;
    if (w->bw < 0 || n < 2 * w->bw + 1) {
        if (bd[0] < -tl) ++st;
        if (bd[n + 1] > ql) --en;
    } else {
        int32_t lo = -w->bw > -tl? -w->bw : -tl;
        int32_t hi = w->bw > ql? w->bw : ql;       /* as written in the reference: the LARGER of bw and ql */
        while (bd[st] < lo) ++st;
        while (bd[en - 1] > hi) --en;
    }
    memcpy(w->d, bd + st, sizeof(int32_t) * (en - st));
    memcpy(w->k, b + st, sizeof(int32_t) * (en - st));
    w->n = en - st;
    return 0;
}

/* levdist.c:265-310: step until an end is hit or the score exceeds bw; results are 1-based lengths,
 * (0, 0) when the band was exhausted */
void orc_wf_step(orc_wf_t *w, const char *qs, int32_t ql, int32_t *out3)
{
    for (;;) {
        if (wf_step(w, qs, ql)) break;
        ++w->score;
        if (w->bw >= 0 && w->score > w->bw) break;
    }
    w->t_end += 1, w->q_end += 1;
    out3[0] = w->score, out3[1] = w->t_end, out3[2] = w->q_end;
}

void orc_wf_ed(int32_t tl, const char *ts, int32_t ql, const char *qs, int32_t bw, int32_t *out3)
{
    orc_wf_t *w = orc_wf_new(ts, tl, bw);
    orc_wf_step(w, qs, ql, out3);
    orc_wf_free(w);
}

/* snapshot / restore used by the error-correction DFS (syncerr.c:165-171, 277-284) */
orc_wf_t *orc_wf_clone(const orc_wf_t *w)
{
    orc_wf_t *c = (orc_wf_t *) calloc(1, sizeof(orc_wf_t));
    *c = *w;
    c->ts = (char *) malloc((size_t) w->tl + 1);
    memcpy(c->ts, w->ts, w->tl);
    c->d = c->k = c->nd = c->nk = 0; c->cap = 0;
    wf_reserve(c, w->n + 4);
    memcpy(c->d, w->d, sizeof(int32_t) * w->n);
    memcpy(c->k, w->k, sizeof(int32_t) * w->n);
    return c;
}

void orc_wf_state(const orc_wf_t *w, int32_t *out3)
{
    out3[0] = w->score, out3[1] = w->t_end, out3[2] = w->q_end;
}

void orc_ed_bruteforce(int32_t tl, const char *ts, int32_t ql, const char *qs, int32_t *out3)
{
    int32_t i, j, W = ql + 1;
    int32_t *D = (int32_t *) malloc(sizeof(int32_t) * (size_t) (tl + 1) * W);
    for (j = 0; j <= ql; ++j) D[j] = j;
    for (i = 1; i <= tl; ++i) {
        D[i * W] = i;
        for (j = 1; j <= ql; ++j) {
            int32_t v = D[(i - 1) * W + j - 1] + (ts[i - 1] != qs[j - 1]);
            if (D[(i - 1) * W + j] + 1 < v) v = D[(i - 1) * W + j] + 1;
            if (D[i * W + j - 1] + 1 < v) v = D[i * W + j - 1] + 1;
            D[i * W + j] = v;
        }
    }
    /* boundary cells by ascending diagonal d = j - i: (tl, tl+d) while tl+d <= ql, else (ql-d, ql) */
    int32_t best = INT32_MAX, bi = 0, bj = 0, d;
    for (d = -tl; d <= ql; ++d) {
        if (tl + d <= ql) i = tl, j = tl + d; else i = ql - d, j = ql;
        if (i < 0 || j < 0) continue;
        if (D[i * W + j] < best) best = D[i * W + j], bi = i, bj = j;
    }
    out3[0] = best, out3[1] = bi, out3[2] = bj;
    free(D);
}
