/*
 * tests/trace/ec_trace.c -- DEVELOPMENT AID (CPU; test-side: it reads the oracle's structs; not part of any product library).
 *
 * What does the error-block search (dfs_search + wf_ed_core, syncerr.c:144-286, levdist.c:156-310) DO on the blocks that cost the
 * device seconds?  The search of oracle/ec.c with counters: arcs tried, dead ends, wavefront steps and their widths, bases compared,
 * lengths of the appended extensions, depth, levels that branch, distinct vertices / arcs visited, and how often a level is entered in a
 * state another level was entered in before (same vertex, same consensus string: the subtree below is then identical -- the hit rate
 * an exact memo would have).  One line per heavy block on `fo`.   Built and driven by tests/trace/ec_trace.py.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/oracle.h"

#define MAX_DFS_PATH 10000
#define MIN_ERR_SEQ_LEN 10
#define MIN_ERR_BASE 6

typedef struct {
    uint64_t tried, dead, steps, diag, cmp, ext_sum, ext_max, frames, levels, max_depth, memo_hit, memo_hit_arcs, succ_events;
    uint64_t n_hist[8];           /* wavefront width at a step: 1, 2, <=4, <=8, <=16, <=32, <=64, more */
    uint64_t steps_hist[8];       /* steps per arc: 1, 2, <=4, <=8, <=16, <=32, <=64, more */
    uint64_t ext_hist[8];         /* extension length: <=1, <=2, <=4, <=8, <=16, <=64, <=256, more */
    uint64_t post_cap_arcs;       /* arcs tried after the dead-end counter hit its cap */
    uint64_t depth_at_dead_sum;
    uint64_t ret_dist_sum, ret_cnt;    /* after a dead end: how many levels up is the next arc taken */
    uint64_t sub_steps[8], sub_cnt[8];                 /* at levels with several live arcs: wavefront steps in the subtree below the i-th live arc (7: seventh and later) */
    uint64_t cat_arcs[6], cat_steps[6], cat_diag[6];
    /* the extension's cost as ec_fused.hpp pays it, per step the slowest of the waves (56 diagonals each): window turns of 16 bases (the first, up to three more lane by
     * lane), lanes that still go on after 64 bases (run down one after the other by the whole wave), and the turns windows of 64 bases would take */
    uint64_t x_steps, x_w16_turns, x_long_lanes, x_long_steps, x_w64_turns, x_w64_long, x_act_lanes, x_ext_hist[8];   /* alive first / alive sibling / dead-by-score first / dead-by-score sibling / dead-otherwise first / sibling */
} ctr_t;

typedef struct { uint64_t *k; uint32_t *sub_arcs; size_t m, n; } memo_t;       /* open addressing: key -> arcs in the subtree below */

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static int bucket8(uint64_t v, const uint64_t *lim) { int i; for (i = 0; i < 7; ++i) if (v <= lim[i]) return i; return 7; }

typedef struct {
    const orc_graph_t *g;
    const char *ts;
    int32_t tl, bw;
    uint64_t sink;
    /* search state */
    char *cs; size_t cl, cm;
    uint64_t chash;               /* rolling hash of the consensus (so that the memo key is O(1)) */
    uint64_t *hstack;             /* hash before each level's append */
    int32_t n_path;
    /* wavefront: d0, n, k[] */
    int32_t *k, *nk; int32_t n, d0, cap;
    int32_t score, t_end, q_end;
    ctr_t *c;
    memo_t *memo;
    uint8_t *vseen; uint8_t *aseen; uint64_t nv_seen, na_seen; uint64_t *vlist, *alist; size_t vl_m, al_m;
    int32_t last_dead_depth;
} S;

static void wf_need(S *s, int32_t need)
{
    if (need <= s->cap) return;
    s->cap = need * 2 + 16;
    s->k = (int32_t *) realloc(s->k, sizeof(int32_t) * s->cap);
    s->nk = (int32_t *) realloc(s->nk, sizeof(int32_t) * s->cap);
}

static int wf_step(S *s, int32_t ql)
{
    static const uint64_t NL[7] = {1, 2, 4, 8, 16, 32, 64};
    const char *ts = s->ts, *qs = s->cs;
    const int32_t tl = s->tl, n = s->n;
    int32_t j;
    s->t_end = s->q_end = -1;
    s->c->steps++, s->c->diag += (uint64_t) n, s->c->n_hist[bucket8((uint64_t) n, NL)]++;
    uint64_t w_turn16 = 0, w_long = 0, w_turn64 = 0, w_long64 = 0, m_turn16 = 0, m_long = 0, m_turn64 = 0, m_long64 = 0;
    static const uint64_t XL[7] = {0, 15, 31, 63, 127, 255, 1023};
    for (j = 0; j < n; ++j) {
        int32_t k = s->k[j], d = s->d0 + j;
        if (j % 56 == 0) {
            if (w_turn16 > m_turn16) m_turn16 = w_turn16;
            if (w_long > m_long) m_long = w_long;
            if (w_turn64 > m_turn64) m_turn64 = w_turn64;
            if (w_long64 > m_long64) m_long64 = w_long64;
            w_turn16 = w_long = w_turn64 = w_long64 = 0;
        }
        if (k >= tl || k + d >= ql) continue;
        int32_t lim = (ql - d < tl? ql - d : tl) - 1;
        const int32_t kb = k;
        while (k < lim && ts[k + 1] == qs[k + d + 1]) ++k, s->c->cmp++;
        s->c->cmp++;
        {
            const uint64_t e = (uint64_t) (k - kb);                      /* bases matched */
            const uint64_t t16 = e / 16 + 1 < 4? e / 16 + 1 : 4, t64 = e / 64 + 1 < 4? e / 64 + 1 : 4;
            s->c->x_act_lanes++, s->c->x_ext_hist[bucket8(e, XL)]++;
            if (t16 > w_turn16) w_turn16 = t16;
            if (e >= 64) ++w_long;
            if (t64 > w_turn64) w_turn64 = t64;
            if (e >= 256) ++w_long64;
        }
        if (k + d == ql - 1 || k == tl - 1) { s->t_end = k, s->q_end = k + d; s->c->x_steps++; goto acc; }
        s->k[j] = k;
    }
    s->c->x_steps++;
acc:
    if (w_turn16 > m_turn16) m_turn16 = w_turn16;
    if (w_long > m_long) m_long = w_long;
    if (w_turn64 > m_turn64) m_turn64 = w_turn64;
    if (w_long64 > m_long64) m_long64 = w_long64;
    s->c->x_w16_turns += m_turn16, s->c->x_long_lanes += m_long, s->c->x_long_steps += m_long > 0, s->c->x_w64_turns += m_turn64, s->c->x_w64_long += m_long64;
    if (s->t_end >= 0) return 1;
    wf_need(s, n + 4);
    int32_t *a = s->k, *b = s->nk;
    for (j = 0; j < n + 2; ++j) {
        int32_t jj = j - 1, v = INT32_MIN;
        if (jj - 1 >= 0) v = a[jj - 1];
        if (jj >= 0 && jj < n && a[jj] + 1 > v) v = a[jj] + 1;
        if (jj + 1 < n && a[jj + 1] + 1 > v) v = a[jj + 1] + 1;
        b[j] = v;
    }
    int32_t st = 0, en = n + 2, nd0 = s->d0 - 1;
    if (s->bw < 0 || n < 2 * s->bw + 1) {
        if (nd0 < -tl) ++st;
        if (nd0 + n + 1 > ql) --en;
    } else {
        int32_t lo = -s->bw > -tl? -s->bw : -tl, hi = s->bw > ql? s->bw : ql;
        while (nd0 + st < lo) ++st;
        while (nd0 + en - 1 > hi) --en;
    }
    memmove(s->k, b + st, sizeof(int32_t) * (size_t) (en - st));
    s->n = en - st, s->d0 = nd0 + st;
    return 0;
}

static char comp(char c) { return c == 'A'? 'T' : c == 'C'? 'G' : c == 'G'? 'C' : c == 'T'? 'A' : c; }

static uint32_t *memo_slot(memo_t *m, uint64_t key, int *found)
{
    if (m->n * 2 >= m->m) {
        size_t om = m->m, i;
        uint64_t *ok = m->k; uint32_t *ov = m->sub_arcs;
        m->m = om? om * 2 : 1 << 10;
        m->k = (uint64_t *) calloc(m->m, 8), m->sub_arcs = (uint32_t *) calloc(m->m, 4);
        for (i = 0; i < om; ++i) if (ok[i]) { size_t h = mix64(ok[i]) & (m->m - 1); while (m->k[h]) h = (h + 1) & (m->m - 1); m->k[h] = ok[i], m->sub_arcs[h] = ov[i]; }
        free(ok), free(ov);
    }
    size_t h = mix64(key) & (m->m - 1);
    while (m->k[h] && m->k[h] != key) h = (h + 1) & (m->m - 1);
    *found = m->k[h] == key;
    if (!*found) m->k[h] = key, m->n++;
    return &m->sub_arcs[h];
}

/* frame-scoped memo WITH skipping (what the device could do exactly): a state (vertex, consensus) met before under the SAME nearest branching level, within
 * `hop_max` arcs of it, whose subtree held no in-band arrival and fits the budget that is left, is not searched again: its dead ends are added.  ECT_SKIP=1 switches it on. */
typedef struct { uint64_t key, frame; uint32_t dead, ok; } m2_t;
static m2_t *g_m2; static size_t g_m2_m, g_m2_n; static int g_skip = -1, g_hop_max = 4; static uint64_t g_frame_serial, g_m2_hits, g_m2_saved_dead;
static m2_t *m2_slot(uint64_t key, uint64_t frame)
{
    if (!g_m2_m) { g_m2_m = 1 << 20; g_m2 = (m2_t *) calloc(g_m2_m, sizeof(m2_t)); }
    size_t h = mix64(key ^ frame * 0x9E3779B97F4A7C15ULL) & (g_m2_m - 1);
    while (g_m2[h].key && !(g_m2[h].key == key && g_m2[h].frame == frame)) h = (h + 1) & (g_m2_m - 1);
    return &g_m2[h];
}
static uint64_t g_cur_frame; static int g_cur_hops;
static uint64_t g_fs_serial[4096]; static size_t g_fs_l0[4096]; static int g_fs_n; static size_t g_win = 2048;

/* ---- ECT_DP=1: is the search's wavefront alignment (wf_ed_core, resumed arc by arc) the banded edit-distance matrix in disguise?  Beside the wavefront a plain DP row per
 * query base is kept (rows are pushed and popped with the consensus), and after every arc the call's outcome is PREDICTED from the matrix alone: the score is the least value
 * on the matrix's boundary (the query's last row, the target's last column) but not below the parent's score, the end is the boundary cell of the LOWEST diagonal that has
 * reached that value, and the arc dies by score when that least value is beyond the band.  Counted: arcs where prediction and wavefront differ. ---- */
static int g_dp = -1;
static int32_t *g_dp_rows; static size_t g_dp_m;
static int32_t g_dp_R, g_dp_W;
static uint64_t g_dp_checked, g_dp_bad, g_dp_bad_dead, g_dp_cells;
#define DP_INF 1000000
static inline int32_t dp_get(const S *s, int32_t q, int32_t t)
{
    if (q < -1 || t < -1) return DP_INF;
    const int32_t d = q - t;
    if (d > g_dp_R || d < -g_dp_R) return DP_INF;
    if (q == -1) return t + 1;
    if (t == -1) return q + 1;
    if (t >= s->tl) return DP_INF;
    return g_dp_rows[(size_t) q * (size_t) g_dp_W + (size_t) (t - q + g_dp_R)];
}
static uint64_t g_dp_rows_all, g_dp_rows_cut, g_dp_cut_arcs;
static void dp_rows(const S *s, size_t from, size_t to)
{
    if (to * (size_t) g_dp_W > g_dp_m) { g_dp_m = to * (size_t) g_dp_W * 2; g_dp_rows = (int32_t *) realloc(g_dp_rows, g_dp_m * sizeof(int32_t)); }
    size_t q;
    int cut = 0;
    g_dp_rows_all += to - from;
    for (q = from; q < to; ++q) {
        int32_t t, lo = (int32_t) q - g_dp_R, hi = (int32_t) q + g_dp_R, rmin = DP_INF;
        int32_t *row = g_dp_rows + q * (size_t) g_dp_W;
        if (!cut) g_dp_rows_cut++;
        for (t = lo; t <= hi; ++t) {
            int32_t v = DP_INF;
            if (t >= 0 && t < s->tl) {
                const int32_t a = dp_get(s, (int32_t) q - 1, t - 1) + (s->ts[t] != s->cs[q]), b = dp_get(s, (int32_t) q - 1, t) + 1;
                const int32_t c = t - 1 >= lo? (t - 1 == -1? (int32_t) q + 1 : row[t - 1 - lo]) + 1 : DP_INF;
                v = a < b? a : b; v = v < c? v : c;
                if (v > DP_INF) v = DP_INF;
                g_dp_cells++;
                if (v < rmin) rmin = v;
            }
            row[t - lo] = v;
        }
        /* (a row whose least value is beyond the band: so is every later row's -- a solver that works by rows could stop here and call the arc dead, once the
         *  target's last column is out of reach of the rows that are left) */
        if (!cut && rmin > s->bw && (int32_t) to + s->bw < s->tl) cut = 1, g_dp_cut_arcs++;
    }
}
/* the call's predicted outcome: *score (bw + 1: dead by score), *t_end, *q_end as wf_ed_core leaves them (one past the last aligned base; 0 0 when it dies) */
static void dp_predict(const S *s, int32_t ql, int32_t parent_score, int32_t *score, int32_t *t_end, int32_t *q_end)
{
    int32_t best = DP_INF, t, q;
    const int32_t qr = ql - 1, tc = s->tl - 1;
    for (t = qr - g_dp_R < 0? 0 : qr - g_dp_R; t <= qr + g_dp_R && t < s->tl; ++t) { const int32_t v = dp_get(s, qr, t); if (v < best) best = v; }
    for (q = tc - g_dp_R < 0? 0 : tc - g_dp_R; q <= tc + g_dp_R && q < ql; ++q) { const int32_t v = dp_get(s, q, tc); if (v < best) best = v; }
    int32_t sc = best > parent_score? best : parent_score;
    if (sc > s->bw) { *score = s->bw + 1, *t_end = 0, *q_end = 0; return; }
    /* the lowest diagonal (q - t) whose boundary cell is within sc: on the last column diagonals descend with q, on the last row they descend as t grows */
    int32_t bd = DP_INF, bt = -1, bq = -1;
    for (t = qr - g_dp_R < 0? 0 : qr - g_dp_R; t <= qr + g_dp_R && t < s->tl; ++t) if (dp_get(s, qr, t) <= sc && qr - t < bd) bd = qr - t, bt = t, bq = qr;
    for (q = tc - g_dp_R < 0? 0 : tc - g_dp_R; q <= tc + g_dp_R && q < ql; ++q) if (dp_get(s, q, tc) <= sc && q - tc < bd) bd = q - tc, bt = tc, bq = q;
    *score = sc, *t_end = bt + 1, *q_end = bq + 1;
}

/* ---- ECT_CERT=1: how many of the arcs that die by score could be KNOWN to die without aligning anything?  By (ECT_DP) the call's outcome is the matrix's: an arc whose
 * appended string cannot be fitted ANYWHERE into the target within bw edits (semi-global edit distance of the string against the whole target, free ends in the target)
 * leaves every cell of the query's new last row beyond the band -- it dies by score, whatever the path before it, as long as the target's last column is out of the new
 * rows' reach.  That distance depends on (oriented vertex, overlap) and the block only: computed once per block and arc target, here by plain DP. ---- */
static int g_cert = -1;
static int32_t *g_cert_m; static uint64_t *g_cert_key; static size_t g_cert_n, g_cert_cap;       /* per block: key (w, ls) -> least cost (capped at bw + 1) */
static uint64_t g_cert_arcs, g_cert_steps, g_cert_wrong, g_cert_tables, g_cert_cells, g_cert_dead_arcs, g_cert_dead_steps;
static int32_t cert_cost(const S *s, uint64_t key, const char *ext, int32_t m)
{
    size_t i;
    for (i = 0; i < g_cert_n; ++i) if (g_cert_key[i] == key) return g_cert_m[i];
    if (g_cert_n == g_cert_cap) { g_cert_cap = g_cert_cap? 2 * g_cert_cap : 64; g_cert_key = (uint64_t *) realloc(g_cert_key, 8 * g_cert_cap); g_cert_m = (int32_t *) realloc(g_cert_m, 4 * g_cert_cap); }
    const int32_t tl = s->tl;
    int32_t *prev = (int32_t *) malloc(sizeof(int32_t) * (size_t) (tl + 1)), *cur = (int32_t *) malloc(sizeof(int32_t) * (size_t) (tl + 1)), r, t, best;
    for (t = 0; t <= tl; ++t) prev[t] = 0;                              /* the string may begin anywhere in the target */
    for (r = 1; r <= m; ++r) {
        cur[0] = r;
        int32_t rmin = cur[0];
        for (t = 1; t <= tl; ++t) {
            int32_t v = prev[t - 1] + (ext[r - 1] != s->ts[t - 1]);
            if (prev[t] + 1 < v) v = prev[t] + 1;
            if (cur[t - 1] + 1 < v) v = cur[t - 1] + 1;
            cur[t] = v;
            if (v < rmin) rmin = v;
        }
        g_cert_cells += (uint64_t) tl;
        { int32_t *x = prev; prev = cur, cur = x; }
        if (rmin > s->bw) { r = m + 1; break; }                           /* (row minima only grow) */
    }
    best = s->bw + 1;
    if (r == m + 1 && 0) {}
    else for (t = 0; t <= tl; ++t) if (prev[t] < best) best = prev[t];
    free(prev), free(cur);
    g_cert_key[g_cert_n] = key, g_cert_m[g_cert_n] = best, ++g_cert_n, ++g_cert_tables;
    return best;
}

/* ---- ECT_ROWS=1 (with ECT_DP=1): what a solver that keeps the matrix's last ROW instead of a wavefront would have to compute.  A short appended string costs its rows.
 * A long one (>= 32 bases) is first asked whether it can be alive at all: the least value of the new last row is min over t' of (parent's row at t' + the least cost of
 * fitting the string into the target FROM t' on) -- the second term a table per (vertex, overlap) and block (one pass over the target), the minimum one pass over the band;
 * beyond the band: the arc dies by score, no row computed.  Only when the target's last column is within the new rows' reach, or the arc is alive, its rows are computed.
 * Counted: row-equivalents, and arcs where this prediction of "dies by score" differs from the wavefront's (must be none). ---- */
static int g_rows = -1;
static int32_t **g_prof; static uint64_t *g_prof_key; static size_t g_prof_n, g_prof_cap;
static uint64_t g_rw_rows, g_rw_tests, g_rw_dead_by_test, g_rw_wrong, g_rw_tables;
static uint64_t g_lvc_arcs, g_lvc_steps, g_lvc_wrong, g_lvc_dead_arcs, g_lvc_dead_steps, g_rw_near;
/* (the kernels skip the alignment of a vertex on an unbranched stretch -- ec_wave.hpp, DESIGN.md 8.3 -- so their wavefront may stand for a shorter consensus than the one a
 *  long string is appended to: then they do not ask.  g_aligned = the consensus length the kernel's wavefront would stand for; g_lvk_*: what the kernel's switch would know) */
static int64_t g_aligned;
static uint64_t g_lvk_arcs, g_lvk_steps, g_lvk_lag;
/* round 6: the same with a reached cell priced at the parent's SCORE (the call ended in the step that first reached the row): g_lv2_*; and how often that price is the cell's true value */
static uint64_t g_lv2_arcs, g_lv2_steps, g_lv2_wrong, g_lv2k_arcs, g_lv2k_steps, g_hyp_cells, g_hyp_below, g_hyp_above;
static const int32_t *prof_of(const S *s, uint64_t key, const char *ext, int32_t m)
{
    size_t i;
    for (i = 0; i < g_prof_n; ++i) if (g_prof_key[i] == key) return g_prof[i];
    if (g_prof_n == g_prof_cap) { g_prof_cap = g_prof_cap? 2 * g_prof_cap : 64; g_prof_key = (uint64_t *) realloc(g_prof_key, 8 * g_prof_cap); g_prof = (int32_t **) realloc(g_prof, sizeof(int32_t *) * g_prof_cap); }
    const int32_t tl = s->tl;
    /* both strings backwards: F[r][j] = least cost of the last r bases of ext against a piece of the target that ENDS (backwards) at j, i.e. begins (forwards) at tl - j */
    int32_t *prev = (int32_t *) malloc(sizeof(int32_t) * (size_t) (tl + 1)), *cur = (int32_t *) malloc(sizeof(int32_t) * (size_t) (tl + 1)), r, j;
    for (j = 0; j <= tl; ++j) prev[j] = 0;
    for (r = 1; r <= m; ++r) {
        cur[0] = r;
        for (j = 1; j <= tl; ++j) {
            int32_t v = prev[j - 1] + (ext[m - r] != s->ts[tl - j]);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        { int32_t *x = prev; prev = cur, cur = x; }
    }
    /* prof[t' + 1], t' = -1 .. tl - 1: the string fitted into target[t' + 1 ..]: begins at tl - j = t' + 1 */
    int32_t *prof = (int32_t *) malloc(sizeof(int32_t) * (size_t) (2 * (tl + 1)));
    for (j = 0; j <= tl; ++j) prof[tl - j] = prev[j];
    /* ... and behind it prof2[u], u = t' + 1: the least cost of some PREFIX of the string against target[u ..] TO ITS END (what a cell of the target's last column in one of
     * the new rows costs on top of the parent's row): G[i][u] = min(G[i+1][u+1] + mismatch, G[i+1][u] + 1, G[i][u+1] + 1), G[i][tl] = 0, G[m][u] = tl - u */
    {
        int32_t *g1 = prev, *g0 = cur, i, u;
        for (u = 0; u <= tl; ++u) g1[u] = tl - u;                      /* i = m */
        for (i = m - 1; i >= 0; --i) {
            g0[tl] = 0;
            for (u = tl - 1; u >= 0; --u) {
                int32_t v = g1[u + 1] + (ext[i] != s->ts[u]);
                if (g1[u] + 1 < v) v = g1[u] + 1;
                if (g0[u + 1] + 1 < v) v = g0[u + 1] + 1;
                g0[u] = v;
            }
            { int32_t *x = g1; g1 = g0, g0 = x; }
        }
        for (u = 0; u <= tl; ++u) prof[tl + 1 + u] = g1[u];
        prev = g1, cur = g0;
    }
    free(prev), free(cur);
    g_prof_key[g_prof_n] = key, g_prof[g_prof_n] = prof, ++g_prof_n, ++g_rw_tables;
    return prof;
}

/* ---- ECT_SLOTS=1: the row solver's state as a kernel would hold it -- one value per DIAGONAL (slot = diagonal + bw + 2, like the wavefront's slots), nothing else: the
 * cell of the current row on that diagonal, or, once the diagonal has run past the target's last column, the value it had THERE (frozen).  Every diagonal then holds its one
 * boundary cell when the arc's last row is done, and the call's outcome is read off the slots: least value (not below the parent's score), lowest slot within it.  Saved and
 * restored per level like the wavefront.  Checked against the wavefront arc by arc. ---- */
static int g_slots = -1;
static int32_t *g_sl; static int32_t g_sl_n, g_sl_off;
static uint64_t g_sl_checked, g_sl_bad;
static void slots_init(const S *s)
{
    int32_t i;
    g_sl_off = s->bw + 2, g_sl_n = 2 * g_sl_off + 1;
    g_sl = (int32_t *) realloc(g_sl, sizeof(int32_t) * (size_t) (g_sl_n + 2));
    for (i = 0; i < g_sl_n; ++i) { const int32_t d = i - g_sl_off, t = -1 - d; g_sl[i] = t >= -1 && t < s->tl? t + 1 : DP_INF; }       /* the row before the first: D(-1, t) = t + 1 */
}
static void slots_rows(const S *s, size_t from, size_t to)
{
    size_t qq;
    const int32_t tl = s->tl;
    for (qq = from; qq < to; ++qq) {
        const int32_t q = (int32_t) qq;
        const char c = s->cs[q];
        int32_t i, carry = DP_INF;                       /* the new value of the slot above (the cell to the left in the row) */
        for (i = g_sl_n - 1; i >= 0; --i) {
            const int32_t d = i - g_sl_off, t = q - d;
            int32_t v;
            if (t >= tl) break;                          /* this diagonal and all below it have left the matrix: they keep what they had on its last column */
            if (t < -1) v = DP_INF;
            else if (t == -1) v = q + 1;
            else {
                const int32_t a = g_sl[i] + (s->ts[t] != c), b = (i >= 1? g_sl[i - 1] : DP_INF) + 1;      /* (old values: slot i is read before it is written, slot i - 1 is written later) */
                v = a < b? a : b;
                if (carry + 1 < v) v = carry + 1;
                if (v > DP_INF) v = DP_INF;
            }
            g_sl[i] = v, carry = v;
        }
    }
}
static void slots_predict(const S *s, int32_t ql, int32_t parent_score, int32_t *score, int32_t *t_end, int32_t *q_end)
{
    int32_t i, best = DP_INF;
    for (i = 0; i < g_sl_n; ++i) { const int32_t d = i - g_sl_off, t = ql - 1 - d; if (t >= 0 && g_sl[i] < best) best = g_sl[i]; }       /* (t >= tl: frozen on the last column; t = -1: no cell of the target) */
    const int32_t sc = best > parent_score? best : parent_score;
    if (sc > s->bw) { *score = s->bw + 1, *t_end = 0, *q_end = 0; return; }
    for (i = 0; i < g_sl_n; ++i) {
        const int32_t d = i - g_sl_off, t = ql - 1 - d;
        if (t < 0 || g_sl[i] > sc) continue;
        *score = sc;
        if (t < s->tl) *t_end = t + 1, *q_end = ql;
        else *t_end = s->tl, *q_end = s->tl + d;
        return;
    }
}

static void dfs(S *s, uint64_t source, int depth)
{
    static const uint64_t SL[7] = {1, 2, 4, 8, 16, 32, 64}, EL[7] = {1, 2, 4, 8, 16, 64, 256};
    if (s->n_path >= MAX_DFS_PATH) return;
    if (g_skip < 0) { g_skip = getenv("ECT_SKIP") != 0; if (getenv("ECT_HOPS")) g_hop_max = atoi(getenv("ECT_HOPS")); }
    m2_t *m2 = 0;
    uint64_t my_frame = 0; const int my_hops = 0;
    if (getenv("ECT_WIN")) g_win = (size_t) atoi(getenv("ECT_WIN"));
    { int i_; for (i_ = 0; i_ < g_fs_n; ++i_) if (g_fs_l0[i_] + g_win >= s->cl) { my_frame = g_fs_serial[i_]; break; } }      /* the shallowest live branching level within the window */
    const int32_t np_at_entry = s->n_path; const uint64_t succ_at_entry = s->c->succ_events;
    if (g_skip && my_frame && my_hops <= g_hop_max && (!g_m2_m || g_m2_n * 2 < g_m2_m)) {
        m2 = m2_slot(mix64(source * 0x9E3779B97F4A7C15ULL ^ s->chash ^ ((uint64_t) s->cl << 40)) | 1ULL, my_frame);
        if (m2->key) {
            if (m2->ok && s->n_path + (int32_t) m2->dead < MAX_DFS_PATH) { s->n_path += (int32_t) m2->dead, s->c->dead += m2->dead, g_m2_hits++, g_m2_saved_dead += m2->dead; return; }
            m2 = 0;
        }
    }
    const orc_graph_t *g = s->g;
    const size_t l0 = s->cl;
    const uint64_t p = g->idx_p[source], na = g->idx_n[source], h0 = s->chash;
    const int32_t n0 = s->n, d00 = s->d0, sc0 = s->score, te0 = s->t_end, qe0 = s->q_end;
    int32_t *sv = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n0 + 1));
    memcpy(sv, s->k, sizeof(int32_t) * (size_t) n0);
    const int64_t al0 = g_aligned;
    if (g_slots < 0) g_slots = getenv("ECT_SLOTS") != 0;
    int32_t *sl_sv = 0;
    if (g_slots) { sl_sv = (int32_t *) malloc(sizeof(int32_t) * (size_t) g_sl_n); memcpy(sl_sv, g_sl, sizeof(int32_t) * (size_t) g_sl_n); }
    uint64_t i, live = 0; int live_seen = 0;
    for (i = 0; i < na; ++i) if (!g->arc_del[p + i]) ++live;
    s->c->levels++;
    if (live > 1) s->c->frames++;
    const uint64_t child_frame = 0; const int child_hops = 0;
    if (live > 1 && g_fs_n < 4096) g_fs_serial[g_fs_n] = ++g_frame_serial, g_fs_l0[g_fs_n] = l0, ++g_fs_n;
    if ((uint64_t) depth > s->c->max_depth) s->c->max_depth = (uint64_t) depth;
    /* memo probe: (vertex, consensus) -- with the same string the wavefront, score and ends are the same too */
    int found = 0;
    uint64_t arcs_before = s->c->tried;
    uint32_t *slot = memo_slot(s->memo, mix64(source * 0x9E3779B97F4A7C15ULL ^ h0 ^ ((uint64_t) l0 << 40)) | 1ULL, &found);
    size_t slot_idx = (size_t) (slot - s->memo->sub_arcs);
    if (found) s->c->memo_hit++, s->c->memo_hit_arcs += *slot;
    if (found && getenv("ECT_DEBUG") && s->c->memo_hit <= 6 && s->tl == 7746) {
        int i_; fprintf(stderr, "[hit] vertex %llu l0 %zu depth %d n_path %d sub_arcs %u my_frame %llu skipflag %d; live frames:", (unsigned long long) source, l0, depth, s->n_path, *slot, (unsigned long long) my_frame, g_skip);
        for (i_ = 0; i_ < g_fs_n; ++i_) fprintf(stderr, " (%llu, l0 %zu)", (unsigned long long) g_fs_serial[i_], g_fs_l0[i_]);
        fprintf(stderr, "\n");
    }
    for (i = 0; i < na; ++i) {
        if (g->arc_del[p + i]) continue;
        const uint64_t w = g->arc_w[p + i];
        const int64_t ls = (int64_t) g->arc_ls[p + i], l_seq = (int64_t) g->vtx_len[w >> 1];
        const char *k_seq = g->seq + g->vtx_seq_off[w >> 1];
        const size_t ext = (size_t) (l_seq - ls);
        if (s->n_path >= MAX_DFS_PATH) s->c->post_cap_arcs++;
        if (s->last_dead_depth >= 0) { s->c->ret_dist_sum += (uint64_t) (s->last_dead_depth - depth), s->c->ret_cnt++; s->last_dead_depth = -1; }
        s->c->tried++;
        s->c->ext_sum += ext; if (ext > s->c->ext_max) s->c->ext_max = ext;
        s->c->ext_hist[bucket8(ext, EL)]++;
        if (!s->vseen[w >> 1]) { s->vseen[w >> 1] = 1; if (s->nv_seen == s->vl_m) { s->vl_m = s->vl_m? 2 * s->vl_m : 1024; s->vlist = (uint64_t *) realloc(s->vlist, 8 * s->vl_m); } s->vlist[s->nv_seen++] = w >> 1; }
        if (!s->aseen[p + i]) { s->aseen[p + i] = 1; if (s->na_seen == s->al_m) { s->al_m = s->al_m? 2 * s->al_m : 1024; s->alist = (uint64_t *) realloc(s->alist, 8 * s->al_m); } s->alist[s->na_seen++] = p + i; }
        if (s->cl + ext + 1 > s->cm) { s->cm = (s->cl + ext + 1) * 2; s->cs = (char *) realloc(s->cs, s->cm); }
        size_t t;
        if (w & 1) for (t = 0; t < ext; ++t) s->cs[s->cl + t] = comp(k_seq[ext - 1 - t]);
        else memcpy(s->cs + s->cl, k_seq + ls, ext);
        for (t = 0; t < ext; ++t) s->chash = (s->chash ^ (uint64_t) (unsigned char) s->cs[s->cl + t]) * 0x100000001B3ULL;
        s->cl += ext;
        const int32_t ql = (int32_t) s->cl;
        uint64_t st_before = s->c->steps, dg_before = s->c->diag; const int sib = live_seen++ > 0;
        const int64_t al_before = g_aligned;
        {   /* would the kernel align here?  (ec_fused.hpp: no success so far, a middle block whose end this is not, one way on, the counter below its cap, lengths in range) */
            uint64_t w_live = 0, q_;
            for (q_ = 0; q_ < g->idx_n[w]; ++q_) if (!g->arc_del[g->idx_p[w] + q_]) ++w_live;
            const int skip = s->c->succ_events == 0 && s->sink != UINT64_MAX && s->sink != w && w_live == 1 && s->n_path < MAX_DFS_PATH && (int64_t) s->cl - l_seq <= (int64_t) s->tl + s->bw && (int64_t) s->cl >= s->bw + 3;
            if (!skip) g_aligned = (int64_t) s->cl;
        }
        for (;;) {
            if (wf_step(s, ql)) break;
            ++s->score;
            if (s->score > s->bw) break;
        }
        s->c->steps_hist[bucket8(s->c->steps - st_before, SL)]++;
        s->t_end += 1, s->q_end += 1;
        if (g_slots && ext > 0) {
            int32_t ps = 0, pt = 0, pq = 0;
            slots_rows(s, l0, s->cl);
            slots_predict(s, ql, sc0, &ps, &pt, &pq);
            const int dead_a = s->score > s->bw, dead_p = ps > s->bw;
            g_sl_checked++;
            if (dead_a != dead_p || (!dead_a && (ps != s->score || pt != s->t_end || pq != s->q_end))) {
                if (g_sl_bad++ < 12) fprintf(stderr, "[slots] tl %d bw %d ql %d ext %zu parent score %d: wavefront score %d t_end %d q_end %d | slots score %d t_end %d q_end %d\n", s->tl, s->bw, ql, ext, sc0, s->score, s->t_end, s->q_end, ps, pt, pq);
            }
        }
        if (g_cert < 0) g_cert = getenv("ECT_CERT") != 0;
        if (g_cert && ext >= 32) {
            /* (the last column out of the new rows' reach: no cell (q, tl - 1) with q < ql lies within bw of the diagonal) */
            const int far = ql - 1 + s->bw < s->tl - 1;
            const int dead_by_score = s->score > s->bw;
            if (dead_by_score) g_cert_dead_arcs++, g_cert_dead_steps += s->c->steps - st_before;
            if (far) {
                const int32_t m = cert_cost(s, (w << 20) ^ (uint64_t) ls, s->cs + l0, (int32_t) ext);
                if (m > s->bw) {
                    g_cert_arcs++, g_cert_steps += s->c->steps - st_before;
                    if (!dead_by_score) { g_cert_wrong++; if (g_cert_wrong < 6) fprintf(stderr, "[cert] WRONG: tl %d bw %d ql %d ext %zu least cost %d, wavefront score %d\n", s->tl, s->bw, ql, ext, m, s->score); }
                }
            }
        }
        if (g_dp < 0) g_dp = getenv("ECT_DP") != 0;
        if (g_rows < 0) g_rows = getenv("ECT_ROWS") != 0;
        if (g_dp && g_rows && ext > 0) {
            const int dead_a = s->score > s->bw;
            const int far = ql - 1 + s->bw < s->tl - 1;
            const int near_ok = !far && (int32_t) l0 - 1 + g_dp_R < s->tl - 1;      /* the last column is within the NEW rows' reach only: no cell of it in the rows before */
            if (near_ok && ext >= 32) g_rw_near++;
            if (ext >= 32 && (far || near_ok)) {
                const int32_t *prof0 = prof_of(s, (w << 20) ^ (uint64_t) ls, s->cs + l0, (int32_t) ext);
                int32_t *prof = (int32_t *) prof0, *pmix = 0;
                if (near_ok) { int32_t u; pmix = (int32_t *) malloc(sizeof(int32_t) * (size_t) (s->tl + 1)); for (u = 0; u <= s->tl; ++u) pmix[u] = prof0[u] < prof0[s->tl + 1 + u]? prof0[u] : prof0[s->tl + 1 + u]; prof = pmix; }
                int32_t t, best = DP_INF;
                const int32_t qp = (int32_t) l0 - 1;
                for (t = qp - g_dp_R < -1? -1 : qp - g_dp_R; t <= qp + g_dp_R && t < s->tl; ++t) { const int32_t v = dp_get(s, qp, t) + prof[t + 1]; if (v < best) best = v; }
                g_rw_tests++;
                {   /* the same test with what a WAVEFRONT knows of the parent's row instead of the row itself: a cell the parent's wavefront has passed (t' <= its k on that
                     * diagonal) costs at least |diagonal|, one it has not reached at least the parent's score + 1 */
                    int32_t lb_best = DP_INF, lb2_best = DP_INF;
                    for (t = qp - g_dp_R < -1? -1 : qp - g_dp_R; t <= qp + g_dp_R && t < s->tl; ++t) {
                        const int32_t d = qp - t, j = d - d00;
                        int reached = 0;
                        if (j >= 0 && j < n0) {                        /* (the parent's call stored only the diagonals below the one it ended on extended: the others are extended here, as its child's first step would) */
                            int32_t k = sv[j];
                            const int32_t qlp = (int32_t) l0;
                            if (!(k >= s->tl || k + d >= qlp)) { const int32_t lim = (qlp - d < s->tl? qlp - d : s->tl) - 1; while (k < lim && s->ts[k + 1] == s->cs[k + d + 1]) ++k; }
                            reached = k >= t;
                        }
                        int32_t lb = d < 0? -d : d;
                        if (!reached && sc0 + 1 > lb) lb = sc0 + 1;
                        if (lb + prof[t + 1] < lb_best) lb_best = lb + prof[t + 1];
                        {
                            int32_t lb2 = d < 0? -d : d;
                            const int32_t floor2 = reached? sc0 : sc0 + 1;
                            if (floor2 > lb2) lb2 = floor2;
                            if (lb2 + prof[t + 1] < lb2_best) lb2_best = lb2 + prof[t + 1];
                            if (reached) { const int32_t tv = dp_get(s, qp, t); g_hyp_cells++; if (tv < sc0) { if (g_hyp_below++ < 8) fprintf(stderr, "[hyp] reached cell BELOW the parent's score: tl %d bw %d l0 %d t %d d %d value %d score %d\n", s->tl, s->bw, (int) l0, t, d, tv, sc0); } else if (tv > sc0) g_hyp_above++; }
                        }
                    }
                    if (lb2_best > s->bw) { g_lv2_arcs++, g_lv2_steps += s->c->steps - st_before; if (!dead_a) { if (g_lv2_wrong++ < 8) fprintf(stderr, "[lv2] WRONG: tl %d bw %d l0 %d ext %zu parent score %d, bound %d, wavefront score %d\n", s->tl, s->bw, (int) l0, ext, sc0, lb2_best, s->score); } }
                    if (lb2_best > s->bw && ext <= 1024 && l0 >= 1 && al_before == (int64_t) l0) g_lv2k_arcs++, g_lv2k_steps += s->c->steps - st_before;
                    if (lb_best > s->bw) { g_lvc_arcs++, g_lvc_steps += s->c->steps - st_before; if (!dead_a) g_lvc_wrong++; }
                    if (lb_best > s->bw && ext <= 1024 && l0 >= 1) { if (al_before == (int64_t) l0) g_lvk_arcs++, g_lvk_steps += s->c->steps - st_before; else g_lvk_lag++; }
                    if (dead_a) g_lvc_dead_arcs++, g_lvc_dead_steps += s->c->steps - st_before;
                }
                const int dead_p = best > s->bw || (best < sc0? sc0 : best) > s->bw;
                if (dead_p != dead_a) { if (g_rw_wrong++ < 8) fprintf(stderr, "[rows] tl %d bw %d ql %d ext %zu parent score %d: least value of the new last row by the table %d, wavefront score %d\n", s->tl, s->bw, ql, ext, sc0, best, s->score); }
                if (dead_p) g_rw_dead_by_test++, g_rw_rows += 2;            /* (a pass over the band: about two rows' worth) */
                else if (far) g_rw_rows += ext;
                else { const int64_t reach = (int64_t) s->tl + s->bw - (int64_t) l0; g_rw_rows += reach > 0 && (uint64_t) reach < ext? (uint64_t) reach : ext; }
                free(pmix);
            } else if (ext >= 32) {
                const int64_t reach = (int64_t) s->tl + s->bw - (int64_t) l0;    /* rows beyond which the band has left the matrix */
                g_rw_rows += reach > 0 && (uint64_t) reach < ext? (uint64_t) reach : ext;
            } else g_rw_rows += ext;
        }
        if (g_dp && ext > 0) {
            int32_t ps, pt, pq;
            dp_rows(s, l0, s->cl);
            dp_predict(s, ql, sc0, &ps, &pt, &pq);
            const int dead_a = s->score > s->bw, dead_p = ps > s->bw;
            g_dp_checked++;
            if (dead_a != dead_p || (!dead_a && (ps != s->score || pt != s->t_end || pq != s->q_end))) {
                if (dead_a != dead_p) g_dp_bad_dead++;
                if (g_dp_bad++ < 12)
                    fprintf(stderr, "[dp] tl %d bw %d ql %d ext %zu parent score %d: wavefront score %d t_end %d q_end %d | matrix score %d t_end %d q_end %d\n", s->tl, s->bw, ql, ext, sc0, s->score, s->t_end, s->q_end, ps, pt, pq);
            }
        }
        const int32_t score = s->score + s->tl - s->t_end;
        if (score <= s->bw && (s->sink == UINT64_MAX || s->sink == w)) s->c->succ_events++;
        const int alive = s->score <= s->bw && ql - l_seq <= s->tl + s->bw && ((s->sink != UINT64_MAX && s->sink != w) || s->t_end < s->tl);
        { const int cat = (alive? 0 : (s->score > s->bw? 2 : 4)) + sib; s->c->cat_arcs[cat]++, s->c->cat_steps[cat] += s->c->steps - st_before, s->c->cat_diag[cat] += s->c->diag - dg_before; }
        if (alive) {
            g_cur_frame = child_frame, g_cur_hops = child_hops;
            dfs(s, w, depth + 1);
        }
        else {
            s->n_path++, s->c->dead++, s->c->depth_at_dead_sum += (uint64_t) depth;
            s->last_dead_depth = depth;
        }
        if (live > 1) { const int li = live_seen - 1 < 7? live_seen - 1 : 7; s->c->sub_steps[li] += s->c->steps - st_before, s->c->sub_cnt[li]++; }
        s->cl = l0, s->chash = h0;
        s->n = n0, s->d0 = d00, s->score = sc0, s->t_end = te0, s->q_end = qe0;
        wf_need(s, n0 + 4);
        memcpy(s->k, sv, sizeof(int32_t) * (size_t) n0);
        if (g_slots) memcpy(g_sl, sl_sv, sizeof(int32_t) * (size_t) g_sl_n);
        g_aligned = al0;
    }
    free(sl_sv);
    if (live > 1 && g_fs_n > 0) --g_fs_n;
    if (m2 && !m2->key) {
        m2->key = mix64(source * 0x9E3779B97F4A7C15ULL ^ h0 ^ ((uint64_t) l0 << 40)) | 1ULL, m2->frame = my_frame, m2->dead = (uint32_t) (s->n_path - np_at_entry);
        m2->ok = s->c->succ_events == succ_at_entry && s->n_path < MAX_DFS_PATH;
        g_m2_n++;
    }
    if (!found) s->memo->sub_arcs[slot_idx] = (uint32_t) (s->c->tried - arcs_before);      /* (the table may have moved: index, not pointer -- and it may have been rehashed; good enough for a hit rate) */
    free(sv);
}

static void hoco_dna(const uint8_t *hoco_s, uint32_t pos, int32_t l, int rev, char *out)
{
    int32_t i;
    for (i = 0; i < l; ++i) { uint32_t p = pos + (uint32_t) i; out[i] = "ACGT"[(hoco_s[p >> 2] >> (((p & 3) ^ 3) << 1)) & 3]; }
    if (rev) { int32_t a = 0, b = l - 1; for (; a < b; ++a, --b) { char t = out[a]; out[a] = comp(out[b]); out[b] = comp(t); } if (a == b) out[a] = comp(out[a]); }
}

/* every block of every read (the block finder of syncerr.c:339-612 as in oracle/ec.c, without the splice); prints blocks with >= min_tried arcs */
uint64_t ect_trace(const orc_graph_t *g, const uint8_t *scm_del, int K, double max_edist, uint64_t n_reads, const uint32_t *hoco_l, const uint8_t *hoco_s,
                   const uint64_t *hoco_byte_off, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t min_tried, const char *out_path)
{
    FILE *fo = fopen(out_path, "w");
    uint64_t rd, in_off = 0, n_blocks = 0, tot_tried = 0;
    S s;
    memset(&s, 0, sizeof(s));
    s.g = g;
    s.vseen = (uint8_t *) calloc(g->n_vtx + 1, 1), s.aseen = (uint8_t *) calloc(g->n_arc + 1, 1);
    char *seq = 0; size_t seq_m = 0;
    memo_t memo = {0, 0, 0, 0};
    for (rd = 0; rd < n_reads; ++rd) {
        const int32_t n = (int32_t) n_scm[rd];
        const uint64_t *km = k_mer + in_off;
        const uint32_t *mp = m_pos + in_off;
        const uint8_t *hs = hoco_s + hoco_byte_off[rd];
        int32_t beg = -1, end, l;
        for (;;) {
            uint32_t beg_pos = beg < 1? 0 : (mp[beg - 1] >> 1) + (uint32_t) K;
            beg_pos += MIN_ERR_SEQ_LEN;
            for (end = beg + 1; end < n; ++end)
                if (!scm_del[km[end] >> 1] && !(km[end] & 1) && (mp[end] >> 1) >= beg_pos) break;
            if (beg >= 0 || end < n) {
                uint64_t beg_utg, end_utg;
                int r;
                if (beg < 0) {
                    beg = end;
                    beg_utg = (km[beg] & ~1ULL) | (uint64_t) !(mp[beg] & 1);
                    beg_pos = 0, end_utg = UINT64_MAX, l = (int32_t) (mp[beg] >> 1), r = 1;
                } else {
                    --beg;
                    beg_utg = (km[beg] & ~1ULL) | (mp[beg] & 1);
                    beg_pos = (mp[beg] >> 1) + (uint32_t) K;
                    if (end >= n) end_utg = UINT64_MAX, l = (int32_t) hoco_l[rd] - (int32_t) beg_pos;
                    else end_utg = (km[end] & ~1ULL) | (mp[end] & 1), l = (int32_t) (mp[end] >> 1) - (int32_t) beg_pos;
                    r = 0;
                }
                if (l >= MIN_ERR_SEQ_LEN) {
                    if (seq_m < (size_t) l + 1) { seq_m = (size_t) l + 1; seq = (char *) realloc(seq, seq_m); }
                    hoco_dna(hs, beg_pos, l, r, seq);
                    int32_t bw = (int32_t) ceil(l * max_edist);
                    if (bw < MIN_ERR_BASE) bw = MIN_ERR_BASE;
                    ctr_t c;
                    memset(&c, 0, sizeof(c));
                    s.ts = seq, s.tl = l, s.bw = bw, s.sink = end_utg, s.cl = 0, s.chash = 0xCBF29CE484222325ULL, s.n_path = 0, s.c = &c;
                    wf_need(&s, 8);
                    s.n = 1, s.d0 = 0, s.k[0] = -1, s.score = 0, s.t_end = 0, s.q_end = 0;
                    memo.n = 0; if (memo.m > 4096) { free(memo.k); free(memo.sub_arcs); memo.k = 0, memo.sub_arcs = 0, memo.m = 0; } else if (memo.m) memset(memo.k, 0, memo.m * 8);
                    s.memo = &memo;
                    s.last_dead_depth = -1;
                    s.nv_seen = s.na_seen = 0;
                    if (g_m2) memset(g_m2, 0, g_m2_m * sizeof(m2_t));
                    g_m2_n = 0, g_frame_serial = 0, g_cur_frame = 0, g_cur_hops = 0, g_m2_hits = 0, g_m2_saved_dead = 0, g_fs_n = 0;
                    { size_t i_; for (i_ = 0; i_ < g_prof_n; ++i_) free(g_prof[i_]); g_prof_n = 0; }
                    const uint64_t lv_a0 = g_lvc_arcs, lv_s0 = g_lvc_steps;
                    const uint64_t rw_r0 = g_rw_rows, rw_t0 = g_rw_tests, rw_d0 = g_rw_dead_by_test, rw_b0 = g_rw_tables;
                    g_cert_n = 0, g_aligned = 0;
                    const uint64_t c_a0 = g_cert_arcs, c_s0 = g_cert_steps, c_t0 = g_cert_tables, c_c0 = g_cert_cells, c_da0 = g_cert_dead_arcs, c_ds0 = g_cert_dead_steps;
                    if (getenv("ECT_SLOTS")) slots_init(&s);
                    g_dp_R = bw + 2, g_dp_W = 2 * g_dp_R + 1;
                    const uint64_t ra0 = g_dp_rows_all, rc0 = g_dp_rows_cut, ca0 = g_dp_cut_arcs;
                    dfs(&s, beg_utg, 0);
                    if (g_cert > 0 && c.tried >= min_tried) fprintf(fo, "cert_arcs %llu cert_steps %llu cert_tables %llu cert_cells %llu long_dead_arcs %llu long_dead_steps %llu ", (unsigned long long) (g_cert_arcs - c_a0), (unsigned long long) (g_cert_steps - c_s0),
                                                                   (unsigned long long) (g_cert_tables - c_t0), (unsigned long long) (g_cert_cells - c_c0), (unsigned long long) (g_cert_dead_arcs - c_da0), (unsigned long long) (g_cert_dead_steps - c_ds0));
                    if (g_rows > 0 && c.tried >= min_tried) fprintf(fo, "lvc_arcs %llu lvc_steps %llu ", (unsigned long long) (g_lvc_arcs - lv_a0), (unsigned long long) (g_lvc_steps - lv_s0));
                    if (g_rows > 0 && c.tried >= min_tried) fprintf(fo, "rw_rows %llu rw_tests %llu rw_dead_by_test %llu rw_tables %llu ", (unsigned long long) (g_rw_rows - rw_r0), (unsigned long long) (g_rw_tests - rw_t0),
                                                                    (unsigned long long) (g_rw_dead_by_test - rw_d0), (unsigned long long) (g_rw_tables - rw_b0));
                    if (g_dp > 0 && c.tried >= min_tried) fprintf(fo, "dp_rows %llu dp_rows_cut %llu dp_cut_arcs %llu ", (unsigned long long) (g_dp_rows_all - ra0), (unsigned long long) (g_dp_rows_cut - rc0), (unsigned long long) (g_dp_cut_arcs - ca0));
                    ++n_blocks, tot_tried += c.tried;
                    if (c.tried >= min_tried) {
                        int i;
                        fprintf(fo, "read %llu beg_pos %u tl %d bw %d %s tried %llu dead %llu steps %llu diag %llu cmp %llu ext_sum %llu ext_max %llu frames %llu levels %llu max_depth %llu "
                                "memo_hit %llu memo_hit_arcs %llu succ %llu vtx %llu arcs %llu post_cap %llu mean_dead_depth %.1f mean_ret_dist %.2f",
                                (unsigned long long) rd, beg_pos, l, bw, r? "leading" : (end_utg == UINT64_MAX? "trailing" : "middle"),
                                (unsigned long long) c.tried, (unsigned long long) c.dead, (unsigned long long) c.steps, (unsigned long long) c.diag, (unsigned long long) c.cmp,
                                (unsigned long long) c.ext_sum, (unsigned long long) c.ext_max, (unsigned long long) c.frames, (unsigned long long) c.levels, (unsigned long long) c.max_depth,
                                (unsigned long long) c.memo_hit, (unsigned long long) c.memo_hit_arcs, (unsigned long long) c.succ_events, (unsigned long long) s.nv_seen, (unsigned long long) s.na_seen,
                                (unsigned long long) c.post_cap_arcs, c.dead? (double) c.depth_at_dead_sum / (double) c.dead : 0.0, c.ret_cnt? (double) c.ret_dist_sum / (double) c.ret_cnt : 0.0);
                        fprintf(fo, " n_hist"); for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.n_hist[i]);
                        fprintf(fo, " steps_hist"); for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.steps_hist[i]);
                        fprintf(fo, " ext_hist"); for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.ext_hist[i]);
                        fprintf(fo, " cat_arcs"); for (i = 0; i < 6; ++i) fprintf(fo, " %llu", (unsigned long long) c.cat_arcs[i]);
                        fprintf(fo, " cat_steps"); for (i = 0; i < 6; ++i) fprintf(fo, " %llu", (unsigned long long) c.cat_steps[i]);
                        fprintf(fo, " cat_diag"); for (i = 0; i < 6; ++i) fprintf(fo, " %llu", (unsigned long long) c.cat_diag[i]);
                        fprintf(fo, " m2_hits %llu m2_saved_dead %llu", (unsigned long long) g_m2_hits, (unsigned long long) g_m2_saved_dead);
                        fprintf(fo, " sub_steps"); for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.sub_steps[i]);
                        fprintf(fo, " sub_cnt"); for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.sub_cnt[i]);
                        fprintf(fo, " x_steps %llu x_w16_turns %llu x_long_lanes %llu x_long_steps %llu x_w64_turns %llu x_w64_long %llu x_act_lanes %llu x_ext_hist", (unsigned long long) c.x_steps, (unsigned long long) c.x_w16_turns,
                                (unsigned long long) c.x_long_lanes, (unsigned long long) c.x_long_steps, (unsigned long long) c.x_w64_turns, (unsigned long long) c.x_w64_long, (unsigned long long) c.x_act_lanes);
                        for (i = 0; i < 8; ++i) fprintf(fo, " %llu", (unsigned long long) c.x_ext_hist[i]);
                        fprintf(fo, "\n");
                        fflush(fo);
                    }
                    { uint64_t t_; for (t_ = 0; t_ < s.nv_seen; ++t_) s.vseen[s.vlist[t_]] = 0; for (t_ = 0; t_ < s.na_seen; ++t_) s.aseen[s.alist[t_]] = 0; }
                }
            }
            for (beg = end + 1; beg < n; ++beg)
                if (scm_del[km[beg] >> 1] || (km[end] & 1)) break;
            if (beg > n) break;
        }
        in_off += (uint64_t) n;
    }
    fprintf(fo, "# %llu blocks, %llu arcs tried in all\n", (unsigned long long) n_blocks, (unsigned long long) tot_tried);
    if (g_cert > 0) {
        fprintf(fo, "# ECT_CERT: %llu arcs (%llu wavefront steps) known to die by score from their string's least cost in the target alone; %llu of them did NOT die (must be 0); arcs of >= 32 bases that died by score: %llu (%llu steps); %llu tables, %llu cells\n",
                (unsigned long long) g_cert_arcs, (unsigned long long) g_cert_steps, (unsigned long long) g_cert_wrong, (unsigned long long) g_cert_dead_arcs, (unsigned long long) g_cert_dead_steps, (unsigned long long) g_cert_tables, (unsigned long long) g_cert_cells);
        fprintf(stderr, "ECT_CERT: %llu arcs / %llu steps certified, %llu wrong; long dead arcs %llu / %llu steps; %llu tables %llu cells\n", (unsigned long long) g_cert_arcs, (unsigned long long) g_cert_steps, (unsigned long long) g_cert_wrong,
                (unsigned long long) g_cert_dead_arcs, (unsigned long long) g_cert_dead_steps, (unsigned long long) g_cert_tables, (unsigned long long) g_cert_cells);
    }
    if (g_slots > 0) {
        fprintf(fo, "# ECT_SLOTS: one value per diagonal, frozen on the target's last column: %llu arcs checked against the wavefront, %llu differ\n", (unsigned long long) g_sl_checked, (unsigned long long) g_sl_bad);
        fprintf(stderr, "ECT_SLOTS: %llu arcs checked, %llu differ\n", (unsigned long long) g_sl_checked, (unsigned long long) g_sl_bad);
    }
    if (g_rows > 0) {
        fprintf(fo, "# ECT_ROWS: a solver by rows: %llu row-equivalents in all; %llu long arcs asked by table whether they can be alive, %llu of them die by score there; %llu differ from the wavefront (must be 0); %llu tables\n",
                (unsigned long long) g_rw_rows, (unsigned long long) g_rw_tests, (unsigned long long) g_rw_dead_by_test, (unsigned long long) g_rw_wrong, (unsigned long long) g_rw_tables);
        fprintf(fo, "# ECT_ROWS, the table test fed from the parent's WAVEFRONT (lower bounds) instead of its row: %llu of %llu long arcs that die by score are known to (%llu of %llu wavefront steps); known to die but alive (must be 0): %llu\n",
                (unsigned long long) g_lvc_arcs, (unsigned long long) g_lvc_dead_arcs, (unsigned long long) g_lvc_steps, (unsigned long long) g_lvc_dead_steps, (unsigned long long) g_lvc_wrong);
        fprintf(fo, "# ... and as the kernel's switch would ask (only where its wavefront stands for the consensus before the string; strings of <= 1024 bases): %llu arcs, %llu steps known; %llu not asked for a lagging wavefront\n",
                (unsigned long long) g_lvk_arcs, (unsigned long long) g_lvk_steps, (unsigned long long) g_lvk_lag);
        fprintf(stderr, "ECT_ROWS/reached cells priced at the parent's score: %llu arcs (%llu steps) known, %llu WRONG; as the kernel would ask: %llu arcs, %llu steps; reached cells %llu, %llu below the score, %llu above\n",
                (unsigned long long) g_lv2_arcs, (unsigned long long) g_lv2_steps, (unsigned long long) g_lv2_wrong, (unsigned long long) g_lv2k_arcs, (unsigned long long) g_lv2k_steps, (unsigned long long) g_hyp_cells, (unsigned long long) g_hyp_below, (unsigned long long) g_hyp_above);
        fprintf(fo, "# round 6 -- reached cells priced at the parent's score: %llu arcs (%llu steps) known, %llu wrong (must be 0); as the kernel would ask: %llu arcs, %llu steps; reached cells %llu, %llu below the score, %llu above\n",
                (unsigned long long) g_lv2_arcs, (unsigned long long) g_lv2_steps, (unsigned long long) g_lv2_wrong, (unsigned long long) g_lv2k_arcs, (unsigned long long) g_lv2k_steps, (unsigned long long) g_hyp_cells, (unsigned long long) g_hyp_below, (unsigned long long) g_hyp_above);
        fprintf(stderr, "ECT_ROWS/kernel's switch: %llu arcs, %llu steps known; %llu not asked (lag)\n", (unsigned long long) g_lvk_arcs, (unsigned long long) g_lvk_steps, (unsigned long long) g_lvk_lag);
        fprintf(stderr, "ECT_ROWS/wavefront bounds: %llu of %llu dead long arcs certified (%llu of %llu steps), wrong %llu\n", (unsigned long long) g_lvc_arcs, (unsigned long long) g_lvc_dead_arcs, (unsigned long long) g_lvc_steps,
                (unsigned long long) g_lvc_dead_steps, (unsigned long long) g_lvc_wrong);
        fprintf(stderr, "ECT_ROWS: %llu of the tests with the last column within the new rows' reach\n", (unsigned long long) g_rw_near);
        fprintf(stderr, "ECT_ROWS: %llu row-equivalents, %llu tests, %llu dead by test, %llu wrong, %llu tables\n", (unsigned long long) g_rw_rows, (unsigned long long) g_rw_tests, (unsigned long long) g_rw_dead_by_test, (unsigned long long) g_rw_wrong, (unsigned long long) g_rw_tables);
    }
    if (g_dp > 0) {
        fprintf(fo, "# ECT_DP: %llu arcs checked against the banded matrix (%llu cells): %llu differ, %llu of them in whether the arc dies by score\n", (unsigned long long) g_dp_checked, (unsigned long long) g_dp_cells,
                (unsigned long long) g_dp_bad, (unsigned long long) g_dp_bad_dead);
        fprintf(stderr, "ECT_DP: %llu arcs checked, %llu differ (%llu in dead-by-score); rows %llu, with the cut at the first row beyond the band %llu (%llu arcs cut)\n", (unsigned long long) g_dp_checked, (unsigned long long) g_dp_bad,
                (unsigned long long) g_dp_bad_dead, (unsigned long long) g_dp_rows_all, (unsigned long long) g_dp_rows_cut, (unsigned long long) g_dp_cut_arcs);
    }
    fclose(fo);
    free(seq); free(s.cs); free(s.k); free(s.nk); free(s.vseen); free(s.aseen); free(memo.k); free(memo.sub_arcs);
    return n_blocks;
}
