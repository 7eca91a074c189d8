/*
 * tests/c/rows_bitpar_test.c -- the row solver's row in BITS (round 5; CPU prototype for the next round's kernel, built and run by tests/test_rows_bitpar.py).
 *
 * tools/experiments/ec_rows.hpp keeps one VALUE per diagonal and pays ~67 instructions per 64 diagonals and row.  The same row as differences: along a row of the
 * edit-distance matrix neighbouring cells differ by -1, 0 or +1, so a band of W diagonals is two bit vectors (Myers 1999), and a band that moves one cell down the
 * target per query base is Hyyro's diagonal band (2003): previous row's vectors shifted by one, one multiword addition, a dozen logical operations -- a few dozen
 * instructions per row whatever W <= 512 is.  Worked out here on 64-bit words and checked against the plain matrix:
 *   band cell b = 0 .. W - 1 of row q is the cell (q, t) with t = q - OFF + b  (diagonal q - t = OFF - b; b grows with t; OFF = bw + 2, W = 2 OFF + 1)
 *   Pv / Mv bit b: D(q, t_b) - D(q, t_b - 1) = +1 / -1;  the value of the middle cell (diagonal 0) is carried as a number: it moves by 1 - D0[OFF] per row
 *   cells with t <= -1 continue the matrix upwards as q - t (what its first column is), cells with t >= tl continue it with a base that matches nothing:
 *   neither feeds a cell of the matrix, so nothing has to be masked
 *   above the band's top and below its bottom the neighbour is taken as one more than the cell inside: never less than the truth, so every value <= bw is exact
 * Checked: after every row all values min(., bw + 1) against the matrix with infinite cells outside the band, and the outcome of every resumed call (score, t_end,
 * q_end as wf_ed_core leaves them) against the closed form of tests/test_oracle_golden.py.  Test infrastructure: nothing in the product links this.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NWMAX 10
#define INF 1000000

static uint64_t rng_s = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

typedef struct {
    int W, OFF, nw, tl;
    uint64_t pv[NWMAX], mv[NWMAX], eq[4][NWMAX];     /* eq[x] bit b: target[t_b] == x (0 outside the target) */
    int mid;                                        /* D(q, q): the cell of diagonal 0 */
    int q;                                          /* rows done: the state is row q - 1 */
    const uint8_t *ts;
} bp_t;

static inline int bit(const uint64_t *v, int b) { return (int) (v[b >> 6] >> (b & 63) & 1); }
static void shr1(uint64_t *v, int nw) { int w; for (w = 0; w < nw; ++w) v[w] = v[w] >> 1 | (w + 1 < nw? v[w + 1] << 63 : 0); }
static void shl1(uint64_t *v, int nw, int in) { int w; for (w = nw - 1; w >= 0; --w) v[w] = v[w] << 1 | (w? v[w - 1] >> 63 : (uint64_t) in); }
static void setbit(uint64_t *v, int b, int x) { v[b >> 6] = (v[b >> 6] & ~(1ULL << (b & 63))) | (uint64_t) (x & 1) << (b & 63); }

static void bp_init(bp_t *p, const uint8_t *ts, int tl, int bw)
{
    int b, x;
    memset(p, 0, sizeof(*p));
    p->OFF = bw + 2, p->W = 2 * p->OFF + 1, p->nw = (p->W + 63) / 64, p->tl = tl, p->ts = ts, p->q = 0, p->mid = 0;
    /* the row before the first, q = -1: t_b = -1 - OFF + b; D = |t + 1| */
    for (b = 0; b < p->W; ++b) { if (b <= p->OFF) setbit(p->mv, b, 1); else setbit(p->pv, b, 1); }
    /* eq for row q = 0 is built by the first shift: here the window of row -1 */
    for (b = 0; b < p->W; ++b) { const int t = -1 - p->OFF + b; for (x = 0; x < 4; ++x) setbit(p->eq[x], b, t >= 0 && t < tl && ts[t] == x); }
}

/* one row: the query's base c */
static void bp_row(bp_t *p, int c)
{
    const int nw = p->nw, W = p->W, q = p->q;
    uint64_t pva[NWMAX], mva[NWMAX], xv[NWMAX], xh[NWMAX], ph[NWMAX], mh[NWMAX], d0[NWMAX];
    const uint64_t *eq;
    int w, x;
    const uint64_t last = W & 63? (1ULL << (W & 63)) - 1 : ~0ULL;
    /* the band moves one cell down the target: the window of target bases, and the previous row's differences seen from the new band (cell b's neighbour in the
     * previous row at the same t was cell b + 1 there); below the bottom: one more */
    {
        const int t_new = q - p->OFF + (W - 1);
        for (x = 0; x < 4; ++x) { shr1(p->eq[x], nw); setbit(p->eq[x], W - 1, t_new >= 0 && t_new < p->tl && p->ts[t_new] == x); }
    }
    memcpy(pva, p->pv, sizeof(pva)), memcpy(mva, p->mv, sizeof(mva));
    shr1(pva, nw), shr1(mva, nw);
    setbit(pva, W - 1, 1), setbit(mva, W - 1, 0);
    eq = p->eq[c];
    /* Myers' column step on the aligned vectors; one multiword addition */
    {
        unsigned carry = 0;
        for (w = 0; w < nw; ++w) {
            const uint64_t a = eq[w] & pva[w], b2 = pva[w];
            const uint64_t s1 = a + b2, s2 = s1 + carry;
            carry = (s1 < a) | (s2 < s1);
            xh[w] = (s2 ^ pva[w]) | eq[w];
            xv[w] = eq[w] | mva[w];
            d0[w] = xh[w] | mva[w];
            ph[w] = mva[w] | ~(xh[w] | pva[w]);
            mh[w] = pva[w] & xh[w];
        }
    }
    ph[nw - 1] &= last, mh[nw - 1] &= last;
    p->mid += 1 - bit(d0, p->OFF);
    /* the differences along the new row: the row above the band's top counts as one more */
    shl1(ph, nw, 1), shl1(mh, nw, 0);
    for (w = 0; w < nw; ++w) p->pv[w] = mh[w] | ~(xv[w] | ph[w]), p->mv[w] = ph[w] & xv[w];
    p->pv[nw - 1] &= last, p->mv[nw - 1] &= last;
    p->q = q + 1;
}

/* the value of band cell b of the row at hand */
static void bp_values(const bp_t *p, int *val)
{
    int b;
    val[p->OFF] = p->mid;
    for (b = p->OFF + 1; b < p->W; ++b) val[b] = val[b - 1] + bit(p->pv, b) - bit(p->mv, b);
    for (b = p->OFF - 1; b >= 0; --b) val[b] = val[b + 1] - bit(p->pv, b + 1) + bit(p->mv, b + 1);
}

/* ---- the plain matrix, cells outside the band infinite ---- */
typedef struct { int W, OFF, tl, q; int *row; const uint8_t *ts; } ref_t;
static void ref_init(ref_t *r, const uint8_t *ts, int tl, int bw)
{
    int b;
    r->OFF = bw + 2, r->W = 2 * r->OFF + 1, r->tl = tl, r->ts = ts, r->q = 0;
    r->row = (int *) malloc(sizeof(int) * (size_t) r->W);
    for (b = 0; b < r->W; ++b) { const int t = -1 - r->OFF + b; r->row[b] = t >= -1 && t < tl? t + 1 : INF; }
}
static void ref_row(ref_t *r, int c)
{
    const int q = r->q, W = r->W;
    int b, *n = (int *) malloc(sizeof(int) * (size_t) W);
    for (b = 0; b < W; ++b) {
        const int t = q - r->OFF + b;
        int v;
        if (t < -1 || t >= r->tl) v = INF;
        else if (t == -1) v = q + 1;
        else {
            v = r->row[b] + (r->ts[t] != c);
            if (b + 1 < W && r->row[b + 1] + 1 < v) v = r->row[b + 1] + 1;
            if (b >= 1 && n[b - 1] + 1 < v) v = n[b - 1] + 1;
        }
        n[b] = v > INF? INF : v;
    }
    free(r->row), r->row = n, r->q = q + 1;
}

/* ---- the read-off as the device does it (tools/experiments/ec_rows.hpp: ecb_read): 32-bit words, per-word sums of differences, a cell's value from popcounts under a
 * mask, the last row's qualifying cell with the LARGEST band index against the last column's first qualifying row ---- */
static uint32_t upto(int k) { return k == 31? 0xFFFFFFFFu : (1u << (k + 1)) - 1u; }
static void dev_read(const bp_t *p, int tl, int ql, int bw, const int *lc /* by row */, int before, int *out)
{
    const int W = p->W, OFF = p->OFF, nw = (W + 31) / 32;
    uint32_t pv[2 * NWMAX], mv[2 * NWMAX];
    int wpre[2 * NWMAX], w, b, acc = 0;
    for (w = 0; w < nw; ++w) pv[w] = (uint32_t) (p->pv[w >> 1] >> ((w & 1) * 32)), mv[w] = (uint32_t) (p->mv[w >> 1] >> ((w & 1) * 32));
    for (w = 0; w < nw; ++w) { wpre[w] = acc; acc += __builtin_popcount(pv[w]) - __builtin_popcount(mv[w]); }
    const int c_off = wpre[OFF >> 5] + __builtin_popcount(pv[OFF >> 5] & upto(OFF & 31)) - __builtin_popcount(mv[OFF >> 5] & upto(OFF & 31));
    const int lc0 = tl - 1 - OFF, q_lo = lc0 > 0? lc0 : 0;
    int best = INF, qq;
#define VAL(b) (p->mid + wpre[(b) >> 5] + __builtin_popcount(pv[(b) >> 5] & upto((b) & 31)) - __builtin_popcount(mv[(b) >> 5] & upto((b) & 31)) - c_off)
    for (b = 0; b < W; ++b) { const int t = ql - 1 - OFF + b; if (t >= 0 && t < tl && VAL(b) < best) best = VAL(b); }
    for (qq = q_lo; qq < ql && qq < lc0 + W; ++qq) if (lc[qq] < best) best = lc[qq];
    const int sc = best > before? best : before;
    if (sc > bw) { out[0] = bw + 1, out[1] = out[2] = 0; return; }
    int d_col = INF, q_col = 0, d_row = INF, t_row = 0;
    for (qq = q_lo; qq < ql && qq < lc0 + W; ++qq) if (lc[qq] <= sc) { q_col = qq, d_col = qq - (tl - 1); break; }
    for (b = W - 1; b >= 0; --b) { const int t = ql - 1 - OFF + b; if (t >= 0 && t < tl && VAL(b) <= sc) { t_row = t, d_row = OFF - b; break; } }
#undef VAL
    out[0] = sc;
    if (d_col <= d_row) out[1] = tl, out[2] = q_col + 1; else out[1] = t_row + 1, out[2] = ql;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1? atoi(argv[1]) : 300;
    int r, n_rows = 0, n_calls = 0;
    if (argc > 2) rng_s ^= (uint64_t) atoll(argv[2]) * 0xD1B54A32D192ED03ULL;
    for (r = 0; r < rounds; ++r) {
        const int tl = 20 + (int) (rnd() % (r % 7 == 0? 3000 : 400));
        int bw = r % 3 == 0? 6 + (int) (rnd() % 20) : (tl / 50 > 6? tl / 50 : 6);
        if (r % 11 == 0) bw = 100 + (int) (rnd() % 150);
        if (2 * (bw + 2) + 1 > 64 * NWMAX) bw = 64 * NWMAX / 2 - 3;
        uint8_t *ts = (uint8_t *) malloc((size_t) tl), *qs = (uint8_t *) malloc((size_t) tl * 2 + 400);
        int i, ql = 0;
        const int alpha = r % 5 == 1? 2 : 4, period = r % 4 == 2? 2 + (int) (rnd() % 9) : 0;
        for (i = 0; i < tl; ++i) ts[i] = (uint8_t) (period && i >= period && rnd() % 40? ts[i - period] : rnd() % (uint64_t) alpha);
        /* the query: the target with edits, or a stretch of it followed by something else, or unrelated */
        {
            const int kind = (int) (rnd() % 4), edits = (int) (rnd() % (uint64_t) (2 * bw + 2));
            int t = 0;
            if (kind == 3) for (i = 0; i < tl; ++i) qs[ql++] = (uint8_t) (rnd() % (uint64_t) alpha);
            else {
                const int keep = kind == 1? (int) (rnd() % (uint64_t) tl) : tl;
                while (t < keep) {
                    const int e = edits && (int) (rnd() % (uint64_t) tl) < edits? 1 + (int) (rnd() % 3) : 0;
                    if (e == 1) qs[ql++] = (uint8_t) (rnd() % (uint64_t) alpha), ++t;
                    else if (e == 2) qs[ql++] = (uint8_t) (rnd() % (uint64_t) alpha);
                    else if (e == 3) ++t;
                    else qs[ql++] = ts[t++];
                }
                if (kind == 1 || kind == 2) { const int more = (int) (rnd() % 300); for (i = 0; i < more; ++i) qs[ql++] = (uint8_t) (rnd() % (uint64_t) alpha); }
            }
            if (ql == 0) qs[ql++] = 0;
        }
        bp_t p;
        ref_t f;
        bp_init(&p, ts, tl, bw), ref_init(&f, ts, tl, bw);
        int *val = (int *) malloc(sizeof(int) * (size_t) p.W), *lastcol = (int *) malloc(sizeof(int) * (size_t) (ql + 1)), *lastcol_b = (int *) malloc(sizeof(int) * (size_t) (ql + 1));
        int q = 0, score_f = 0, score_b = 0, score_dev = 0;
        while (q < ql) {
            int to = q + 1 + (int) (rnd() % 60), b;
            if (to > ql) to = ql;
            for (; q < to; ++q) {
                bp_row(&p, qs[q]), ref_row(&f, qs[q]);
                bp_values(&p, val);
                ++n_rows;
                for (b = 0; b < p.W; ++b) {
                    const int t = q - p.OFF + b;
                    if (t < 0 || t >= tl) continue;
                    const int a = val[b] > bw? bw + 1 : val[b], c = f.row[b] > bw? bw + 1 : f.row[b];
                    if (a != c) { fprintf(stderr, "round %d tl %d bw %d row %d cell t %d (band %d): bits %d, matrix %d\n", r, tl, bw, q, t, b, val[b], f.row[b]); return 1; }
                }
                { const int bl = tl - 1 - q + p.OFF; lastcol[q] = bl >= 0 && bl < p.W? f.row[bl] : INF, lastcol_b[q] = bl >= 0 && bl < p.W? val[bl] : INF; }
            }
            /* the call's outcome, from either */
            {
                int which;
                int out[2][3];
                for (which = 0; which < 2; ++which) {
                    const int *row = which? val : f.row, *lc = which? lastcol_b : lastcol;
                    int best = INF, qq, *score = which? &score_b : &score_f;
                    for (b = 0; b < p.W; ++b) { const int t = to - 1 - p.OFF + b; if (t >= 0 && t < tl && row[b] < best) best = row[b]; }
                    for (qq = 0; qq < to; ++qq) if (lc[qq] < best) best = lc[qq];
                    int sc = best > *score? best : *score;
                    if (sc > bw) { *score = bw + 1, out[which][0] = bw + 1, out[which][1] = out[which][2] = 0; continue; }
                    int bd = INF, bt = 0, bq = 0;
                    for (b = 0; b < p.W; ++b) { const int t = to - 1 - p.OFF + b; if (t >= 0 && t < tl && row[b] <= sc && to - 1 - t < bd) bd = to - 1 - t, bt = t, bq = to - 1; }
                    for (qq = 0; qq < to; ++qq) if (lc[qq] <= sc && qq - (tl - 1) < bd) bd = qq - (tl - 1), bt = tl - 1, bq = qq;
                    *score = sc, out[which][0] = sc, out[which][1] = bt + 1, out[which][2] = bq + 1;
                }
                ++n_calls;
                {
                    int dv[3];
                    dev_read(&p, tl, to, bw, lastcol_b, score_dev, dv);
                    score_dev = dv[0];
                    if (memcmp(out[0], dv, sizeof(dv))) { fprintf(stderr, "round %d tl %d bw %d ql %d: matrix (%d %d %d), the device's read-off (%d %d %d)\n", r, tl, bw, to, out[0][0], out[0][1], out[0][2], dv[0], dv[1], dv[2]); return 1; }
                }
                if (memcmp(out[0], out[1], sizeof(out[0]))) { fprintf(stderr, "round %d tl %d bw %d ql %d: matrix (%d %d %d), bits (%d %d %d)\n", r, tl, bw, to, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2]); return 1; }
                if (score_f > bw && rnd() % 3 == 0) break;
            }
        }
        free(val), free(lastcol), free(lastcol_b), free(f.row), free(ts), free(qs);
    }
    printf("ok: %d rows, %d calls\n", n_rows, n_calls);
    return 0;
}
