/*
 * tests/c/prof_bitpar_test.c -- the row solver's TABLE in bits (round 5; CPU prototype, run by tests/test_rows_bitpar.py).
 * For an arc that appends a long string (the K - overlap bases of a vertex) the search asks whether the arc can be alive at all: the new last row's least value is
 * min over t' of (parent's row at t' + prof[t']), prof[t'] = the least cost of fitting the whole string into the target FROM position t' + 1 on (any end) --
 * tests/trace/ec_trace.c ECT_ROWS: 339 082 such tests on the config-1 surrogate, none differs from the wavefront.  prof is approximate string matching of the
 * REVERSED string against the REVERSED target with a free start (Myers 1999, multiword): one pass over the target, a dozen operations per word and base.
 * Checked here against the plain matrix on random and repetitive strings.  Test infrastructure: nothing in the product links this.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_s = 0x853C49E6748FEA9BULL;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
#define NW 32                                   /* 32-bit words: strings of up to 1024 bases (K - overlap <= 1001) */

/* prof[t' + 1], t' = -1 .. tl - 1, by the plain matrix (as tests/trace/ec_trace.c prof_of) */
static void prof_plain(const uint8_t *ts, int tl, const uint8_t *ext, int m, int *prof)
{
    int *prev = malloc(sizeof(int) * (size_t) (tl + 1)), *cur = malloc(sizeof(int) * (size_t) (tl + 1)), r, j;
    for (j = 0; j <= tl; ++j) prev[j] = 0;
    for (r = 1; r <= m; ++r) {
        cur[0] = r;
        for (j = 1; j <= tl; ++j) {
            int v = prev[j - 1] + (ext[m - r] != ts[tl - j]);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        { int *x = prev; prev = cur, cur = x; }
    }
    for (j = 0; j <= tl; ++j) prof[tl - j] = prev[j];
    free(prev), free(cur);
}

/* prof2[u], u = t' + 1 = 0 .. tl, by the plain matrix (as tests/trace/ec_trace.c): G[i][u] = min(G[i+1][u+1] + mismatch, G[i+1][u] + 1, G[i][u+1] + 1), G[i][tl] = 0, G[m][u] = tl - u */
static void prof2_plain(const uint8_t *ts, int tl, const uint8_t *ext, int m, int *prof)
{
    int *g1 = malloc(sizeof(int) * (size_t) (tl + 1)), *g0 = malloc(sizeof(int) * (size_t) (tl + 1)), i, u;
    for (u = 0; u <= tl; ++u) g1[u] = tl - u;
    for (i = m - 1; i >= 0; --i) {
        g0[tl] = 0;
        for (u = tl - 1; u >= 0; --u) {
            int v = g1[u + 1] + (ext[i] != ts[u]);
            if (g1[u] + 1 < v) v = g1[u] + 1;
            if (g0[u + 1] + 1 < v) v = g0[u + 1] + 1;
            g0[u] = v;
        }
        { int *x = g1; g1 = g0, g0 = x; }
    }
    for (u = 0; u <= tl; ++u) prof[u] = g1[u];
    free(g1), free(g0);
}

/* the same in bits: word w of a vector = pattern positions 32 w .. 32 w + 31 of the reversed string */
/* second = 0: the whole string into the target from t' + 1 on, any end (rows start at i, the first row at 0);  second = 1: some PREFIX of the string against the target
 * from t' + 1 TO ITS END (backwards: the piece of the reversed target begins at its first base -- the first row counts the columns -- and the reversed string may be
 * entered at any row for nothing) */
static void prof_bits(const uint8_t *ts, int tl, const uint8_t *ext, int m, int *prof, int second)
{
    uint32_t peq[4][NW], pv[NW], mv[NW];
    const int nw = (m + 31) / 32;
    const uint32_t last = m & 31? (1u << (m & 31)) - 1u : 0xFFFFFFFFu, top = 1u << ((m - 1) & 31);
    int w, i, j, score = second? 0 : m;
    memset(peq, 0, sizeof(peq));
    for (i = 0; i < m; ++i) peq[ext[m - 1 - i]][i >> 5] |= 1u << (i & 31);
    for (w = 0; w < nw; ++w) pv[w] = second? 0u : (w == nw - 1? last : 0xFFFFFFFFu), mv[w] = 0;
    prof[tl] = score;                                                     /* j = 0: nothing of the target */
    for (j = 1; j <= tl; ++j) {
        const uint32_t *eq = peq[ts[tl - j]];
        uint32_t xv[NW], ph[NW], mh[NW];
        uint64_t G = 0, P = 0;
        uint32_t s1[NW];
        for (w = 0; w < nw; ++w) {                                    /* the addition word by word, its carries as the device resolves them: C = (P + (G << 1)) ^ P */
            const uint32_t a = eq[w] & pv[w];
            s1[w] = a + pv[w];
            G |= (uint64_t) (s1[w] < a) << w, P |= (uint64_t) (s1[w] == 0xFFFFFFFFu) << w;
        }
        const uint64_t U = G << 1, C = (P + U) ^ P;
        for (w = 0; w < nw; ++w) {
            const uint32_t s2 = s1[w] + (uint32_t) (C >> w & 1), xh = (s2 ^ pv[w]) | eq[w];
            xv[w] = eq[w] | mv[w];
            ph[w] = mv[w] | ~(xh | pv[w]);
            mh[w] = pv[w] & xh;
        }
        ph[nw - 1] &= last, mh[nw - 1] &= last;
        score += (ph[nw - 1] & top) != 0, score -= (mh[nw - 1] & top) != 0;
        for (w = nw - 1; w >= 0; --w) ph[w] = ph[w] << 1 | (w? ph[w - 1] >> 31 : (uint32_t) second), mh[w] = mh[w] << 1 | (w? mh[w - 1] >> 31 : 0u);      /* (free start in the target: nothing comes in; the second table's first row counts the columns: +1) */
        for (w = 0; w < nw; ++w) pv[w] = mh[w] | ~(xv[w] | ph[w]), mv[w] = ph[w] & xv[w];
        pv[nw - 1] &= last, mv[nw - 1] &= last;
        prof[tl - j] = score;
    }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1? atoi(argv[1]) : 200;
    int r, n = 0;
    if (argc > 2) rng_s ^= (uint64_t) atoll(argv[2]) * 0x9E3779B97F4A7C15ULL;
    for (r = 0; r < rounds; ++r) {
        const int tl = 1 + (int) (rnd() % 3000), m = 1 + (int) (rnd() % (r % 3? 1001 : 70)), alpha = r % 5 == 1? 2 : 4, period = r % 4 == 2? 2 + (int) (rnd() % 9) : 0;
        uint8_t *ts = malloc((size_t) tl), *ext = malloc((size_t) m);
        int *a = malloc(sizeof(int) * (size_t) (tl + 1)), *b = malloc(sizeof(int) * (size_t) (tl + 1)), i;
        for (i = 0; i < tl; ++i) ts[i] = (uint8_t) (period && i >= period && rnd() % 40? ts[i - period] : rnd() % (uint64_t) alpha);
        if (r % 2 && m < tl) { const int at = (int) (rnd() % (uint64_t) (tl - m + 1)); for (i = 0; i < m; ++i) ext[i] = (uint8_t) (rnd() % 25? ts[at + i] : rnd() % (uint64_t) alpha); }
        else for (i = 0; i < m; ++i) ext[i] = (uint8_t) (rnd() % (uint64_t) alpha);
        prof_plain(ts, tl, ext, m, a), prof_bits(ts, tl, ext, m, b, 0);
        for (i = 0; i <= tl; ++i) if (a[i] != b[i]) { fprintf(stderr, "round %d tl %d m %d: prof[%d] plain %d bits %d\n", r, tl, m, i, a[i], b[i]); return 1; }
        prof2_plain(ts, tl, ext, m, a), prof_bits(ts, tl, ext, m, b, 1);
        for (i = 0; i <= tl; ++i) if (a[i] != b[i]) { fprintf(stderr, "round %d tl %d m %d: prof2[%d] plain %d bits %d\n", r, tl, m, i, a[i], b[i]); return 1; }
        n += tl + 1;
        free(ts), free(ext), free(a), free(b);
    }
    printf("ok: %d entries\n", n);
    return 0;
}
