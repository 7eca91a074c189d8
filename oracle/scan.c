/*
 * oracle/scan.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates the per-read scan of oatk's syncasm: homopolymer compression into
 * 2-bit hoco_s + run lengths, rolling canonical s-mer hash, closed-syncmer
 * emission (reference syncmer.c:243-421, `sr_read_analysis_thread`).
 *
 * Two formulations that must agree bit for bit:
 *   mode 0  scan_streaming()  one pass with a Q-slot ring and a tracked
 *           minimum, the way the reference does it;
 *   mode 1  scan_stateless()  per-position predicates over the array of s-mer
 *           hashes -- what the HIP kernel evaluates in parallel.
 *
 * Naming: K = k-mer size (1001), S = s-mer size (31), Q = K-S+1 s-mers per
 * k-mer.  (Inside the reference function `k` is S and `w` is K, syncmer.c:250.)
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- small growable arrays ---- */
typedef struct { uint8_t *a; size_t n, m; } vec8;
typedef struct { uint32_t *a; size_t n, m; } vec32;
typedef struct { uint64_t *a; size_t n, m; } vec64;
#define VPUSH(v, T, x) do { if ((v).n == (v).m) { (v).m = (v).m? (v).m * 2 : 64; \
    (v).a = (T *) realloc((v).a, (v).m * sizeof(T)); } (v).a[(v).n++] = (x); } while (0)

/* seq_nt4_table, syncmer.c:47-64: A/a 0, C/c 1, G/g 2, T/t/U/u 3, bytes 0..3 map to themselves, rest 4 */
uint8_t orc_nt4(uint8_t ch)
{
    if (ch < 4) return ch;
    switch (ch | 0x20) {
        case 'a': return 0;
        case 'c': return 1;
        case 'g': return 2;
        case 't': case 'u': return 3;
    }
    return 4;
}

/* invertible integer mix confined to 2S bits, syncmer.c:116-126 */
uint64_t orc_hash64(uint64_t x, uint64_t mask)
{
    x = (~x + (x << 21)) & mask;
    x ^= x >> 24;
    x = (x + (x << 3) + (x << 8)) & mask;
    x ^= x >> 14;
    x = (x + (x << 2) + (x << 4)) & mask;
    x ^= x >> 28;
    x = (x + (x << 31)) & mask;
    return x;
}

typedef struct {
    vec8 hoco_s, ho_rl;
    vec32 ho_l_rl, n_nucl, m_pos;
    vec64 s_mer, k_mer;
    uint32_t hoco_l;
} read_out_t;

/* hoco layout: base i lives in byte i/4 at bit offset ((i&3)^3)*2, syncmer.c:290 */
static inline void put_base(read_out_t *o, uint32_t i, uint8_t c)
{
    if ((i & 3) == 0) VPUSH(o->hoco_s, uint8_t, 0);
    o->hoco_s.a[o->hoco_s.n - 1] |= (uint8_t) (c << (((i & 3) ^ 3) << 1));
}

static void emit(read_out_t *o, uint32_t pos, uint64_t scode, uint32_t strand, int K)
{
    uint32_t p = pos << 1 | strand;
    VPUSH(o->s_mer, uint64_t, scode);
    VPUSH(o->m_pos, uint32_t, p);
    VPUSH(o->k_mer, uint64_t, orc_kmer_hash(o->hoco_s.a, pos, strand, K));
}

/* HPC + packing shared by both modes: fills hoco_s/ho_rl/ho_l_rl/n_nucl, and per hoco position the
 * base code b[] (N -> 0) and isn[] flags.  syncmer.c:284-323 */
static void compress(const uint8_t *seq, int64_t len, read_out_t *o, vec8 *b, vec8 *isn)
{
    int64_t i = 0;
    uint32_t h = 0;
    while (i < len) {
        uint8_t c = orc_nt4(seq[i]);
        if (c < 4) {
            int64_t j = i + 1;
            while (j < len && orc_nt4(seq[j]) == c) ++j;       /* run [i, j) :294-299 */
            int64_t rl = j - i;
            put_base(o, h, c);
            if (rl > 255) VPUSH(o->ho_l_rl, uint32_t, (uint32_t) (rl - 1)); /* :301-302 */
            VPUSH(o->ho_rl, uint8_t, (uint8_t) ((rl > 256? 256 : rl) - 1)); /* :303-304 */
            if (b) { VPUSH(*b, uint8_t, c); VPUSH(*isn, uint8_t, 0); }
            i = j;
        } else {                                               /* ambiguous: stored as A, rl 0, never merged :316-321 */
            put_base(o, h, 0);
            VPUSH(o->ho_rl, uint8_t, 0);
            VPUSH(o->n_nucl, uint32_t, (uint32_t) i);
            if (b) { VPUSH(*b, uint8_t, 0); VPUSH(*isn, uint8_t, 1); }
            i += 1;
        }
        ++h;
    }
    o->hoco_l = h;
}

/* per hoco END position e: hash of the canonical s-mer ending there (UINT64_MAX if it spans an N,
 * is shorter than S, or equals its own reverse complement) and its code canon<<1|strand. :306-315 */
static void smer_hashes(const uint8_t *b, const uint8_t *isn, uint32_t n, int S, uint64_t *m, uint64_t *sc, uint32_t *lrun)
{
    uint64_t mask = (1ULL << 2 * S) - 1, fw = 0, rv = 0;
    uint32_t e, l = 0;
    int sh = 2 * (S - 1);
    for (e = 0; e < n; ++e) {
        m[e] = sc[e] = UINT64_MAX;
        if (isn[e]) { l = 0; lrun[e] = 0; continue; }
        ++l;
        fw = (fw << 2 | b[e]) & mask;
        rv = rv >> 2 | (uint64_t) (3 ^ b[e]) << sh;
        if (fw != rv && l >= (uint32_t) S) {
            uint64_t z = fw < rv? 0 : 1, c = z? rv : fw;
            m[e] = orc_hash64(c, mask);
            sc[e] = c << 1 | z;
        }
        lrun[e] = l;
    }
}

/* ---------------- mode 0: streaming ---------------- */
static void scan_streaming(const uint8_t *seq, int64_t len, int K, int S, read_out_t *o)
{
    /* The reference interleaves compression and emission; emission at hoco position e only looks at
     * hoco_s bytes up to e, so compressing first and replaying positions is equivalent. */
    vec8 b = {0, 0, 0}, isn = {0, 0, 0};
    compress(seq, len, o, &b, &isn);
    uint32_t n = o->hoco_l, e;
    if (n == 0) { free(b.a); free(isn.a); return; }
    uint64_t *m = (uint64_t *) malloc(sizeof(uint64_t) * n), *sc = (uint64_t *) malloc(sizeof(uint64_t) * n);
    uint32_t *lrun = (uint32_t *) malloc(sizeof(uint32_t) * n);
    smer_hashes(b.a, isn.a, n, S, m, sc, lrun);

    int Q = K - S + 1, slot = 0, best_slot = 0, j;
    uint64_t *ring_m = (uint64_t *) malloc(sizeof(uint64_t) * Q), *ring_s = (uint64_t *) malloc(sizeof(uint64_t) * Q);
    uint64_t best = UINT64_MAX;
    memset(ring_m, 0xff, sizeof(uint64_t) * Q);                /* buffers start all-ones :280-281 */
    memset(ring_s, 0xff, sizeof(uint64_t) * Q);
    uint32_t l = 0;
    for (e = 0; e < n; ++e) {
        uint64_t me = m[e], se = sc[e];
        l = lrun[e];
        /* the slot about to be overwritten holds the s-mer that starts k-mer (e-K): open test :325-338 */
        if (slot == best_slot && best != UINT64_MAX && l > (uint32_t) K) {
            emit(o, e - K, ring_s[slot], (uint32_t) (ring_s[slot] & 1), K);
            size_t c = o->m_pos.n;
            if (c >= 2 && o->m_pos.a[c - 1] >> 1 == o->m_pos.a[c - 2] >> 1)   /* first == last s-mer: drop both :337 */
                o->m_pos.n -= 2, o->s_mer.n -= 2, o->k_mer.n -= 2;
        }
        ring_m[slot] = me, ring_s[slot] = se;
        if (me <= best && me != UINT64_MAX) {                                    /* :342-355 */
            if (l >= (uint32_t) K) emit(o, e + 1 - K, se ^ 1, (uint32_t) (se & 1), K);
            if (me < best) best = me, best_slot = slot;
        }
        if (me >= best && slot == best_slot) {                                   /* tracked minimum left the window :356-377 */
            int differs = me != best;
            best = UINT64_MAX;
            for (j = slot + 1; j < Q; ++j) if (ring_m[j] < best) best = ring_m[j], best_slot = j;
            for (j = 0; j <= slot; ++j) if (ring_m[j] < best) best = ring_m[j], best_slot = j;
            int nxt = slot + 1 == Q? 0 : slot + 1;
            if (differs && ((best_slot == nxt && best == me) || best_slot == slot) && best != UINT64_MAX && l >= (uint32_t) K)
                emit(o, e + 1 - K, se ^ 1, (uint32_t) (se & 1), K);
        }
        slot = slot + 1 == Q? 0 : slot + 1;
    }
    /* the k-mer that ends exactly at the read end can still be open :383-394 */
    if (slot == best_slot && best != UINT64_MAX && l >= (uint32_t) K) {
        emit(o, n - K, ring_s[slot], (uint32_t) (ring_s[slot] & 1), K);
        size_t c = o->m_pos.n;
        if (c >= 2 && o->m_pos.a[c - 1] >> 1 == o->m_pos.a[c - 2] >> 1)
            o->m_pos.n -= 2, o->s_mer.n -= 2, o->k_mer.n -= 2;
    }
    free(ring_m); free(ring_s); free(m); free(sc); free(lrun); free(b.a); free(isn.a);
}

/* ---------------- mode 1: stateless ----------------
 * With s-mers indexed by START p (M[p] = hash of s-mer [p, p+S), UINT64_MAX when invalid / off the
 * read; M[-1] = UINT64_MAX), and for k-mer start j:
 *     e = j+Q-1 (last s-mer), x = M[j-1], f = M[j], y = M[e], b = min M[j .. e-1]
 *     valid(j)  : no N in [j, j+K)
 *     next(j)   : j+K == hoco_l, or base j+K exists and is not N            (:325 `l > w`, :383)
 *     Close(j) <=> valid && y != MAX && ( y < b || (y == b && (x >= b || f == b)) )
 *     Open(j)  <=> valid && next && f != MAX && f <= b && f <= y
 * Close and Open together at one j cancel (:337,:393).  Close is listed before Open, positions ascending.
 * Derivation: DESIGN.md "Stateless closed-syncmer rule".
 */
static void scan_stateless(const uint8_t *seq, int64_t len, int K, int S, read_out_t *o)
{
    vec8 bb = {0, 0, 0}, isn = {0, 0, 0};
    compress(seq, len, o, &bb, &isn);
    uint32_t n = o->hoco_l;
    if (n < (uint32_t) K) { free(bb.a); free(isn.a); return; }
    uint64_t *mend = (uint64_t *) malloc(sizeof(uint64_t) * n), *send = (uint64_t *) malloc(sizeof(uint64_t) * n);
    uint32_t *lrun = (uint32_t *) malloc(sizeof(uint32_t) * n);
    smer_hashes(bb.a, isn.a, n, S, mend, send, lrun);
    int Q = K - S + 1;
    uint32_t j, nk = n - K + 1;
#define M(p)  ((int64_t) (p) < 0 || (uint64_t) (p) + S > n? UINT64_MAX : mend[(p) + S - 1])
#define SC(p) (send[(p) + S - 1])
    for (j = 0; j < nk; ++j) {
        uint32_t e = j + Q - 1, t;
        if (lrun[j + K - 1] < (uint32_t) K) continue;                         /* valid(j) */
        uint64_t x = M((int64_t) j - 1), f = M(j), y = M(e), b = UINT64_MAX;
        for (t = j; t < e; ++t) { uint64_t v = M(t); if (v < b) b = v; }       /* O(Q) on purpose: this is the checker */
        int nextok = (j + K == n) || !isn.a[j + K];
        int cl = y != UINT64_MAX && (y < b || (y == b && (x >= b || f == b)));
        int op = nextok && f != UINT64_MAX && f <= b && f <= y;
        if (cl && op) continue;
        if (cl) emit(o, j, SC(e) ^ 1, (uint32_t) (SC(e) & 1), K);
        if (op) emit(o, j, SC(j), (uint32_t) (SC(j) & 1), K);
    }
#undef M
#undef SC
    free(mend); free(send); free(lrun); free(bb.a); free(isn.a);
}

static void append(void **dst, uint64_t *cnt, const void *src, size_t n, size_t sz)
{
    if (!n) return;
    *dst = realloc(*dst, (*cnt + n) * sz);
    memcpy((char *) *dst + *cnt * sz, src, n * sz);
    *cnt += n;
}

orc_scan_t *orc_scan_batch(const uint8_t *seq, const uint64_t *off, uint64_t n_reads, int K, int S, int mode)
{
    orc_scan_t *r = (orc_scan_t *) calloc(1, sizeof(orc_scan_t));
    uint64_t i, c_scm2 = 0, c_scm3 = 0;
    r->n_reads = n_reads;
    r->hoco_l = (uint32_t *) calloc(n_reads + 1, sizeof(uint32_t));
    r->n_scm = (uint32_t *) calloc(n_reads + 1, sizeof(uint32_t));
    r->n_lrl = (uint32_t *) calloc(n_reads + 1, sizeof(uint32_t));
    r->n_nn = (uint32_t *) calloc(n_reads + 1, sizeof(uint32_t));
    for (i = 0; i < n_reads; ++i) {
        read_out_t o;
        memset(&o, 0, sizeof(o));
        if (mode == 0) scan_streaming(seq + off[i], (int64_t) (off[i + 1] - off[i]), K, S, &o);
        else scan_stateless(seq + off[i], (int64_t) (off[i + 1] - off[i]), K, S, &o);
        r->hoco_l[i] = o.hoco_l;
        r->n_scm[i] = (uint32_t) o.m_pos.n;
        r->n_lrl[i] = (uint32_t) o.ho_l_rl.n;
        r->n_nn[i] = (uint32_t) o.n_nucl.n;
        append((void **) &r->hoco_s, &r->tot_bytes, o.hoco_s.a, o.hoco_s.n, 1);
        append((void **) &r->ho_rl, &r->tot_hoco, o.ho_rl.a, o.ho_rl.n, 1);
        append((void **) &r->ho_l_rl, &r->tot_lrl, o.ho_l_rl.a, o.ho_l_rl.n, 4);
        append((void **) &r->n_nucl, &r->tot_nn, o.n_nucl.a, o.n_nucl.n, 4);
        append((void **) &r->m_pos, &r->tot_scm, o.m_pos.a, o.m_pos.n, 4);
        append((void **) &r->s_mer, &c_scm2, o.s_mer.a, o.s_mer.n, 8);
        append((void **) &r->k_mer, &c_scm3, o.k_mer.a, o.k_mer.n, 8);
        free(o.hoco_s.a); free(o.ho_rl.a); free(o.ho_l_rl.a); free(o.n_nucl.a);
        free(o.m_pos.a); free(o.s_mer.a); free(o.k_mer.a);
    }
    return r;
}

void orc_scan_free(orc_scan_t *r)
{
    if (!r) return;
    free(r->hoco_l); free(r->n_scm); free(r->n_lrl); free(r->n_nn);
    free(r->hoco_s); free(r->ho_rl); free(r->ho_l_rl); free(r->n_nucl);
    free(r->m_pos); free(r->s_mer); free(r->k_mer);
    free(r);
}
