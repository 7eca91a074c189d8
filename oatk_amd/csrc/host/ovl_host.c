/*
 * oatk_amd/csrc/host/ovl_host.c -- host side of the drop-in boundary for calc_syncmer_overlap (syncasm.c:477-582) and
 * scg_unitig_consensus (syncasm.c:1004-1046).
 *
 * The device has reduced every pair of adjacent syncmers to its distinct distances in first-appearance order with counts
 * (oatk_hip_overlap_hist, include/oatk_hip_cons.h).  Here: the table of all pairs copied to the host once, a lookup by oriented pair,
 * a replica of the khashl<int,int> the reference tabulates in (identity hash, khashl.h:82-218) so that the most frequent distance comes
 * out with the reference's tie-break (bucket order) -- INCLUDING the table a caller keeps across calls: scg_unitig_consensus passes one
 * table for all pairs of a unitig and kh_clear keeps its size, so the bucket order of a later pair depends on the earlier ones -- and
 * scg_unitig_consensus itself on top of that and of the consensus arrays (cons_host.c).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_cons.h"
#include "oatk_syncasm.h"
#include "host_internal.h"

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

oatk_overlap_t *oatk_overlap_fetch(oatk_hip_ctx *ctx, int *rc)
{
    int r = 0;
    uint64_t np = 0, ne = 0;
    if (!rc) rc = &r;
    *rc = oatk_hip_overlap_hist(ctx, &np, &ne);
    if (*rc) return 0;
    return oatk_host_overlap_from_resident(ctx, np, ne, rc);
}

oatk_overlap_t *oatk_host_overlap_from_resident(oatk_hip_ctx *ctx, uint64_t np, uint64_t ne, int *rc)
{
    uint64_t b;
    oatk_overlap_t *o = (oatk_overlap_t *) calloc(1, sizeof(oatk_overlap_t));
    o->n_pairs = np, o->n_entries = ne;
    o->key = (uint64_t *) fetch(ctx, OATK_BUF_OVL_KEY, &b, rc); if (*rc) return 0;
    o->off = (uint64_t *) fetch(ctx, OATK_BUF_OVL_OFF, &b, rc); if (*rc) return 0;
    o->dist = (int32_t *) fetch(ctx, OATK_BUF_OVL_DIST, &b, rc); if (*rc) return 0;
    o->cnt = (uint32_t *) fetch(ctx, OATK_BUF_OVL_CNT, &b, rc); if (*rc) return 0;
    o->tail = (uint8_t *) fetch(ctx, OATK_BUF_OVL_TAIL, &b, rc); if (*rc) return 0;
    return o;
}

void oatk_overlap_destroy(oatk_overlap_t *o)
{
    if (!o) return;
    free(o->key); free(o->off); free(o->dist); free(o->cnt); free(o->tail);
    free(o);
}

/* the table of the pair v -> w (oriented syncmers, id << 1 | strand): the complementary pair w^1 -> v^1 is the same table */
int oatk_overlap_lookup(const oatk_overlap_t *o, uint64_t v, uint64_t w, const int32_t **dist, const uint32_t **cnt, int *tail_repeat)
{
    const uint64_t key = v <= w? v << 32 | w : (w ^ 1ULL) << 32 | (v ^ 1ULL);
    uint64_t lo = 0, hi = o->n_pairs;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (o->key[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo == o->n_pairs || o->key[lo] != key) { *dist = 0, *cnt = 0, *tail_repeat = 0; return 0; }
    *dist = o->dist + o->off[lo], *cnt = o->cnt + o->off[lo], *tail_repeat = o->tail[lo];
    return (int) (o->off[lo + 1] - o->off[lo]);
}

/* ---- khashl<int,int> with the identity hash (syncasm.c:63), as far as put / clear / iteration go ---- */
static uint32_t h2b(uint32_t hash, uint32_t bits) { return (hash * 2654435769U) >> (32 - bits); }      /* khashl.h:82 */
static int t_used(const oatk_ovl_table_t *h, uint32_t i) { return h->used[i >> 5] >> (i & 31U) & 1U; }
static void t_set(uint32_t *used, uint32_t i, int on) { if (on) used[i >> 5] |= 1U << (i & 31U); else used[i >> 5] &= ~(1U << (i & 31U)); }
static uint32_t fsize(uint32_t m) { return m < 32? 1 : m >> 5; }

oatk_ovl_table_t *oatk_ovl_table_new(void) { return (oatk_ovl_table_t *) calloc(1, sizeof(oatk_ovl_table_t)); }
void oatk_ovl_table_destroy(oatk_ovl_table_t *h) { if (h) { free(h->used); free(h->keys); free(h->vals); free(h); } }
static void t_clear(oatk_ovl_table_t *h)                                   /* khashl.h:121-127: the size stays */
{
    if (h->used) memset(h->used, 0, fsize(1U << h->bits) * 4), h->count = 0;
}
static void t_resize(oatk_ovl_table_t *h, uint32_t want)                   /* khashl.h:150-192, growth only */
{
    uint32_t j = 0, x = want;
    while ((x >>= 1) != 0) ++j;
    if (want & (want - 1)) ++j;
    const uint32_t nbits = j > 2? j : 2, n_new = 1U << nbits, n_old = h->keys? 1U << h->bits : 0U;
    if (h->count > (n_new >> 1) + (n_new >> 2)) return;
    uint32_t *nused = (uint32_t *) calloc(fsize(n_new), 4);
    if (n_old < n_new) {
        h->keys = (int32_t *) realloc(h->keys, 4 * (size_t) n_new);
        h->vals = (int32_t *) realloc(h->vals, 4 * (size_t) n_new);
    }
    for (j = 0; j != n_old; ++j) {
        if (!t_used(h, j)) continue;
        int32_t key = h->keys[j], val = h->vals[j];
        t_set(h->used, j, 0);
        for (;;) {                                                         /* the kick-out walk */
            uint32_t i = h2b((uint32_t) key, nbits);
            while (nused[i >> 5] >> (i & 31U) & 1U) i = (i + 1) & (n_new - 1);
            t_set(nused, i, 1);
            if (i < n_old && t_used(h, i)) {
                const int32_t tk = h->keys[i], tv = h->vals[i];
                h->keys[i] = key, h->vals[i] = val, key = tk, val = tv;
                t_set(h->used, i, 0);
            } else {
                h->keys[i] = key, h->vals[i] = val;
                break;
            }
        }
    }
    free(h->used);
    h->used = nused, h->bits = nbits;
}
static uint32_t t_put(oatk_ovl_table_t *h, int32_t key, int *absent)       /* khashl.h:195-218 */
{
    uint32_t n = h->keys? 1U << h->bits : 0U;
    if (h->count >= (n >> 1) + (n >> 2)) { t_resize(h, n + 1U); n = 1U << h->bits; }
    uint32_t i = h2b((uint32_t) key, h->bits);
    const uint32_t last = i;
    while (t_used(h, i) && h->keys[i] != key) { i = (i + 1U) & (n - 1); if (i == last) break; }
    if (!t_used(h, i)) h->keys[i] = key, t_set(h->used, i, 1), ++h->count, *absent = 1;
    else *absent = 0;
    return i;
}

/* calc_syncmer_overlap(sr_db, m1, rc1, m2, rc2, hm): v = id(m1) << 1 | rc1, w = id(m2) << 1 | rc2; hm == NULL: a fresh table */
int oatk_calc_syncmer_overlap(const oatk_overlap_t *o, uint64_t v, uint64_t w, oatk_ovl_table_t *hm)
{
    oatk_ovl_table_t *h = hm? hm : oatk_ovl_table_new();
    const int32_t *dist;
    const uint32_t *cnt;
    int tail, absent, i, n = oatk_overlap_lookup(o, v, w, &dist, &cnt, &tail);
    t_clear(h);
    for (i = 0; i < n; ++i) { const uint32_t k = t_put(h, dist[i], &absent); h->vals[k] = (int32_t) cnt[i]; }
    if (n && tail) (void) t_put(h, dist[0], &absent);                      /* the walk's last call was a repeat: it may still grow the table */
    int movl = 0, mcnt = 0;                                                /* syncasm.c:558-571 */
    uint32_t k, nb = h->keys? 1U << h->bits : 0U;
    for (k = 0; k < nb; ++k) if (t_used(h, k) && h->vals[k] > mcnt) mcnt = h->vals[k], movl = h->keys[k];
    if (!hm) oatk_ovl_table_destroy(h);
    return movl;
}

/* scg_unitig_consensus (syncasm.c:1004-1046): v[0..n) are the oriented syncmers of the unitig (vtx.a).  One pass: `here` is where syncmer i
 * starts on the unitig (the running sum of the pair distances, each looked up once, in order, through ONE table whose size carries over),
 * `done` where the sequence written so far ends.  A syncmer whose successor still starts inside the written part adds nothing; the others
 * contribute their consensus from the first position not yet written. */
int64_t oatk_scg_unitig_consensus(const oatk_consensus_t *cs, const oatk_overlap_t *o, const oatk_sr_db_t *sr_db, const uint64_t *v, uint64_t n,
                                  oatk_kstring_t *c_seq, int hoco_seq)
{
    int64_t here = 0, done = 0, total = 0;
    oatk_ovl_table_t *tab = n? oatk_ovl_table_new() : NULL;
    for (uint64_t i = 0; i < n; ++i) {
        const int last = i + 1 == n;
        const int64_t next = last? 0 : here + oatk_calc_syncmer_overlap(o, v[i], v[i + 1], tab);
        if (last || next > done) {
            const int64_t got = oatk_scg_syncmer_consensus(cs, sr_db, v[i] >> 1, (int) (v[i] & 1), done - here, c_seq, hoco_seq);
            if (got < 0) { total = -1; break; }                            /* a syncmer without prepared consensus: the caller's own routine */
            total += got;
            done = here + sr_db->k;
        }
        here = next;
    }
    if (tab) oatk_ovl_table_destroy(tab);
    return total;
}
