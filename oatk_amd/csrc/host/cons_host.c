/*
 * oatk_amd/csrc/host/cons_host.c -- host side of the drop-in boundary for scg_syncmer_consensus (syncasm.c:888-1003).
 *
 * The part of that function that costs -- adding up the run lengths of every occurrence of a syncmer at every k-mer position --
 * is done once for all live syncmers on the MI355X (oatk_hip_consensus).  What is left is string assembly: 'N' padding for a
 * negative `beg`, the bases of the first uncorrected occurrence (:912-936), each repeated 1 + rounded mean run length times.
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

oatk_consensus_t *oatk_consensus_fetch(oatk_hip_ctx *ctx, uint32_t min_cov, int k, int *rc)
{
    *rc = oatk_hip_consensus(ctx, min_cov);
    if (*rc) return 0;
    return oatk_host_consensus_from_resident(ctx, k, rc);
}

oatk_consensus_t *oatk_host_consensus_from_resident(oatk_hip_ctx *ctx, int k, int *rc)
{
    uint64_t b;
    oatk_consensus_t *c = (oatk_consensus_t *) calloc(1, sizeof(oatk_consensus_t));
    c->k = k;
    c->slot = (uint32_t *) fetch(ctx, OATK_BUF_CONS_SLOT, &b, rc); if (*rc) { oatk_consensus_destroy(c); return 0; }
    c->n_scm = b / 4;
    c->m_seq = (uint32_t *) fetch(ctx, OATK_BUF_CONS_MSEQ, &b, rc); if (*rc) { oatk_consensus_destroy(c); return 0; }
    c->n_sel = b / 4;
    c->rl = (uint32_t *) fetch(ctx, OATK_BUF_CONS_RL, &b, rc); if (*rc) { oatk_consensus_destroy(c); return 0; }
    c->first = (uint64_t *) fetch(ctx, OATK_BUF_CONS_FIRST, &b, rc); if (*rc) { oatk_consensus_destroy(c); return 0; }
    return c;
}

void oatk_consensus_destroy(oatk_consensus_t *c)
{
    if (!c) return;
    free(c->slot); free(c->m_seq); free(c->rl); free(c->first);
    free(c);
}

static void ks_put(oatk_kstring_t *s, char ch)          /* kputc_ (kstring.h): grow to the next power of two, no terminator */
{
    if (s->l + 1 > s->m) {
        size_t m = s->l + 2;
        --m, m |= m >> 1, m |= m >> 2, m |= m >> 4, m |= m >> 8, m |= m >> 16, m |= m >> 32, ++m;
        s->s = (char *) realloc(s->s, m);
        if (!s->s) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
        s->m = m;
    }
    s->s[s->l++] = ch;
}

int64_t oatk_scg_syncmer_consensus(const oatk_consensus_t *cs, const oatk_sr_db_t *sr_db, uint64_t scm_id, int rev, int64_t beg,
                                   oatk_kstring_t *c_seq, int hoco_seq)
{
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    const int w = cs->k;
    if (scm_id >= cs->n_scm || cs->slot[scm_id] == 0xFFFFFFFFu || beg >= w) return -1;      /* not prepared: use the reference's own routine */
    const uint32_t sl = cs->slot[scm_id];
    int64_t bl = beg < 0? -beg : 0, i;
    while (beg < 0) ks_put(c_seq, 'N'), ++beg;
    const int64_t l = w - beg;
    bl += l;
    if (cs->first[sl] == UINT64_MAX) {                     /* every occurrence was corrected away (:926-932) */
        for (i = 0; i < l; ++i) ks_put(c_seq, 'N');
        return bl;
    }
    const uint64_t o = cs->first[sl];
    const oatk_sr_t *s = &sr_db->a[o >> 32];
    uint64_t p = s->m_pos[(o >> 1) & 0x7FFFFFFFULL];
    const uint64_t r = (p & 1) ^ (uint64_t) rev;
    p >>= 1;
    const uint32_t *rl = cs->rl + (size_t) sl * (size_t) w;
    const uint32_t m = cs->m_seq[sl];
    for (i = 0; i < l; ++i) {
        const uint64_t q = r? p + (uint64_t) (l - 1 - i) : p + (uint64_t) (beg + i);      /* get_kmer_seq, syncmer.c:1218-1235 */
        uint32_t c = (s->hoco_s[q >> 2] >> (((q & 3) ^ 3) << 1)) & 3;
        if (r) c ^= 3;
        ks_put(c_seq, nt[c]);
        if (!hoco_seq && m) {
            const uint32_t b = rl[rev? w - 1 - (beg + i) : beg + i], j = 0;
            uint32_t t;
            (void) j;
            for (t = 0; t < b; ++t) ks_put(c_seq, nt[c]);
            bl += b;
        }
    }
    return bl;
}
