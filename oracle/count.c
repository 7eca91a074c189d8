/*
 * oracle/count.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates the syncmer count / ID assignment of oatk's syncasm
 * (reference syncmer.c:1397-1451 `collect_syncmer_from_reads`,
 * :1270-1393 `process_kmer_cluster`):
 *   - one 128-bit record per syncmer occurrence, (hash, sid<<32 | idx<<1 | rev)  :1410
 *   - sort ascending                                                            :1419
 *   - equal-hash groups; inside a group, occurrences are split by exact k-mer
 *     sequence in first-seen order (true 64-bit hash collisions)               :1293-1335
 *   - each cluster becomes one syncmer; IDs are dense in that order            :1353-1360
 *   - occurrence lists keep sorted (sid, idx) order; reads get k_mer = id<<1   :1365-1378
 *   - identical k-mers that carry different s-mers are fatal                   :1370-1376
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct { uint64_t h, lo, src; } rec_t;

static int rec_cmp(const void *a, const void *b)
{
    const rec_t *x = (const rec_t *) a, *y = (const rec_t *) b;
    if (x->h != y->h) return x->h < y->h? -1 : 1;
    if (x->lo != y->lo) return x->lo < y->lo? -1 : 1;
    return 0;
}

orc_count_t *orc_count(const orc_scan_t *sc, int K)
{
    orc_count_t *c = (orc_count_t *) calloc(1, sizeof(orc_count_t));
    uint64_t tot = sc->tot_scm, i, j, g;
    c->tot_occ = tot;
    if (tot == 0) return c;

    /* where each read's arrays start */
    uint64_t *scm_off = (uint64_t *) malloc(sizeof(uint64_t) * (sc->n_reads + 1));
    uint64_t *byte_off = (uint64_t *) malloc(sizeof(uint64_t) * (sc->n_reads + 1));
    scm_off[0] = byte_off[0] = 0;
    for (i = 0; i < sc->n_reads; ++i) {
        scm_off[i + 1] = scm_off[i] + sc->n_scm[i];
        byte_off[i + 1] = byte_off[i] + ((uint64_t) sc->hoco_l[i] + 3) / 4;
    }

    rec_t *r = (rec_t *) malloc(sizeof(rec_t) * tot);
    for (i = 0; i < sc->n_reads; ++i)
        for (j = 0; j < sc->n_scm[i]; ++j) {
            uint64_t t = scm_off[i] + j;
            r[t].h = sc->k_mer[t];
            r[t].lo = i << 32 | j << 1 | (sc->m_pos[t] & 1);
            r[t].src = t;
        }
    qsort(r, tot, sizeof(rec_t), rec_cmp);

    uint32_t nb = (uint32_t) (K - 1) / 4 + 1;
    uint64_t *clus = (uint64_t *) malloc(sizeof(uint64_t) * tot);   /* dense syncmer id of each sorted record */
    uint64_t n_scm = 0;
    uint8_t *reps = 0, *cur = (uint8_t *) malloc(nb);
    size_t reps_m = 0;
    for (g = 0; g < tot; ) {
        uint64_t e = g + 1;
        while (e < tot && r[e].h == r[g].h) ++e;
        if (e - g == 1) {
            clus[g] = n_scm++;
        } else {
            uint64_t n_clus = 0, t, q;
            for (t = g; t < e; ++t) {
                uint64_t sid = r[t].lo >> 32;
                uint32_t pos = sc->m_pos[r[t].src] >> 1, rev = (uint32_t) (r[t].lo & 1);
                orc_kmer_pack(sc->hoco_s + byte_off[sid], pos, rev, K, cur);
                for (q = 0; q < n_clus; ++q)
                    if (memcmp(cur, reps + q * nb, nb) == 0) break;
                if (q == n_clus) {
                    if ((n_clus + 1) * nb > reps_m) { reps_m = (n_clus + 1) * nb * 2; reps = (uint8_t *) realloc(reps, reps_m); }
                    memcpy(reps + n_clus * nb, cur, nb);
                    ++n_clus;
                }
                clus[t] = n_scm + q;
            }
            n_scm += n_clus;
        }
        g = e;
    }
    free(reps); free(cur);

    c->n_scm = n_scm;
    c->h = (uint64_t *) calloc(n_scm, sizeof(uint64_t));
    c->s = (uint64_t *) malloc(sizeof(uint64_t) * n_scm);
    c->cov = (uint32_t *) calloc(n_scm, sizeof(uint32_t));
    c->occ_off = (uint64_t *) calloc(n_scm + 1, sizeof(uint64_t));
    c->occ = (uint64_t *) malloc(sizeof(uint64_t) * tot);
    c->k_id = (uint64_t *) malloc(sizeof(uint64_t) * tot);
    memset(c->s, 0xff, sizeof(uint64_t) * n_scm);
    for (i = 0; i < tot; ++i) ++c->cov[clus[i]];
    for (i = 0; i < n_scm; ++i) c->occ_off[i + 1] = c->occ_off[i] + c->cov[i];
    uint64_t *fill = (uint64_t *) calloc(n_scm, sizeof(uint64_t));
    for (i = 0; i < tot; ++i) {
        uint64_t id = clus[i], sm = sc->s_mer[r[i].src];
        c->h[id] = r[i].h;
        c->occ[c->occ_off[id] + fill[id]++] = r[i].lo;
        if (c->s[id] == UINT64_MAX) c->s[id] = sm;
        else if (c->s[id] != sm) c->err = 1;
        c->k_id[r[i].src] = id << 1;
    }
    free(fill); free(clus); free(r); free(scm_off); free(byte_off);
    return c;
}

void orc_count_free(orc_count_t *c)
{
    if (!c) return;
    free(c->h); free(c->s); free(c->cov); free(c->occ_off); free(c->occ); free(c->k_id);
    free(c);
}
