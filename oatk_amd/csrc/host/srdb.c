/*
 * oatk_amd/csrc/host/srdb.c -- host side of the drop-in boundary: device results -> the reference's structs.
 *
 * oatk_sr_read_packed          is the body of sr_read (syncmer.c:487-556) once the reads are in memory,
 * oatk_collect_syncmer_from_reads is collect_syncmer_from_reads (syncmer.c:1397-1451),
 * both computed on the MI355X through the C ABI of include/oatk_hip.h.  What remains on the host is what the struct
 * layout forces: one malloc + memcpy per member array per read (sr_destroy frees each of them, syncmer.c:1047-1058).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_syncasm.h"

static void *xmalloc(size_t n)
{
    void *p = malloc(n? n : 1);
    if (!p) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    return p;
}

/* copy one resident device buffer to a fresh host array */
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

int oatk_sr_read_packed(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint8_t *seq, const uint64_t *off, const uint32_t *len,
                        uint64_t n_reads, uint64_t seq_bytes, char **names)
{
    int rc = oatk_hip_scan_host(ctx, seq, off, len, n_reads, seq_bytes, 0, sr_db->k, sr_db->s);
    if (rc) return rc;
    return oatk_sr_db_fill_resident(ctx, sr_db, off, n_reads, names);
}

/* sr_db->a[0 .. n_reads) from the scan resident in ctx; off[i] = offset of read i in the packed stream that was scanned */
int oatk_sr_db_fill_resident(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint64_t *off, uint64_t n_reads, char **names)
{
    int rc = 0;
    if (n_reads == 0) return OATK_OK;

    uint64_t b;
    uint32_t *hoco_l = 0, *n_scm = 0, *n_nn = 0, *n_lrl = 0, *lrl_val = 0, *m_pos = 0;
    uint64_t *nn_key = 0, *s_mer = 0, *k_hash = 0;
    hoco_l = (uint32_t *) fetch(ctx, OATK_BUF_HOCO_L, &b, &rc); if (rc) goto done;
    n_scm = (uint32_t *) fetch(ctx, OATK_BUF_N_SCM, &b, &rc); if (rc) goto done;
    n_nn = (uint32_t *) fetch(ctx, OATK_BUF_N_NN, &b, &rc); if (rc) goto done;
    n_lrl = (uint32_t *) fetch(ctx, OATK_BUF_N_LRL, &b, &rc); if (rc) goto done;
    /* the two big per-base arrays (1 + 1/4 byte per raw base) come over in pieces through page-locked memory, straight into the reads' own
     * blocks: no gigabyte-sized pageable landing buffer, PCIe at full speed */
    const void *d_rl = 0, *d_hs = 0;
    uint64_t b_rl = 0, b_hs = 0;
    rc = oatk_hip_buffer(ctx, OATK_BUF_HO_RL, &d_rl, &b_rl); if (rc) goto done;
    rc = oatk_hip_buffer(ctx, OATK_BUF_HOCO_S, &d_hs, &b_hs); if (rc) goto done;
    (void) b_rl; (void) b_hs;
    const uint64_t STAGE = 64ULL << 20;
    uint8_t *stage = (uint8_t *) oatk_hip_staging(ctx, STAGE + STAGE / 4 + 4096);
    if (!stage) { rc = OATK_E_NOMEM; goto done; }
    nn_key = (uint64_t *) fetch(ctx, OATK_BUF_NN_KEY, &b, &rc); if (rc) goto done;
    lrl_val = (uint32_t *) fetch(ctx, OATK_BUF_LRL_VAL, &b, &rc); if (rc) goto done;
    m_pos = (uint32_t *) fetch(ctx, OATK_BUF_POS_MPOS, &b, &rc); if (rc) goto done;
    s_mer = (uint64_t *) fetch(ctx, OATK_BUF_POS_SMER, &b, &rc); if (rc) goto done;
    k_hash = (uint64_t *) fetch(ctx, OATK_BUF_POS_HASH, &b, &rc); if (rc) goto done;

    /* zeroed, and counted only as far as it is filled: after a failure half way sr_db_destroy / oatk_sr_db_clean free what exists */
    sr_db->a = (oatk_sr_t *) calloc(n_reads, sizeof(oatk_sr_t));
    if (!sr_db->a) { rc = OATK_E_NOMEM; goto done; }
    sr_db->n = 0, sr_db->m = n_reads;
    uint64_t i, o_scm = 0, o_nn = 0, o_lrl = 0, stage_end = 0, stage_o0 = 0;
    uint8_t *stage_hs = stage;
    for (i = 0; i < n_reads; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        const uint32_t hl = hoco_l[i], ns = n_scm[i];
        const size_t nb = ((size_t) hl + 3) / 4;
        r->sid = i;                                        /* reads are numbered in input order, syncmer.c:525 */
        r->sname = names? names[i] : 0;
        r->hoco_l = hl;
        /* empty arrays are NULL in the reference (kvec never allocated), syncmer.c:396-412 */
        if (i == stage_end) {                              /* next piece: reads [i, j) whose packed range fits the staging block */
            uint64_t j = i + 1;
            const uint64_t o0 = off[i];
            while (j < n_reads && off[j] + (((uint64_t) hoco_l[j] + 63) & ~63ULL) - o0 <= STAGE) ++j;
            const uint64_t o1 = off[j - 1] + hoco_l[j - 1], bytes = o1 - o0;      /* one read longer than the block: the block grows */
            if (bytes > STAGE) { stage = (uint8_t *) oatk_hip_staging(ctx, bytes + bytes / 4 + 4096); if (!stage) { rc = OATK_E_NOMEM; goto done; } }
            stage_hs = stage + (((bytes > STAGE? bytes : STAGE) + 63) & ~63ULL);
            rc = oatk_hip_d2h(ctx, stage, (const uint8_t *) d_rl + o0, bytes); if (rc) goto done;
            rc = oatk_hip_d2h(ctx, stage_hs, (const uint8_t *) d_hs + o0 / 4, (bytes + 3) / 4 + 1); if (rc) goto done;
            stage_o0 = o0, stage_end = j;
        }
        r->hoco_s = nb? (uint8_t *) memcpy(xmalloc(nb), stage_hs + (off[i] - stage_o0) / 4, nb) : 0;
        r->ho_rl = hl? (uint8_t *) memcpy(xmalloc(hl), stage + (off[i] - stage_o0), hl) : 0;
        r->ho_l_rl = n_lrl[i]? (uint32_t *) memcpy(xmalloc(4 * (size_t) n_lrl[i]), lrl_val + o_lrl, 4 * (size_t) n_lrl[i]) : 0;
        r->n_nucl = 0;
        if (n_nn[i]) {
            uint32_t t;
            r->n_nucl = (uint32_t *) xmalloc(4 * (size_t) n_nn[i]);
            for (t = 0; t < n_nn[i]; ++t) r->n_nucl[t] = (uint32_t) nn_key[o_nn + t];   /* low word = raw coordinate */
        }
        r->n = ns;
        r->m_pos = ns? (uint32_t *) memcpy(xmalloc(4 * (size_t) ns), m_pos + o_scm, 4 * (size_t) ns) : 0;
        r->s_mer = ns? (uint64_t *) memcpy(xmalloc(8 * (size_t) ns), s_mer + o_scm, 8 * (size_t) ns) : 0;
        r->k_mer = ns? (uint64_t *) memcpy(xmalloc(8 * (size_t) ns), k_hash + o_scm, 8 * (size_t) ns) : 0;
        o_scm += ns, o_nn += n_nn[i], o_lrl += n_lrl[i];
        sr_db->n = i + 1;
    }
done:
    free(hoco_l); free(n_scm); free(n_nn); free(n_lrl); free(nn_key); free(lrl_val);
    free(m_pos); free(s_mer); free(k_hash);
    return rc;
}

oatk_syncmer_db_t *oatk_collect_syncmer_from_reads(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, int *rc_out)
{
    int rc = oatk_hip_count(ctx);
    if (rc_out) *rc_out = rc;
    if (rc == OATK_E_SMER) {                               /* fatal in the reference, syncmer.c:1370-1375 */
        fprintf(stderr, "[E::%s] identical kmers have different smers\n", __func__);
        exit(EXIT_FAILURE);
    }
    if (rc) return 0;
    oatk_hip_info_t inf;
    oatk_hip_info(ctx, &inf);
    if (inf.n_occ == 0) return 0;                          /* syncmer.c:1414-1417 */

    uint64_t b;
    uint64_t *h = 0, *s = 0, *occ_off = 0, *occ = 0, *kid = 0;
    uint32_t *cov = 0;
    h = (uint64_t *) fetch(ctx, OATK_BUF_SCM_H, &b, &rc); if (rc) goto fail;
    s = (uint64_t *) fetch(ctx, OATK_BUF_SCM_S, &b, &rc); if (rc) goto fail;
    cov = (uint32_t *) fetch(ctx, OATK_BUF_SCM_COV, &b, &rc); if (rc) goto fail;
    occ_off = (uint64_t *) fetch(ctx, OATK_BUF_SCM_OCC_OFF, &b, &rc); if (rc) goto fail;
    occ = (uint64_t *) fetch(ctx, OATK_BUF_SCM_OCC, &b, &rc); if (rc) goto fail;
    kid = (uint64_t *) fetch(ctx, OATK_BUF_POS_KID, &b, &rc); if (rc) goto fail;

    oatk_syncmer_db_t *db = (oatk_syncmer_db_t *) xmalloc(sizeof(oatk_syncmer_db_t));
    db->n = db->m = inf.n_scm;
    db->a = (oatk_syncmer_t *) xmalloc(sizeof(oatk_syncmer_t) * inf.n_scm);
    db->c = (uint16_t *) xmalloc(sizeof(uint16_t) * inf.n_scm);
    db->h = 0;
    uint64_t i, j, o = 0;
    for (i = 0; i < inf.n_scm; ++i) {
        oatk_syncmer_t *m = &db->a[i];
        m->h = h[i], m->s = s[i], m->cov = cov[i], m->del = 0;
        m->m_pos = (uint64_t *) memcpy(xmalloc(8 * (size_t) cov[i]), occ + occ_off[i], 8 * (size_t) cov[i]);
        db->c[i] = 1;                                      /* syncmer.c:1443-1444 */
    }
    /* reads: k-mer hash -> syncmer id << 1 (syncmer.c:1378) */
    for (i = 0; i < sr_db->n; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        for (j = 0; j < r->n; ++j) r->k_mer[j] = kid[o + j];
        o += r->n;
    }
    free(h); free(s); free(cov); free(occ_off); free(occ); free(kid);
    return db;
fail:
    free(h); free(s); free(cov); free(occ_off); free(occ); free(kid);
    if (rc_out) *rc_out = rc;
    return 0;
}

void oatk_sr_db_clean(oatk_sr_db_t *sr_db)
{
    size_t i;
    if (!sr_db) return;
    for (i = 0; i < sr_db->n; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        free(r->sname); free(r->hoco_s); free(r->ho_rl); free(r->ho_l_rl); free(r->n_nucl);
        free(r->m_pos); free(r->s_mer); free(r->k_mer);
    }
    free(sr_db->a);
    free(sr_db->stats);
    sr_db->a = 0, sr_db->n = sr_db->m = 0, sr_db->stats = 0;
}

void oatk_syncmer_db_destroy(oatk_syncmer_db_t *db)
{
    size_t i;
    if (!db) return;
    for (i = 0; i < db->n; ++i) free(db->a[i].m_pos);
    free(db->a); free(db->c); free(db->h); free(db);
}

/* malloc'ed, initialised like sr_db_init (syncmer.c:1060-1067); freed by the reference's sr_db_destroy or oatk_sr_db_clean + free */
oatk_sr_db_t *oatk_sr_db_new(int k, int s)
{
    oatk_sr_db_t *db = (oatk_sr_db_t *) xmalloc(sizeof(oatk_sr_db_t));
    db->n = db->m = 0, db->a = 0, db->k = k, db->s = s, db->stats = 0;
    return db;
}
