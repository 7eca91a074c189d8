/*
 * oracle/ec.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates the per-read syncmer-chain error correction of oatk's syncasm on flat arrays:
 *   orc_find_error_syncmers   syncerr.c:679-757  (+ asmg_vtx_del, graph.h:101-122)
 *   orc_ec_reads              syncerr.c:339-668 (block finder), :144-332 (DFS over the good-syncmer graph with the
 *                             resumable wavefront edit distance of oracle/levdist.c)
 *   orc_update_db             syncerr.c:769-814
 * The graph is the reference's asmg_t flattened in arc-array order (sorted (v,w), graph.c:70-83): CSR idx_p/idx_n per
 * oriented vertex, arcs {w, ls, cov, del}, vertices {len, seq}.  In this graph every vertex is one syncmer
 * (utg id == syncmer id, syncerr.c:421-423).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define EC_FAILURE 0
#define EC_SUCCESS 1
#define EC_AMBISNQ 2
#define EC_AMBISEQ 3
#define MAX_DFS_PATH 10000      /* syncerr.c:142 */
#define MIN_ERR_SEQ_LEN 10      /* syncerr.c:334 */
#define MIN_ERR_BASE 6          /* syncerr.c:335 */

typedef struct { char *s; size_t l, m; } str_t;
typedef struct { uint64_t *a; size_t n, m; } v64_t;
typedef struct { uint32_t *a; size_t n, m; } v32_t;

static void str_put(str_t *s, const char *p, size_t l)
{
    if (s->l + l + 1 > s->m) { s->m = (s->l + l + 1) * 2; s->s = (char *) realloc(s->s, s->m); }
    memcpy(s->s + s->l, p, l);
    s->l += l;
    s->s[s->l] = 0;
}
static char comp(char c) { return c == 'A'? 'T' : c == 'C'? 'G' : c == 'G'? 'C' : c == 'T'? 'A' : c; }
static void str_put_rc(str_t *s, const char *p, size_t l)
{
    size_t i;
    if (s->l + l + 1 > s->m) { s->m = (s->l + l + 1) * 2; s->s = (char *) realloc(s->s, s->m); }
    for (i = 0; i < l; ++i) s->s[s->l + i] = comp(p[l - 1 - i]);
    s->l += l;
    s->s[s->l] = 0;
}
#define VPUSH(v, T, x) do { if ((v).n == (v).m) { (v).m = (v).m? (v).m * 2 : 16; (v).a = (T *) realloc((v).a, (v).m * sizeof(T)); } (v).a[(v).n++] = (x); } while (0)

/* ---- find_error_syncmers, syncerr.c:679-757 ---- */
int64_t orc_find_error_syncmers(const orc_graph_t *g, const uint32_t *scm_cov, uint8_t *scm_del, uint32_t err_mer_c, uint32_t max_err_c,
                                uint32_t err_arc_c, double max_arc_f)
{
    uint64_t i, j, k, n = g->n_vtx;
    int64_t n_err = 0;
    for (i = 0; i < n; ++i) {
        if (scm_del[i] || scm_cov[i] >= max_err_c) continue;
        if (scm_cov[i] < err_mer_c) { scm_del[i] = 1; continue; }
        uint32_t nv = scm_cov[i];
        int b[2] = {-1, -1};
        for (k = 0; k < 2; ++k) {
            uint64_t v = i << 1 | k, p = g->idx_p[v], na = g->idx_n[v], live = 0;
            for (j = 0; j < na; ++j) if (!g->arc_del[p + j]) ++live;
            if (!live) continue;
            b[k] = 0;
            for (j = 0; j < na; ++j) {
                if (g->arc_del[p + j]) continue;
                uint32_t nw = scm_cov[g->arc_w[p + j] >> 1], mn = nv < nw? nv : nw;
                if (g->arc_cov[p + j] >= err_arc_c && g->arc_cov[p + j] >= mn * max_arc_f) { b[k] = 1; break; }
            }
        }
        if (!b[0] || !b[1]) scm_del[i] = 1;
    }
    /* asmg_vtx_del(asmg, i, 1) for every marked syncmer: the vertex, its arcs on both strands, and their complements */
    for (i = 0; i < n; ++i) {
        if (!scm_del[i]) continue;
        ++n_err;
        g->vtx_del[i] = 1;
        for (k = 0; k < 2; ++k) {
            uint64_t v = i << 1 | k, p = g->idx_p[v], na = g->idx_n[v];
            for (j = 0; j < na; ++j) {
                g->arc_del[p + j] = 1;
                uint64_t cv = g->arc_w[p + j] ^ 1, cw = v ^ 1, q = g->idx_p[cv], nq = g->idx_n[cv], t;
                for (t = 0; t < nq; ++t) if (g->arc_w[q + t] == cw) g->arc_del[q + t] = 1;
            }
        }
    }
    return n_err;
}

/* ---- DFS, syncerr.c:144-286 ---- */
typedef struct {
    int status, n_path, edist, s_edist;
    str_t c_seq, opt_seq;
    v64_t c_path, opt_path;
} dfs_t;

static int v64_differ(const v64_t *a, const v64_t *b)
{
    if (a->n != b->n) return 1;
    return a->n && memcmp(a->a, b->a, a->n * sizeof(uint64_t)) != 0;
}

static void dfs_search(const orc_graph_t *g, dfs_t *d, uint64_t sink, orc_wf_t **pw, const char *ts, int32_t tl, int32_t bw)
{
    if (d->n_path >= MAX_DFS_PATH) return;
    const size_t l0 = d->c_seq.l, n0 = d->c_path.n;
    const uint64_t source = d->c_path.a[n0 - 1], p = g->idx_p[source], na = g->idx_n[source];
    orc_wf_t *saved = orc_wf_clone(*pw);                   /* snapshot of score, ends and the wavefront, :165-171 */
    int32_t st0[3];
    orc_wf_state(*pw, st0);                                /* score, t_end, q_end before this level */
    uint64_t i;
    for (i = 0; i < na; ++i) {
        if (g->arc_del[p + i]) continue;
        const uint64_t w = g->arc_w[p + i];
        const int64_t ls = (int64_t) g->arc_ls[p + i], l_seq = (int64_t) g->vtx_len[w >> 1];
        const char *k_seq = g->seq + g->vtx_seq_off[w >> 1];
        VPUSH(d->c_path, uint64_t, w);
        if (w & 1) str_put_rc(&d->c_seq, k_seq, (size_t) (l_seq - ls));    /* :186-190 */
        else str_put(&d->c_seq, k_seq + ls, (size_t) (l_seq - ls));
        int32_t r[3];
        orc_wf_step(*pw, d->c_seq.s, (int32_t) d->c_seq.l, r);            /* r = score, t_end, q_end */
        const int32_t ql = (int32_t) d->c_seq.l;
        const int32_t score = r[0] + tl - r[1];                            /* :209 */
        if (score <= bw && (sink == UINT64_MAX || sink == w)) {
            d->status = EC_SUCCESS;
            if (score <= d->edist) {
                if (r[1] > st0[1]) d->s_edist = d->edist;                  /* otherwise only an extension */
                d->edist = score;
                if (sink == UINT64_MAX && r[2] < ql) --d->c_path.n;        /* last syncmer only partially covered :232-233 */
                if (d->edist == d->s_edist) {
                    if ((size_t) r[2] != d->opt_seq.l || strncmp(d->c_seq.s, d->opt_seq.s, (size_t) r[2])) d->status = EC_AMBISEQ;
                    if (d->status == EC_SUCCESS && v64_differ(&d->c_path, &d->opt_path)) d->status = EC_AMBISNQ;
                }
                d->opt_seq.l = 0;
                str_put(&d->opt_seq, d->c_seq.s, (size_t) r[2]);
                d->opt_path.n = 0;
                { size_t t; for (t = 0; t < d->c_path.n; ++t) VPUSH(d->opt_path, uint64_t, d->c_path.a[t]); }
            } else if (score < d->s_edist) {
                d->s_edist = score;
            }
        }
        if (r[0] <= bw && ql - l_seq <= tl + bw && ((sink != UINT64_MAX && sink != w) || r[1] < tl))   /* :267-272 */
            dfs_search(g, d, sink, pw, ts, tl, bw);
        else
            d->n_path++;
        /* back to the state before this arc */
        d->c_path.n = n0;
        d->c_seq.l = l0;
        orc_wf_free(*pw);
        *pw = orc_wf_clone(saved);
    }
    orc_wf_free(saved);
}

static void hoco_dna(const uint8_t *hoco_s, uint32_t pos, int32_t l, int rev, char *out)   /* get_kmer_dna_seq, syncmer.c:1237-1254 */
{
    int32_t i;
    for (i = 0; i < l; ++i) {
        uint32_t p = pos + (uint32_t) i;
        out[i] = "ACGT"[(hoco_s[p >> 2] >> (((p & 3) ^ 3) << 1)) & 3];
    }
    if (rev) {
        int32_t a = 0, b = l - 1;
        for (; a < b; ++a, --b) { char t = out[a]; out[a] = comp(out[b]); out[b] = comp(t); }
        if (a == b) out[a] = comp(out[a]);
    }
}

/* ---- per-read driver, syncerr.c:339-612 ---- */
void orc_ec_reads(const orc_graph_t *g, const uint8_t *scm_del, const uint64_t *scm_s, int K, double max_edist, uint64_t n_reads,
                  const uint32_t *hoco_l, const uint8_t *hoco_s, const uint64_t *hoco_byte_off, const uint32_t *n_scm,
                  const uint64_t *k_mer, const uint32_t *m_pos, const uint64_t *s_mer, orc_ec_out_t *out)
{
    uint64_t rd, in_off = 0;
    v64_t okm = {0, 0, 0}, osm = {0, 0, 0};
    v32_t omp = {0, 0, 0};
    memset(out, 0, sizeof(*out));
    out->n_scm = (uint32_t *) calloc(n_reads + 1, sizeof(uint32_t));
    dfs_t d;
    memset(&d, 0, sizeof(d));
    str_t seq = {0, 0, 0};
    for (rd = 0; rd < n_reads; ++rd) {
        const int32_t n = (int32_t) n_scm[rd];
        const uint64_t *km = k_mer + in_off;
        const uint32_t *mp = m_pos + in_off;
        const uint8_t *hs = hoco_s + hoco_byte_off[rd];
        v64_t ck = {0, 0, 0};
        v32_t cm = {0, 0, 0};
        int updated = 1;
        int32_t beg = -1, end, j, l;
        for (;;) {
            uint32_t beg_pos = beg < 1? 0 : (mp[beg - 1] >> 1) + (uint32_t) K;
            beg_pos += MIN_ERR_SEQ_LEN;
            for (end = beg + 1; end < n; ++end)
                if (!scm_del[km[end] >> 1] && !(km[end] & 1) && (mp[end] >> 1) >= beg_pos) break;     /* :407-415 */
            if (beg >= 0 || end < n) {
                uint64_t beg_utg, end_utg;
                int r;
                if (beg < 0) {                                           /* no left anchor: search backwards from `end` :427-436 */
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
                int err = EC_FAILURE;
                if (l >= MIN_ERR_SEQ_LEN) {
                    if (seq.m < (size_t) l + 1) { seq.m = (size_t) l + 1; seq.s = (char *) realloc(seq.s, seq.m); }
                    hoco_dna(hs, beg_pos, l, r, seq.s);
                    int32_t bw = (int32_t) ceil(l * max_edist);
                    if (bw < MIN_ERR_BASE) bw = MIN_ERR_BASE;
                    d.status = EC_FAILURE, d.n_path = 0, d.edist = d.s_edist = INT32_MAX;
                    d.c_seq.l = d.opt_seq.l = 0, d.c_path.n = d.opt_path.n = 0;
                    VPUSH(d.c_path, uint64_t, beg_utg);
                    orc_wf_t *wf = orc_wf_new(seq.s, l, bw);
                    dfs_search(g, &d, end_utg, &wf, seq.s, l, bw);
                    orc_wf_free(wf);
                    err = d.status;
                    if (end_utg == UINT64_MAX) ++out->stats[0], ++out->stats[1 + err];
                    else ++out->stats[5], ++out->stats[6 + err];
                } else {
                    ++out->stats[10];
                }
                if (err == EC_SUCCESS) {                                   /* splice the path's interior :513-532 */
                    const int32_t np = (int32_t) d.opt_path.n;
                    if (r) {
                        for (j = np - 1; j > 0; --j) {
                            VPUSH(ck, uint64_t, (d.opt_path.a[j] & ~1ULL) | 1);
                            VPUSH(cm, uint32_t, UINT32_MAX ^ (uint32_t) (d.opt_path.a[j] & 1));
                        }
                    } else {
                        for (j = 1; j < np - 1; ++j) {
                            VPUSH(ck, uint64_t, (d.opt_path.a[j] & ~1ULL) | 1);
                            VPUSH(cm, uint32_t, 0xFFFFFFFEu | (uint32_t) (d.opt_path.a[j] & 1));
                        }
                        if (end_utg == UINT64_MAX && np > 1) {
                            VPUSH(ck, uint64_t, (d.opt_path.a[j] & ~1ULL) | 1);
                            VPUSH(cm, uint32_t, 0xFFFFFFFEu | (uint32_t) (d.opt_path.a[j] & 1));
                        }
                    }
                } else if (r) {                                            /* keep the originals :533-542 */
                    for (j = 0; j < beg; ++j) { VPUSH(ck, uint64_t, km[j]); VPUSH(cm, uint32_t, mp[j]); }
                } else if (beg + 1 < n) {
                    for (j = beg + 1; j < end; ++j) { VPUSH(ck, uint64_t, km[j]); VPUSH(cm, uint32_t, mp[j]); }
                }
            } else {
                updated = 0;                                               /* no good syncmer on the read */
            }
            for (beg = end + 1; beg < n; ++beg)
                if (scm_del[km[beg] >> 1] || (km[end] & 1)) break;         /* [end], as written :579 */
            if (beg > n) break;
            for (j = end; j < beg; ++j) { VPUSH(ck, uint64_t, km[j]); VPUSH(cm, uint32_t, mp[j]); }
        }
        if (updated) {
            size_t t;
            out->n_scm[rd] = (uint32_t) ck.n;
            for (t = 0; t < ck.n; ++t) { VPUSH(okm, uint64_t, ck.a[t]); VPUSH(omp, uint32_t, cm.a[t]); VPUSH(osm, uint64_t, scm_s[ck.a[t] >> 1]); }
        } else {
            int32_t t;
            out->n_scm[rd] = (uint32_t) n;
            for (t = 0; t < n; ++t) { VPUSH(okm, uint64_t, km[t]); VPUSH(omp, uint32_t, mp[t]); VPUSH(osm, uint64_t, s_mer[in_off + (uint64_t) t]); }   /* untouched read keeps its arrays */
        }
        out->updated_reads += (uint64_t) updated;
        free(ck.a); free(cm.a);
        in_off += (uint64_t) n;
    }
    out->tot = okm.n;
    out->k_mer = okm.a, out->m_pos = omp.a, out->s_mer = osm.a;
    free(seq.s); free(d.c_seq.s); free(d.opt_seq.s); free(d.c_path.a); free(d.opt_path.a);
}

void orc_ec_out_free(orc_ec_out_t *o)
{
    free(o->n_scm); free(o->k_mer); free(o->m_pos); free(o->s_mer);
    memset(o, 0, sizeof(*o));
}

/* ---- update_syncmer_db, syncerr.c:769-814: recount coverage, rebuild occurrence lists in (sid, idx) order;
 * `del` becomes "no occurrence on the forward strand" (:803, :811-812) ---- */
void orc_update_db(uint64_t n_reads, const uint32_t *n_scm, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t n_syncmers,
                   uint32_t *cov, uint8_t *del, uint64_t *occ_off, uint64_t *occ)
{
    uint64_t i, j, o = 0;
    memset(cov, 0, sizeof(uint32_t) * n_syncmers);
    for (i = 0; i < n_reads; ++i) { for (j = 0; j < n_scm[i]; ++j) ++cov[k_mer[o + j] >> 1]; o += n_scm[i]; }
    occ_off[0] = 0;
    for (i = 0; i < n_syncmers; ++i) occ_off[i + 1] = occ_off[i] + cov[i];
    uint32_t *fill = (uint32_t *) calloc(n_syncmers, sizeof(uint32_t)), *fwd = (uint32_t *) calloc(n_syncmers, sizeof(uint32_t));
    for (i = 0, o = 0; i < n_reads; ++i) {
        for (j = 0; j < n_scm[i]; ++j) {
            uint64_t k = k_mer[o + j] >> 1;
            occ[occ_off[k] + fill[k]++] = i << 32 | j << 1 | (m_pos[o + j] & 1);
            if (!(m_pos[o + j] & 1)) ++fwd[k];
        }
        o += n_scm[i];
    }
    for (i = 0; i < n_syncmers; ++i) del[i] = !fwd[i];
    free(fill); free(fwd);
}
