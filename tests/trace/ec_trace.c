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
    uint64_t cat_arcs[6], cat_steps[6], cat_diag[6];   /* alive first / alive sibling / dead-by-score first / dead-by-score sibling / dead-otherwise first / sibling */
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
    for (j = 0; j < n; ++j) {
        int32_t k = s->k[j], d = s->d0 + j;
        if (k >= tl || k + d >= ql) continue;
        int32_t lim = (ql - d < tl? ql - d : tl) - 1;
        while (k < lim && ts[k + 1] == qs[k + d + 1]) ++k, s->c->cmp++;
        s->c->cmp++;
        if (k + d == ql - 1 || k == tl - 1) { s->t_end = k, s->q_end = k + d; return 1; }
        s->k[j] = k;
    }
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
        for (;;) {
            if (wf_step(s, ql)) break;
            ++s->score;
            if (s->score > s->bw) break;
        }
        s->c->steps_hist[bucket8(s->c->steps - st_before, SL)]++;
        s->t_end += 1, s->q_end += 1;
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
    }
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
                    dfs(&s, beg_utg, 0);
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
    fclose(fo);
    free(seq); free(s.cs); free(s.k); free(s.nk); free(s.vseen); free(s.aseen); free(memo.k); free(memo.sub_arcs);
    return n_blocks;
}
