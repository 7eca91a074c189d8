// oatk_amd/csrc/align.hpp -- read -> unitig alignment, one lane per read.
//
// Replaces scg_ra_analysis_thread (alignment.c:180-594).  A read is its chain of <= a few dozen syncmers and the whole routine is a
// few hundred integer operations on arrays of that size -- the reference spends its time in malloc / qsort / pointer chasing per read,
// not in arithmetic -- so the mapping is the plain one: a lane owns a read, its working arrays live in a per-lane slab in HBM (L2
// resident: 10 KB per lane in flight), reads are taken round-robin.  Every step keeps the reference's order of operations because the
// results depend on it: the sort of the hits, the order fragments are pushed in, the STABLE sort of the fragments (glibc's qsort is a
// merge sort here), the chaining loop's early exit, the order predecessors are recorded and walked in.
#pragma once
#include "common.hpp"

namespace oatk {

constexpr int RA_MAXS = 160;      // syncmer hits per read
constexpr int RA_MAXF = 128;      // fragments per read
constexpr int RA_PREV = 6;        // recorded predecessors per fragment
constexpr int RA_DEPTH = 48;      // fragments per alignment
constexpr uint64_t RA_NONE = 0xFFFFFFFFFFFFFFFEULL;

struct RaScm { uint64_t uid, next; uint32_t u_pos, s_pos; };
struct RaFrg {
    uint64_t uid;
    uint32_t u_beg, u_end, s_beg, s_end, s_cnt;
    int32_t score0, score;
    uint16_t prev_n, prev[RA_PREV];
};

struct RaArgs {
    uint64_t n_reads, n_scm;
    const uint64_t *chain_off, *k_mer;
    const uint32_t *m_pos;
    const uint64_t *su_off, *su_uid;
    const uint32_t *su_pos, *utg_n;
    const uint64_t *idx_p, *idx_n, *arc_w, *arc_ln;
    const uint8_t *arc_del;
    const int64_t *old_ra;                    // may be null
    RaScm *scm_slab;                          // [threads * RA_MAXS]
    RaFrg *frg_slab;                          // [threads * RA_MAXF]
    uint32_t *cnt_aln, *cnt_frg;              // [n_reads] pass 1
    uint8_t *skipped;                         // [n_reads]
    const uint64_t *aln_off, *frg_off;        // [n_reads + 1] pass 2
    uint32_t *o_sid;
    uint64_t *o_off;                          // [n_aln + 1]
    double *o_s;
    uint64_t *o_uid;
    uint32_t *o_ubeg, *o_uend, *o_sbeg, *o_send;
};

// asmg_arc1 (graph.h:193-205): the first arc v -> w that is not deleted
__device__ inline int64_t ra_arc_ln(const RaArgs &a, uint64_t v, uint64_t w)
{
    const uint64_t p = a.idx_p[v], n = a.idx_n[v];
    for (uint64_t i = 0; i < n; ++i) if (a.arc_w[p + i] == w && !a.arc_del[p + i]) return (int64_t) a.arc_ln[p + i];
    return -1;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void ra_kernel(RaArgs a)
{
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t) gridDim.x * blockDim.x;
    RaScm *S = a.scm_slab + tid * RA_MAXS;
    RaFrg *F = a.frg_slab + tid * RA_MAXF;
    for (uint64_t r = tid; r < a.n_reads; r += nthr) {
        if (!WRITE) a.cnt_aln[r] = 0, a.cnt_frg[r] = 0, a.skipped[r] = 0;
        const int64_t old = a.old_ra? a.old_ra[r] : 1;
        if ((old & 1) == 0) continue;                                              // alignment.c:225
        const uint64_t co = a.chain_off[r], n = a.chain_off[r + 1] - co;
        if (n == 0) continue;
        if (WRITE && (a.skipped[r] || a.cnt_aln[r] == 0)) continue;
        bool over = false;
        // ---- every position of every syncmer of the read on the unitigs (alignment.c:233-251) ----
        uint32_t ns = 0;
        for (uint64_t j = 0; j < n && !over; ++j) {
            const uint64_t s = a.k_mer[co + j] >> 1;
            for (uint64_t k = a.su_off[s]; k < a.su_off[s + 1]; ++k) {
                if (ns == RA_MAXS) { over = true; break; }
                const uint64_t x = a.su_uid[k], u = x >> 1, t = (x & 1ULL) ^ (a.m_pos[co + j] & 1u);
                const uint32_t p = a.su_pos[k];
                S[ns].uid = u << 1 | t, S[ns].u_pos = t? a.utg_n[u] - p - 1u : p, S[ns].s_pos = (uint32_t) j, S[ns].next = RA_NONE;
                ++ns;
            }
        }
        if (over) { if (!WRITE) a.skipped[r] = 1; continue; }
        if (ns == 0) continue;
        // sort by unitig, read position, unitig position (sr_scm_cmpfunc :93-107; the order is total)
        for (uint32_t i = 1; i < ns; ++i) {
            const RaScm x = S[i];
            uint32_t j = i;
            while (j > 0) {
                const RaScm &y = S[j - 1];
                const bool gt = y.uid != x.uid? y.uid > x.uid : (y.s_pos != x.s_pos? y.s_pos > x.s_pos : y.u_pos > x.u_pos);
                if (!gt) break;
                S[j] = y, --j;
            }
            S[j] = x;
        }
        // ---- fragments, unitig by unitig (:259-342) ----
        uint32_t nf = 0;
        for (uint32_t j = 0; j < ns && !over; ) {
            const uint64_t u = S[j].uid;
            uint32_t p = j;
            while (++p < ns && S[p].uid == u) {}
            // next mapping position of every hit: the closest larger unitig position among the hits of the next read position (:279-292)
            {
                uint32_t g0 = j, g1 = j;                                           // [g0, g1): hits of one read position
                while (g1 < p && S[g1].s_pos == S[g0].s_pos) ++g1;
                while (g1 < p) {
                    uint32_t g2 = g1;
                    while (g2 < p && S[g2].s_pos == S[g1].s_pos) ++g2;
                    uint32_t s1 = g0, t1 = g1;
                    while (s1 < g1) {
                        while (t1 < g2 && S[t1].u_pos <= S[s1].u_pos) ++t1;
                        if (t1 < g2 && S[t1].u_pos > S[s1].u_pos) S[s1].next = (uint64_t) t1 << 1;
                        ++s1;
                    }
                    g0 = g1, g1 = g2;
                }
            }
            // walk the links into fragments (:295-326)
            for (uint32_t k = j; k < p && !over; ++k) {
                uint32_t s = k;
                if (S[s].next & 1ULL) continue;                                    // not a starting point
                const uint32_t u_beg = S[s].u_pos, s_beg = S[s].s_pos;
                uint32_t s_cnt = 1;
                int64_t u_gap = 0, s_gap = 0;
                for (;;) {
                    const uint64_t t = S[s].next >> 1;
                    if (t == 0x7FFFFFFFFFFFFFFFULL) break;
                    const int64_t du = (int64_t) S[t].u_pos - (int64_t) S[s].u_pos, ds = (int64_t) S[t].s_pos - (int64_t) S[s].s_pos;
                    u_gap += (du < 0? -du : du) - 1, s_gap += (ds < 0? -ds : ds) - 1;
                    S[s].next |= 1ULL;
                    ++s_cnt;
                    s = (uint32_t) t;
                }
                if (s_cnt == 1) continue;                                          // singletons come below
                S[s].next |= 1ULL;
                if (s_gap > u_gap) u_gap = s_gap;
                if (u_gap < 0) u_gap = 0;
                const int64_t score = (int64_t) s_cnt - u_gap;                     // match_score = gap_penalty = 1 (:159-160)
                if (score >= 0) {
                    if (nf == RA_MAXF) { over = true; break; }
                    RaFrg &f = F[nf++];
                    f.uid = u, f.s_beg = s_beg, f.s_end = S[s].s_pos, f.s_cnt = s_cnt, f.u_beg = u_beg, f.u_end = S[s].u_pos;
                    f.score0 = f.score = (int32_t) score, f.prev_n = 0;
                }
            }
            for (uint32_t k = j; k < p && !over; ++k) {                            // :329-336
                if (S[k].next != RA_NONE) continue;
                if (nf == RA_MAXF) { over = true; break; }
                RaFrg &f = F[nf++];
                f.uid = u, f.s_beg = f.s_end = S[k].s_pos, f.s_cnt = 1, f.u_beg = f.u_end = S[k].u_pos, f.score0 = f.score = 1, f.prev_n = 0;
            }
            j = p;
        }
        if (over) { if (!WRITE) a.skipped[r] = 1; continue; }
        if (nf == 0) continue;
        // stable sort by the read interval (sr_frg_cmpfunc :109-119 under glibc's merge sort)
        for (uint32_t i = 1; i < nf; ++i) {
            const RaFrg x = F[i];
            uint32_t j = i;
            while (j > 0) {
                const RaFrg &y = F[j - 1];
                const bool gt = y.s_beg != x.s_beg? y.s_beg > x.s_beg : y.s_end > x.s_end;
                if (!gt) break;
                F[j] = y, --j;
            }
            F[j] = x;
        }
        // ---- chaining across arcs: no clipping, no gap, no overlap beyond the arc's (:440-476) ----
        for (uint32_t j = 0; j < nf && !over; ++j) {
            const RaFrg &f = F[j];
            const int64_t p = f.s_end;
            if ((int64_t) a.utg_n[f.uid >> 1] - (int64_t) f.u_end - 1 > 0) continue;
            const int64_t score = f.score;
            for (uint32_t k = j + 1; k < nf; ++k) {
                RaFrg &f1 = F[k];
                if (f1.u_beg > 0) continue;
                const int64_t ln = ra_arc_ln(a, f.uid, f1.uid);
                if (ln < 0) continue;
                const int64_t u_ovl = ln < p + 1? ln : p + 1, p1 = f1.s_beg;
                if (p1 > p + 1) break;
                if (p1 + u_ovl != p + 1) continue;
                const int64_t score1 = score + f1.score0 - u_ovl;
                if (score1 <= score || score1 < f1.score || (score1 == f1.score && f1.prev_n == 0)) continue;
                if (score1 > f1.score) f1.score = (int32_t) score1, f1.prev_n = 0;
                if (f1.prev_n == RA_PREV) { over = true; break; }
                f1.prev[f1.prev_n++] = (uint16_t) j;
            }
        }
        if (over) { if (!WRITE) a.skipped[r] = 1; continue; }
        int64_t max_score = 0;
        for (uint32_t j = 0; j < nf; ++j) if (F[j].score > max_score) max_score = F[j].score;
        if (max_score < (old >> 1)) continue;                                      // :505
        // ---- all chains of maximal score, predecessors first (aln_frg_backtrace :132-157), kept when they cover 90 % of the read ----
        uint32_t n_a = 0, n_fr = 0;
        const uint32_t tot_a = WRITE? a.cnt_aln[r] : 0;
        uint64_t wa = WRITE? a.aln_off[r] : 0, wf = WRITE? a.frg_off[r] : 0;
        uint16_t st_node[RA_DEPTH], st_child[RA_DEPTH];
        for (uint32_t j = 0; j < nf && !over; ++j) {
            if (F[j].score < max_score) continue;
            int d = 0;
            st_node[0] = (uint16_t) j, st_child[0] = 0;
            while (d >= 0) {
                const RaFrg &f = F[st_node[d]];
                if (f.prev_n == 0) {                                               // a chain is complete: its fragments are st_node[d .. 0]
                    uint64_t s = 0;
                    for (int t = d; t >= 0; --t) s += F[st_node[t]].s_cnt;
                    if (!((double) s / (double) n < 0.9)) {                        // min_a_frac (:161, :547)
                        if (WRITE) {
                            a.o_sid[wa] = (uint32_t) r, a.o_off[wa] = wf, a.o_s[wa] = 1.0 / (double) tot_a + (double) max_score;
                            for (int t = d; t >= 0; --t, ++wf) {
                                const RaFrg &q = F[st_node[t]];
                                a.o_uid[wf] = q.uid, a.o_ubeg[wf] = q.u_beg, a.o_uend[wf] = q.u_end, a.o_sbeg[wf] = q.s_beg, a.o_send[wf] = q.s_end;
                            }
                            ++wa;
                        }
                        ++n_a, n_fr += (uint32_t) d + 1;
                    }
                    --d;
                } else if (st_child[d] < f.prev_n) {
                    if (d + 1 == RA_DEPTH) { over = true; break; }
                    const uint16_t c = f.prev[st_child[d]++];
                    ++d;
                    st_node[d] = c, st_child[d] = 0;
                } else --d;
            }
        }
        if (!WRITE) {
            if (over) a.skipped[r] = 1;
            else a.cnt_aln[r] = n_a, a.cnt_frg[r] = n_fr;
        }
    }
}

} // namespace oatk
