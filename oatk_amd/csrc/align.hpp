// oatk_amd/csrc/align.hpp -- read -> unitig alignment, one lane per read.
//
// Replaces scg_ra_analysis_thread (alignment.c:180-594).  A read is its chain of <= a few dozen syncmers and the whole routine is a
// few hundred integer operations on arrays of that size -- the reference spends its time in malloc / qsort / pointer chasing per read,
// not in arithmetic -- so the mapping is the plain one: a lane owns a read, its working arrays live in a per-lane slab in HBM (L2
// resident: 10 KB per lane in flight), reads are taken round-robin.  Every step keeps the reference's order of operations because the
// results depend on it: the sort of the hits, the order fragments are pushed in, the STABLE sort of the fragments (glibc's qsort is a
// merge sort here), the chaining loop's early exit, the order predecessors are recorded and walked in.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "scan_hpc.hpp"

namespace oatk {

// The per-read working arrays of the lane-per-read routine: what almost every read needs (a read is a few dozen syncmers, each on one or two unitigs) and
// what the layout is tuned for -- 10 KB per lane, the walk's stack in LDS.  Reads over these limits (reads through tandem arrays, whose syncmers sit on
// hundreds of positions of one unitig: tens of thousands of hits) go through align_big.hpp, data-parallel inside the read.
struct RaSmall {
    static constexpr int MAXS = 160;      // syncmer hits per read
    static constexpr int MAXF = 128;      // fragments per read
    static constexpr int PREV = 6;        // recorded predecessors per fragment
    static constexpr int DEPTH = 48;      // fragments per alignment
    static constexpr int FBITS = 8;       // stack entry: the fragment in the low FBITS bits, the next predecessor to visit above them
    static constexpr bool LDS_STACK = true;
    static constexpr int THREADS = 256;
    typedef uint16_t st_t;
};
constexpr uint64_t RA_NONE = 0xFFFFFFFFFFFFFFFEULL;

// A lane's working arrays, element i of every lane next to each other: lanes that walk their arrays in step touch consecutive words
// (the routine is bound by memory transactions per lane, not by arithmetic; arrays of structs per lane made every access its own line).
template <typename T>
struct RaCol {
    T *p;                 // already offset by the lane
    uint64_t stride;      // lanes in the grid
    __device__ T &operator[](uint32_t i) const { return p[(uint64_t) i * stride]; }
};
struct RaHits { RaCol<uint64_t> uid, next; RaCol<uint32_t> u_pos, s_pos; };
struct RaFrgs {
    RaCol<uint64_t> uid;
    RaCol<uint32_t> u_beg, u_end, s_beg, s_end, s_cnt;
    RaCol<int32_t> score0, score;
    RaCol<uint16_t> prev_n, prev;         // prev[i * L::PREV + k]
    __device__ void copy(uint32_t dst, const RaFrgs &o, uint32_t src) const
    {
        uid[dst] = o.uid[src], u_beg[dst] = o.u_beg[src], u_end[dst] = o.u_end[src], s_beg[dst] = o.s_beg[src], s_end[dst] = o.s_end[src];
        s_cnt[dst] = o.s_cnt[src], score0[dst] = o.score0[src], score[dst] = o.score[src], prev_n[dst] = o.prev_n[src];
    }
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
    uint8_t *slab;                            // working arrays of all lanes, column by column (ra_slab_bytes)
    uint32_t *cnt_aln, *cnt_frg;              // [n_reads] pass 1
    uint8_t *skipped;                         // [n_reads]
    const uint64_t *aln_off, *frg_off;        // [n_reads + 1] pass 2
    const uint32_t *list;                     // align_big.hpp: the reads to take, n_list of them
    uint64_t n_list;
    unsigned long long *pool_used;            // [0] alignments, [1] fragments taken from the pool, [2] it ran short
    uint64_t pool_cap_a, pool_cap_f;
    uint64_t *pool_a, *pool_f;                // [n_reads] where a read's block starts in the pool
    uint32_t *o_sid;
    uint64_t *o_off;                          // [n_aln + 1]
    double *o_s;
    uint64_t *o_uid;
    uint32_t *o_ubeg, *o_uend, *o_sbeg, *o_send;
};

// bytes of working arrays per lane: hits, fragments as collected, fragments in sorted order, the sorted order itself (and RaBig's stack)
template <class L>
static inline uint64_t ra_slab_bytes(uint64_t lanes)
{
    const uint64_t lane_bytes = (uint64_t) L::MAXS * 24 + 2 * ((uint64_t) L::MAXF * (8 + 5 * 4 + 2 * 4 + 2 + 2 * L::PREV)) + (uint64_t) L::MAXF * 2 + (L::LDS_STACK? 0 : (uint64_t) L::DEPTH * 4);
    return lanes * lane_bytes + 256;
}

struct RaWork { RaHits S; RaFrgs F, G; RaCol<uint16_t> P; RaCol<uint32_t> stack; };
template <class L>
__device__ inline RaWork ra_work(uint8_t *slab, uint64_t tid, uint64_t nthr)
{
    constexpr int RA_MAXS = L::MAXS, RA_MAXF = L::MAXF, RA_PREV = L::PREV;
    RaWork w;
    uint8_t *q = slab;
    auto col64 = [&](uint64_t n) { RaCol<uint64_t> c = {(uint64_t *) q + tid, nthr}; q += n * nthr * 8; return c; };
    auto col32 = [&](uint64_t n) { RaCol<uint32_t> c = {(uint32_t *) q + tid, nthr}; q += n * nthr * 4; return c; };
    auto coli32 = [&](uint64_t n) { RaCol<int32_t> c = {(int32_t *) q + tid, nthr}; q += n * nthr * 4; return c; };
    auto col16 = [&](uint64_t n) { RaCol<uint16_t> c = {(uint16_t *) q + tid, nthr}; q += n * nthr * 2; return c; };
    w.S.uid = col64(RA_MAXS), w.S.next = col64(RA_MAXS);                           // 8-byte columns first: everything stays aligned
    w.F.uid = col64(RA_MAXF), w.G.uid = col64(RA_MAXF);
    w.S.u_pos = col32(RA_MAXS), w.S.s_pos = col32(RA_MAXS);
    for (RaFrgs *f : {&w.F, &w.G}) {
        f->u_beg = col32(RA_MAXF), f->u_end = col32(RA_MAXF), f->s_beg = col32(RA_MAXF), f->s_end = col32(RA_MAXF), f->s_cnt = col32(RA_MAXF);
        f->score0 = coli32(RA_MAXF), f->score = coli32(RA_MAXF);
    }
    for (RaFrgs *f : {&w.F, &w.G}) f->prev_n = col16(RA_MAXF), f->prev = col16((uint64_t) RA_MAXF * RA_PREV);
    w.P = col16(RA_MAXF);
    if (!L::LDS_STACK) { q = (uint8_t *) (((uintptr_t) q + 3) & ~(uintptr_t) 3); w.stack = col32(L::DEPTH); }
    else w.stack = RaCol<uint32_t>{nullptr, 0};
    return w;
}

// asmg_arc1 (graph.h:193-205): the first arc v -> w that is not deleted
__device__ inline int64_t ra_arc_ln(const RaArgs &a, uint64_t v, uint64_t w)
{
    const uint64_t p = a.idx_p[v], n = a.idx_n[v];
    for (uint64_t i = 0; i < n; ++i) if (a.arc_w[p + i] == w && !a.arc_del[p + i]) return (int64_t) a.arc_ln[p + i];
    return -1;
}

// Depth-first walk over the recorded predecessors from every fragment of maximal score; a chain's fragments come out earliest first.
// WR: write alignments from slot wa / fragments from slot wf on (offsets stored relative to f_base); otherwise only count.  Returns true
// when a chain is longer than the stack.
template <bool WR, class L, class Stack>
__device__ inline bool ra_backtrace(const RaArgs &a, const Stack &st, const RaFrgs &G, uint32_t nf, int64_t max_score, uint64_t n, uint64_t r,
                                    uint32_t tot_a, uint64_t wa, uint64_t wf, uint64_t f_base, uint32_t &n_a, uint32_t &n_fr)
{
    constexpr int RA_DEPTH = L::DEPTH, RA_PREV = L::PREV;
    constexpr uint32_t FMASK = (1u << L::FBITS) - 1u, CHILD1 = 1u << L::FBITS;
    typedef typename L::st_t st_t;
    // the walk's stack lives in LDS (st_mem, RA_DEPTH * 256 entries per workgroup), one column per lane: an array in registers indexed by a per-lane depth turns every access into a
    // loop over all its elements (measured: the walk took three times as long as everything before it).  An entry is the fragment
    // (< RA_MAXF = 128) in its low byte and the next predecessor to visit (<= RA_PREV) above it: 24 KB per workgroup, so that registers, not
    // LDS, decide how many of the 782 workgroups of 200 k reads are resident at once (all of them at four waves per SIMD).
    static_assert(L::MAXF <= (1 << L::FBITS) && (uint64_t) (L::PREV + 1) << L::FBITS <= (1ull << (8 * sizeof(st_t))), "stack entry: the fragment in the low bits, the child counter above them");
    n_a = 0, n_fr = 0;
    for (uint32_t j = 0; j < nf; ++j) {
        if (G.score[j] < max_score) continue;
        int d = 0;
        st[0] = (st_t) j;
        while (d >= 0) {
            const uint32_t e = st[d], f = e & FMASK, child = e >> L::FBITS;
            const uint32_t pn = G.prev_n[f];
            if (pn == 0) {                                                         // a chain is complete: its fragments are st[d .. 0]
                uint64_t s = 0;
                for (int t = d; t >= 0; --t) s += G.s_cnt[st[t] & FMASK];
                if (!((double) s / (double) n < 0.9)) {                            // min_a_frac (:161, :547)
                    if (WR) {
                        a.o_sid[wa] = (uint32_t) r, a.o_off[wa] = wf - f_base, a.o_s[wa] = 1.0 / (double) tot_a + (double) max_score;
                        for (int t = d; t >= 0; --t, ++wf) {
                            const uint32_t q = st[t] & FMASK;
                            a.o_uid[wf] = G.uid[q], a.o_ubeg[wf] = G.u_beg[q], a.o_uend[wf] = G.u_end[q], a.o_sbeg[wf] = G.s_beg[q], a.o_send[wf] = G.s_end[q];
                        }
                        ++wa;
                    }
                    ++n_a, n_fr += (uint32_t) d + 1;
                }
                --d;
            } else if (child < pn) {
                if (d + 1 == RA_DEPTH) return true;
                const uint16_t c = G.prev[f * RA_PREV + child];
                st[d] = (st_t) (e + CHILD1);
                ++d;
                st[d] = (st_t) c;
            } else --d;
        }
    }
    return false;
}

// Everything of scg_ra_analysis_thread up to the chained fragments for read r.  Returns 0: nothing to report, 1: the fragments G[0 .. nf) in
// sorted order with their best scores and predecessors are ready and max_score passes the read's threshold, 2: the read is over the limits (skipped[r] says which).
template <int MODE, class L>
__device__ inline int ra_prepare(const RaArgs &a, uint64_t r, const RaWork &w, uint32_t &nf, int64_t &max_score, uint64_t &n)
{
    constexpr int RA_MAXS = L::MAXS, RA_MAXF = L::MAXF, RA_PREV = L::PREV;
    const RaHits &S = w.S;
    const RaFrgs &F = w.F, &G = w.G;
    const RaCol<uint16_t> &P = w.P;
    const int64_t old = a.old_ra? a.old_ra[r] : 1;
    if ((old & 1) == 0) return 0;                                              // alignment.c:225
    const uint64_t co = a.chain_off[r];
    n = a.chain_off[r + 1] - co;
    if (n == 0) return 0;
    if (MODE == 1 && (a.skipped[r] || a.cnt_aln[r] == 0)) return 0;
    bool over = false;
    // (which limit a read is over goes straight into skipped[r]: 1 hits, 2 fragments, 3 predecessors; the caller adds 4, the depth of the walk)
    // ---- every position of every syncmer of the read on the unitigs (alignment.c:233-251), kept sorted by unitig, read position, unitig
    //      position as they arrive (sr_scm_cmpfunc :93-107; the order is total, so inserting in place equals sorting afterwards) ----
    uint32_t ns = 0;
    for (uint64_t j = 0; j < n && !over; ++j) {
        const uint64_t s = a.k_mer[co + j] >> 1;
        for (uint64_t k = a.su_off[s]; k < a.su_off[s + 1]; ++k) {
            if (ns == RA_MAXS) { over = true; if (MODE != 1) a.skipped[r] = 1; break; }
            const uint64_t x = a.su_uid[k], u = x >> 1, t = (x & 1ULL) ^ (a.m_pos[co + j] & 1u);
            const uint32_t p = a.su_pos[k];
            const uint64_t xu = u << 1 | t;
            const uint32_t xp = t? a.utg_n[u] - p - 1u : p, xs = (uint32_t) j;
            uint32_t i = ns;
            while (i > 0) {
                const uint64_t yu = S.uid[i - 1];
                bool gt = yu > xu;
                if (yu == xu) { const uint32_t ys = S.s_pos[i - 1]; gt = ys != xs? ys > xs : S.u_pos[i - 1] > xp; }
                if (!gt) break;
                S.uid[i] = yu, S.s_pos[i] = S.s_pos[i - 1], S.u_pos[i] = S.u_pos[i - 1], --i;
            }
            S.uid[i] = xu, S.s_pos[i] = xs, S.u_pos[i] = xp;
            ++ns;
        }
    }
    if (over) return 2;
    if (ns == 0) return 0;
    for (uint32_t i = 0; i < ns; ++i) S.next[i] = RA_NONE;
    // ---- fragments, unitig by unitig (:259-342) ----
    nf = 0;
    for (uint32_t j = 0; j < ns && !over; ) {
        const uint64_t u = S.uid[j];
        uint32_t p = j;
        while (++p < ns && S.uid[p] == u) {}
        // next mapping position of every hit: the closest larger unitig position among the hits of the next read position (:279-292)
        {
            uint32_t g0 = j, g1 = j;                                           // [g0, g1): hits of one read position
            while (g1 < p && S.s_pos[g1] == S.s_pos[g0]) ++g1;
            while (g1 < p) {
                uint32_t g2 = g1;
                while (g2 < p && S.s_pos[g2] == S.s_pos[g1]) ++g2;
                uint32_t s1 = g0, t1 = g1;
                while (s1 < g1) {
                    const uint32_t up = S.u_pos[s1];
                    while (t1 < g2 && S.u_pos[t1] <= up) ++t1;
                    if (t1 < g2 && S.u_pos[t1] > up) S.next[s1] = (uint64_t) t1 << 1;
                    ++s1;
                }
                g0 = g1, g1 = g2;
            }
        }
        // walk the links into fragments (:295-326)
        for (uint32_t k = j; k < p && !over; ++k) {
            uint32_t s = k;
            if (S.next[s] & 1ULL) continue;                                    // not a starting point
            const uint32_t u_beg = S.u_pos[s], s_beg = S.s_pos[s];
            uint32_t s_cnt = 1;
            int64_t u_gap = 0, s_gap = 0;
            for (;;) {
                const uint64_t nx = S.next[s], t = nx >> 1;
                if (t == 0x7FFFFFFFFFFFFFFFULL) break;
                const int64_t du = (int64_t) S.u_pos[(uint32_t) t] - (int64_t) S.u_pos[s], ds = (int64_t) S.s_pos[(uint32_t) t] - (int64_t) S.s_pos[s];
                u_gap += (du < 0? -du : du) - 1, s_gap += (ds < 0? -ds : ds) - 1;
                S.next[s] = nx | 1ULL;
                ++s_cnt;
                s = (uint32_t) t;
            }
            if (s_cnt == 1) continue;                                          // singletons come below
            S.next[s] |= 1ULL;
            if (s_gap > u_gap) u_gap = s_gap;
            if (u_gap < 0) u_gap = 0;
            const int64_t score = (int64_t) s_cnt - u_gap;                     // match_score = gap_penalty = 1 (:159-160)
            if (score >= 0) {
                if (nf == RA_MAXF) { over = true; if (MODE != 1) a.skipped[r] = 2; break; }
                F.uid[nf] = u, F.s_beg[nf] = s_beg, F.s_end[nf] = S.s_pos[s], F.s_cnt[nf] = s_cnt, F.u_beg[nf] = u_beg, F.u_end[nf] = S.u_pos[s];
                F.score0[nf] = F.score[nf] = (int32_t) score, F.prev_n[nf] = 0;
                ++nf;
            }
        }
        for (uint32_t k = j; k < p && !over; ++k) {                            // :329-336
            if (S.next[k] != RA_NONE) continue;
            if (nf == RA_MAXF) { over = true; if (MODE != 1) a.skipped[r] = 2; break; }
            const uint32_t sp = S.s_pos[k], up = S.u_pos[k];
            F.uid[nf] = u, F.s_beg[nf] = F.s_end[nf] = sp, F.s_cnt[nf] = 1, F.u_beg[nf] = F.u_end[nf] = up, F.score0[nf] = F.score[nf] = 1, F.prev_n[nf] = 0;
            ++nf;
        }
        j = p;
    }
    if (over) return 2;
    if (nf == 0) return 0;
    // stable sort by the read interval (sr_frg_cmpfunc :109-119 under glibc's merge sort): the order first, then the fragments moved once
    for (uint32_t i = 0; i < nf; ++i) {
        const uint32_t xb = F.s_beg[i], xe = F.s_end[i];
        uint32_t j = i;
        while (j > 0) {
            const uint32_t y = P[j - 1], yb = F.s_beg[y];
            const bool gt = yb != xb? yb > xb : F.s_end[y] > xe;
            if (!gt) break;
            P[j] = (uint16_t) y, --j;
        }
        P[j] = (uint16_t) i;
    }
    for (uint32_t i = 0; i < nf; ++i) G.copy(i, F, P[i]);
    // ---- chaining across arcs: no clipping, no gap, no overlap beyond the arc's (:440-476) ----
    for (uint32_t j = 0; j < nf && !over; ++j) {
        const uint64_t fu = G.uid[j];
        const int64_t p = G.s_end[j];
        if ((int64_t) a.utg_n[fu >> 1] - (int64_t) G.u_end[j] - 1 > 0) continue;
        const int64_t score = G.score[j];
        for (uint32_t k = j + 1; k < nf; ++k) {
            if (G.u_beg[k] > 0) continue;
            const int64_t ln = ra_arc_ln(a, fu, G.uid[k]);
            if (ln < 0) continue;
            const int64_t u_ovl = ln < p + 1? ln : p + 1, p1 = G.s_beg[k];
            if (p1 > p + 1) break;
            if (p1 + u_ovl != p + 1) continue;
            const int64_t score1 = score + G.score0[k] - u_ovl, sk = G.score[k];
            uint32_t pn = G.prev_n[k];
            if (score1 <= score || score1 < sk || (score1 == sk && pn == 0)) continue;
            if (score1 > sk) G.score[k] = (int32_t) score1, pn = 0;
            if (pn == RA_PREV) { over = true; if (MODE != 1) a.skipped[r] = 3; break; }
            G.prev[k * RA_PREV + pn] = (uint16_t) j;
            G.prev_n[k] = (uint16_t) (pn + 1);
        }
    }
    if (over) return 2;
    max_score = 0;
    for (uint32_t j = 0; j < nf; ++j) { const int32_t sc = G.score[j]; if (sc > max_score) max_score = sc; }
    if (max_score < (old >> 1)) return 0;                                      // :505
    return 1;
}

// MODE 0: count a read's alignments and fragments; 1: write them at the scanned offsets (a second run of the whole routine);
//      2: count, take room in a pool and write there (the normal path; 0 + 1 is the fallback when the pool is short).  The pool is
//      handed out per WAVE -- the lanes' needs are summed with DPP and one lane asks -- because 200 k lanes asking one by one serialise
//      on the two counters (measured: as long as the whole routine again).
template <class L> struct RaStackLds { typename L::st_t *p; __device__ typename L::st_t &operator[](int i) const { return p[i * L::THREADS]; } };
template <bool WR, class L>
__device__ __forceinline__ bool ra_walk(const RaArgs &a, const RaStackLds<L> &st_lds, const RaWork &w, uint32_t nf, int64_t max_score, uint64_t n, uint64_t r,
                                        uint32_t tot_a, uint64_t wa, uint64_t wf, uint64_t f_base, uint32_t &n_a, uint32_t &n_fr)
{
    if (L::LDS_STACK) return ra_backtrace<WR, L>(a, st_lds, w.G, nf, max_score, n, r, tot_a, wa, wf, f_base, n_a, n_fr);
    return ra_backtrace<WR, L>(a, w.stack, w.G, nf, max_score, n, r, tot_a, wa, wf, f_base, n_a, n_fr);
}
template <int MODE, class L>
__global__ __launch_bounds__(L::THREADS, L::LDS_STACK? 4 : 1) void ra_kernel(RaArgs a)
{
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t) gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ typename L::st_t st_mem[L::LDS_STACK? L::DEPTH * L::THREADS : 1];
    const RaWork w = ra_work<L>(a.slab, tid, nthr);
    const RaStackLds<L> st_lds = {st_mem + threadIdx.x};
    const uint64_t n_items = L::LDS_STACK? a.n_reads : a.n_list;
    for (uint64_t base = tid - lane; base < n_items; base += nthr) {                   // the same trip count in every lane of a wave
        const bool have = base + lane < n_items;
        const uint64_t r = L::LDS_STACK? base + lane : (have? (uint64_t) a.list[base + lane] : 0);
        uint32_t nf = 0, n_a = 0, n_fr = 0;
        int64_t max_score = 0;
        uint64_t n = 0;
        int st = 0;
        if (have) {
            if (MODE != 1) a.cnt_aln[r] = 0, a.cnt_frg[r] = 0, a.skipped[r] = 0;
            st = ra_prepare<MODE, L>(a, r, w, nf, max_score, n);
            if (MODE == 1) {
                if (st == 1) ra_walk<true, L>(a, st_lds, w, nf, max_score, n, r, a.cnt_aln[r], a.aln_off[r], a.frg_off[r], 0, n_a, n_fr);
            } else {
                if (st == 1 && ra_walk<false, L>(a, st_lds, w, nf, max_score, n, r, 0, 0, 0, 0, n_a, n_fr)) st = 2, a.skipped[r] = 4;
                if (st == 2) n_a = n_fr = 0;
                else a.cnt_aln[r] = n_a, a.cnt_frg[r] = n_fr;
            }
        }
        if (MODE == 2) {                                                           // every lane of the wave is here
            const uint32_t ia = wave_incl_sum_dpp(n_a, lane), ifr = wave_incl_sum_dpp(n_fr, lane);
            const uint32_t ta = (uint32_t) __builtin_amdgcn_readlane((int) ia, 63), tf = (uint32_t) __builtin_amdgcn_readlane((int) ifr, 63);
            uint64_t wa = 0, wf = 0;
            if (ta) {
                if (lane == 0) wa = atomicAdd(&a.pool_used[0], (unsigned long long) ta), wf = atomicAdd(&a.pool_used[1], (unsigned long long) tf);
                wa = (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (wa >> 32)) << 32 | (uint32_t) __builtin_amdgcn_readfirstlane((int) wa);
                wf = (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (wf >> 32)) << 32 | (uint32_t) __builtin_amdgcn_readfirstlane((int) wf);
                if (wa + ta > a.pool_cap_a || wf + tf > a.pool_cap_f) { if (lane == 0) a.pool_used[2] = 1ULL; }
                else if (n_a) {
                    const uint64_t pa = wa + ia - n_a, pf = wf + ifr - n_fr;
                    uint32_t x, y;
                    a.pool_a[r] = pa, a.pool_f[r] = pf;
                    ra_walk<true, L>(a, st_lds, w, nf, max_score, n, r, n_a, pa, pf, pf, x, y);
                }
            }
        }
    }
}

// pool -> read order (MODE 2): one lane per read moves its block; fragment offsets become absolute
struct RaOut { uint32_t *sid; uint64_t *off; double *s; uint64_t *uid; uint32_t *ubeg, *uend, *sbeg, *send; };
__global__ void ra_gather_kernel(uint64_t n_reads, const uint32_t *cnt_aln, const uint32_t *cnt_frg, const uint64_t *pool_a, const uint64_t *pool_f,
                                 const uint64_t *aln_off, const uint64_t *frg_off, RaOut src, RaOut dst)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t na = cnt_aln[r], nf = cnt_frg[r];
    if (na == 0) return;
    const uint64_t pa = pool_a[r], pf = pool_f[r], A = aln_off[r], Fo = frg_off[r];
    for (uint32_t i = 0; i < na; ++i) dst.sid[A + i] = src.sid[pa + i], dst.s[A + i] = src.s[pa + i], dst.off[A + i] = Fo + src.off[pa + i];
    for (uint32_t i = 0; i < nf; ++i) {
        dst.uid[Fo + i] = src.uid[pf + i], dst.ubeg[Fo + i] = src.ubeg[pf + i], dst.uend[Fo + i] = src.uend[pf + i];
        dst.sbeg[Fo + i] = src.sbeg[pf + i], dst.send[Fo + i] = src.send[pf + i];
    }
}


} // namespace oatk
