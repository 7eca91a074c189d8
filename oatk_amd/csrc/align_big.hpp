// oatk_amd/csrc/align_big.hpp -- read -> unitig alignment for the reads the lane-per-read routine (align.hpp) reports as over its limits.
//
// Same routine, same results (scg_ra_analysis_thread, alignment.c:180-594), for reads whose syncmers sit on hundreds of unitig positions each: a 15 kb read
// inside a tandem array has ~400 syncmers, all copies of one or two k-mers, the array's unitig carries each of them ~400 times, and the read has 10^5 hits
// (round 4, the config-1 surrogate: 1900 such reads in 100 k, 8 k - 160 k hits each; the reference sorts them with qsort and walks them on one thread per
// read).  One lane per read cannot do that (its insertion sorts are quadratic), so here the READ is the unit of a wave and the quadratic steps become
// device-wide sorts:
//
//   hits of all big reads of a batch, unsorted       rab_expand_kernel        (a wave per read: syncmer by syncmer, lanes over its unitig positions)
//   sorted by (read, unitig, read position, unitig position)   two stable radix sorts (rocPRIM), 48 + 55 key bits -- sr_scm_cmpfunc's order is total
//   next mapping position of every hit               rab_links_kernel         (a lane per hit: three binary searches in the read's sorted hits)
//   fragments: chains from hits nobody points to, singletons   rab_frags_kernel  (a lane per start, the chain walked by that lane; appended in any order, each
//                                                                               with its place in the reference's push order in the sort key)
//   sorted by (read, s_beg, s_end, push order)       one radix sort            -- the reference's stable qsort by (s_beg, s_end)
//   chaining, best score, the walk over predecessors rab_chain_kernel         (a wave per read: fragments j in order, lanes over the fragments k behind j)
//
// What the reference does in order and what can be done in any order: the hits' sort order is total, so how they are produced does not matter; a hit starts
// a fragment iff no hit points to it (alignment.c:295-326 walks starts in index order and marks what it passes: by induction a hit is passed iff it has a
// predecessor); fragments are pushed unitig by unitig, chains before singletons, in index order (:295-336) -- that rank is part of the sort key; the chaining
// loop (:440-476) is sequential in j, and for one j the fragments k are independent of each other (each k updates only its own score and list), the `break`
// cuts a prefix of the k because they are sorted by s_beg; the predecessors of k are recorded in j order, which is the order they arrive in.
#pragma once
#include "align.hpp"

namespace oatk {

constexpr int RAB_PREV = 32;                      // recorded predecessors per fragment (over it: the read is reported as skipped, code 3)
constexpr uint32_t RAB_NONE = 0xFFFFFFFFu;
constexpr int RAB_IDX_BITS = 12;                  // reads per batch < 4096
constexpr uint32_t RAB_MAX_SPOS = 1u << 15;       // syncmers per read the fragment sort key has room for
constexpr uint64_t RAB_MAX_HITS = 1ull << 21;     // hits per read the fragment sort key has room for (twice the index)

struct RabArgs {
    RaArgs a;
    const uint32_t *list;             // the batch's reads (read indices)
    uint32_t n_list;
    const uint64_t *hoff;             // [n_list + 1] hit segments
    const uint64_t *foff;             // [n_list + 1] fragment segments (sorted fragments), from the counts of rab_frags_kernel
    // hits
    uint64_t *key_lo, *key_hi;        // unsorted -> sorted: read position << 32 | unitig position; batch index << 43 | unitig << 1 | strand
    uint32_t *next;                   // index in the read's segment of the hit's next mapping position
    uint8_t *haspred;
    // fragments as collected (slots of the hit segments)
    uint64_t *f_uid, *f_key;
    uint32_t *f_sbeg, *f_send, *f_scnt, *f_ubeg, *f_uend;
    int32_t *f_score;
    uint32_t *f_cnt;                  // [n_list]
    const uint32_t *f_order;          // slots in sorted order
    // fragments in sorted order (segments foff)
    uint64_t *g_uid;
    uint32_t *g_sbeg, *g_send, *g_scnt, *g_ubeg, *g_uend, *g_prevn, *g_prev;
    int32_t *g_score0, *g_score;
    uint64_t *stack;                  // a walk's stack: fragment | child << 32
    int mode;                         // as ra_kernel's MODE
};

// hits of every big read: a wave per read
__global__ __launch_bounds__(256) void rab_count_hits_kernel(RaArgs a, const uint32_t *list, uint32_t n_list, unsigned long long *hits)
{
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= n_list) return;
    const uint64_t r = list[b], co = a.chain_off[r], n = a.chain_off[r + 1] - co;
    unsigned long long h = 0;
    for (uint64_t j = lane; j < n; j += 64) { const uint64_t s = a.k_mer[co + j] >> 1; h += a.su_off[s + 1] - a.su_off[s]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d);
    if (lane == 0) hits[b] = n >= RAB_MAX_SPOS? ~0ULL : h;             // (a read with 2^15 syncmers: no room in the fragments' sort key -- left to the caller like one with 2^21 hits)
}

// every position of every syncmer of the read on the unitigs (alignment.c:233-251)
__global__ __launch_bounds__(256) void rab_expand_kernel(RabArgs q)
{
    const RaArgs &a = q.a;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= q.n_list) return;
    const uint64_t r = q.list[b], co = a.chain_off[r], n = a.chain_off[r + 1] - co;
    uint64_t at = q.hoff[b];
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t s = a.k_mer[co + j] >> 1, k0 = a.su_off[s], k1 = a.su_off[s + 1];
        const uint32_t mp = a.m_pos[co + j] & 1u;
        for (uint64_t k = k0 + lane; k < k1; k += 64) {
            const uint64_t x = a.su_uid[k], u = x >> 1, t = (x & 1ULL) ^ mp;
            const uint32_t p = a.su_pos[k], xp = t? a.utg_n[u] - p - 1u : p;
            q.key_lo[at + (k - k0)] = j << 32 | xp;
            q.key_hi[at + (k - k0)] = (uint64_t) b << 43 | u << 1 | t;
        }
        at += k1 - k0;
    }
}

// first index in [lo, hi) whose (key_hi, key_lo) is greater than (kh, kl); the segment is sorted by that pair
__device__ __forceinline__ uint64_t rab_upper(const uint64_t *key_hi, const uint64_t *key_lo, uint64_t lo, uint64_t hi, uint64_t kh, uint64_t kl)
{
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1, mh = key_hi[mid];
        const bool gt = mh != kh? mh > kh : key_lo[mid] > kl;
        if (gt) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// next mapping position of every hit: the closest larger unitig position among the hits of the next read position on the same unitig (:279-292)
__global__ __launch_bounds__(256) void rab_links_kernel(RabArgs q)
{
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint64_t h0 = q.hoff[b], h1 = q.hoff[b + 1];
    for (uint64_t i = h0 + tid; i < h1; i += blockDim.x) {
        const uint64_t kh = q.key_hi[i], kl = q.key_lo[i];
        // the end of this read position's hits, the end of the next read position's
        const uint64_t g1 = rab_upper(q.key_hi, q.key_lo, i + 1, h1, kh, kl | 0xFFFFFFFFULL);
        uint32_t nx = RAB_NONE;
        if (g1 < h1 && q.key_hi[g1] == kh) {
            const uint64_t sp1 = q.key_lo[g1] >> 32;
            const uint64_t g2 = rab_upper(q.key_hi, q.key_lo, g1 + 1, h1, kh, sp1 << 32 | 0xFFFFFFFFULL);
            const uint64_t t1 = rab_upper(q.key_hi, q.key_lo, g1, g2, kh, sp1 << 32 | (kl & 0xFFFFFFFFULL));     // first with a larger unitig position
            if (t1 < g2) nx = (uint32_t) (t1 - h0);
        }
        q.next[i] = nx;
        if (nx != RAB_NONE) q.haspred[h0 + nx] = 1;
    }
}

// fragments (:295-336): the chain from every hit nobody points to; a hit nobody points to and that points nowhere is a fragment of its own.  Appended to the
// read's slots in any order; the key orders them as the reference's stable sort by (s_beg, s_end) of its push order does -- unitig by unitig (the hits
// are sorted by unitig, so a unitig is a range [ua, ub) of them), chains in index order, then singletons in index order: ranks ua + i and ub + i
__global__ __launch_bounds__(64) void rab_frags_kernel(RabArgs q)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint64_t h0 = q.hoff[b], h1 = q.hoff[b + 1];
    uint32_t nf = 0;
    for (uint64_t base = h0; base < h1; base += 64) {
        const uint64_t i = base + lane;
        bool have = false;
        uint64_t uid = 0, key = 0;
        uint32_t s_beg = 0, s_end = 0, s_cnt = 0, u_beg = 0, u_end = 0;
        int32_t score = 0;
        if (i < h1 && !q.haspred[i]) {
            const uint64_t kh = q.key_hi[i];
            uint64_t s = i, kl = q.key_lo[s];
            s_beg = (uint32_t) (kl >> 32), u_beg = (uint32_t) kl, s_cnt = 1;
            int64_t u_gap = 0, s_gap = 0;
            uint32_t nx;
            while ((nx = q.next[s]) != RAB_NONE) {
                const uint64_t t = h0 + nx, tl = q.key_lo[t];
                const int64_t du = (int64_t) (uint32_t) tl - (int64_t) (uint32_t) kl, ds = (int64_t) (tl >> 32) - (int64_t) (kl >> 32);
                u_gap += (du < 0? -du : du) - 1, s_gap += (ds < 0? -ds : ds) - 1;
                ++s_cnt;
                s = t, kl = tl;
            }
            s_end = (uint32_t) (kl >> 32), u_end = (uint32_t) kl;
            if (s_cnt == 1) score = 1, have = true;
            else {
                if (s_gap > u_gap) u_gap = s_gap;
                if (u_gap < 0) u_gap = 0;
                const int64_t sc = (int64_t) s_cnt - u_gap;                    // match_score = gap_penalty = 1 (:159-160)
                score = (int32_t) sc, have = sc >= 0;
            }
            if (have) {
                uid = kh & ((1ULL << 43) - 1);
                // the unitig's range of hits
                uint64_t lo = h0, hi = i;
                while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (q.key_hi[mid] < kh) lo = mid + 1; else hi = mid; }
                const uint64_t ua = lo;
                const uint64_t ub = rab_upper(q.key_hi, q.key_lo, i + 1, h1, kh, ~0ULL);
                const uint64_t rank = (s_cnt == 1? ub : ua) + i - 2 * h0;      // < 2 x the read's hits
                key = (uint64_t) b << 52 | (uint64_t) s_beg << 37 | (uint64_t) s_end << 22 | rank;
            }
        }
        const uint64_t m = __ballot(have);
        if (have) {
            const uint64_t slot = h0 + nf + (uint64_t) __builtin_popcountll(m & ((1ULL << lane) - 1));
            q.f_uid[slot] = uid, q.f_key[slot] = key, q.f_sbeg[slot] = s_beg, q.f_send[slot] = s_end, q.f_scnt[slot] = s_cnt, q.f_ubeg[slot] = u_beg, q.f_uend[slot] = u_end;
            q.f_score[slot] = score;
        }
        nf += (uint32_t) __builtin_popcountll(m);
    }
    if (lane == 0) q.f_cnt[b] = nf;
}

// Depth-first walk over the recorded predecessors from every fragment of maximal score (align.hpp: ra_backtrace, with the stack and the fragments in HBM)
template <bool WR>
__device__ inline void rab_backtrace(const RabArgs &q, uint64_t f0, uint32_t nf, int64_t max_score, uint64_t n, uint64_t r,
                                     uint32_t tot_a, uint64_t wa, uint64_t wf, uint64_t f_base, uint32_t &n_a, uint32_t &n_fr)
{
    const RaArgs &a = q.a;
    uint64_t *st = q.stack + f0;
    n_a = 0, n_fr = 0;
    // (round 6: the whole wave looks for the fragments of maximal score, sixty-four at a time; lane 0 walks from each.  One lane used to read every fragment's score, a round
    //  trip to HBM each)
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t jb = 0; jb < nf; jb += 64) {
      uint64_t top = __ballot(jb + lane < nf && !(q.g_score[f0 + jb + lane] < max_score));
      if (lane == 0) while (top) {          // (the others wait at the end of this statement: the next turn's ballot is the whole wave's again)
        const uint32_t j = jb + (uint32_t) __builtin_ctzll(top);
        top &= top - 1;
        int64_t d = 0;
        st[0] = j;
        while (d >= 0) {
            const uint64_t e = st[d];
            const uint32_t f = (uint32_t) e, child = (uint32_t) (e >> 32);
            const uint32_t pn = q.g_prevn[f0 + f];
            if (pn == 0) {                                                         // a chain is complete: its fragments are st[d .. 0]
                uint64_t s = 0;
                for (int64_t t = d; t >= 0; --t) s += q.g_scnt[f0 + (uint32_t) st[t]];
                if (!((double) s / (double) n < 0.9)) {                            // min_a_frac (:161, :547)
                    if (WR) {
                        a.o_sid[wa] = (uint32_t) r, a.o_off[wa] = wf - f_base, a.o_s[wa] = 1.0 / (double) tot_a + (double) max_score;
                        for (int64_t t = d; t >= 0; --t, ++wf) {
                            const uint64_t g = f0 + (uint32_t) st[t];
                            a.o_uid[wf] = q.g_uid[g], a.o_ubeg[wf] = q.g_ubeg[g], a.o_uend[wf] = q.g_uend[g], a.o_sbeg[wf] = q.g_sbeg[g], a.o_send[wf] = q.g_send[g];
                        }
                        ++wa;
                    }
                    ++n_a, n_fr += (uint32_t) d + 1;
                }
                --d;
            } else if (child < pn) {
                const uint32_t c = q.g_prev[(f0 + f) * RAB_PREV + child];
                st[d] = e + (1ULL << 32);
                ++d;                                                               // (a chain is shorter than the read has fragments: the stack is nf long)
                st[d] = c;
            } else --d;
        }
      }
    }
}

// chaining across arcs (:440-476), the best score, the alignments: a wave per read
__global__ __launch_bounds__(64) void rab_chain_kernel(RabArgs q)
{
    const RaArgs &a = q.a;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint64_t r = q.list[b], f0 = q.foff[b];
    const uint32_t nf = (uint32_t) (q.foff[b + 1] - f0);
    const uint64_t n = a.chain_off[r + 1] - a.chain_off[r];
    const int64_t old = a.old_ra? a.old_ra[r] : 1;
    if (q.mode == 1 && (a.skipped[r] || a.cnt_aln[r] == 0)) return;
    // the fragments in sorted order
    for (uint32_t i = lane; i < nf; i += 64) {
        const uint64_t s = q.f_order[f0 + i];                                      // (a slot of the hit segments)
        const uint64_t g = f0 + i;
        q.g_uid[g] = q.f_uid[s], q.g_sbeg[g] = q.f_sbeg[s], q.g_send[g] = q.f_send[s], q.g_scnt[g] = q.f_scnt[s], q.g_ubeg[g] = q.f_ubeg[s], q.g_uend[g] = q.f_uend[s];
        q.g_score0[g] = q.g_score[g] = q.f_score[s], q.g_prevn[g] = 0;
    }
    __threadfence_block();
    bool over = false;
    // (round 6: what decides whether fragment j can be a predecessor at all -- its unitig, its end on the read, whether it reaches its unitig's end -- never changes: sixty-four
    //  fragments' worth is fetched at once, a lane each, and only those that can go on take a turn.  The loop used to fetch them fragment by fragment: three dependent
    //  round trips to HBM per fragment, 10^4 - 10^5 fragments a read, and a batch of reads with many hits costs what its slowest read costs.)
    for (uint32_t jb = 0; jb < nf && !over; jb += 64) {
        const uint32_t jm = jb + lane;
        uint64_t my_fu = 0;
        int64_t my_p = 0;
        bool my_go = false;
        if (jm < nf) {
            my_fu = q.g_uid[f0 + jm], my_p = q.g_send[f0 + jm];
            my_go = !((int64_t) a.utg_n[my_fu >> 1] - (int64_t) q.g_uend[f0 + jm] - 1 > 0);
        }
        uint64_t go = __ballot(my_go);
        while (go && !over) {
            const int l = __builtin_ctzll(go);
            go &= go - 1;
            const uint32_t j = jb + (uint32_t) l;
            const uint64_t fu = (uint64_t) __shfl((long long) my_fu, l);
            const int64_t p = (int64_t) __shfl((long long) my_p, l);
            const int64_t score = q.g_score[f0 + j];                                // (as the fragments before it have left it)
            for (uint32_t kb = j + 1; kb < nf; kb += 64) {
                const uint32_t k = kb + lane;
                bool past = false, ov = false;
                if (k < nf) {
                    const uint64_t g = f0 + k;
                    const int64_t p1 = q.g_sbeg[g];
                    if (p1 > p + 1) past = true;                                   // (sorted by s_beg: so is everything behind k)
                    else if (q.g_ubeg[g] == 0) {
                        const int64_t ln = ra_arc_ln(a, fu, q.g_uid[g]);
                        if (ln >= 0) {
                            const int64_t u_ovl = ln < p + 1? ln : p + 1;
                            if (p1 + u_ovl == p + 1) {
                                const int64_t score1 = score + q.g_score0[g] - u_ovl, sk = q.g_score[g];
                                uint32_t pn = q.g_prevn[g];
                                if (!(score1 <= score || score1 < sk || (score1 == sk && pn == 0))) {
                                    if (score1 > sk) q.g_score[g] = (int32_t) score1, pn = 0;
                                    if (pn == RAB_PREV) ov = true;
                                    else q.g_prev[g * RAB_PREV + pn] = j, q.g_prevn[g] = pn + 1;
                                }
                            }
                        }
                    }
                } else past = true;
                if (__ballot(ov)) { over = true; break; }
                if (__ballot(past)) break;
            }
            __threadfence_block();                                                 // the scores this j raised are the next j's input
        }
    }
    if (q.mode != 1 && lane == 0) a.cnt_aln[r] = 0, a.cnt_frg[r] = 0, a.skipped[r] = over? 3 : 0;
    if (over) return;
    int64_t max_score = 0;
    for (uint32_t i = lane; i < nf; i += 64) { const int64_t sc = q.g_score[f0 + i]; if (sc > max_score) max_score = sc; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const int64_t o = __shfl_xor(max_score, d); if (o > max_score) max_score = o; }
    if (max_score < (old >> 1)) return;                                            // :505
    // (the walk is lane 0's; the wave finds where it starts)
    uint32_t n_a = 0, n_fr = 0, x, y;
    if (q.mode == 1) {
        rab_backtrace<true>(q, f0, nf, max_score, n, r, a.cnt_aln[r], a.aln_off[r], a.frg_off[r], 0, x, y);
        return;
    }
    rab_backtrace<false>(q, f0, nf, max_score, n, r, 0, 0, 0, 0, n_a, n_fr);
    n_a = (uint32_t) __shfl((int) n_a, 0), n_fr = (uint32_t) __shfl((int) n_fr, 0);
    if (lane == 0) a.cnt_aln[r] = n_a, a.cnt_frg[r] = n_fr;
    if (q.mode == 2 && n_a) {
        unsigned long long wa = 0, wf = 0;
        if (lane == 0) wa = atomicAdd(&a.pool_used[0], (unsigned long long) n_a), wf = atomicAdd(&a.pool_used[1], (unsigned long long) n_fr);
        wa = (unsigned long long) __shfl((long long) wa, 0), wf = (unsigned long long) __shfl((long long) wf, 0);
        if (wa + n_a > a.pool_cap_a || wf + n_fr > a.pool_cap_f) { if (lane == 0) a.pool_used[2] = 1ULL; return; }
        if (lane == 0) a.pool_a[r] = wa, a.pool_f[r] = wf;
        rab_backtrace<true>(q, f0, nf, max_score, n, r, n_a, wa, wf, wf, x, y);
    }
}

} // namespace oatk
