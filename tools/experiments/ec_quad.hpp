// oatk_amd/csrc/ec_quad.hpp -- the error-block solver with FOUR blocks per wavefront, one 16-lane row each.
//
// ec_wave.hpp gives a block the whole wave; what that costs is instruction issue: ~1900 wave-instructions per block with 1 - 7 live diagonals, most
// of them executed with most lanes idle.  A first-tier block never needs 64 lanes -- its target is ~100 - 190 sixteen-base words, an
// appended k-mer ~30, a wavefront 1 - 7 diagonals -- so here every row of sixteen lanes runs the SAME search (dfs_search + wf_ed_core,
// syncerr.c:144-286, levdist.c:75-310; the statements of ecw_solve_block in the same order) on its own block, with its own carve-up of LDS.  The four
// rows of a wave share one instruction stream: control flow is uniform inside a row and differs between rows, so every loop runs as long as the
// slowest of the four rows needs (the rows wait for each other at loop exits), and a wave's cost per quartet is the maximum over its rows at every
// nesting level instead of the sum.  What used to be scalar state (the frame's fields, the arc record, the wave's d0 / n) is per-row vector state
// here: all sixteen lanes of a row compute or load the same value; ballots are cut down to the row's sixteen bits, broadcasts are row-relative.
//
// The optimum consensus -- compared only when two paths tie (syncerr.c:225-243), written once or twice per block -- lives in an HBM slab per row
// instead of LDS, and the frame arena is 1 KB: 4.8 KB of LDS per block, eight waves = 32 blocks in flight per CU.  A block that outgrows the
// carve-up (frames, path, consensus length) is handed to the next tier (ec_wave.hpp) exactly like before.
#pragma once
#include "ec_wave.hpp"

namespace oatk {

#define ECQ_BATCH 4               // rounds of blocks (one per row) taken from the queue per atomic

struct EcqScratch {
    uint32_t *ts, *cs;            // LDS: target, consensus (packed)
    uint32_t *os;                 // HBM: optimum consensus
    int32_t *ka, *kb;             // LDS: two wavefront buffers
    uint64_t *c_path, *o_path;    // LDS
    uint8_t *frames;              // LDS
    int32_t cap_t, cap_c, cap_w, cap_path, cap_f;
};

// LDS accesses of one wave issue in order, so what a lane wrote is there for another lane's later read; the fences only keep the compiler
// from moving accesses across the hand-over (no instruction comes out of them)
#define ECQ_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <int ROW> __device__ __forceinline__ uint32_t ecq_ballot(bool p, int gsh) { return (uint32_t) (__ballot(p) >> gsh) & (ROW == 32? 0xFFFFFFFFu : 0xFFFFu); }

// ecw_step for one row of ROW lanes: lanes per diagonal = ROW / next_pow2(n) while n <= ROW / 2, one lane per diagonal and ROW diagonals per pass beyond
template <int ROW>
__device__ __forceinline__ int ecq_step(const uint32_t *ts, int32_t tl, const uint32_t *qs, int32_t ql, int32_t bw, EcwWave &wv, int32_t *buf_a, int32_t *buf_b,
                                        int32_t &t_end, int32_t &q_end, const int c, const int gsh)
{
    const int32_t n = wv.n, d0 = wv.d0;
    int32_t *k = wv.k;
    t_end = q_end = -1;
    int lg = 0;
    constexpr int LOG = ROW == 32? 5 : 4;
    if (n <= ROW / 2) lg = n <= 1? LOG : __builtin_clz((uint32_t) (n - 1)) - (32 - LOG);
    const int G = 1 << lg, cc = c & (G - 1), sub = gsh + (c & ~(G - 1));
    const uint32_t gmask = G == 32? 0xFFFFFFFFu : (1u << G) - 1u;
    int found = 0;
    for (int32_t base = 0; base < n && !found; base += ROW) {
        const int32_t j = base + (c >> lg);
        const bool valid = j < n;
        int32_t kk = valid? k[j] : 0;
        const int32_t dd = d0 + j;
        const bool act0 = valid && !(kk >= tl || kk + dd >= ql);
        bool act = act0;
        const int32_t lim = (ql - dd < tl? ql - dd : tl) - 1;
        while (ecq_ballot<ROW>(act, gsh)) {
            const int32_t rem = lim - kk, o = cc << 4;
            int32_t m = 0;
            if (act && o < rem) {
                const uint32_t x = ecw_win16(ts, kk + 1 + o) ^ ecw_win16(qs, kk + dd + 1 + o);
                m = x? __builtin_ctz(x) >> 1 : 16;
                if (m > rem - o) m = rem - o;
            }
            const bool full = act && m == 16;
            const uint32_t gb = (uint32_t) (__ballot(!full) >> sub) & gmask;
            const int first = gb? __builtin_ctz(gb) : 0;
            const int32_t mm = __shfl(m, sub + first);
            if (act) {
                if (gb == 0) kk += G << 4;
                else kk += (first << 4) + mm, act = false;
            }
        }
        const bool reached = act0 && (kk + dd == ql - 1 || kk == tl - 1);
        const uint32_t rmask = ecq_ballot<ROW>(reached && cc == 0, gsh);
        if (rmask) {
            const int fl = __builtin_ctz(rmask);
            const int32_t jf = base + (fl >> lg);
            if (act0 && cc == 0 && j < jf) k[j] = kk;
            t_end = __shfl(kk, gsh + fl);
            q_end = t_end + d0 + jf;
            found = 1;
        } else if (act0 && cc == 0) k[j] = kk;
    }
    ECQ_SYNC();
    if (found) return 1;
    // next wavefront: diagonals d0 - 1 .. d0 + n (levdist.c:183-205)
    int32_t *nk = wv.spare;
    for (int32_t i = c; i < n + 2; i += ROW) {
        const int32_t jj = i - 1;
        int32_t v = INT32_MIN;
        if (jj - 1 >= 0) v = k[jj - 1];
        if (jj >= 0 && jj < n) { const int32_t u = k[jj] + 1; v = u > v? u : v; }
        if (jj + 1 < n) { const int32_t u = k[jj + 1] + 1; v = u > v? u : v; }
        nk[i] = v;
    }
    int32_t st = 0, en = n + 2;
    const int32_t nd0 = d0 - 1;
    if (bw < 0 || n < 2 * bw + 1) {
        if (nd0 < -tl) ++st;
        if (nd0 + n + 1 > ql) --en;
    } else {
        const int32_t lo = -bw > -tl? -bw : -tl, hi = bw > ql? bw : ql;     // the LARGER of bw and ql, as in levdist.c:108
        while (nd0 + st < lo) ++st;
        while (nd0 + en - 1 > hi) --en;
    }
    wv.n = en - st, wv.d0 = nd0 + st;
    wv.k = nk + st;
    wv.spare = nk == buf_a? buf_b : buf_a;
    ECQ_SYNC();
    return 0;
}

// ecw_solve_block for one row.  Returns false when the carve-up is too small for the block.
template <int ROW>
__device__ __forceinline__ bool ecq_solve_block(const EcLive &lv, const EcReads &rd, const EcWork &wk, const EcqScratch &s, double max_edist,
                                                uint32_t &status_out, uint32_t &np_out, const int c, const int gsh)
{
    const int K = rd.K;
    const int32_t tl = wk.l;
    int32_t bw = (int32_t) ceil((double) tl * max_edist);
    if (bw < EC_MIN_ERR_BASE) bw = EC_MIN_ERR_BASE;
    if (tl > s.cap_t || 2 * bw + 8 > s.cap_w) return false;
    EcwArcRegs pre;
    pre.a = make_uint4(0, 0, 0, 0), pre.b = make_uint2(0, 0);
    uint32_t pre_idx = 0xFFFFFFFFu;
    if (wk.ln) pre = ecw_arc_load(lv.arc, wk.lp), pre_idx = wk.lp;
    // target: the read segment, reverse-complemented for a leading block (get_kmer_dna_seq, syncmer.c:1237)
    const uint8_t *hs = rd.hoco_s + ((uint64_t) wk.hs16 << 4);
    for (int32_t wb = 0; (wb << 4) < tl; wb += 4 * ROW) {      // four windows per lane with their loads in flight together
        uint32_t w0[4], w1[4], pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t wi = wb + c + ROW * u;
            const int64_t start = wk.r? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
            pp[u] = ecw_gather16_at(start, wk.r != 0);
            w0[u] = w1[u] = 0;
            if ((wi << 4) < tl) { const uint32_t *q = (const uint32_t *) hs + (pp[u] >> 4); w0[u] = q[0], w1[u] = q[1]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t wi = wb + c + ROW * u;
            const int64_t start = wk.r? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
            if ((wi << 4) < tl) s.ts[wi] = ecw_gather16_fin(w0[u], w1[u], pp[u], start, wk.r != 0);
        }
    }
    int32_t status = EC_FAILURE, n_path = 0, edist = INT32_MAX, s_edist = INT32_MAX;
    int32_t c_len = 0, o_len = 0, np = 0;
    int32_t score = 0, t_end = 0, q_end = 0;
    EcwWave wv;
    wv.k = s.ka, wv.spare = s.kb, wv.n = 1, wv.d0 = 0;
    if (c == 0) s.ka[0] = -1, s.c_path[0] = wk.beg_utg;
    int32_t fsz = 0, top = -1, nfr = 0;
    bool ok = true, vpend = false;                    // (a level with one arc needs no frame: ecw_solve_block)
    uint32_t v_arc = 0;
    int32_t v_depth = 0;
    ECQ_SYNC();

    auto push_frame = [&](uint32_t lp, uint32_t ln, int32_t depth) -> bool {
        const int32_t need = ((int32_t) sizeof(EcwFrame) + 4 * wv.n + 7) & ~7;
        if (fsz + need > s.cap_f) return false;
        EcwFrame *f = (EcwFrame *) (s.frames + fsz);
        if (c == 0) {
            f->arc_i = lp, f->arc_end = lp + ln;
            f->l0 = c_len, f->score = score, f->t_end = t_end, f->q_end = q_end, f->n = wv.n, f->d0 = wv.d0, f->prev_off = top, f->depth = depth;
        }
        int32_t *sv = (int32_t *) (f + 1);
        for (int32_t j = c; j < wv.n; j += ROW) sv[j] = wv.k[j];
        top = fsz;
        fsz += need;
        ++nfr;
        return true;
    };
    if (wk.ln == 1) vpend = true, v_arc = wk.lp, v_depth = 0;
    else if (!push_frame(wk.lp, wk.ln, 0)) return false;

    while ((nfr > 0 || vpend) && ok) {
        ECQ_SYNC();
        uint32_t a;
        int32_t depth;
        if (vpend) {
            vpend = false;
            a = v_arc, depth = v_depth;
        } else {
            EcwFrame *f = (EcwFrame *) (s.frames + top);
            a = f->arc_i;                             // (every lane of the row reads the same words: LDS broadcasts them)
            const uint32_t a_end = f->arc_end;
            if (a == a_end) {                         // level exhausted: return to the nearest level with siblings left
                fsz = top;
                top = f->prev_off;
                --nfr;
                continue;
            }
            ECQ_SYNC();                               // every lane has read arc_i before it moves on
            if (c == 0) f->arc_i = a + 1;
            // restore the state this level was entered with (syncerr.c:277-284)
            depth = f->depth;
            c_len = f->l0, score = f->score, t_end = f->t_end, q_end = f->q_end;
            wv.n = f->n, wv.d0 = f->d0, wv.k = s.ka, wv.spare = s.kb;
            const int32_t *sv = (const int32_t *) (f + 1);
            for (int32_t j = c; j < wv.n; j += ROW) s.ka[j] = sv[j];
        }
        if (pre_idx != a) pre = ecw_arc_load(lv.arc, a);
        const uint64_t w = pre.a.x;
        const int32_t ls = (int32_t) pre.a.y, ext = K - ls;
        const uint32_t w_hs16 = pre.a.z, w_mpos = pre.a.w, w_lp = pre.b.x, w_ln = pre.b.y;
        const int32_t t_end0 = t_end;
        if (depth + 2 > s.cap_path || c_len + ext > s.cap_c) { ok = false; break; }
        if (c == 0) s.c_path[depth + 1] = w;
        int32_t cn = depth + 2;                       // entries in c_path
        // the arc most likely to be tried next: the first one out of w (in flight during the gather and the alignment)
        pre_idx = 0xFFFFFFFFu;
        if (w_ln) pre = ecw_arc_load(lv.arc, w_lp), pre_idx = w_lp;
        {   // append the part of w's k-mer that lies beyond the overlap (syncerr.c:186-190); see ecw_solve_block
            const uint8_t *vs = rd.hoco_s + ((uint64_t) w_hs16 << 4);
            const uint32_t pos = w_mpos >> 1;
            const bool asc = (uint32_t) (w & 1ULL) == (w_mpos & 1u);
            const int32_t w0 = c_len >> 4, w1 = (c_len + ext - 1) >> 4;
            for (int32_t wi = w0 + c; wi <= w1; wi += ROW) {
                const int32_t t0 = (wi << 4) - c_len;
                uint32_t x = asc? ecw_gather16(vs, (int64_t) pos + ls + t0, false) : ecw_gather16(vs, (int64_t) pos + K - 1 - ls - t0, true);
                if (t0 < 0) {
                    const uint32_t keep = (1u << ((uint32_t) (-t0) << 1)) - 1u;
                    x = (s.cs[wi] & keep) | (x & ~keep);
                }
                s.cs[wi] = x;
            }
            c_len += ext;
        }
        ECQ_SYNC();
        // wf_ed_core (levdist.c:265-310)
        for (;;) {
            if (ecq_step<ROW>(s.ts, tl, s.cs, c_len, bw, wv, s.ka, s.kb, t_end, q_end, c, gsh)) break;
            ++score;
            if (score > bw) break;
        }
        t_end += 1, q_end += 1;
        const int32_t ql = c_len;
        const int32_t sc = score + tl - t_end;        // syncerr.c:209
        if (sc <= bw && (wk.end_utg == EC_NONE || wk.end_utg == w)) {
            status = EC_SUCCESS;
            if (sc <= edist) {
                if (t_end > t_end0) s_edist = edist;
                edist = sc;
                if (wk.end_utg == EC_NONE && q_end < ql) --cn;
                if (edist == s_edist) {
                    bool diff = q_end != o_len;
                    if (!diff) {
                        bool d = false;
                        const int32_t nw = (q_end + 15) >> 4;
                        for (int32_t wi = c; wi < nw; wi += ROW) {       // (a lane reads the words of `os` it wrote itself: same stride)
                            uint32_t x = s.cs[wi] ^ s.os[wi];
                            if (wi == nw - 1 && (q_end & 15)) x &= (1u << ((q_end & 15) << 1)) - 1u;
                            d |= x != 0;
                        }
                        diff = ecq_ballot<ROW>(d, gsh) != 0;
                    }
                    if (diff) status = EC_AMBISEQ;
                    if (status == EC_SUCCESS) {
                        bool pd = cn != np;
                        if (!pd) {
                            bool d = false;
                            for (int32_t i = c; i < cn; i += ROW) d |= s.c_path[i] != s.o_path[i];
                            pd = ecq_ballot<ROW>(d, gsh) != 0;
                        }
                        if (pd) status = EC_AMBISNQ;
                    }
                }
                ECQ_SYNC();
                for (int32_t wi = c; wi < ((q_end + 15) >> 4); wi += ROW) s.os[wi] = s.cs[wi];
                o_len = q_end;
                for (int32_t i = c; i < cn; i += ROW) s.o_path[i] = s.c_path[i];
                np = cn;
            } else if (sc < s_edist) {
                s_edist = sc;
            }
        }
        if (score <= bw && ql - K <= tl + bw && ((wk.end_utg != EC_NONE && wk.end_utg != w) || t_end < tl)) {
            if (n_path < EC_MAX_DFS_PATH) {           // the callee would return at once otherwise (syncerr.c:146-148)
                if (w_ln == 1) vpend = true, v_arc = w_lp, v_depth = depth + 1;
                else if (w_ln > 1 && !push_frame(w_lp, w_ln, depth + 1)) { ok = false; break; }
            }
        } else {
            ++n_path;
        }
    }
    ECQ_SYNC();
    status_out = (uint32_t) status, np_out = (uint32_t) np;
    return ok;
}

// 32-bit words of one ROW's carve-up of LDS: ts, cs, two wavefronts, two paths, frames (the optimum consensus is in HBM)
__host__ __device__ inline uint32_t ecq_scratch_words(int32_t cap_t, int32_t cap_c, int32_t cap_w, int32_t cap_path, int32_t cap_f)
{
    return ((ecw_words(cap_t) + ecw_words(cap_c) + 2u * (uint32_t) (cap_w + 2) + 1u) & ~1u) + 4u * (uint32_t) cap_path + (uint32_t) cap_f / 4u;
}

// a.slabs: one HBM slab of ecw_words(cap_c) words per row for the optimum consensus (a.slab_bytes apart)
template <int ROW>
__global__ __launch_bounds__(64) void ec_quad_kernel(EcwArgs a)
{
    extern __shared__ uint32_t ecq_lds[];
    constexpr int NR = 64 / ROW;                       // rows = blocks per wave
    const int lane = threadIdx.x, c = lane & (ROW - 1), gsh = lane & ~(ROW - 1), g = lane / ROW;
    EcqScratch s;
    s.cap_t = a.cap_t, s.cap_c = a.cap_c, s.cap_w = a.cap_w, s.cap_path = a.cap_path, s.cap_f = a.cap_f;
    uint32_t *const p0 = ecq_lds + (size_t) g * ecq_scratch_words(a.cap_t, a.cap_c, a.cap_w, a.cap_path, a.cap_f);
    uint32_t *p = p0;
    s.ts = p, p += ecw_words(a.cap_t);
    s.cs = p, p += ecw_words(a.cap_c);
    s.ka = (int32_t *) p, p += a.cap_w + 2;
    s.kb = (int32_t *) p, p += a.cap_w + 2;
    p += (p - p0) & 1;
    s.c_path = (uint64_t *) p, p += 2 * a.cap_path;
    s.o_path = (uint64_t *) p, p += 2 * a.cap_path;
    s.frames = (uint8_t *) p;
    s.os = (uint32_t *) (a.slabs + ((uint64_t) blockIdx.x * NR + (uint64_t) g) * a.slab_bytes);
    const uint64_t total = a.todo? a.n_todo : a.n_work;
    uint64_t pool_at = 0, pool_end = 0;                // this ROW's chunk of the path pool
    for (;;) {
        unsigned long long t0 = 0;
        if (lane == 0) t0 = atomicAdd(a.next, (unsigned long long) (NR * ECQ_BATCH));
        t0 = ecw_uni64(t0);
        if (t0 >= total) break;
        for (int i = 0; i < ECQ_BATCH; ++i) {
            const uint64_t at = t0 + (uint64_t) (NR * i + g);      // row g takes every NR-th block of the batch
            const bool on = at < total;
            EcWork wk;
            uint64_t wi = 0;
            if (on) {
                wi = a.todo? a.todo[at] : at;
                const uint4 *q = (const uint4 *) (a.work + wi);    // (sixteen lanes, one address: one fetch)
                const uint4 m0 = q[0], m1 = q[1], m2 = q[2];
                wk.beg_utg = (uint64_t) m0.y << 32 | m0.x, wk.end_utg = (uint64_t) m0.w << 32 | m0.z;
                wk.read = m1.x, wk.beg_pos = m1.y, wk.l = (int32_t) m1.z, wk.r = (int32_t) m1.w;
                wk.hs16 = m2.x, wk.lp = m2.y, wk.ln = m2.z, wk.pad = 0;
            }
            if (on && wk.l <= a.skip_l) {
                EcBlockOut o;
                o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0;
                if (wk.l < EC_MIN_ERR_SEQ_LEN) {
                    o.short_block = 1;                 // syncerr.c:502-504
                } else {
                    uint32_t st = 0, np = 0;
                    if (!ecq_solve_block<ROW>(a.lv, a.rd, wk, s, a.max_edist, st, np, c, gsh)) {
                        o.flags = 1;
                        if (c == 0) a.todo_out[atomicAdd(a.todo_cnt, 1ULL)] = (uint32_t) wi;
                    } else {
                        o.status = st, o.np = np;
                        if (st == EC_SUCCESS && np) {
                            if (pool_at + np > pool_end) {
                                const unsigned long long want = np > ECW_POOL_CHUNK? np : ECW_POOL_CHUNK;
                                unsigned long long off = 0;
                                if (c == 0) off = atomicAdd(a.pool_cursor, want);
                                pool_at = (uint64_t) __shfl((long long) off, gsh), pool_end = pool_at + want;
                            }
                            o.path_off = pool_at;
                            if (pool_at + np <= a.pool_cap) for (uint32_t j = c; j < np; j += ROW) a.path_pool[pool_at + j] = s.o_path[j];
                            pool_at += np;
                        }
                    }
                }
                if (c == 0) a.out[wi] = o;
            }
            ECQ_SYNC();
        }
    }
}

}  // namespace oatk
