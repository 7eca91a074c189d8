// oatk_amd/csrc/ec_wave.hpp -- the error-block solver, one WAVEFRONT per block.
//
// Same search as dfs_search + wf_ed_core (syncerr.c:144-286, levdist.c:75-310), organised for a 64-wide wave:
//   * strings are 2-bit packed, sixteen bases per 32-bit word, field s of word i = base 16 i + s; the read segment (ts),
//     the growing consensus (cs) and the optimum consensus (os) live in LDS, gathered sixteen bases at a time from the
//     resident hoco strings;
//   * a Landau-Vishkin step runs with one lane GROUP per diagonal: the 64 lanes are split evenly over the (power-of-two
//     rounded) diagonals, every lane XORs one 16-base window of its diagonal, a ballot finds the first mismatch of each
//     group -- the common case (a correct path, 1-7 diagonals, hundreds of matching bases) takes one or two rounds;
//   * the diagonals of a wavefront are consecutive, so a wavefront is (d0, n, k[n]); DFS frames keep it in an LDS arena;
//   * the work is latency-bound (a block is a chain of dependent random HBM reads), so the chain is kept short: a block
//     descriptor and a live-arc record carry everything the search needs next (ec.hpp: EcWork, EcLiveArc), and the first
//     arc of the vertex being entered is fetched while its k-mer is gathered and aligned;
//   * waves pull blocks eight at a time from a shared counter (block costs vary by orders of magnitude) and take space
//     for optimum paths from the pool in chunks -- one contended atomic per many blocks.
// Blocks that outgrow a tier's LDS carve-up are flagged and re-run by a larger tier; the last tier (BIG) keeps every
// array in an HBM slab and has no limit but the slab.
//
// The order of evaluation that the results depend on is kept exactly: a step ends at the LOWEST diagonal that reaches an
// end, with only the diagonals below it extended (levdist.c:166-180); ties between optimum paths compare the consensus
// and then the path (syncerr.c:216-243).
#pragma once
#include <type_traits>
#include "ec.hpp"

namespace oatk {

__device__ __forceinline__ uint32_t ecw_rev_in_bytes(uint32_t x)      // reverse the four 2-bit fields of every byte
{
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    return ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
}
__device__ __forceinline__ uint32_t ecw_rev16(uint32_t x)             // reverse the order of the sixteen 2-bit fields
{
    x = __builtin_bitreverse32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
// field s = base p + s of a hoco string (MSB-first bytes, syncmer.c:268-283); hs is 4-byte aligned, 8 bytes of slack behind
__device__ __forceinline__ uint32_t ecw_src16(const uint8_t *hs, uint32_t p)
{
    const uint32_t *w = (const uint32_t *) hs + (p >> 4);
    const uint64_t v = (uint64_t) ecw_rev_in_bytes(w[1]) << 32 | ecw_rev_in_bytes(w[0]);
    return (uint32_t) (v >> ((p & 15u) << 1));
}
// ascending: field s = base(start + s); descending: field s = 3 ^ base(start - s).  Fields that fall before base 0 are garbage.
__device__ __forceinline__ uint32_t ecw_gather16(const uint8_t *hs, int64_t start, bool desc)
{
    if (!desc) return start >= 0? ecw_src16(hs, (uint32_t) start) : ecw_src16(hs, 0) << ((uint32_t) (-start) << 1);
    const int64_t lo = start - 15;
    return ~(lo >= 0? ecw_rev16(ecw_src16(hs, (uint32_t) lo)) : ecw_rev16(ecw_src16(hs, 0)) >> ((uint32_t) (-lo) << 1));
}
// the same in two halves, so that a loop can have the loads of several windows in flight before it finishes the first: where to read ...
__device__ __forceinline__ uint32_t ecw_gather16_at(int64_t start, bool desc)
{
    const int64_t p = desc? start - 15 : start;
    return p >= 0? (uint32_t) p : 0u;
}
// ... and what to make of the two words w[p >> 4], w[(p >> 4) + 1]
__device__ __forceinline__ uint32_t ecw_gather16_fin(uint32_t w0, uint32_t w1, uint32_t p, int64_t start, bool desc)
{
    const uint64_t v = (uint64_t) ecw_rev_in_bytes(w1) << 32 | ecw_rev_in_bytes(w0);
    const uint32_t x = (uint32_t) (v >> ((p & 15u) << 1));
    if (!desc) return start >= 0? x : x << ((uint32_t) (-start) << 1);
    const int64_t lo = start - 15;
    return ~(lo >= 0? ecw_rev16(x) : ecw_rev16(x) >> ((uint32_t) (-lo) << 1));
}
// sixteen fields starting at base p of a packed array (one pad word behind the data)
__device__ __forceinline__ uint32_t ecw_win16(const uint32_t *W, int32_t p)
{
    const int32_t i = p >> 4;
    return (uint32_t) (((uint64_t) W[i + 1] << 32 | W[i]) >> ((p & 15) << 1));
}
// The solver's waves work on their own: what one lane wrote is read by another lane of the SAME wave, never by another wave.  A wave's memory
// instructions are issued in order, so all it takes is that the compiler does not move accesses across this point (and, for the HBM slabs, that
// the stores have left).  No s_barrier: a workgroup may hold several waves (the hardware places at most sixteen workgroups on a CU, whatever their
// size -- with one wave each, sixteen waves per CU was all this latency-bound kernel ever got), and they are at different blocks.
__device__ __forceinline__ void ecw_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int32_t ecw_uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t ecw_uniu(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) v); }
__device__ __forceinline__ uint32_t ecw_lane(uint32_t v, int l) { return (uint32_t) __builtin_amdgcn_readlane((int32_t) v, l); }      // l uniform
__device__ __forceinline__ uint64_t ecw_uni64(uint64_t v) { return (uint64_t) ecw_uniu((uint32_t) (v >> 32)) << 32 | ecw_uniu((uint32_t) v); }

// sharded reads: the k-mer of vertex ids[i] as a stand-alone hoco string (base 0 = first base as read, same byte layout),
// `stride` bytes apart, with the strand it was read on -- what another shard needs to use this vertex (one wave per vertex)
__global__ __launch_bounds__(64) void ecw_export_kmer_kernel(uint64_t n, const uint32_t *ids, const uint64_t *vtx_hs_off, const uint32_t *vtx_mpos,
                                                             const uint8_t *hoco_s, int K, uint32_t stride, uint8_t *out, uint8_t *rev)
{
    const uint64_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t v = ids[i];
    const uint64_t src = vtx_hs_off[v];
    uint32_t *o = (uint32_t *) (out + i * stride);
    const uint32_t mp = vtx_mpos[v];
    for (uint32_t wi = threadIdx.x; wi < stride / 4; wi += 64)
        o[wi] = src != EC_NO_SRC && (int) (wi << 4) < K? ecw_rev_in_bytes(ecw_src16(hoco_s + src, (mp >> 1) + (wi << 4))) : 0u;
    if (threadIdx.x == 0) rev[i] = src != EC_NO_SRC? (uint8_t) (mp & 1u) : (uint8_t) 0xFF;
}
__global__ void ecw_import_kmer_kernel(uint64_t n, const uint32_t *ids, const uint8_t *rev, uint64_t base_off, uint32_t stride, uint64_t *vtx_hs_off, uint32_t *vtx_mpos)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vtx_hs_off[ids[i]] = base_off + i * stride;
    vtx_mpos[ids[i]] = rev[i] & 1u;                            // position 0 on its own string
}

#ifdef ECW_PROF
__device__ unsigned long long ecw_dyn[16];          // dynamic counts of the loops of ecw_step (development builds only)
#define ECW_D(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&ecw_dyn[i], 1ULL); } while (0)
#define ECW_T(i) do { const unsigned long long _t = __builtin_readcyclecounter(); s.prof[i] += _t - s.t_last; s.t_last = _t; } while (0)
#define ECW_C(i, v) do { s.prof[i] += (v); } while (0)
#else
#define ECW_T(i) do {} while (0)
#define ECW_C(i, v) do {} while (0)
#define ECW_D(i) do {} while (0)
#endif
#ifdef ECW_CENSUS                                    // development builds: how many of the launched waves were resident early enough to find work
__device__ unsigned long long ecw_census[2];
#endif
// block placement: the usual path of the search falls through (a taken branch is what costs here)
#define ECW_LIKELY(x) __builtin_expect(!!(x), 1)
#define ECW_RARE(x) __builtin_expect(!!(x), 0)

struct EcwScratch {
    uint32_t *ts, *cs, *os;       // target, consensus, optimum consensus (packed)
    int32_t *ka, *kb;             // two wavefront buffers, cap_w + 2 entries each
    uint64_t *c_path, *o_path;    // current and optimum path
    uint8_t *frames;              // DFS frame arena
    int32_t cap_t, cap_c, cap_w, cap_path, cap_f;
    int32_t arc_budget;           // arcs after which a block is given up here (0: never): a search inside a repeat tries tens of thousands (round 6)
#ifdef ECW_PROF
    mutable unsigned long long prof[32], t_last;
#endif
};

struct EcwFrame {                 // state at the entry of one DFS level (syncerr.c:158-171), followed by k[n]
    uint32_t arc_i, arc_end;
    int32_t l0, score, t_end, q_end, n, d0, prev_off, depth;
};

struct EcwWave {                  // the working wavefront: diagonals d0 .. d0 + n - 1, furthest target index per diagonal in k[]
    int32_t *k, *spare;
    int32_t n, d0;
};

// one wavefront step (levdist.c:156-224, extension mode, no traceback); returns 1 when an end was reached
__device__ int ecw_step(const uint32_t *ts, int32_t tl, const uint32_t *qs, int32_t ql, int32_t bw, EcwWave &wv, int32_t *buf_a, int32_t *buf_b,
                        int32_t &t_end, int32_t &q_end)
{
    const int lane = threadIdx.x & 63;
    const int32_t n = wv.n, d0 = wv.d0;
    int32_t *k = wv.k;
    t_end = q_end = -1;
    ECW_D(0);
    if (n == 1) ECW_D(6); else if (n <= 2) ECW_D(7); else if (n <= 4) ECW_D(8); else if (n <= 8) ECW_D(9); else if (n <= 16) ECW_D(10); else ECW_D(11);
    int lg = 0;                                        // lanes per diagonal = 1 << lg
    if (n <= 32) lg = n <= 1? 6 : __builtin_clz((uint32_t) (n - 1)) - 26;      // 64 / next_pow2(n)
    const int G = 1 << lg, c = lane & (G - 1), gbase = lane & ~(G - 1);
    const uint64_t gmask = G == 64? ~0ULL : (1ULL << G) - 1ULL;
    // More than 64 diagonals (a long block late in a failing path: up to 2 bw + 1 = several hundred): one lane per diagonal, and FOUR rounds of 64 diagonals
    // side by side -- the rounds are independent chains of LDS round trips (wavefront entry, two windows of each string, the comparison), and one after the
    // other they left the wave waiting for LDS most of the time: a long block of the config-1 surrogate took 2.5 s, forty times a CPU core (r04g).  What
    // the step returns depends on the LOWEST diagonal that reaches an end and on the diagonals below it alone, so the rounds are finished in order.
    if (n > 64) {
        for (int32_t base = 0; base < n; base += 256) {
            int32_t kk[4], dd[4], lim[4];
            bool a0[4], act[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t j = base + 64 * u + lane;
                const bool valid = j < n;
                kk[u] = valid? k[j] : 0, dd[u] = valid? d0 + j : 0;
                a0[u] = valid && !(kk[u] >= tl || kk[u] + dd[u] >= ql);
                act[u] = a0[u];
                lim[u] = (ql - dd[u] < tl? ql - dd[u] : tl) - 1;
            }
            ECW_D(1);
            while (__ballot(act[0] | act[1] | act[2] | act[3])) {
                ECW_D(2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t r = lim[u] - kk[u];
                    const uint32_t x = ecw_win16(ts, kk[u] + 1) ^ ecw_win16(qs, kk[u] + dd[u] + 1);
                    int32_t m = x? __builtin_ctz(x) >> 1 : 16;
                    m = m < r? m : r;
                    m = act[u] && r > 0? m : 0;
                    kk[u] += m;
                    act[u] = act[u] && m == 16;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t j = base + 64 * u + lane;
                const bool reached = a0[u] && (kk[u] + dd[u] == ql - 1 || kk[u] == tl - 1);
                const uint64_t rmask = __ballot(reached);
                if (rmask) {
                    const int fl = __builtin_ctzll(rmask);
                    if (a0[u] && lane < fl) k[j] = kk[u];
                    t_end = (int32_t) ecw_lane((uint32_t) kk[u], fl);
                    q_end = t_end + d0 + base + 64 * u + fl;
                    ECW_D(3);
                    ecw_sync();
                    return 1;
                }
                if (a0[u]) k[j] = kk[u];
            }
        }
    } else
    for (int32_t base = 0; base < n; base += 64) {
        const int32_t j = base + (lane >> lg);
        ECW_D(1);
        const bool valid = j < n;
        int32_t kk = valid? k[j] : 0;
        const int32_t dd = valid? d0 + j : 0;          // (0: lanes without a diagonal read, harmlessly, inside the strings)
        const bool act0 = valid && !(kk >= tl || kk + dd >= ql);
        bool act = act0;
        const int32_t lim = (ql - dd < tl? ql - dd : tl) - 1;
        // (straight-line turns: a taken branch costs this kernel more than the handful of instructions it skips -- 64 extra taken branches per
        //  arc measured +22 %, 64 extra scalar or vector instructions +1.5 % / +1 % -- so lanes that have nothing to compare compare anyway)
        if (__ballot(act)) {
            const int32_t o = c << 4;
            for (;;) {
                ECW_D(2);
                const int32_t r = lim - kk - o;                 // bases left from this lane's window on
                const int32_t oo = r > 0? o : 0;                // (a window behind the end is not read: the strings may lie in HBM, at the end of a buffer)
                const uint32_t x = ecw_win16(ts, kk + 1 + oo) ^ ecw_win16(qs, kk + dd + 1 + oo);
                int32_t m = x? __builtin_ctz(x) >> 1 : 16;
                m = m < r? m : r;
                m = act && r > 0? m : 0;
                const uint64_t gb = (__ballot(m != 16) >> gbase) & gmask;      // lanes of the group that did not match all sixteen
                const int first = gb? __builtin_ctzll(gb) : 0;
                const int32_t mm = __shfl(m, gbase + first);
                const int32_t adv = gb? (first << 4) + mm : G << 4;
                kk += act? adv : 0;
                act = act && gb == 0;
                if (!__ballot(act)) break;
            }
        }
        const bool reached = act0 && (kk + dd == ql - 1 || kk == tl - 1);
        const uint64_t rmask = __ballot(reached && c == 0);
        if (rmask) {
            const int fl = __builtin_ctzll(rmask);
            const int32_t jf = base + (fl >> lg);
            if (act0 && c == 0 && j < jf) k[j] = kk;
            t_end = (int32_t) ecw_lane((uint32_t) kk, fl);
            q_end = t_end + d0 + jf;
            ECW_D(3);
            ecw_sync();
            return 1;
        }
        if (act0 && c == 0) k[j] = kk;
    }
    ecw_sync();
    // next wavefront: diagonals d0 - 1 .. d0 + n (levdist.c:183-205)
    int32_t *__restrict__ nk = wv.spare;
    ECW_D(4);
#pragma unroll 4
    for (int32_t ib = 0; ib < n + 2; ib += 64) {      // (a uniform trip count -- one turn up to 62 diagonals -- and no branches inside: reads with clamped indices, selects)
        const int32_t i = ib + lane, jj = i - 1;
        const int32_t ja = jj - 1 < 0? 0 : (jj - 1 < n? jj - 1 : n - 1), jb = jj < 0? 0 : (jj < n? jj : n - 1), jc = jj + 1 < n? jj + 1 : n - 1;
        const int32_t ka = k[ja], kb = k[jb], kc = k[jc];
        int32_t v = jj - 1 >= 0? ka : INT32_MIN;
        const int32_t ub = kb + 1, uc = kc + 1;
        v = jj >= 0 && jj < n && ub > v? ub : v;
        v = jj + 1 < n && uc > v? uc : v;
        if (i < n + 2) nk[i] = v;
    }
    int32_t st = 0, en = n + 2;
    const int32_t nd0 = d0 - 1;
    if (ECW_LIKELY(bw < 0 || n < 2 * bw + 1)) {
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
    ecw_sync();
    return 0;
}

// wf_ed_core on its own (levdist.c:265-312): one wave per job, every array in HBM; the wavefront survives from one query length to the next
// exactly as the DFS keeps it from one appended k-mer to the next (include/oatk_hip_ec.h: oatk_hip_debug_wf_ed)
__global__ __launch_bounds__(64) void ecw_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                       const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *kbuf, const uint64_t *k_off,
                                                       int32_t *out3)
{
    const uint64_t j = blockIdx.x;
    const uint32_t *ts = tw + tw_off[j], *qs = qw + qw_off[j];
    int32_t *ka = kbuf + k_off[j], *kb = ka + (k_off[j + 1] - k_off[j]) / 2;
    const int32_t tlen = tl[j], band = bw[j];
    EcwWave wv;
    wv.k = ka, wv.spare = kb, wv.n = 1, wv.d0 = 0;              // the caller's initial state: diagonal 0, nothing matched, score 0 (syncerr.c:465-482)
    if (threadIdx.x == 0) ka[0] = -1;
    ecw_sync();
    int32_t score = 0, t_end = -1, q_end = -1;
    for (uint64_t s = step_off[j]; s < step_off[j + 1]; ++s) {
        const int32_t ql = step_ql[s];
        for (;;) {
            if (ecw_step(ts, tlen, qs, ql, band, wv, ka, kb, t_end, q_end)) break;
            ++score;
            if (band >= 0 && score > band) break;
        }
        if (threadIdx.x == 0) out3[3 * s] = score, out3[3 * s + 1] = t_end + 1, out3[3 * s + 2] = q_end + 1;
    }
}

#ifdef OATK_EXPERIMENTS            // (tools/experiments/: built into a library of its own, not into liboatk_hip.so)
// ---- Myers' bit-vector edit distance (north_star names it; SURVEY.md 7-5: "benchmark both") ----
// Measured beside the wavefront routine, not used by the correction (the search RESUMES an alignment after every appended k-mer, saves
// it at a branch and restores it, and its results are defined on wavefronts; a column-wise bit-vector DP can do neither cheaply).  One LANE per pair --
// "batching reads per wavefront" -- with the bit-vectors over the TARGET, sixty-four pairs per wave, the vectors of a wave interleaved in an HBM slab
// (word w of lane l at w * 64 + l: every access of the wave is one coalesced 512-byte row).  Global alignment (the first row counts up: carry-in +1
// per column), Hyyro's block formulation with the horizontal delta handed from word to word.  The result is the closed form wf_ed equals
// (the full-matrix DP of the test suite): the minimum over the last row, then the last column upwards, first minimum wins = smallest diagonal;
// beyond the band (bw >= 0 and minimum > bw): score bw + 1 with both ends 0, as wf_ed_core leaves them.
__global__ __launch_bounds__(64) void myers_ed_kernel(uint64_t n_jobs, const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                      const int32_t *ql_, const int32_t *bw, uint64_t *slab, uint64_t slab_words, int32_t *out3)
{
    const uint64_t j = (uint64_t) blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t *ts = tw + tw_off[j], *qs = qw + qw_off[j];
    const int32_t T = tl[j], Q = ql_[j], nw = (T + 63) >> 6;
    uint64_t *S = slab + (uint64_t) blockIdx.x * slab_words;        // per wave: Peq[4][nw], Pv[nw], Mv[nw], all interleaved over the lanes
    auto at = [&](int32_t plane, int32_t w) -> uint64_t & { return S[((uint64_t) plane * (uint64_t) (slab_words / (6 * 64)) + (uint64_t) w) * 64 + lane]; };
    for (int32_t w = 0; w < nw; ++w) {
        uint64_t eq[4] = {0, 0, 0, 0};
        for (int32_t b = 0; b < 64 && 64 * w + b < T; ++b) {
            const int32_t p = 64 * w + b;
            eq[(ts[p >> 4] >> ((p & 15) << 1)) & 3u] |= 1ULL << b;
        }
        for (int c = 0; c < 4; ++c) at(c, w) = eq[c];
        at(4, w) = ~0ULL, at(5, w) = 0ULL;                             // Pv = all ones (the first column counts up), Mv = 0
    }
    const int32_t lastw = nw - 1;
    const uint64_t lastbit = 1ULL << ((T - 1) & 63);
    int32_t score = T, best = T, bi = T, bj = 0;                        // last row, j = 0: D[T][0] = T
    for (int32_t col = 0; col < Q; ++col) {
        const uint32_t c = (qs[col >> 4] >> ((col & 15) << 1)) & 3u;
        int hin = 1;
        for (int32_t w = 0; w < nw; ++w) {
            uint64_t Eq = at((int32_t) c, w), Pv = at(4, w), Mv = at(5, w);
            const uint64_t Xv = Eq | Mv;
            if (hin < 0) Eq |= 1ULL;
            const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
            const uint64_t top = w == lastw? lastbit : 1ULL << 63;
            const int hout = (Ph & top? 1 : 0) - (Mh & top? 1 : 0);
            Ph <<= 1, Mh <<= 1;
            if (hin < 0) Mh |= 1ULL; else if (hin > 0) Ph |= 1ULL;
            at(4, w) = Mh | ~(Xv | Ph);
            at(5, w) = Ph & Xv;
            hin = hout;
        }
        score += hin;
        if (score < best) best = score, bi = T, bj = col + 1;
    }
    // the last column upwards: D[i][Q] = D[T][Q] - sum of the vertical deltas of rows i + 1 .. T
    int32_t v = score;
    for (int32_t i = T - 1; i >= 0; --i) {
        const uint64_t bit = 1ULL << (i & 63);
        v -= (at(4, i >> 6) & bit? 1 : 0) - (at(5, i >> 6) & bit? 1 : 0);
        if (v < best) best = v, bi = i, bj = Q;
    }
    const int32_t band = bw[j];
    if (band >= 0 && best > band) best = band + 1, bi = 0, bj = 0;
    out3[3 * j] = best, out3[3 * j + 1] = bi, out3[3 * j + 2] = bj;
}
#endif


// a live-arc record in flight: issued as two loads, made uniform only where it is used
struct EcwArcRegs {
    uint4 a;                      // w, ls, hs16, mpos
    uint2 b;                      // lp, ln
};
__device__ __forceinline__ EcwArcRegs ecw_arc_load(const EcLiveArc *arc, uint32_t i)
{
    EcwArcRegs r;
    const uint4 *q = (const uint4 *) (arc + i);
    r.a = q[0];
    r.b = *(const uint2 *) (q + 1);
    return r;
}

// Solve one block with the whole wave.  Returns false when the scratch is too small (the block is then re-run by a larger tier).
__device__ bool ecw_solve_block(const EcLive &lv, const EcReads &rd, const EcWork &wk, const EcwScratch &s, double max_edist,
                                uint32_t &status_out, uint32_t &np_out, uint32_t &tried_out, uint32_t &n_path_out, uint32_t &wf_steps_out, uint32_t &wf_diag_out)
{
    const int lane = threadIdx.x & 63;
    const int K = rd.K;
    const int32_t tl = wk.l;
    int32_t bw = (int32_t) ceil((double) tl * max_edist);
    if (bw < EC_MIN_ERR_BASE) bw = EC_MIN_ERR_BASE;
    ECW_C(16 + (31 - __builtin_clz((uint32_t) tl | 1u)), 1);                  // 16..: histogram of log2(tl)
    if (ECW_RARE(tl > s.cap_t || 2 * bw + 8 > s.cap_w)) { ECW_C(11, 1); return false; }
    // the first arc out of the source is fetched while the target is gathered
    EcwArcRegs pre;
    pre.a = make_uint4(0, 0, 0, 0), pre.b = make_uint2(0, 0);
    uint32_t pre_idx = 0xFFFFFFFFu;
    if (wk.ln) pre = ecw_arc_load(lv.arc, wk.lp), pre_idx = wk.lp;
    // target: the read segment, reverse-complemented for a leading block (get_kmer_dna_seq, syncmer.c:1237)
    const uint8_t *hs = rd.hoco_s + ((uint64_t) wk.hs16 << 4);
    // (a wave's 64 windows are 1024 bases and two blocks in three are shorter than that: the common case is one window per lane, and the
    //  arithmetic of windows that lie behind the target -- a uniform condition -- is skipped, not just their loads)
    auto gather_target = [&](auto nwin, auto rev, int32_t wb) __attribute__((always_inline)) {
        constexpr int NW = decltype(nwin)::value;
        constexpr bool R = decltype(rev)::value;
        uint32_t w0[NW], w1[NW], pp[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int32_t wi = wb + lane + 64 * u;
            const int64_t start = R? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
            pp[u] = ecw_gather16_at(start, R);
            w0[u] = w1[u] = 0;
            if ((wi << 4) < tl) { const uint32_t *q = (const uint32_t *) hs + (pp[u] >> 4); w0[u] = q[0], w1[u] = q[1]; }
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int32_t wi = wb + lane + 64 * u;
            const int64_t start = R? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
            if ((wi << 4) < tl) s.ts[wi] = ecw_gather16_fin(w0[u], w1[u], pp[u], start, R);
        }
    };
    auto gather_all = [&](auto rev) __attribute__((always_inline)) {
        if (tl <= 1024) gather_target(std::integral_constant<int, 1>(), rev, 0);
        else if (tl <= 2048) gather_target(std::integral_constant<int, 2>(), rev, 0);
        else for (int32_t wb = 0; (wb << 4) < tl; wb += 256) gather_target(std::integral_constant<int, 4>(), rev, wb);   // four windows per lane with their loads in flight together
    };
    if (wk.r) gather_all(std::true_type()); else gather_all(std::false_type());
    ECW_T(0);                                          // 0: target gather
    int32_t status = EC_FAILURE, n_path = 0, edist = INT32_MAX, s_edist = INT32_MAX;
    int32_t c_len = 0, o_len = 0, np = 0;
    uint32_t tried = 0, wf_steps = 0;
    uint64_t wf_diag = 0;
    int32_t score = 0, t_end = 0, q_end = 0;
    EcwWave wv;
    wv.k = s.ka, wv.spare = s.kb, wv.n = 1, wv.d0 = 0;
    if (lane == 0) s.ka[0] = -1, s.c_path[0] = wk.beg_utg;
    int32_t fsz = 0, top = -1, nfr = 0;
    // A level with ONE arc needs no frame: nothing has to be restored for a sibling, and when its only child returns the level is exhausted, so the
    // return goes straight to the nearest level that does have siblings.  The search then simply carries on with the state it is in (vpend: the arc
    // to take next, at depth v_depth).  After the marking the graph of HiFi reads is a line almost everywhere, so most blocks never build a frame.
    bool vpend = false;
    uint32_t v_arc = 0;
    int32_t v_depth = 0;
    ecw_sync();

    auto push_frame = [&](uint32_t lp, uint32_t ln, int32_t depth) -> bool {
        const int32_t need = ((int32_t) sizeof(EcwFrame) + 4 * wv.n + 7) & ~7;
        if (fsz + need > s.cap_f) return false;
        EcwFrame *f = (EcwFrame *) (s.frames + fsz);
        if (lane == 0) {
            f->arc_i = lp, f->arc_end = lp + ln;
            f->l0 = c_len, f->score = score, f->t_end = t_end, f->q_end = q_end, f->n = wv.n, f->d0 = wv.d0, f->prev_off = top, f->depth = depth;
        }
        int32_t *sv = (int32_t *) (f + 1);
        for (int32_t j = lane; j < wv.n; j += 64) sv[j] = wv.k[j];
        top = fsz;
        fsz += need;
        ++nfr;
        return true;
    };
    if (ECW_LIKELY(wk.ln == 1)) vpend = true, v_arc = wk.lp, v_depth = 0;
    else if (!push_frame(wk.lp, wk.ln, 0)) { ECW_C(12, 1); return false; }

    while (nfr > 0 || vpend) {
        ecw_sync();
        uint32_t a;
        int32_t depth;
        if (ECW_LIKELY(vpend)) {                      // carry on where the search stands: nothing to restore
            vpend = false;
            a = v_arc, depth = v_depth;
        } else {
            EcwFrame *f = (EcwFrame *) (s.frames + top);
            a = ecw_uniu(f->arc_i);
            const uint32_t a_end = ecw_uniu(f->arc_end);
            if (ECW_RARE(a == a_end)) {                         // level exhausted: return to the nearest level with siblings left
                fsz = top;
                top = ecw_uni(f->prev_off);
                --nfr;
                ECW_T(1);                              // 1: pops
                continue;
            }
            if (lane == 0) f->arc_i = a + 1;
            // restore the state this level was entered with (syncerr.c:277-284)
            depth = ecw_uni(f->depth);
            c_len = ecw_uni(f->l0), score = ecw_uni(f->score), t_end = ecw_uni(f->t_end), q_end = ecw_uni(f->q_end);
            wv.n = ecw_uni(f->n), wv.d0 = ecw_uni(f->d0), wv.k = s.ka, wv.spare = s.kb;
            const int32_t *sv = (const int32_t *) (f + 1);
            for (int32_t j = lane; j < wv.n; j += 64) s.ka[j] = sv[j];
        }
        ECW_C(8, 1);                                   // 8: arcs tried
        ++tried;
        if (ECW_RARE(s.arc_budget > 0 && (tried > (uint32_t) s.arc_budget || wf_steps > 2u * (uint32_t) s.arc_budget))) return false;      // a search inside a repeat (arcs, or twice as many wavefront steps): it starts again where it is given a workgroup
        if (ECW_RARE(pre_idx != a)) pre = ecw_arc_load(lv.arc, a);
        const uint64_t w = ecw_uniu(pre.a.x);
        const int32_t ls = (int32_t) ecw_uniu(pre.a.y), ext = K - ls;
        const uint32_t w_hs16 = ecw_uniu(pre.a.z), w_mpos = ecw_uniu(pre.a.w), w_lp = ecw_uniu(pre.b.x), w_ln = ecw_uniu(pre.b.y);
        const int32_t t_end0 = t_end;
        if (ECW_RARE(depth + 2 > s.cap_path || c_len + ext > s.cap_c)) { ECW_C(depth + 2 > s.cap_path? 13 : 14, 1); return false; }
        if (lane == 0) s.c_path[depth + 1] = w;
        int32_t cn = depth + 2;                       // entries in c_path
        // the arc most likely to be tried next: the first one out of w (in flight during the gather and the alignment)
        pre_idx = 0xFFFFFFFFu;
        if (ECW_LIKELY(w_ln)) pre = ecw_arc_load(lv.arc, w_lp), pre_idx = w_lp;
        ECW_T(3);                                      // 3: restore + arc fetch
        {   // append the part of w's k-mer that lies beyond the overlap (syncerr.c:186-190).  With F the vertex's forward
            // string, base t of the extension is F[ls + t] for a forward w and comp(F[K - ls - 1 - t]) for a reverse one; F itself is
            // the first occurrence's k-mer, reverse-complemented when that occurrence is reverse: two cases remain.
            const uint8_t *vs = rd.hoco_s + ((uint64_t) w_hs16 << 4);
            const uint32_t pos = w_mpos >> 1;
            const bool asc = (uint32_t) (w & 1ULL) == (w_mpos & 1u);
            const int32_t w0 = c_len >> 4, w1 = (c_len + ext - 1) >> 4;
            for (int32_t wb = w0; wb <= w1; wb += 64) {             // (uniform trip count: one turn for an extension of up to 1000 bases)
                const int32_t wi = wb + lane;
                if (wi > w1) continue;
                ECW_D(5);
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
        ecw_sync();
        ECW_T(4);                                      // 4: consensus append
        // A vertex on an unbranched stretch that cannot be the end of the path needs no alignment of its own.  wf_ed_core RESUMES: run on the
        // consensus up to w and then on the consensus up to w's successor, it leaves the wavefront (and score, ends) that a single run on the longer
        // consensus leaves -- extending a diagonal in two goes or in one is the same run of matches, and a step that ends early stores nothing
        // for the diagonals it did not finish.  What the level would have decided besides is reproduced here: no success is possible (w is not the
        // end vertex), the search goes on iff the score is in the band -- if it is not, the next level finds that out, counts the one dead end and
        // returns just the same -- and the length test is plain arithmetic.  Two things do depend on the shorter query and keep the level: a
        // consensus shorter than the band (the next wavefront is trimmed by the query's length, levdist.c:183-205), and `t_end0` of the next level
        // once an optimum exists (syncerr.c:211).  Most blocks are source - two or three such vertices - sink: one alignment instead of three or four.
        if (edist == INT32_MAX && wk.end_utg != EC_NONE && wk.end_utg != w && w_ln == 1 && n_path < EC_MAX_DFS_PATH && c_len - K <= tl + bw && c_len >= bw + 3) {
            vpend = true, v_arc = w_lp, v_depth = depth + 1;
            ECW_C(15, 1);                              // 15: levels without an alignment
            continue;
        }
        // wf_ed_core (levdist.c:265-310)
        for (;;) {
            ++wf_steps, wf_diag += (uint64_t) wv.n;
            if (ecw_step(s.ts, tl, s.cs, c_len, bw, wv, s.ka, s.kb, t_end, q_end)) break;
            ++score;
            ECW_C(9, 1);                               // 9: wavefront steps beyond the first
            if (ECW_RARE(score > bw)) break;
        }
        ECW_T(5);                                      // 5: wavefront steps
        t_end += 1, q_end += 1;
        const int32_t ql = c_len;
        const int32_t sc = score + tl - t_end;        // syncerr.c:209
        bool new_opt = false;
        if (sc <= bw && (wk.end_utg == EC_NONE || wk.end_utg == w)) {
            status = EC_SUCCESS;
            if (sc <= edist) {
                if (t_end > t_end0) s_edist = edist;
                edist = sc;
                if (wk.end_utg == EC_NONE && q_end < ql) --cn;
                if (ECW_RARE(edist == s_edist)) {
                    bool diff = q_end != o_len;
                    if (!diff) {
                        bool d = false;
                        const int32_t nw = (q_end + 15) >> 4;
                        for (int32_t wi = lane; wi < nw; wi += 64) {
                            uint32_t x = s.cs[wi] ^ s.os[wi];
                            if (wi == nw - 1 && (q_end & 15)) x &= (1u << ((q_end & 15) << 1)) - 1u;
                            d |= x != 0;
                        }
                        diff = __ballot(d) != 0;
                    }
                    if (diff) status = EC_AMBISEQ;
                    if (status == EC_SUCCESS) {
                        bool pd = cn != np;
                        if (!pd) {
                            bool d = false;
                            for (int32_t i = lane; i < cn; i += 64) d |= s.c_path[i] != s.o_path[i];
                            pd = __ballot(d) != 0;
                        }
                        if (pd) status = EC_AMBISNQ;
                    }
                }
                ecw_sync();
                new_opt = true;
                o_len = q_end;
                for (int32_t i = lane; i < cn; i += 64) s.o_path[i] = s.c_path[i];
                np = cn;
            } else if (sc < s_edist) {
                s_edist = sc;
            }
        }
        if (score <= bw && ql - K <= tl + bw && ((wk.end_utg != EC_NONE && wk.end_utg != w) || t_end < tl)) {
            if (n_path < EC_MAX_DFS_PATH) {           // the callee would return at once otherwise (syncerr.c:146-148)
                if (ECW_LIKELY(w_ln == 1)) vpend = true, v_arc = w_lp, v_depth = depth + 1;
                else if (w_ln > 1 && !push_frame(w_lp, w_ln, depth + 1)) { ECW_C(12, 1); return false; }      // (no arcs: the callee's loop does not run)
            }
        } else {
            ++n_path;
        }
        // the optimum consensus is only ever compared with a LATER path's (a tie): when the search ends here -- the usual case, a block with one
        // path -- nobody reads it and the copy is left out
        if (new_opt && (nfr > 0 || vpend)) {
            for (int32_t wi = lane; wi < ((o_len + 15) >> 4); wi += 64) s.os[wi] = s.cs[wi];
        }
        ECW_T(6);                                      // 6: result handling + push
    }
    ecw_sync();
    status_out = (uint32_t) status, np_out = (uint32_t) np, tried_out = tried, n_path_out = (uint32_t) n_path, wf_steps_out = wf_steps, wf_diag_out = (uint32_t) (wf_diag >> 6);
    return true;
}

#ifndef ECW_WPB
#define ECW_WPB 2                 // waves per workgroup of the LDS tiers (independent of each other: ecw_sync)
#endif
#ifndef ECW_BATCH
#define ECW_BATCH 16              // blocks taken from the queue per atomic (config 3: 8 -> 14.13 ms, 16 -> 13.83, 32 -> 13.86)
#endif
#define ECW_POOL_CHUNK 512        // path-pool entries taken per atomic

struct EcwArgs {
    EcLive lv;
    EcReads rd;
    const EcWork *work;
    uint64_t n_work;
    const uint32_t *todo;         // optional list of work indices (larger tiers); NULL = all
    uint64_t n_todo;
    double max_edist;
    uint8_t *slabs;               // BIG: one HBM slab per wave with every array
    uint64_t slab_bytes;
    uint32_t *os_slabs;           // LDS tiers: the optimum consensus of every wave, os_words apart
    uint64_t os_words;
    int32_t cap_t, cap_c, cap_w, cap_path, cap_f;
    EcBlockOut *out;              // [n_work]
    uint64_t *path_pool;          // optimum paths; bump-allocated in chunks
    uint64_t pool_cap;
    unsigned long long *pool_cursor;
    unsigned long long *next;     // work counter
    uint32_t *todo_out;           // blocks that did not fit this tier's carve-up: the next tier's list, appended by the waves themselves
    unsigned long long *todo_cnt;
    int32_t skip_l;               // blocks longer than this were routed to a larger tier before the launch (ec_route_kernel): not this launch's business
    int32_t batch;                // blocks taken from the queue per atomic: ECW_BATCH where the blocks are millions and small, 1 where they are few and long
    int32_t arc_budget;           // first tier: arcs after which a block is left to the classes behind it (0: never)
#ifdef ECW_PROF
    unsigned long long *prof;
#endif
};

// Blocks too long for the first tier's carve-up are known before anything runs (the length is in the work item): they go straight onto the
// list of the first tier that holds them, and those tiers run BESIDE the first one instead of after it.  cap[t] = longest block of tier t
// (0 for a tier that is not in use); list[t] / cnt[t] for t = 1 .. 3.
struct EcRoute {
    int32_t cap[4];
    uint32_t *list[4];
    unsigned long long *cnt[4];
};
// (r03: one global atomic per WORKGROUP and tier.  With one per wave and tier -- ~250 k atomics on three addresses at config 3 -- the kernel took 1.4 ms for a
//  pass over 7.9 M lengths; the lists are work queues, so their order is free.)
#define ECW_ROUTE_ITEMS 8          // work items per thread
// a class's blocks, longest first: the longest search of a batch should not be the last one to start (keys for a radix sort: ~length)
__global__ void ec_route_keys_kernel(const EcWork *work, const uint32_t *list, uint64_t n, uint32_t *keys)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ~(uint32_t) work[list[i]].l;
}
// ... and how many of them are longer than `cap` (in a list sorted longest first: the head of the list)
__global__ void ec_route_longer_kernel(const EcWork *work, const uint32_t *list, uint64_t n, int32_t cap, unsigned long long *cnt)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b = __ballot(i < n && work[list[i]].l > cap);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(cnt, (unsigned long long) __builtin_popcountll(b));
}
__global__ __launch_bounds__(256) void ec_route_kernel(const EcWork *work, uint64_t n_work, EcRoute rt)
{
    __shared__ uint32_t cnt[4], base_lo[4], base_hi[4];
    __shared__ uint32_t stage[4][256 * ECW_ROUTE_ITEMS / 2];          // a workgroup routes at most half of its items through the staging area; the rest go one by one
    constexpr uint32_t CAP = 256 * ECW_ROUTE_ITEMS / 2;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t) blockIdx.x * (256 * ECW_ROUTE_ITEMS);
#pragma unroll
    for (int k = 0; k < ECW_ROUTE_ITEMS; ++k) {
        const uint64_t i = i0 + (uint64_t) k * 256 + threadIdx.x;
        if (i >= n_work) continue;
        const int32_t l = work[i].l;
        int t = 0;
        while (t < 3 && l > rt.cap[t]) ++t;
        if (t == 0) continue;
        const uint32_t at = atomicAdd(&cnt[t], 1u);
        if (at < CAP) stage[t][at] = (uint32_t) i;
        else rt.list[t][atomicAdd(rt.cnt[t], 1ULL)] = (uint32_t) i;   // (a workgroup full of long blocks: straight to the list)
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= 3) {
        const uint32_t n = cnt[threadIdx.x] < CAP? cnt[threadIdx.x] : CAP;
        const unsigned long long b = n? atomicAdd(rt.cnt[threadIdx.x], (unsigned long long) n) : 0ULL;
        base_lo[threadIdx.x] = (uint32_t) b, base_hi[threadIdx.x] = (uint32_t) (b >> 32);
    }
    __syncthreads();
    for (int t = 1; t <= 3; ++t) {
        const uint32_t n = cnt[t] < CAP? cnt[t] : CAP;
        const uint64_t b = (uint64_t) base_hi[t] << 32 | base_lo[t];
        for (uint32_t j = threadIdx.x; j < n; j += 256) rt.list[t][b + j] = stage[t][j];
    }
}

__host__ __device__ inline uint32_t ecw_words(int32_t bases) { return (uint32_t) ((bases + 15) / 16 + 2); }
// MODE 2 of ec_wave_kernel: what stays in LDS (ts, cs, two wavefronts) and what goes to the wave's HBM slab (two paths, frames)
__host__ __device__ inline uint32_t ecw_lds_words_hybrid(int32_t cap_t, int32_t cap_c, int32_t cap_w)
{
    return (ecw_words(cap_t) + ecw_words(cap_c) + 2u * (uint32_t) (cap_w + 2) + 1u) & ~1u;
}
__host__ __device__ inline uint64_t ecw_slab_bytes_hybrid(int32_t cap_path, int32_t cap_f) { return ((uint64_t) 16 * (uint64_t) cap_path + (uint64_t) cap_f + 63) & ~63ULL; }
// 32-bit words of one wave's carve-up: ts, cs, os, two wavefronts, two paths, frames
// (os_too: the optimum consensus as well -- the HBM slabs of the last tier; the LDS tiers keep theirs in a slab of its own, EcwArgs::os_slabs)
__host__ __device__ inline uint32_t ecw_scratch_words(int32_t cap_t, int32_t cap_c, int32_t cap_w, int32_t cap_path, int32_t cap_f, bool os_too = true)
{
    return ((ecw_words(cap_t) + (os_too? 2u : 1u) * ecw_words(cap_c) + 2u * (uint32_t) (cap_w + 2) + 1u) & ~1u) + 4u * (uint32_t) cap_path + (uint32_t) cap_f / 4u;
}

// MODE 0: every array in LDS.  MODE 1 (BIG): every array in an HBM slab.  MODE 2 (round 4): the strings and the wavefronts -- what the alignment
// reads in every step, 89 % of a long search's cycles -- in LDS, the paths and the DFS frames -- touched once per arc -- in an HBM slab.  A search
// through a tandem array is hundreds of levels deep with a frame at every level: it outgrew every LDS carve-up and ran in the slabs at a tenth of
// the speed, eight seconds for 40 k reads of the config-1 surrogate (profiles/r04e_solver_config1s.txt).
template <int MODE, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void ec_wave_kernel(EcwArgs a)
{
    constexpr bool BIG = MODE == 1;
    extern __shared__ uint32_t ecw_lds[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (uint32_t) WPB + ecw_uniu(threadIdx.x >> 6);      // the waves of a workgroup share nothing but the launch
    EcwScratch s;
    s.cap_t = a.cap_t, s.cap_c = a.cap_c, s.cap_w = a.cap_w, s.cap_path = a.cap_path, s.cap_f = a.cap_f;
    s.arc_budget = MODE == 0? a.arc_budget : 0;
    uint32_t *const p0 = BIG? (uint32_t *) (a.slabs + (uint64_t) wave * a.slab_bytes)
                            : ecw_lds + (WPB > 1? ecw_uniu(threadIdx.x >> 6) * (MODE == 2? ecw_lds_words_hybrid(a.cap_t, a.cap_c, a.cap_w)
                                                                                                      : ecw_scratch_words(a.cap_t, a.cap_c, a.cap_w, a.cap_path, a.cap_f, false)) : 0u);
    uint32_t *p = p0;
    s.ts = p, p += ecw_words(a.cap_t);
    s.cs = p, p += ecw_words(a.cap_c);
    // The optimum consensus is written when a second path may follow and read when one ties with the first -- hardly ever (a block with one path
    // does neither): in the LDS tiers it lives in HBM, and its 1.4 KB (first tier) buy four more waves per CU.
    if (BIG) s.os = p, p += ecw_words(a.cap_c);
    else s.os = a.os_slabs + (uint64_t) wave * a.os_words;
    s.ka = (int32_t *) p, p += a.cap_w + 2;
    s.kb = (int32_t *) p, p += a.cap_w + 2;
    p += (p - p0) & 1;                 // (the base is 8-byte aligned; no detour through an integer, which would turn every access behind it into a flat one)
    if (MODE == 2) p = (uint32_t *) (a.slabs + (uint64_t) wave * a.slab_bytes);
    s.c_path = (uint64_t *) p, p += 2 * a.cap_path;
    s.o_path = (uint64_t *) p, p += 2 * a.cap_path;
    s.frames = (uint8_t *) p;
#ifdef ECW_PROF
    for (int i = 0; i < 32; ++i) s.prof[i] = 0;
    s.t_last = __builtin_readcyclecounter();
#endif
    const uint64_t total = a.todo? a.n_todo : a.n_work;
    uint64_t pool_at = 0, pool_end = 0;                // this wave's chunk of the path pool
    ECW_D(13);                                         // 13: waves launched, 12: waves that found work, 14: batches
#if defined(ECW_PROF) || defined(ECW_CENSUS)
    bool first_batch = true;
#endif
#ifdef ECW_CENSUS
    if (lane == 0) atomicAdd(&ecw_census[0], 1ULL);
#endif
    for (;;) {
        ECW_T(7);                                      // 7: queue + output
        unsigned long long t0 = 0;
        if (lane == 0) t0 = atomicAdd(a.next, (unsigned long long) a.batch);
        t0 = ecw_uni64(t0);
        if (t0 >= total) break;
#ifdef ECW_CENSUS
        if (first_batch && lane == 0) atomicAdd(&ecw_census[1], 1ULL);
        first_batch = false;
#endif
#ifdef ECW_PROF
        if (first_batch) { ECW_D(12); first_batch = false; }
        ECW_D(14);
#endif
        const int cnt = total - t0 < (uint64_t) a.batch? (int) (total - t0) : a.batch;
        // lane i holds block i of the batch
        uint64_t my_wi = 0;
        uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0, m2 = m0;
        if (lane < cnt) {
            my_wi = a.todo? a.todo[t0 + lane] : t0 + lane;
            const uint4 *q = (const uint4 *) (a.work + my_wi);
            m0 = q[0], m1 = q[1], m2 = q[2];
        }
        for (int i = 0; i < cnt; ++i) {
            EcWork wk;
            const uint64_t wi = (uint64_t) ecw_lane((uint32_t) (my_wi >> 32), i) << 32 | ecw_lane((uint32_t) my_wi, i);
            wk.beg_utg = (uint64_t) ecw_lane(m0.y, i) << 32 | ecw_lane(m0.x, i);
            wk.end_utg = (uint64_t) ecw_lane(m0.w, i) << 32 | ecw_lane(m0.z, i);
            wk.read = ecw_lane(m1.x, i), wk.beg_pos = ecw_lane(m1.y, i);
            wk.l = (int32_t) ecw_lane(m1.z, i), wk.r = (int32_t) ecw_lane(m1.w, i);
            wk.hs16 = ecw_lane(m2.x, i), wk.lp = ecw_lane(m2.y, i), wk.ln = ecw_lane(m2.z, i), wk.pad = 0;
            if (ECW_RARE(wk.l > a.skip_l)) continue;
            EcBlockOut o;
            o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0, o.tried = 0, o.n_path = 0, o.wf_steps = 0, o.wf_diag = 0, o.tier = (uint32_t) MODE;
            const uint64_t tick0 = MODE != 0? __builtin_amdgcn_s_memrealtime() : 0;      // (the first tier does without: millions of small blocks)
            if (ECW_RARE(wk.l < EC_MIN_ERR_SEQ_LEN)) {
                o.short_block = 1;                     // syncerr.c:502-504
            } else {
                uint32_t st = 0, np = 0;
                if (ECW_RARE(!ecw_solve_block(a.lv, a.rd, wk, s, a.max_edist, st, np, o.tried, o.n_path, o.wf_steps, o.wf_diag))) {
                    o.flags = 1;
                    if (lane == 0) a.todo_out[atomicAdd(a.todo_cnt, 1ULL)] = (uint32_t) wi;
                } else {
                    o.status = st, o.np = np;
                    if (st == EC_SUCCESS && np) {
                        if (ECW_RARE(pool_at + np > pool_end)) {
                            const unsigned long long want = np > ECW_POOL_CHUNK? np : ECW_POOL_CHUNK;
                            unsigned long long off = 0;
                            if (lane == 0) off = atomicAdd(a.pool_cursor, want);
                            pool_at = ecw_uni64(off), pool_end = pool_at + want;
                        }
                        o.path_off = pool_at;
                        if (pool_at + np <= a.pool_cap) for (uint32_t j = lane; j < np; j += 64) a.path_pool[pool_at + j] = s.o_path[j];
                        pool_at += np;
                    }
                }
            }
            o.ticks = MODE != 0? (uint32_t) (__builtin_amdgcn_s_memrealtime() - tick0) : 0u;
            if (lane == 0) a.out[wi] = o;
            ecw_sync();
            ECW_C(10, 1);                              // 10: blocks
        }
    }
#ifdef ECW_PROF
    if (lane == 0) for (int i = 0; i < 32; ++i) atomicAdd(a.prof + i, s.prof[i]);
#endif
}

}  // namespace oatk
