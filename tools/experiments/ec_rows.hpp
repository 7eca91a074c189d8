// oatk_amd/csrc/ec_rows.hpp -- the alignment of the error-block search by MATRIX ROWS instead of wavefront steps (round 5; EXPERIMENTAL: the debug entry
// only -- oatk_hip_debug_wf_ed_wg with variant 32 -- nothing of the product path calls it yet; written after the round's last GPU run and never executed: its
// tests in tests/test_gpu_levdist.py run on request, OATK_TEST_EC_ROWS=1).
//
// wf_ed_core (levdist.c:265-310), resumed arc by arc with a longer and longer query, returns what the banded edit-distance matrix of (target, query) says
// (DESIGN.md 8.3 "What the alignment IS"; tests/trace/ec_trace.c ECT_DP / ECT_SLOTS: 7.6 M arcs of the config-1 surrogate, none differs;
// tests/test_oracle_golden.py: the reference's own resumed traces):
//     score = the least value on the matrix's boundary -- the query's last row, the target's last column -- but not below the score of the call before;
//     (t_end, q_end) = the boundary cell of the LOWEST diagonal within that score;  beyond the band: score bw + 1, no end.
// The state is ONE value per diagonal, in the wavefront's own slots (slot = diagonal + OFF, diagonal = q - t): the cell of the current row on that diagonal,
// or, once the diagonal has run past the target's last column, the value it had THERE (frozen).  When a call's last row is done every diagonal holds its one
// boundary cell and the outcome is read off the slots.  A row costs no LDS exchange, no barrier and no extension loop, whatever the strings are:
//     A[s]  = min(D[s] + (target[t] != base), D[s - 1] + 1)                      (the cell above-left on the same diagonal; the cell above: the slot below)
//     D'[s] = min over j >= s of (A[j] + j) - s                                    (the cell to the left is the slot above: a min-plus scan from the high slots down)
// One wave, R registers per lane: slot s = r * 64 + lane, 2 bw + 5 <= 64 R.
#pragma once

#include "../../oatk_amd/csrc/ec_heavy.hpp"
#include "../../oatk_amd/csrc/ec_tables.hpp"

namespace oatk {

#define ECR_INF (1 << 24)
#define ECR_R 8


// the row before the first (the empty query): D(-1, t) = t + 1 on diagonal -1 - t
template <int R>
__device__ __forceinline__ void ecr_init(int32_t tl, int32_t OFF, int32_t (&D)[R])
{
    const int lane = (int) threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int32_t d = r * 64 + lane - OFF, t = -1 - d;
        D[r] = t >= -1 && t < tl? t + 1 : ECR_INF;
    }
}

// rows `from` .. `to` - 1 of the query (cs, sixteen bases to a word) onto the slots
template <int R>
__device__ __forceinline__ void ecr_rows(const uint32_t *ts, const uint32_t *cs, int32_t tl, int32_t OFF, int32_t (&D)[R], int32_t from, int32_t to)
{
    const int lane = (int) threadIdx.x & 63;
    for (int32_t q = from; q < to; ++q) {
        const int32_t c = (int32_t) ecw_uniu(ecr_base(cs, q));
        int32_t carry = ECR_INF;                                                     // min over the slots of the registers above of (A[j] + j)
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const int32_t s = r * 64 + lane, d = s - OFF, t = q - d;
            if ((r * 64 + 63 - OFF) < q - tl) continue;                              // (uniform) every diagonal of this register has left the matrix: frozen
            const int32_t old = D[r];
            int32_t below = ech_dpp<0x138>(ECR_INF, old);                            // wave_shr:1 -- lane i takes lane i - 1 (slot s - 1, still the old row); lane 0 keeps `old`
            if (r > 0) { const int32_t e = (int32_t) ecw_lane((uint32_t) D[r > 0? r - 1 : 0], 63); below = lane == 0? e : below; }
            const bool in = t >= 0 && t < tl;
            const int32_t tb = (int32_t) ecr_base(ts, in? t : 0);
            int32_t a = old + (tb != c? 1 : 0);
            a = below + 1 < a? below + 1 : a;
            a = t == -1? q + 1 : a;                                                  // the matrix's first column: D(q, -1) = q + 1
            a = t < -1? ECR_INF : a;
            const bool frozen = t >= tl;
            int32_t x = (frozen? ECR_INF : a) + s;
            // suffix minimum over the wave (towards the higher lanes): within rows of sixteen by DPP, the rows' minima by readlane
            { const int32_t y = ech_dpp<0x101>(ECR_INF + s, x); x = y < x? y : x; }  // row_shl:1 -- lane i takes lane i + 1 of its row
            { const int32_t y = ech_dpp<0x102>(ECR_INF + s, x); x = y < x? y : x; }
            { const int32_t y = ech_dpp<0x104>(ECR_INF + s, x); x = y < x? y : x; }
            { const int32_t y = ech_dpp<0x108>(ECR_INF + s, x); x = y < x? y : x; }
            {
                const int32_t m3 = (int32_t) ecw_lane((uint32_t) x, 48);
                int32_t m2 = (int32_t) ecw_lane((uint32_t) x, 32), m1 = (int32_t) ecw_lane((uint32_t) x, 16);
                m2 = m3 < m2? m3 : m2, m1 = m2 < m1? m2 : m1;
                int32_t hi = lane < 16? m1 : (lane < 32? m2 : (lane < 48? m3 : ECR_INF + 64 * R));
                hi = carry < hi? carry : hi;
                x = hi < x? hi : x;
            }
            carry = (int32_t) ecw_lane((uint32_t) x, 0);
            int32_t v = x - s;
            v = v > ECR_INF? ECR_INF : v;
            D[r] = frozen? old : v;
        }
    }
}

// the call's outcome off the slots (query length ql, the score of the call before): score, t_end, q_end as wf_ed_core leaves them, one past the last aligned base
template <int R>
__device__ __forceinline__ void ecr_read(int32_t tl, int32_t ql, int32_t bw, int32_t OFF, const int32_t (&D)[R], int32_t before, int32_t &score, int32_t &t_end, int32_t &q_end)
{
    const int lane = (int) threadIdx.x & 63;
    bool ok[R];
    int32_t best = ECR_INF;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int32_t d = r * 64 + lane - OFF, t = ql - 1 - d;
        ok[r] = t >= 0 && (t < tl || d >= 1 - tl);                                   // a cell of the last row, or of the last column in a row >= 0 (frozen)
        const int32_t v = ok[r]? D[r] : ECR_INF;
        best = v < best? v : best;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int32_t y = __shfl_xor(best, o, 64); best = y < best? y : best; }
    best = ecw_uni(best);
    const int32_t sc = best > before? best : before;
    if (bw >= 0 && sc > bw) { score = bw + 1, t_end = 0, q_end = 0; return; }
    score = sc, t_end = 0, q_end = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t b = __ballot(ok[r] && D[r] <= sc);
        if (b) {
            const int32_t d = r * 64 + __builtin_ctzll(b) - OFF, t = ql - 1 - d;
            if (t < tl) t_end = t + 1, q_end = ql; else t_end = tl, q_end = tl + d;
            return;
        }
    }
}

// test entry (include/oatk_hip_ec.h: oatk_hip_debug_wf_ed_wg, variant 32): one wave per job, the query's lengths taken one after the other like the search resumes them
template <int R>
__global__ __launch_bounds__(64) void ecr_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                       const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3, int32_t cap_words)
{
    extern __shared__ uint32_t ecr_lds[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x;
    uint32_t *ts = ecr_lds, *cs = ecr_lds + cap_words;
    const int32_t tlen = tl[j], band = bw[j];
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];
    for (uint64_t i = t; i < nt; i += 64) ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += 64) cs[i] = qw[qw_off[j] + i];
    const int32_t OFF = (band < 0? tlen : band) + 2;
    int32_t D[R];
    ecr_init<R>(tlen, OFF, D);
    ecw_sync();
    int32_t score = 0, t_end = 0, q_end = 0, rq = 0;
    for (uint64_t s = step_off[j]; s < step_off[j + 1]; ++s) {
        const int32_t ql = step_ql[s];
        ecr_rows<R>(ts, cs, tlen, OFF, D, rq, ql);
        rq = ql;
        ecr_read<R>(tlen, ql, band, OFF, D, score, score, t_end, q_end);
        if (t == 0) out3[3 * s] = score, out3[3 * s + 1] = t_end, out3[3 * s + 2] = q_end;
    }
}

// ---------------------------------------------------------------- the row in bits ----------------------------------------------------------------
// The same row as DIFFERENCES (tests/c/rows_bitpar_test.c is this, on the CPU, checked against the plain matrix: 600 k rows, 20 k resumed calls): neighbouring cells of
// a row differ by -1, 0 or +1, so the band is two bit vectors; band cell b = 0 .. W - 1 of row q is the cell (q, t) with t = q - OFF + b (diagonal OFF - b), and the band
// moves one cell down the target per query base -- Hyyro's diagonal band (2003) of Myers' bit-vector step (1999): the previous row's vectors shifted by one, ONE multiword
// addition, a dozen logical operations.  Lane w of the wave holds word w (32 bits) of every vector, W <= 32 x ECB_LANES; the carries of the addition are resolved with two
// ballots and a scalar addition.  The value of the middle cell (diagonal 0) is carried as a number: it moves by 1 - D0[OFF] per row.  Cells with t <= -1 continue the
// matrix upwards as q - t, cells with t >= tl continue it with a base that matches nothing: neither feeds a cell of the matrix, nothing is masked.  What the target's
// last column held in each row is kept in LDS (lc[]: the rows whose band reaches it), the outcome of a call is read off the expanded last row and lc[].
// (variant 33 of the debug entry; never executed: OATK_TEST_EC_ROWS=1)
#define ECB_LANES 20                          // words: W = 2 (bw + 2) + 1 <= 640

struct EcbState { uint32_t pv, mv, e0, e1, e2, e3, wmask; int32_t mid; };

__device__ __forceinline__ uint32_t ecb_upto(int32_t k) { return k == 31? 0xFFFFFFFFu : (1u << (k + 1)) - 1u; }      // bits 0 .. k

__device__ __forceinline__ uint32_t ecb_shr1(uint32_t v) { const uint32_t nx = (uint32_t) ech_dpp<0x130>(0, (int32_t) v); return v >> 1 | nx << 31; }                       // bit b takes bit b + 1 (lane i + 1's lowest bit comes in at the top)

// the row before the first (D(-1, t) = |t + 1|: falling towards t = -1, rising from there) and the window of target bases it would see
__device__ __forceinline__ void ecb_init(EcbState &st, const uint32_t *ts, int32_t tl, int32_t OFF)
{
    const int lane = (int) threadIdx.x & 63;
    const int32_t W = 2 * OFF + 1, wl = (W - 1) >> 5;
    st.wmask = lane < wl? 0xFFFFFFFFu : (lane == wl? ((W & 31)? (1u << (W & 31)) - 1u : 0xFFFFFFFFu) : 0u);
    st.pv = st.mv = st.e0 = st.e1 = st.e2 = st.e3 = 0;
    st.mid = 0;
    for (int i = 0; i < 32; ++i) {
        const int32_t b = (lane << 5) + i, t = -1 - OFF + b;
        if (b >= W) break;
        if (b <= OFF) st.mv |= 1u << i; else st.pv |= 1u << i;
        if (t >= 0 && t < tl) {
            const uint32_t x = ecr_base(ts, t);
            st.e0 |= (uint32_t) (x == 0) << i, st.e1 |= (uint32_t) (x == 1) << i, st.e2 |= (uint32_t) (x == 2) << i, st.e3 |= (uint32_t) (x == 3) << i;
        }
    }
}

// the sum of (pv - mv) over band cells lo .. hi (inclusive; lo > hi: 0), the same in every lane
__device__ __forceinline__ int32_t ecb_sum(const EcbState &st, int32_t lo, int32_t hi)
{
    const int lane = (int) threadIdx.x & 63;
    const int32_t b0 = lane << 5;
    int32_t v = 0;
    if (lo <= hi && b0 <= hi && b0 + 31 >= lo) {
        const int32_t a = lo > b0? lo - b0 : 0, z = hi < b0 + 31? hi - b0 : 31;
        const uint32_t m = (z == 31? 0xFFFFFFFFu : (1u << (z + 1)) - 1u) & ~((1u << a) - 1u);
        v = __popc(st.pv & m) - __popc(st.mv & m);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return ecw_uni(v);
}

// one row: the query's base c (uniform); q = the row's index
__device__ __forceinline__ void ecb_row(EcbState &st, const uint32_t *ts, int32_t tl, int32_t OFF, int32_t q, uint32_t c)
{
    const int lane = (int) threadIdx.x & 63;
    const int32_t W = 2 * OFF + 1, wl = (W - 1) >> 5;
    const uint32_t topbit = 1u << ((W - 1) & 31);
    // the band moves one cell down the target
    {
        const int32_t t_new = q + OFF;
        const uint32_t x = t_new >= 0 && t_new < tl? ecw_uniu(ecr_base(ts, t_new)) : 4u;
        const uint32_t ins = lane == wl? topbit : 0u;
        st.e0 = ecb_shr1(st.e0) | (x == 0? ins : 0u), st.e1 = ecb_shr1(st.e1) | (x == 1? ins : 0u);
        st.e2 = ecb_shr1(st.e2) | (x == 2? ins : 0u), st.e3 = ecb_shr1(st.e3) | (x == 3? ins : 0u);
    }
    // the previous row's differences seen from the new band (its cell at the same t was cell b + 1 there); below the bottom: one more
    const uint32_t pva = ecb_shr1(st.pv) | (lane == wl? topbit : 0u), mva = ecb_shr1(st.mv);
    const uint32_t eq = c == 0? st.e0 : (c == 1? st.e1 : (c == 2? st.e2 : st.e3));
    // Myers' step; the one addition, its carries from word to word by ballots
    const uint32_t a = eq & pva, s1 = a + pva;
    const uint64_t G = __ballot(s1 < a), P = __ballot(s1 == 0xFFFFFFFFu), U = G << 1, C = (P + U) ^ P;
    const uint32_t s2 = s1 + (uint32_t) (C >> lane & 1ULL);
    const uint32_t xh = (s2 ^ pva) | eq, xv = eq | mva, d0 = xh | mva;
    uint32_t ph = (mva | ~(xh | pva)) & st.wmask, mh = pva & xh;
    st.mid += 1 - (int32_t) (ecw_lane(d0, OFF >> 5) >> (OFF & 31) & 1u);
    // the differences along the new row: the row above the band's top counts as one more
    ph = ecb_shl1(ph, 1u) & st.wmask, mh = ecb_shl1(mh, 0u) & st.wmask;
    st.pv = (mh | ~(xv | ph)) & st.wmask, st.mv = ph & xv;
}

// rows from .. to - 1; lc[q - lc0] = what the target's last column holds in row q, for the rows whose band reaches it (lc0 = tl - 1 - OFF)
__device__ __forceinline__ void ecb_rows(EcbState &st, const uint32_t *ts, const uint32_t *cs, int32_t tl, int32_t OFF, int32_t from, int32_t to, int32_t *lc)
{
    const int32_t W = 2 * OFF + 1;
    for (int32_t q = from; q < to; ++q) {
        ecb_row(st, ts, tl, OFF, q, ecw_uniu(ecr_base(cs, q)));
        const int32_t bl = tl - 1 - q + OFF;                                   // the band cell of the last column in this row
        if (bl >= 0 && bl < W) {
            const int32_t v = st.mid + (bl > OFF? ecb_sum(st, OFF + 1, bl) : -ecb_sum(st, bl + 1, OFF));
            if (((int) threadIdx.x & 63) == 0) lc[q - (tl - 1 - OFF)] = v;
        }
    }
}

// the call's outcome (as ecr_read): the last row expanded to numbers, sixty-four cells a turn, and lc[]
__device__ __forceinline__ void ecb_read(const EcbState &st, int32_t tl, int32_t ql, int32_t bw, int32_t OFF, const int32_t *lc, int32_t before, int32_t &score, int32_t &t_end, int32_t &q_end)
{
    const int lane = (int) threadIdx.x & 63;
    const int32_t W = 2 * OFF + 1, nw = (W + 31) >> 5;
    // per word: the sum of the differences of all words before it
    int32_t wsum = __popc(st.pv) - __popc(st.mv), wpre = 0;
    for (int w = 0; w < nw; ++w) { const int32_t x = (int32_t) ecw_lane((uint32_t) wsum, w); wpre += lane > w? x : 0; }
    const uint32_t m_off = ecb_upto(OFF & 31);
    const int32_t c_off = (int32_t) ecw_lane((uint32_t) wpre, OFF >> 5) + (int32_t) ecw_lane((uint32_t) (__popc(st.pv & m_off) - __popc(st.mv & m_off)), OFF >> 5);     // the sum up to the middle cell
    const int32_t lc0 = tl - 1 - OFF;
    const int32_t q_lo = lc0 > 0? lc0 : 0;                                     // rows whose last-column cell is kept: q_lo .. ql - 1 (within lc0 .. lc0 + W - 1)
    int32_t best = ECR_INF;
    const int32_t turns = (W + 63) >> 6;
    for (int r = 0; r < turns; ++r) {
        const int32_t b = (r << 6) + lane, t = ql - 1 - OFF + b, w = b >> 5;
        const uint32_t pw = (uint32_t) __shfl((int32_t) st.pv, w, 64), mw = (uint32_t) __shfl((int32_t) st.mv, w, 64);
        const int32_t pre = __shfl(wpre, w, 64);
        const uint32_t m = ecb_upto(b & 31);
        const int32_t v = st.mid + pre + __popc(pw & m) - __popc(mw & m) - c_off;
        if (b < W && t >= 0 && t < tl) best = v < best? v : best;
    }
    for (int32_t qq = q_lo + lane; qq < ql && qq < lc0 + W; qq += 64) { const int32_t v = lc[qq - lc0]; best = v < best? v : best; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int32_t y = __shfl_xor(best, o, 64); best = y < best? y : best; }
    best = ecw_uni(best);
    const int32_t sc = best > before? best : before;
    if (bw >= 0 && sc > bw) { score = bw + 1, t_end = 0, q_end = 0; return; }
    score = sc;
    // the lowest diagonal within sc: on the last column diagonals grow with the row (the first row that qualifies), on the last row they fall as b grows (the last cell)
    int32_t d_col = ECR_INF, q_col = 0;
    for (int32_t q0 = q_lo; q0 < ql && q0 < lc0 + W && d_col == ECR_INF; q0 += 64) {
        const int32_t qq = q0 + lane;
        const uint64_t bm = __ballot(qq < ql && qq < lc0 + W && lc[qq - lc0] <= sc);
        if (bm) q_col = q0 + __builtin_ctzll(bm), d_col = q_col - (tl - 1);
    }
    int32_t d_row = ECR_INF, t_row = 0;
    for (int r = turns - 1; r >= 0 && d_row == ECR_INF; --r) {
        const int32_t b = (r << 6) + lane, t = ql - 1 - OFF + b, w = b >> 5;
        const uint32_t pw = (uint32_t) __shfl((int32_t) st.pv, w, 64), mw = (uint32_t) __shfl((int32_t) st.mv, w, 64);
        const int32_t pre = __shfl(wpre, w, 64);
        const uint32_t m = ecb_upto(b & 31);
        const int32_t v = st.mid + pre + __popc(pw & m) - __popc(mw & m) - c_off;
        const uint64_t bm = __ballot(b < W && t >= 0 && t < tl && v <= sc);
        if (bm) { const int32_t bb = (r << 6) + 63 - __builtin_clzll(bm); t_row = ql - 1 - OFF + bb, d_row = OFF - bb; }
    }
    if (d_col <= d_row) t_end = tl, q_end = q_col + 1;
    else t_end = t_row + 1, q_end = ql;
}

__global__ __launch_bounds__(64) void ecb_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                       const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3, int32_t cap_words)
{
    extern __shared__ uint32_t ecb_lds[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x;
    uint32_t *ts = ecb_lds, *cs = ecb_lds + cap_words;
    int32_t *lc = (int32_t *) (ecb_lds + 2 * cap_words);
    const int32_t tlen = tl[j], band = bw[j];
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];
    for (uint64_t i = t; i < nt; i += 64) ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += 64) cs[i] = qw[qw_off[j] + i];
    for (int i = t; i < 32 * ECB_LANES + 8; i += 64) lc[i] = ECR_INF;
    const int32_t OFF = band + 2;                                              // (banded jobs only: the host refuses the others for this variant)
    ecw_sync();
    EcbState st;
    ecb_init(st, ts, tlen, OFF);
    int32_t score = 0, t_end = 0, q_end = 0, rq = 0;
    for (uint64_t s = step_off[j]; s < step_off[j + 1]; ++s) {
        const int32_t ql = step_ql[s];
        ecb_rows(st, ts, cs, tlen, OFF, rq, ql, lc);
        rq = ql;
        ecw_sync();
        ecb_read(st, tlen, ql, band, OFF, lc, score, score, t_end, q_end);
        if (t == 0) out3[3 * s] = score, out3[3 * s + 1] = t_end, out3[3 * s + 2] = q_end;
    }
}

}  // namespace oatk
