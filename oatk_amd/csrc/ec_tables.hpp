// oatk_amd/csrc/ec_tables.hpp -- the two tables by which an arc that appends a long string can be KNOWN to die by score without a wavefront step (ec_fused.hpp, CERT;
// DESIGN.md 8.3).  Bit-parallel approximate matching (Myers 1999) of the reversed string against the reversed read segment, the string's words in the lanes of a wave.
// (Until round 6 this was the tail of ec_rows.hpp, whose row solvers -- the alignment by matrix rows, a value per diagonal or the row in bits -- were measured out of the
//  product and live in tools/experiments/ec_rows.hpp.)
#pragma once
#include "ec_heavy.hpp"

namespace oatk {

__device__ __forceinline__ uint32_t ecr_base(const uint32_t *w, int32_t p) { return (w[p >> 4] >> ((uint32_t) (p & 15) << 1)) & 3u; }
__device__ __forceinline__ uint32_t ecb_shl1(uint32_t v, uint32_t in) { const uint32_t pr = (uint32_t) ech_dpp<0x138>((int32_t) (in << 31), (int32_t) v); return v << 1 | pr >> 31; }   // bit b takes bit b - 1, bit 0 takes `in`

// ---------------------------------------------------------------- the tables of a long arc ----------------------------------------------------------------
// An arc that appends a long string can be KNOWN to die by score without a step (tests/trace/ec_trace.c ECT_ROWS: from the parent's wavefront alone 92.8 % of
// the steps of such arcs on the config-1 surrogate, and no arc that lives): min over the band of (a bound on the parent's last row at t' + table[t' + 1]) > bw, with
//   table 0 [u] = the least cost of fitting the WHOLE string into the target from position u on, any end           (a cell of the new last row)
//   table 1 [u] = the least cost of some PREFIX of the string against the target from u TO ITS END                  (a cell of the last column in one of the new rows)
// Both are approximate matching of the reversed string against the reversed target (Myers 1999), one pass over the target each: the string's words in the lanes (<= 1024
// bases), carries by ballots.  tests/c/prof_bitpar_test.c is this on the CPU against the plain recurrences.  (oatk_hip_debug_tables; never executed: OATK_TEST_EC_ROWS=1)
// The scan over target positions p_top .. u_lo (descending), from a fresh state; out[pos] is written for pos < u_hi.  A scan that starts at the target's last base
// (p_top = tl - 1) gives the tables as defined.  Table 0 only: one that starts further down gives, at every u <= p_top + 1 - m - cut, the same number where that number is
// <= cut and some number > cut where it is not -- whatever fits the string into the target from u on within cut ends before u + m + cut (round 6: the stretches of a table
// are built by the waves of a workgroup side by side, each with m + cut positions of run-up).
__device__ __forceinline__ void ecb_table_range(const uint32_t *ts, int32_t tl, const uint32_t *cs, int32_t s0, int32_t m, int second, int32_t p_top, int32_t u_lo, int32_t u_hi, int32_t *out)
{
    const int lane = (int) threadIdx.x & 63;
    const int32_t wl = (m - 1) >> 5;
    const uint32_t top = 1u << ((m - 1) & 31);
    const uint32_t wmask = lane < wl? 0xFFFFFFFFu : (lane == wl? ((m & 31)? (1u << (m & 31)) - 1u : 0xFFFFFFFFu) : 0u);
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < 32; ++i) {                                             // pattern position p = 32 lane + i is the string's base m - 1 - p
        const int32_t pp = (lane << 5) + i;
        if (pp >= m) break;
        const uint32_t x = ecr_base(cs, s0 + m - 1 - pp);
        e0 |= (uint32_t) (x == 0) << i, e1 |= (uint32_t) (x == 1) << i, e2 |= (uint32_t) (x == 2) << i, e3 |= (uint32_t) (x == 3) << i;
    }
    uint32_t pv = second? 0u : wmask, mv = 0;
    int32_t score = second? 0 : m;
    // sixty-four target positions a turn (round 6): lane l fetches the base of the turn's l-th position and keeps the score after it, so a position costs no LDS round trip and no
    // store of its own
    for (int32_t hi = p_top; hi >= u_lo; hi -= 64) {                           // hi = the target position of the turn's first step; lane l's is hi - l
        const int32_t nj = hi - u_lo + 1 < 64? hi - u_lo + 1 : 64;
        const uint32_t cv = hi - lane >= 0? ecr_base(ts, hi - lane) : 0u;
        int32_t mine = 0;
        for (int jj = 0; jj < nj; ++jj) {
            const uint32_t c = ecw_lane(cv, jj);
            const uint32_t eq = c == 0? e0 : (c == 1? e1 : (c == 2? e2 : e3));
            const uint32_t a = eq & pv, s1 = a + pv;
            const uint64_t G = __ballot(s1 < a), P = __ballot(s1 == 0xFFFFFFFFu), U = G << 1, C = (P + U) ^ P;
            const uint32_t s2 = s1 + (uint32_t) (C >> lane & 1ULL);
            const uint32_t xh = (s2 ^ pv) | eq, xv = eq | mv;
            uint32_t ph = (mv | ~(xh | pv)) & wmask, mh = pv & xh;
            score += (ecw_lane(ph, wl) & top) != 0u, score -= (ecw_lane(mh, wl) & top) != 0u;
            ph = ecb_shl1(ph, (uint32_t) second) & wmask, mh = ecb_shl1(mh, 0u) & wmask;      // (a free start in the target: nothing comes in; the second table's first row counts the columns)
            pv = (mh | ~(xv | ph)) & wmask, mv = ph & xv;
            mine = lane == jj? score : mine;
        }
        if (lane < nj && hi - lane < u_hi) out[hi - lane] = mine;
    }
}
__device__ __forceinline__ void ecb_table(const uint32_t *ts, int32_t tl, const uint32_t *cs, int32_t s0, int32_t m, int second, int32_t *out)
{
    if (((int) threadIdx.x & 63) == 0) out[tl] = second? 0 : m;
    ecb_table_range(ts, tl, cs, s0, m, second, tl - 1, 0, tl, out);
}
// table 1 is only ever below `cut` near the target's end (a prefix of m bases against tl - u target bases costs at least tl - u - m): entries below this u are not written,
// and not read (ec_fused.hpp)
__host__ __device__ inline int32_t ecb_table1_lo(int32_t tl, int32_t m, int32_t cut) { const int32_t lo = tl - (m + cut + 1); return lo > 0? lo : 0; }
// Both tables by the NW waves of a workgroup (every wave calls; no barrier inside): wave 0 the second table's stretch near the target's end, the others a stretch of the
// first table each with its run-up.  Table 0 is exact where it is <= cut and > cut elsewhere; NW = 1: both tables in full, one after the other.
template <int NW>
__device__ __forceinline__ void ecb_tables_wg(const uint32_t *ts, int32_t tl, const uint32_t *cs, int32_t s0, int32_t m, int32_t cut, int32_t *t0, int32_t *t1)
{
    const int lane = (int) threadIdx.x & 63;
    const int wave = ecw_uni((int) threadIdx.x >> 6);
    if constexpr (NW == 1) {
        ecb_table(ts, tl, cs, s0, m, 0, t0);
        ecb_table(ts, tl, cs, s0, m, 1, t1);
    } else {
        if (wave == 0) {
            if (lane == 0) t1[tl] = 0;
            ecb_table_range(ts, tl, cs, s0, m, 1, tl - 1, ecb_table1_lo(tl, m, cut), tl, t1);
        } else {
            constexpr int P = NW - 1;
            const int32_t span = (tl + 1 + P - 1) / P, u_lo = (wave - 1) * span;
            int32_t u_hi = u_lo + span < tl + 1? u_lo + span : tl + 1;
            if (u_lo < u_hi) {
                if (u_hi == tl + 1) { if (lane == 0) t0[tl] = m; u_hi = tl; }
                if (u_lo < u_hi) {
                    const int32_t want = u_hi - 1 + m + cut + 2, p_top = want < tl - 1? want : tl - 1;
                    ecb_table_range(ts, tl, cs, s0, m, 0, p_top, u_lo, u_hi, t0);
                }
            }
        }
    }
}

// test entry (include/oatk_hip_ec.h: oatk_hip_debug_tables): one wave per job, both tables behind each other: out[out_off[j] ..] = table 0 [0 .. tl], table 1 [0 .. tl]
template <int NW>
__global__ __launch_bounds__(64 * NW) void ecb_tables_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off, const int32_t *ql,
                                                             int32_t *out, const uint64_t *out_off, int32_t cap_words, int32_t cut)
{
    extern __shared__ uint32_t ecb_lds2[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x;
    uint32_t *ts = ecb_lds2, *cs = ecb_lds2 + cap_words;
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];
    for (uint64_t i = t; i < nt; i += 64 * NW) ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += 64 * NW) cs[i] = qw[qw_off[j] + i];
    if constexpr (NW == 1) ecw_sync(); else __syncthreads();
    ecb_tables_wg<NW>(ts, tl[j], cs, 0, ql[j], cut, out + out_off[j], out + out_off[j] + (uint64_t) tl[j] + 1);
}

}  // namespace oatk
