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

#include "ec_heavy.hpp"

namespace oatk {

#define ECR_INF (1 << 24)
#define ECR_R 8

__device__ __forceinline__ uint32_t ecr_base(const uint32_t *w, int32_t p) { return (w[p >> 4] >> ((uint32_t) (p & 15) << 1)) & 3u; }

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

}  // namespace oatk
