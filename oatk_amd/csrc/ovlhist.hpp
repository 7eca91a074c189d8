// oatk_amd/csrc/ovlhist.hpp -- what calc_syncmer_overlap (syncasm.c:477-582) tabulates, for every pair of adjacent syncmers at once.
//
// The reference answers "how far apart are syncmers m1 and m2 on the reads" by walking the occurrence list of m1, finding m2 next to
// it on the same read and counting the distances in a khashl<int,int>; scg_consensus asks this for every pair of neighbours inside
// every unitig and for every arc, four times per assembly.  All answers are already sitting in the sorted (key, distance) list the EC
// graph is built from (ecgraph.hpp): a run of equal keys is the multiset of one pair.  This file reduces each run to the distinct
// distances IN THE ORDER OF THEIR FIRST APPEARANCE with their counts -- which is all a khashl table's final layout depends on -- plus
// one bit: whether the last add_ovl_count call of the run was a repeat (khashl grows at the call AFTER the insert that filled the
// table, khashl.h:199, so a trailing repeat can still change the bucket order).  Pairs with an error-corrected member do not take part
// (syncasm.c:499, :511).
#pragma once
#include "common.hpp"
#include "ecgraph.hpp"

namespace oatk {

#define OVH_MAX 64        // distinct distances per pair the kernel keeps in LDS; a pair with more (an arc across a tandem array at thousand-fold coverage)
                          // continues in global memory (round 4; OATK_E_SPLIT until then)
#define OVH_SPILL (128ull << 20)

// one wave per run of equal keys.  n_out == nullptr: write the entries at out_off[run]; otherwise only count them.
// W: the run is made of weighted segments -- sval[i] = distance | calls << 32 stands for `calls` consecutive add_ovl_count calls with that distance
// (what shards exchange, include/oatk_hip_multi.h); sdist is then unused.  Order of first appearance, totals and the tail rule are those of the
// expanded list: the last call is a repeat iff the last segment's distance was counted more than once in all.
template <bool W>
__global__ __launch_bounds__(64) void ovh_kernel_t(uint64_t n_runs, const uint64_t *ukeys, const uint32_t *counts, const uint64_t *run_off,
                                                   const uint32_t *sdist, const uint64_t *sval, uint32_t *n_out, const uint64_t *out_off, int32_t *o_dist,
                                                   uint32_t *o_cnt, uint8_t *o_tail, uint32_t *flags, const uint32_t *nd_in, uint8_t *spill, uint64_t spill_bytes)
{
    __shared__ int32_t hk_lds[OVH_MAX];
    __shared__ uint32_t hc_lds[OVH_MAX];
    __shared__ uint64_t s_at;
    const int lane = threadIdx.x;
    const uint64_t run = blockIdx.x;
    if (run >= n_runs) return;
    if (ukeys[run] == EGR_INVALID) { if (n_out && lane == 0) n_out[run] = 0; return; }
    const uint32_t cc = counts[run];
    const uint64_t oo = run_off[run];
    uint32_t nd = 0;
    bool overflow = false;
    // the table: LDS for up to OVH_MAX distances; beyond that the counting pass moves to a piece of `spill` (room for one entry per call, handed out
    // with flags[6] as the bump pointer), and the writing pass -- which knows the final number from the counting pass -- works in its output slots
    int32_t *hk = hk_lds;
    uint32_t *hc = hc_lds;
    uint32_t cap = OVH_MAX;
    const bool in_place = !n_out && nd_in[run] > OVH_MAX;
    if (in_place) hk = o_dist + out_off[run], hc = o_cnt + out_off[run], cap = nd_in[run];
    for (uint32_t t0 = 0; t0 < cc && !overflow; t0 += 64) {
        const bool in = t0 + lane < cc;
        const uint64_t sv = W && in? sval[oo + t0 + lane] : 0ULL;
        const int32_t d = W? (int32_t) (uint32_t) sv : (in? (int32_t) sdist[oo + t0 + lane] : 0);
        const uint32_t wt = W? (uint32_t) (sv >> 32) : 1u;
        auto total = [&](uint64_t mask) -> uint32_t {                    // calls behind the lanes of `mask`
            if (!W) return (uint32_t) __builtin_popcountll(mask);
            uint32_t v = (mask >> lane & 1ULL)? wt : 0u;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            return v;
        };
        uint64_t rest = __ballot(in);
        for (uint32_t j = 0; j < nd && rest; ++j) {                      // distances seen in earlier chunks
            const uint64_t eq = __ballot(in && d == hk[j]) & rest;
            if (eq) { const uint32_t t = total(eq); if (lane == 0) hc[j] += t; }
            rest &= ~eq;
        }
        while (rest) {                                                   // new ones, in the order of their first appearance
            const int f = __builtin_ctzll(rest);
            const int32_t x = __builtin_amdgcn_readfirstlane(__shfl(d, f));
            const uint64_t eq = __ballot(in && d == x) & rest;
            rest &= ~eq;
            if (nd == cap) {
                if (!n_out || cap != OVH_MAX) { overflow = true; break; }           // (cannot happen: the table has room for every call)
                if (lane == 0) s_at = (uint64_t) atomicAdd(&flags[6], (cc * 8u + 15u) >> 4) * 16ull;
                __syncthreads();
                const uint64_t at = s_at;
                if (cc >= (1u << 28) || at + (uint64_t) cc * 8 > spill_bytes) { overflow = true; break; }
                int32_t *gk = (int32_t *) (spill + at);
                uint32_t *gc = (uint32_t *) (gk + cc);
                gk[lane] = hk_lds[lane], gc[lane] = hc_lds[lane];                    // OVH_MAX == 64 lanes
                hk = gk, hc = gc, cap = cc;
                __syncthreads();
            }
            const uint32_t t = total(eq);
            if (lane == 0) hk[nd] = x, hc[nd] = t;
            ++nd;
        }
        __syncthreads();                                                 // lane 0's LDS writes before the next chunk's reads
    }
    __syncthreads();
    if (overflow) { if (lane == 0) flags[1] = 1u; nd = 0; }
    if (n_out) { if (lane == 0) n_out[run] = nd; return; }
    const uint64_t w0 = out_off[run];
    if (!in_place && (uint32_t) lane < nd) o_dist[w0 + lane] = hk[lane], o_cnt[w0 + lane] = hc[lane];
    if (lane == 0 && nd) {
        // the last call was a repeat unless the last distance of the run occurs exactly once (then the call inserted it)
        const int32_t last = W? (int32_t) (uint32_t) sval[oo + cc - 1] : (int32_t) sdist[oo + cc - 1];
        uint32_t c = 0;
        for (uint32_t j = 0; j < nd; ++j) if (hk[j] == last) c = hc[j];
        o_tail[run] = c > 1;
    }
}

} // namespace oatk
