// oatk_amd/csrc/consensus.hpp -- base-space consensus of syncmers on the device (the part that costs).
//
// scg_syncmer_consensus (syncasm.c:888-1003) expands the hoco k-mer of a syncmer back to base space: at every hoco position it
// repeats the base 1 + lround(mean run length) times, the mean taken over all occurrences of the syncmer that were not
// error-corrected (:949-1001).  On the host that is sum(coverage) x K byte reads per call of scg_consensus, four calls per
// assembly -- the second largest share of the reference's CPU time.  Here the totals of EVERY live syncmer are taken once:
//
//   one workgroup per syncmer; its four waves split the occurrence list in batches of 64; a lane first fetches the metadata of one
//   occurrence (chain entry -> read, position, strand), then the wave walks the batch with the addresses broadcast by v_readlane
//   and every lane adds the run lengths of its 16 positions (t = lane + 64 q; coalesced 64-byte reads of ho_rl, ascending or,
//   for a reverse occurrence, descending); run lengths behind the 255 escape are looked up in the sorted long-run list.
//
// Only the FORWARD orientation is stored: index t of a reverse request is index K-1-t (the CPU restatement used by the tests checks that against the
// reference).  The string itself -- bases of the first uncorrected occurrence, 'N' padding for a negative `beg` -- is cheap host
// work (liboatk_host: oatk_scg_syncmer_consensus).
#pragma once
#include "common.hpp"

namespace oatk {

struct ConsArgs {
    uint64_t n_sel;
    const uint32_t *sel;          // syncmer ids
    const uint64_t *occ_off, *occ;  // occurrence lists: sid << 32 | idx << 1 | rev
    uint64_t sid0;
    const uint64_t *chain_off;    // [n_reads + 1] slots of the per-read chains
    const uint64_t *k_mer;        // id << 1 | corrected
    const uint32_t *m_pos;
    const uint64_t *off;          // packed-stream offsets = offsets of ho_rl
    const uint8_t *ho_rl;
    const uint64_t *lrl_key;      // sid << 32 | pos, ascending
    const uint32_t *lrl_val;
    uint64_t n_lrl;
    int K;
    unsigned long long *cons_tot; // [n_sel * K] total run length per forward position (what shards add up)
    uint32_t *cons_rl;            // [n_sel * K] lround(mean run length) per forward position
    uint32_t *m_seq;              // [n_sel] occurrences that took part
    uint64_t *first_occ;          // [n_sel] the first of them, ~0 if none
    // (round 6) a syncmer's occurrence list is walked in chunks of CONS_CHUNK, a workgroup each: ch_off[s] = the first workgroup of syncmer s (n_sel + 1 entries)
    const uint64_t *ch_off;
    uint64_t n_wg;
};

__device__ __forceinline__ uint32_t cons_long_run(const ConsArgs &a, uint64_t key)
{
    uint64_t lo = 0, hi = a.n_lrl;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a.lrl_key[mid] < key) lo = mid + 1; else hi = mid; }
    return lo < a.n_lrl && a.lrl_key[lo] == key? a.lrl_val[lo] : 255u;
}

#define CONS_Q 16                 // positions per lane per pass: a pass covers 1024 positions
// Occurrences a workgroup walks.  (Until round 6 a workgroup walked its syncmer's whole list: a syncmer inside a tandem array of the config-1 surrogate occurs 10^5 times --
// several times per read -- and ONE workgroup spent 131 ms on it, 6 % of that read set's CLI run, while the device idled: tools/prof_cli_config1s.sh.)
#define CONS_CHUNK 1024

__global__ void cons_chunks_kernel(uint64_t n_sel, const uint32_t *sel, const uint64_t *occ_off, uint64_t *nch)
{
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_sel) return;
    if (s == n_sel) { nch[s] = 0; return; }
    const uint64_t n = occ_off[sel[s] + 1] - occ_off[sel[s]];
    nch[s] = n > CONS_CHUNK? (n + CONS_CHUNK - 1) / CONS_CHUNK : 1;
}

__global__ __launch_bounds__(256) void cons_rl_kernel(ConsArgs a)
{
    __shared__ unsigned long long tot[CONS_Q * 64];
    __shared__ uint32_t s_m;
    __shared__ unsigned long long s_first;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // which syncmer, which chunk of its list: the last s with ch_off[s] <= this workgroup
    uint64_t s;
    {
        uint64_t lo = 0, hi = a.n_sel;
        const uint64_t w = blockIdx.x;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.ch_off[mid] <= w) lo = mid; else hi = mid; }
        s = lo;
    }
    const uint64_t chunk = (uint64_t) blockIdx.x - a.ch_off[s];
    const bool shared_row = a.ch_off[s + 1] - a.ch_off[s] > 1;                  // several workgroups add into this syncmer's row (zeroed before the launch)
    const uint32_t id = a.sel[s];
    const uint64_t o0 = a.occ_off[id], n_all = a.occ_off[id + 1] - o0;
    const uint64_t c_lo = shared_row? chunk * CONS_CHUNK : 0, n = shared_row? (c_lo + CONS_CHUNK < n_all? c_lo + CONS_CHUNK : n_all) : n_all;
    const int K = a.K;
    for (int t0 = 0; t0 < K; t0 += CONS_Q * 64) {
        for (uint32_t i = tid; i < CONS_Q * 64; i += 256) tot[i] = 0;
        if (tid == 0) s_m = 0, s_first = ~0ULL;
        __syncthreads();
        uint32_t acc[CONS_Q];
#pragma unroll
        for (int q = 0; q < CONS_Q; ++q) acc[q] = 0;
        uint32_t m_mine = 0;
        uint64_t first_mine = ~0ULL;
        for (uint64_t b0 = c_lo + (uint64_t) wid * 64; b0 < n; b0 += 256) {
            // metadata of one occurrence per lane
            bool valid = b0 + lane < n;
            uint64_t addr = 0, key0 = 0;
            uint32_t r = 0;
            if (valid) {
                const uint64_t o = a.occ[o0 + b0 + lane], rd = (o >> 32) - a.sid0, at = a.chain_off[rd] + ((o >> 1) & 0x7FFFFFFFULL);
                if (a.k_mer[at] & 1ULL) valid = false;             // error-corrected entries carry no position (syncasm.c:958-959)
                else {
                    const uint32_t mp = a.m_pos[at];
                    r = mp & 1u;
                    addr = a.off[rd] + (mp >> 1);
                    key0 = (o >> 32) << 32 | (uint64_t) (mp >> 1);
                    if (first_mine == ~0ULL) first_mine = b0 + lane;
                }
            }
            uint64_t live = __ballot(valid);
            m_mine += (uint32_t) __builtin_popcountll(live);
            while (live) {
                const int i = __builtin_ctzll(live);
                live &= live - 1;
                const uint64_t ai = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (addr >> 32), i) << 32 | (uint32_t) __builtin_amdgcn_readlane((int) addr, i);
                const uint32_t ri = (uint32_t) __builtin_amdgcn_readlane((int) r, i);
                // all sixteen loads first (a test of each value right behind its load would serialise sixteen round trips)
                uint32_t v[CONS_Q];
                bool esc = false;
#pragma unroll
                for (int q = 0; q < CONS_Q; ++q) {
                    const int t = t0 + (int) lane + 64 * q;
                    v[q] = t < K? (uint32_t) a.ho_rl[ai + (uint64_t) (ri? K - 1 - t : t)] : 0u;
                }
#pragma unroll
                for (int q = 0; q < CONS_Q; ++q) esc |= v[q] == 255u;
                if (__any(esc)) {                                    // run lengths that sit in the long-run list (syncmer.c: ho_l_rl)
                    const uint64_t ki = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (key0 >> 32), i) << 32
                                      | (uint32_t) __builtin_amdgcn_readlane((int) key0, i);
#pragma unroll
                    for (int q = 0; q < CONS_Q; ++q) {
                        const int t = t0 + (int) lane + 64 * q;
                        if (v[q] == 255u) v[q] = cons_long_run(a, ki + (uint64_t) (ri? K - 1 - t : t));
                    }
                }
#pragma unroll
                for (int q = 0; q < CONS_Q; ++q) acc[q] += v[q];
            }
        }
#pragma unroll
        for (int q = 0; q < CONS_Q; ++q) if (acc[q]) atomicAdd(&tot[lane + 64 * q], (unsigned long long) acc[q]);
        if (lane == 0 && t0 == 0) atomicAdd(&s_m, m_mine);
        {   // the first occurrence in list order that took part
            unsigned long long f = first_mine;
            for (int d = 32; d; d >>= 1) { const unsigned long long g = __shfl_xor(f, d); f = g < f? g : f; }
            if (lane == 0 && t0 == 0 && f != ~0ULL) atomicMin(&s_first, f);
        }
        __syncthreads();
        const uint32_t m = s_m;
        if (shared_row) {                                  // totals, count and the first occurrence's INDEX are added up over the chunks; cons_finish_kernel does the rest
            for (uint32_t i = tid; i < CONS_Q * 64 && t0 + (int) i < K; i += 256)
                if (tot[i]) atomicAdd(&a.cons_tot[s * (uint64_t) K + (uint64_t) (t0 + (int) i)], tot[i]);
            if (tid == 0 && t0 == 0) {
                if (m) atomicAdd(&a.m_seq[s], m);
                if (s_first != ~0ULL) atomicMin((unsigned long long *) &a.first_occ[s], s_first);
            }
        } else {
            for (uint32_t i = tid; i < CONS_Q * 64 && t0 + (int) i < K; i += 256) {
                a.cons_rl[s * (uint64_t) K + (uint64_t) (t0 + (int) i)] = m? (uint32_t) lround((double) tot[i] / (double) m) : 0u;
                a.cons_tot[s * (uint64_t) K + (uint64_t) (t0 + (int) i)] = tot[i];
            }
            if (tid == 0 && t0 == 0) {
                a.m_seq[s] = m;
                a.first_occ[s] = s_first == ~0ULL? ~0ULL : a.occ[o0 + s_first];
            }
        }
        __syncthreads();
    }
}

// the rows several workgroups add into: zeroed before, finished after (a workgroup per such syncmer)
__global__ __launch_bounds__(256) void cons_zero_shared_kernel(ConsArgs a)
{
    const uint64_t s = blockIdx.x;
    if (a.ch_off[s + 1] - a.ch_off[s] <= 1) return;
    for (int i = (int) threadIdx.x; i < a.K; i += 256) a.cons_tot[s * (uint64_t) a.K + (uint64_t) i] = 0;
    if (threadIdx.x == 0) a.m_seq[s] = 0, a.first_occ[s] = ~0ULL;
}
__global__ __launch_bounds__(256) void cons_finish_shared_kernel(ConsArgs a)
{
    const uint64_t s = blockIdx.x;
    if (a.ch_off[s + 1] - a.ch_off[s] <= 1) return;
    const uint32_t m = a.m_seq[s];
    for (int i = (int) threadIdx.x; i < a.K; i += 256)
        a.cons_rl[s * (uint64_t) a.K + (uint64_t) i] = m? (uint32_t) lround((double) a.cons_tot[s * (uint64_t) a.K + (uint64_t) i] / (double) m) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) { const uint64_t f = a.first_occ[s]; a.first_occ[s] = f == ~0ULL? ~0ULL : a.occ[a.occ_off[a.sel[s]] + f]; }
}

__global__ void cons_flag_kernel(uint64_t n, const uint32_t *cov, const uint8_t *del, uint32_t min_cov, uint32_t *flag)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (!del || !del[i]) && cov[i] >= min_cov && cov[i] > 0;
}
__global__ void cons_select_kernel(uint64_t n, const uint32_t *flag, const uint64_t *slot_of, uint32_t *sel, uint32_t *slot)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) sel[slot_of[i]] = (uint32_t) i, slot[i] = (uint32_t) slot_of[i];
    else slot[i] = 0xFFFFFFFFu;
}

}  // namespace oatk
