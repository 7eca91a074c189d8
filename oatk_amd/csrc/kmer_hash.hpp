// oatk_amd/csrc/kmer_hash.hpp -- MurmurHash64A of oriented k-mers, one LANE per syncmer.
//
// Replaces `kmer_hash64` (syncmer.c:175-226) for all syncmer records of a batch at once.  Inside the scan kernel the
// hash is poison for a wave64 machine: a 251-byte k-mer is a 31-step dependent chain that keeps one lane busy while
// 63 idle.  Here a wave owns 64 records: first all 64 lanes cooperatively fetch and pre-mix the 8-byte Murmur blocks
// (coalesced reads of each record's contiguous hoco bytes, results parked in LDS), then every lane runs the dependent
// chain of ITS record, so the chain phase issues with all 64 lanes active.
#pragma once
#include "common.hpp"
#include "count.hpp"   // kmer_word_global

namespace oatk {

struct KmerHashArgs {
    const uint8_t *hoco_s;        // read r at off[r] / 4
    const uint64_t *off;
    uint64_t sid0;
    const uint64_t *rec_lo;       // sid << 32 | ordinal << 1 | rev
    const uint32_t *rec_mpos;     // pos << 1 | rev
    uint64_t *rec_hash;           // out
    uint32_t n_rec;
    int K;
};

#define KMH_REC 16                // records per wave: most lanes idle in the short chain phase, but four times the waves fit a CU's LDS (64: 1.12 ms, 32: 0.75, 16: 0.56, 8: 0.67)

__global__ __launch_bounds__(64) void kmer_hash_kernel(KmerHashArgs a)
{
    extern __shared__ uint64_t kmix[];          // KMH_REC records x (NW + 1)
    const uint32_t lane = threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t) KMH_REC;
    const int K = a.K;
    const int nbytes = (K - 1) / 4 + 1, nfull = nbytes >> 3, nrem = nbytes & 7, NW = nfull + (nrem? 1 : 0);
    const int stride = NW + 1;
    const uint32_t nrec = a.n_rec - base < (uint32_t) KMH_REC? a.n_rec - base : (uint32_t) KMH_REC;
    const uint32_t items = nrec * (uint32_t) NW;
    // where record `lane` lives: fetched once, handed to the lanes that read its words by ds_bpermute (three dependent gathers
    // in front of every word otherwise)
    uint32_t my_mp = 0, my_hsw = 0;
    if (lane < nrec) {
        my_mp = a.rec_mpos[base + lane];
        my_hsw = (uint32_t) (a.off[(a.rec_lo[base + lane] >> 32) - a.sid0] >> 4);      // 32-bit word index of the read's hoco string
    }
    const uint32_t *hs32 = (const uint32_t *) a.hoco_s;
    for (uint32_t it = lane; it < ((items + 63u) & ~63u); it += 64u) {
        const uint32_t rr0 = it / (uint32_t) NW, rr = rr0 < nrec? rr0 : 0u, wd = it - rr0 * (uint32_t) NW;
        const uint32_t mp = __shfl(my_mp, (int) rr);
        const uint32_t *hs = hs32 + __shfl(my_hsw, (int) rr);
        if (it >= items) continue;
        uint64_t word = bswap64(kmer_word_global(hs, mp >> 1, mp & 1u, K, (int) wd));
        kmix[rr * (uint32_t) stride + wd] = (int) wd < nfull? murmur_mix_word(word) : word;
    }
    __syncthreads();
    if (lane < nrec) {
        const uint64_t *km = &kmix[lane * (uint32_t) stride];
        uint64_t h = OATK_MURMUR_SEED ^ ((uint64_t) (uint32_t) nbytes * OATK_MURMUR_M);
        for (int wd = 0; wd < nfull; ++wd) h = (h ^ km[wd]) * OATK_MURMUR_M;
        if (nrem) h = (h ^ km[nfull]) * OATK_MURMUR_M;
        h ^= h >> 47; h *= OATK_MURMUR_M; h ^= h >> 47;
        a.rec_hash[base + lane] = h;
    }
}

}  // namespace oatk
