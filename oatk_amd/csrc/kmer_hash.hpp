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
#define KMH_WPL 8                 // Murmur blocks a lane prepares in one go (their source words are loaded together; four: 3.68 against 2.95 ms, r04)

// r03: FOUR LANES PER RECORD, each preparing a run of consecutive 8-byte Murmur blocks of its record.  A run of blocks is a run of consecutive source
// words, so a lane loads 2 n + 1 dwords for n blocks (three per block before) with no division, no shuffles and no per-block address arithmetic in
// front of them (the old loop spent ~100 instructions per block: `it / NW` at run time, two ds_bpermute, three guarded loads, three byte swaps, a
// 64-bit funnel shift with a special case, revcomp32 and a second byte swap).  The Murmur block of a REVERSE occurrence needs no reversal of the
// word at all: revcomp32 followed by the byte swap that turns the MSB-first string into the little-endian word Murmur reads is the complement with
// the four fields of every BYTE reversed, bytes staying where they are.
__device__ __forceinline__ uint32_t kmh_rev_in_bytes(uint32_t x)
{
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    return ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
}

#ifndef OATK_KMH_WAVES
#define OATK_KMH_WAVES 8                 // waves per SIMD the register allocation aims at: the kernel is a chain of dependent gathers, its rate is records in flight (r03m, 400 k reads: 0.94 ms at five waves, 0.60 at eight)
#endif
__global__ __launch_bounds__(64, OATK_KMH_WAVES) void kmer_hash_kernel(KmerHashArgs a)
{
    extern __shared__ uint64_t kmix[];          // KMH_REC records x (NW + 1)
    const uint32_t lane = threadIdx.x, rr = lane >> 2, q = lane & 3u;
    const uint32_t base = blockIdx.x * (uint32_t) KMH_REC;
    const int K = a.K;
    const int nbytes = (K - 1) / 4 + 1, nfull = nbytes >> 3, nrem = nbytes & 7, NW = nfull + (nrem? 1 : 0);
    const int stride = NW + 1;
    const uint32_t nrec = a.n_rec - base < (uint32_t) KMH_REC? a.n_rec - base : (uint32_t) KMH_REC;
    if (rr < nrec) {
        const uint32_t mp = a.rec_mpos[base + rr], rev = mp & 1u, pos = mp >> 1;      // (four lanes, one address)
        const int64_t hsw = (int64_t) (a.off[(a.rec_lo[base + rr] >> 32) - a.sid0] >> 4);      // 32-bit word index of the read's hoco string
        const uint32_t *hs32 = (const uint32_t *) a.hoco_s;
        uint64_t *out = &kmix[rr * (uint32_t) stride];
        const int per = (NW + 3) >> 2;                                        // blocks per lane
        for (int w0 = (int) q * per; w0 < (int) (q + 1) * per && w0 < NW; w0 += KMH_WPL) {
            const int nwd = (int) (q + 1) * per - w0 < KMH_WPL? (int) (q + 1) * per - w0 : KMH_WPL;
            const int n = w0 + nwd > NW? NW - w0 : nwd;                       // blocks w0 .. w0 + n - 1
            // their source bases: [t_lo, t_lo + 32 n) of the hoco string -- the oriented bases 32 w0 .. for a forward occurrence, the mirror image
            // (block w0 + j from the chunk n - 1 - j) for a reverse one; bases in front of the k-mer (a reverse occurrence's last block) are masked below
            const int32_t t_lo = rev? (int32_t) pos + K - 32 * (w0 + n) : (int32_t) pos + 32 * w0;
            const int32_t wi = t_lo >> 4;                                     // (arithmetic shift: t_lo may be negative by less than 32)
            const uint32_t sh = ((uint32_t) t_lo & 15u) * 2u;
            // the 2 n + 1 source words, in the order the blocks use them: ascending for a forward occurrence, descending for a reverse one (whose
            // block j is made of chunk n - 1 - j) -- so block j always finds its three words at d[2j], d[2j + 1], d[2j + 2].  A full run (n = 8, every
            // lane at K = 1001) fetches them as four 16-byte loads and one dword: seventeen single-dword loads touch 64 cache lines EACH (the lanes of
            // a wave sit 64 bytes apart) and the vector cache looks up every line of every load -- that, not arithmetic, was the kernel's run time.
            // (r04: the words stay in STRING order in their registers and the loop below goes over the CHUNKS c; which block a chunk makes -- j = c, or n - 1 - c for a
            //  reverse occurrence -- only moves the LDS address it is written to.  Until then a second array held them in block order, seventeen selects put them
            //  there, and nine of its words lived in scratch: 5.6 GB of scratch writes per step at config 3 for 0.34 GB of hashes.)
            uint32_t asc[2 * KMH_WPL + 1];
            const int64_t gbase = hsw + wi;
            if (n == KMH_WPL && gbase >= 0) {
                struct __attribute__((packed, aligned(4))) W4 { uint32_t a, b, c, e; };
                const W4 *src = (const W4 *) (hs32 + gbase);
                const W4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                const uint32_t v4 = hs32[gbase + 16];
                const uint32_t raw[17] = {v0.a, v0.b, v0.c, v0.e, v1.a, v1.b, v1.c, v1.e, v2.a, v2.b, v2.c, v2.e, v3.a, v3.b, v3.c, v3.e, v4};
#pragma unroll
                for (int i = 0; i < 17; ++i) asc[i] = __builtin_bswap32(raw[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 2 * KMH_WPL + 1; ++i) {
                    int64_t gi = gbase + i;
                    gi = gi < 0? 0 : gi;                                      // (only the very first read of a slab: what is read there is masked)
                    asc[i] = i < 2 * n + 1? __builtin_bswap32(hs32[gi]) : 0u;
                }
            }
#pragma unroll
            for (int c = 0; c < KMH_WPL; ++c) {
                if (c >= n) break;
                // sixty-four bits from bit offset sh of the chunk's three words (first, middle, last in string order)
                const uint32_t x0 = asc[2 * c], x1 = asc[2 * c + 1], x2 = asc[2 * c + 2];
                const uint32_t hi = (uint32_t) (((uint64_t) x0 << 32 | x1) >> (32u - sh));
                const uint32_t lo = (uint32_t) (((uint64_t) x1 << 32 | x2) >> (32u - sh));
                const int j = rev? n - 1 - c : c;
                const int wd = w0 + j;
                int nb = K - 32 * wd;
                nb = nb > 32? 32 : nb;
                uint64_t word;
                if (rev) {
                    // the oriented block is revcomp32 of the chunk; bases of the chunk in FRONT of the k-mer (its first 32 - nb) fall behind K
                    uint64_t V = (uint64_t) hi << 32 | lo;
                    if (nb < 32) V &= ~0ULL >> (64 - 2 * nb);                 // keep the chunk's LAST nb bases
                    const uint64_t R = revcomp32(V);                          // (rare tail aside, the two lines below are all a reverse block costs)
                    word = nb < 32? bswap64(R & (~0ULL << (64 - 2 * nb)))
                                  : ~((uint64_t) kmh_rev_in_bytes(hi) << 32 | kmh_rev_in_bytes(lo));
                } else {
                    uint64_t V = (uint64_t) hi << 32 | lo;
                    if (nb < 32) V &= ~0ULL << (64 - 2 * nb);
                    word = bswap64(V);
                }
                out[wd] = wd < nfull? murmur_mix_word(word) : word;
            }
        }
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
