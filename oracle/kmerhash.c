/*
 * oracle/kmerhash.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * 64-bit hash of an oriented k-mer taken out of the 2-bit hoco string
 * (reference syncmer.c:175-226 `kmer_hash64`, :131-170 `MurmurHash64A`).
 * The reference copies the covering bytes, reverse-complements them bytewise
 * through a 256-entry table, shifts the whole buffer left to byte-align the
 * first base and masks the tail; the net effect -- restated here base by base
 * -- is: write the K bases of the oriented k-mer (reverse complement of
 * hoco[pos, pos+K) when rev = 1) MSB-first, four per byte, zero-pad the last
 * byte, and hash those (K-1)/4+1 bytes with MurmurHash64A, seed 1234 (:129).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static inline uint8_t base_at(const uint8_t *hoco_s, uint32_t i)
{
    return (hoco_s[i >> 2] >> (((i & 3) ^ 3) << 1)) & 3;
}

void orc_kmer_pack(const uint8_t *hoco_s, uint32_t pos, uint32_t rev, int K, uint8_t *out)
{
    int i, nb = (K - 1) / 4 + 1;
    memset(out, 0, nb);
    for (i = 0; i < K; ++i) {
        uint8_t c = rev? (3 ^ base_at(hoco_s, pos + K - 1 - i)) : base_at(hoco_s, pos + i);
        out[i >> 2] |= (uint8_t) (c << (((i & 3) ^ 3) << 1));
    }
}

/* Austin Appleby's MurmurHash64A: 8-byte little-endian blocks, then the tail bytes. syncmer.c:131-170 */
uint64_t orc_murmur64a(const void *key, uint32_t len, uint64_t seed)
{
    const uint64_t M = 0xc6a4a7935bd1e995ULL;
    const uint8_t *p = (const uint8_t *) key;
    uint64_t h = seed ^ ((uint64_t) len * M);
    uint32_t nblk = len >> 3, i, t;
    for (i = 0; i < nblk; ++i) {
        uint64_t w = 0;
        for (t = 0; t < 8; ++t) w |= (uint64_t) p[8 * i + t] << (8 * t);
        w *= M; w ^= w >> 47; w *= M;
        h ^= w; h *= M;
    }
    uint32_t rem = len & 7;
    if (rem) {
        uint64_t w = 0;
        for (t = 0; t < rem; ++t) w |= (uint64_t) p[8 * nblk + t] << (8 * t);
        h ^= w; h *= M;
    }
    h ^= h >> 47; h *= M; h ^= h >> 47;
    return h;
}

uint64_t orc_kmer_hash(const uint8_t *hoco_s, uint32_t pos, uint32_t rev, int K)
{
    uint32_t nb = (uint32_t) (K - 1) / 4 + 1;
    uint8_t *buf = (uint8_t *) malloc(nb);
    orc_kmer_pack(hoco_s, pos, rev, K, buf);
    uint64_t h = orc_murmur64a(buf, nb, 1234);
    free(buf);
    return h;
}
