// Host check of hash64_s31_left (oatk_amd/csrc/common.hpp): the 62-bit hash64 of khash / syncmer.c computed on the LEFT-aligned key equals hash64 on the
// right-aligned one, for 2 x 10^8 random keys and the corner values, with every pair of junk bits below the key.   g++ -O2 hash_left_check.cpp && ./a.out
#include <cstdint>
#include <cstdio>
#include <random>
static inline uint32_t alignbit(uint32_t hi, uint32_t lo, int s) { return (uint32_t) ((((uint64_t) hi << 32) | lo) >> s); }
static uint64_t hash64(uint64_t x, uint64_t mask)
{
    x = (~x + (x << 21)) & mask; x ^= x >> 24; x = (x + (x << 3) + (x << 8)) & mask; x ^= x >> 14;
    x = (x + (x << 2) + (x << 4)) & mask; x ^= x >> 28; x = (x + (x << 31)) & mask; return x;
}
// K-form: the 62-bit key LEFT-aligned (K = key << 2); returns K7 = hash << 2
static uint64_t hash64_s31_left(uint64_t cn)
{
    uint32_t lo = (uint32_t) cn & ~3u, hi = (uint32_t) (cn >> 32);
    uint64_t p;
    p = (uint64_t) lo * 0x1FFFFFu + 0xFFFFFFFFFFFFFFFCull;
    hi = (uint32_t) (p >> 32) + hi * 0x1FFFFFu, lo = (uint32_t) p;
    { const uint32_t t = alignbit(hi, lo, 24) & ~3u, u = hi >> 24; lo ^= t, hi ^= u; }
    p = (uint64_t) lo * 265u;
    hi = hi * 265u + (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = alignbit(hi, lo, 14) & ~3u, u = hi >> 14; lo ^= t, hi ^= u; }
    p = ((uint64_t) hi << 32 | lo) * 21u;
    hi = (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = alignbit(hi, lo, 28) & ~3u, u = hi >> 28; lo ^= t, hi ^= u; }
    hi += alignbit(hi, lo, 1);
    return (uint64_t) hi << 32 | lo;
}
int main()
{
    const uint64_t mask = (1ULL << 62) - 1;
    std::mt19937_64 rng(12345);
    uint64_t bad = 0, n = 0;
    auto one = [&](uint64_t key, uint32_t junk) {
        const uint64_t want = hash64(key & mask, mask), got = hash64_s31_left((key & mask) << 2 | (junk & 3));
        if ((got >> 2) != want || (got & 3)) { if (bad < 5) printf("key %016lx want %016lx got %016lx\n", key & mask, want, got); ++bad; }
        ++n;
    };
    const uint64_t corners[] = {0, 1, 2, 3, mask, mask - 1, mask >> 1, 1ULL << 61, 1ULL << 32, 0xFFFFFFFFULL, 0x100000000ULL, 0x3FFFFFFF00000000ULL, 0x00000000FFFFFFFFULL};
    for (uint64_t c : corners) for (uint32_t j = 0; j < 4; ++j) one(c, j);
    for (uint64_t i = 0; i < 200000000ULL; ++i) { const uint64_t x = rng(); one(x, (uint32_t) (x >> 62)); }
    printf("%lu keys, %lu mismatches\n", n, bad);
    return bad != 0;
}
