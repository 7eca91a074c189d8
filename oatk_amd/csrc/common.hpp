// oatk_amd/csrc/common.hpp -- device helpers shared by the gfx950 kernels.
//
// Written for MI355X (gfx950, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OATK_WAVE 64

namespace oatk {

// Exact restatement of the reference's 256-entry base table (syncmer.c:47-64) as arithmetic:
// bytes 0..3 map to themselves, A/a C/c G/g T/t U/u -> 0 1 2 3 3, everything else -> 4.
__device__ __forceinline__ uint32_t nt4_code(uint32_t ch)
{
    uint32_t up = ch & 0xDFu;                 // fold case
    uint32_t t = up - 0x41u;                  // 'A' -> 0, 'C' -> 2, 'G' -> 6, 'T' -> 19, 'U' -> 20
    bool letter = t < 32u && ((0x00180045u >> t) & 1u);
    uint32_t code = (up >> 1) & 3u;           // A 0, C 1, G 3, T/U 2
    code ^= code >> 1;                        // -> A 0, C 1, G 2, T/U 3
    return ch < 4u ? ch : (letter ? code : 4u);
}

// Invertible 64-bit mix confined to 2S bits (syncmer.c:116-126).
__device__ __forceinline__ uint64_t hash64(uint64_t x, uint64_t mask)
{
    x = (~x + (x << 21)) & mask;
    x ^= x >> 24;
    x = (x + (x << 3) + (x << 8)) & mask;
    x ^= x >> 14;
    x = (x + (x << 2) + (x << 4)) & mask;
    x ^= x >> 28;
    x = (x + (x << 31)) & mask;
    return x;
}

// The same function for S = 31 (62 bits), written on 32-bit halves for gfx950: the low word's products as 32 x 32 -> 64, the high word's share
// added in 32 bits, and NO mask between the steps -- what a product leaves above bit 61 never comes down again, because the right shifts
// take the high word's bits through bit-field extracts (v_alignbit / v_bfe / v_bitop3).  23 instructions against 27 (r03h); equal to hash64 on
// 2 x 10^8 random arguments and the corner values on the host, and through every scan test on the device.
__device__ __forceinline__ uint64_t hash64_s31(uint64_t x)
{
    uint32_t lo = (uint32_t) x, hi = (uint32_t) (x >> 32);
    uint64_t p;
    p = (uint64_t) lo * 0x1FFFFFu + 0xFFFFFFFFFFFFFFFFull;              // ~x + (x << 21) = x (2^21 - 1) - 1
    hi = (uint32_t) (p >> 32) + (hi << 21) - hi, lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 24), u = __builtin_amdgcn_ubfe(hi, 24, 6); lo ^= t, hi ^= u; }
    p = (uint64_t) lo * 265u;
    hi = hi * 265u + (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 14), u = __builtin_amdgcn_ubfe(hi, 14, 16); lo ^= t, hi ^= u; }
    p = (uint64_t) hi << 32 | lo;                                       // x 21 = 16 x + (4 x + x): two v_lshl_add_u64 (r04; a 32 x 32 -> 64 product, a move and a second
    {                                                                   // product before -- every one of these instructions issues at the same 4-cycle rate;
        uint64_t p5;                                                    // written as shifts and adds the compiler turns it back into the products)
        asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(p5) : "v"(p));
        asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(p) : "v"(p), "v"(p5));
    }
    hi = (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 28), u = __builtin_amdgcn_ubfe(hi, 28, 2); lo ^= t, hi ^= u; }
    p = (uint64_t) lo * 0x80000001u;
    hi = (hi + (uint32_t) (p >> 32)) & 0x3FFFFFFFu, lo = (uint32_t) p;
    return (uint64_t) hi << 32 | lo;
}

// ... and with the 62-bit key LEFT-aligned: the argument is key << 2 (its two low bits may hold anything: kernel B passes the base that follows the s-mer),
// the result hash64 << 2.  Arithmetic modulo 2^62 on the key is arithmetic modulo 2^64 on four times the key, so no product needs a mask; a right shift takes
// two bits of the key's low end along, which an AND removes.  The last step, x += x << 31, adds nothing to the low word (bit 0 of it is zero) and
// (hi : lo) >> 1 to the high word.  Saves the 64-bit shift that right-aligns the key and two of the last step's three instructions (r04; checked on the host
// against hash64 on 2 x 10^8 random keys and the corner values with every pair of junk bits: tools/ubench/hash_left_check.cpp).
__device__ __forceinline__ uint64_t hash64_s31_left(uint64_t x4)
{
    uint32_t lo = (uint32_t) x4 & ~3u, hi = (uint32_t) (x4 >> 32);
    uint64_t p;
    p = (uint64_t) lo * 0x1FFFFFu + 0xFFFFFFFFFFFFFFFCull;              // 4 (~x + (x << 21)) = 4 x (2^21 - 1) - 4
    hi = (uint32_t) (p >> 32) + (hi << 21) - hi, lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 24) & ~3u, u = hi >> 24; lo ^= t, hi ^= u; }
    p = (uint64_t) lo * 265u;
    hi = hi * 265u + (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 14) & ~3u, u = hi >> 14; lo ^= t, hi ^= u; }
    p = (uint64_t) hi << 32 | lo;
    {
        uint64_t p5;
        asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(p5) : "v"(p));
        asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(p) : "v"(p), "v"(p5));
    }
    hi = (uint32_t) (p >> 32), lo = (uint32_t) p;
    { const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 28) & ~3u, u = hi >> 28; lo ^= t, hi ^= u; }
    hi += __builtin_amdgcn_alignbit(hi, lo, 1);
    return (uint64_t) hi << 32 | lo;
}

// Reverse the order of the 32 two-bit groups of x and complement each (3 ^ c).
__device__ __forceinline__ uint64_t revcomp32(uint64_t x)
{
    uint32_t lo = (uint32_t) x, hi = (uint32_t) (x >> 32);
    uint32_t rlo = __builtin_bitreverse32(hi), rhi = __builtin_bitreverse32(lo);   // full 64-bit bit reversal
    uint64_t r = (uint64_t) rhi << 32 | rlo;
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);  // restore bit order inside each pair
    return ~r;
}

__device__ __forceinline__ uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

// MurmurHash64A constants (syncmer.c:131-170), seed 1234 (syncmer.c:129)
#define OATK_MURMUR_M 0xc6a4a7935bd1e995ULL
#define OATK_MURMUR_SEED 1234ULL

__device__ __forceinline__ uint64_t murmur_mix_word(uint64_t w)
{
    w *= OATK_MURMUR_M;
    w ^= w >> 47;
    w *= OATK_MURMUR_M;
    return w;
}

// inclusive wave scan (sum) over 64 lanes
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < OATK_WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ int32_t wave_incl_max(int32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < OATK_WAVE; d <<= 1) {
        int32_t o = __shfl_up(v, d);
        if (lane >= d) v = v > o? v : o;
    }
    return v;
}

}  // namespace oatk
