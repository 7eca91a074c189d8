// Throughput microbenchmark for the integer VALU instructions the scan kernels lean on (gfx950).
// Prints cycles per wave64 instruction per SIMD, measured with 4 waves/SIMD resident and 8 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 4096
#define REP8(x) x x x x x x x x

template <int OP>
__global__ void k(uint64_t *out, uint64_t seed)
{
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = (uint32_t) seed | 1u;
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) {        // v_mad_u64_u32
#define S(v) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"((uint32_t) v), "v"(b) : "vcc");
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 1) { // v_lshlrev_b64
#define S(v) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(v));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 2) { // v_mul_lo_u32
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 3) { // v_lshl_add_u64
#define S(v) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(v));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 4) { // v_add_co_u32 + v_addc_co_u32 (64-bit add as two 32-bit ops)
#define S(v) { uint32_t lo = (uint32_t) v, hi = (uint32_t) (v >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(b) : "vcc"); v = (uint64_t) hi << 32 | lo; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 5) { // v_alignbit_b32
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 6) { // v_xor_b32 (baseline full-rate op)
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 7) { // v_cmp_lt_u64 + v_cndmask (64-bit min)
#define S(v) { uint32_t lo = (uint32_t) v; asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(lo) : "v"(v), "v"(a7), "v"(b) : "vcc"); v = (v & 0xFFFFFFFF00000000ULL) | lo; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6)
#undef S
        } else if (OP == 8) { // v_lshrrev_b64
#define S(v) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(v));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 9) { // v_mul_hi_u32
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 10) { // v_add_u32
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (OP == 11) { // v_mul_u32_u24
#define S(v) { uint32_t x = (uint32_t) v; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); v = x; }
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint64_t) (t1 - t0);
}

template <int OP>
static void run(const char *name, int per_iter)
{
    uint64_t *d;
    hipMalloc(&d, ((1 << 20) + 8) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, threads = 256;   // 16 waves per CU = 4 per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345ULL);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345ULL);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t cyc; hipMemcpy(&cyc, d + (1 << 20), 8, hipMemcpyDeviceToHost);
    // 4 waves per SIMD share the SIMD: cycles per instruction per SIMD = wall cycles / (instr per wave * 4)
    double instr_per_wave = (double) ITER * per_iter;
    printf("%-28s wall %.3f ms  s_memtime cycles/instr/SIMD %.2f  (event-derived @2.4GHz %.2f)\n", name, ms,
           (double) cyc / (instr_per_wave * 4.0), ms * 1e-3 * 2.4e9 / (instr_per_wave * 4.0));
    hipFree(d);
}

int main()
{
    run<6>("v_xor_b32", 8);
    run<10>("v_add_u32", 8);
    run<5>("v_alignbit_b32", 8);
    run<11>("v_mul_u32_u24", 8);
    run<2>("v_mul_lo_u32", 8);
    run<9>("v_mul_hi_u32", 8);
    run<0>("v_mad_u64_u32", 8);
    run<1>("v_lshlrev_b64", 8);
    run<8>("v_lshrrev_b64", 8);
    run<3>("v_lshl_add_u64", 8);
    run<4>("v_add_co+v_addc (pair)", 8);
    run<7>("v_cmp_lt_u64+cndmask (pair)", 7);
    return 0;
}
