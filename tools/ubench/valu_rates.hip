// Issue-rate microbenchmark for the integer VALU instructions the scan kernels lean on (gfx950), round 4 form.
//
// What it settles: how many SIMD cycles one wave64 VALU instruction occupies (MI355X_MICROARCH.md "Wave scheduling" says 2 for a
// plain 32-bit op on the SIMD-32; the round-3 form of this file measured 2.9 from kernels of 0.2-0.5 ms priced at an assumed 2.4 GHz).
// Here every opcode runs >= 20 ms (the iteration count is calibrated per opcode), at 1, 2, 4 and 8 waves per SIMD, and the clock is
// MEASURED: s_memtime ticks of the timed region (shader cycles) over the region's wall time from wall_clock64() (the 100 MHz
// constant counter), so "cycles per instruction per SIMD" never depends on an assumed frequency.  Two forms per opcode:
//   tput : 8 independent chains per wave -- the issue cost once dependencies are out of the way,
//   dep  : 1 chain, 1 wave per SIMD      -- the dependent-issue latency.
// and interleaves of two opcodes (v_mad_u64_u32 + v_xor_b32, v_cmp_lt_u64 + v_cndmask) to see whether a slow op hides a fast one.
//
// Output: one line per (opcode, waves/SIMD): SIMD cycles per wave-instruction = region cycles * 1 / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum { XOR, ADD, MIN, ALIGNBIT, PERM, CNDMASK, MUL24, MULLO, MULHI, MAD64, SHL64, SHR64, LSHLADD64, ADDC, CMP64SEL, CMP32SEL, DPPMIN, ANDOR, ADD3, XAD, LSHLOR,
       MAD64_XOR, MAD64_2XOR, CMP64_XOR,
       AND, OR, SHL32, SHR32, SUB, MOV, NOT, BFE, CNDS, CMP32, CMPEQ32, CMP64, MAX, MIN3, XOR64E, ADD64E, BITOP3, BFI, LSHLADD, ADDC1, ADDCO, FMA, FMAC, MULF, PKFMA, MOVDPP, XOR_MIN, XOR3_MAD, NOPS };

static const char *NAMES[] = {"v_xor_b32", "v_add_u32", "v_min_u32", "v_alignbit_b32", "v_perm_b32", "v_cndmask_b32 (vcc)", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32",
                              "v_mad_u64_u32", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_add_co+v_addc (2)", "v_cmp_lt_u64+2 cndmask (3)", "v_cmp_lt_u32+cndmask (2)",
                              "v_min_u32 dpp row_shr:1", "v_and_or_b32", "v_add3_u32", "v_xad_u32", "v_lshl_or_b32",
                              "v_mad_u64_u32 + v_xor (2)", "v_mad_u64_u32 + 2 v_xor (3)", "v_cmp_lt_u64 + v_xor (2)",
                              "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_mov_b32", "v_not_b32", "v_bfe_u32", "v_cndmask_b32 (sgpr pair)", "v_cmp_lt_u32 (vcc)", "v_cmp_eq_u32 (vcc)", "v_cmp_lt_u64 (vcc)", "v_max_u32", "v_min3_u32", "v_xor_b32_e64", "v_add_u32_e64", "v_bitop3_b32", "v_bfi_b32", "v_lshl_add_u32", "v_addc_co_u32 (vcc in, vcc out)", "v_add_co_u32 (vcc out)", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_pk_fma_f32", "v_mov_b32 dpp row_shr:1", "v_xor + v_min_u32 (2)", "3 v_xor + v_mad_u64_u32 (4)"};
static const int PER[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 3, 2, 1, 1, 1, 1, 1, 2, 3, 2,
                          1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 4};

template <int OP, int CH>
__device__ __forceinline__ void step(uint64_t &v, uint32_t &x, uint32_t b, uint64_t other)
{
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(x) : "v"(b));
    else if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(0x06070001u));
    else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : );
    else if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(x), "v"(b) : "vcc");
    else if (OP == SHL64) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(v));
    else if (OP == SHR64) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(v));
    else if (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(v));
    else if (OP == ADDC) { uint32_t lo = (uint32_t) v, hi = (uint32_t) (v >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(b) : "vcc"); v = (uint64_t) hi << 32 | lo; }
    else if (OP == CMP64SEL) { uint32_t lo = (uint32_t) v, hi = (uint32_t) (v >> 32); asm volatile("v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(lo), "+v"(hi) : "v"(v), "v"(other), "v"(b) : "vcc"); v = (uint64_t) hi << 32 | lo; }
    else if (OP == CMP32SEL) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
    else if (OP == DPPMIN) {          // (a DPP read needs two wait states behind the VALU write of its source: seven other chains provide them, a lone chain needs the s_nop)
        if (CH == 1) asm volatile("s_nop 1\n v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
        else asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    }
    else if (OP == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
    else if (OP == MAD64_XOR) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(b), "v"(b) : "vcc"); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); }
    else if (OP == MAD64_2XOR) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(b), "v"(b) : "vcc"); asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); }
    else if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == SHL32) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
    else if (OP == SHR32) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
    else if (OP == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
    else if (OP == NOT) asm volatile("v_not_b32 %0, %0" : "+v"(x));
    else if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(x));
    else if (OP == CNDS) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(other));
    else if (OP == CMP32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
    else if (OP == CMPEQ32) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
    else if (OP == CMP64) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(v), "v"(other) : "vcc");
    else if (OP == MAX) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == XOR64E) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == ADD64E) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x) : "v"(b));
    else if (OP == BFI) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
    else if (OP == ADDC1) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
    else if (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc");
    else if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(b));
    else if (OP == MULF) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    else if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
    else if (OP == MOVDPP) { if (CH == 1) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x)); else asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x)); }
    else if (OP == XOR_MIN) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); uint32_t lo = (uint32_t) v; asm volatile("v_min_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); v = lo; }
    else if (OP == XOR3_MAD) { asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(b), "v"(b) : "vcc"); }
    else if (OP == CMP64_XOR) { asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(v), "v"(other) : "vcc"); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)); }
}

// CH independent chains, `iters` rounds of 8 steps per chain.  out[1 << 20 ..] = {s_memtime ticks, wall_clock64 ticks} of block 0's first wave.
template <int OP, int CH>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed, int iters)
{
    uint64_t v[CH];
    uint32_t x[CH];
    for (int c = 0; c < CH; ++c) v[c] = seed * (2 * c + 3) + threadIdx.x, x[c] = (uint32_t) (seed >> 7) * (2 * c + 5) + threadIdx.x;
    const uint32_t b = (uint32_t) seed | 1u;
    const uint64_t other = seed * 977;
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) step<OP, CH>(v[c], x[c], b, other);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint64_t acc = 0;
    for (int c = 0; c < CH; ++c) acc ^= v[c] ^ x[c];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & ((1 << 20) - 1)] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = t1 - t0, out[(1 << 20) + 1] = w1 - w0;
}

static uint64_t *d_out;

template <int OP, int CH>
static void run_one(int waves_per_simd, FILE *fo)
{
    const int blocks = 256 * waves_per_simd, threads = 256;          // one wave per SIMD per block, 256 CUs
    int iters = 2000;
    float ms = 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {                            // pass 0 calibrates, pass 1 runs >= 25 ms
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, d_out, 12345ULL, iters);      // warm: clocks up, code resident
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, d_out, 12345ULL, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) { double f = 30.0 / (ms > 0.01? ms : 0.01); iters = (int) (iters * f) + 1; }
    }
    uint64_t tk[2];
    CHECK(hipMemcpy(tk, d_out + (1 << 20), 16, hipMemcpyDeviceToHost));
    const double instr_per_wave = (double) iters * 8.0 * CH * PER[OP];
    const double wall_s = (double) tk[1] / 100e6;                     // wall_clock64: 100 MHz
    const double ghz = (double) tk[0] / wall_s / 1e9;
    const double cyc = (double) tk[0] / (instr_per_wave * waves_per_simd);
    fprintf(fo, "%-30s chains %d  waves/SIMD %d  kernel %7.2f ms  clock %.3f GHz  SIMD cycles per wave-instruction %6.2f   (event-derived at the measured clock %6.2f)\n",
            NAMES[OP], CH, waves_per_simd, ms, ghz, cyc, ms * 1e-3 * ghz * 1e9 / (instr_per_wave * waves_per_simd));
    fflush(fo);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int OP>
static void run(FILE *fo)
{
    for (int w = 1; w <= 8; w *= 2) run_one<OP, 8>(w, fo);
    run_one<OP, 1>(1, fo);                                            // dependent chain, one wave per SIMD: latency
}

int main(int argc, char **argv)
{
    FILE *fo = stdout;
    CHECK(hipMalloc(&d_out, ((1 << 20) + 8) * 8));
    if (argc > 1 && !strcmp(argv[1], "more")) {       // the wider census (round 4): which opcodes issue at the 2-cycle rate at all?
        run<AND>(fo); run<OR>(fo); run<SHL32>(fo); run<SHR32>(fo); run<SUB>(fo); run<MOV>(fo); run<NOT>(fo); run<BFE>(fo); run<CNDS>(fo); run<CMP32>(fo); run<CMPEQ32>(fo); run<CMP64>(fo); run<MAX>(fo); run<MIN3>(fo); run<XOR64E>(fo); run<ADD64E>(fo); run<BITOP3>(fo); run<BFI>(fo); run<LSHLADD>(fo); run<ADDC1>(fo); run<ADDCO>(fo); run<FMA>(fo); run<FMAC>(fo); run<MULF>(fo); run<PKFMA>(fo); run<MOVDPP>(fo); run<XOR_MIN>(fo); run<XOR3_MAD>(fo);
        return 0;
    }
    run<XOR>(fo); run<ADD>(fo); run<MIN>(fo); run<ALIGNBIT>(fo); run<PERM>(fo); run<CNDMASK>(fo); run<ANDOR>(fo); run<ADD3>(fo); run<XAD>(fo); run<LSHLOR>(fo);
    run<MUL24>(fo); run<MULLO>(fo); run<MULHI>(fo); run<MAD64>(fo);
    run<SHL64>(fo); run<SHR64>(fo); run<LSHLADD64>(fo); run<ADDC>(fo); run<CMP64SEL>(fo); run<CMP32SEL>(fo); run<DPPMIN>(fo);
    run<MAD64_XOR>(fo); run<MAD64_2XOR>(fo); run<CMP64_XOR>(fo);
    return 0;
}
