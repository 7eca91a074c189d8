// checks the DPP helpers of scan_syncmer_fast.hpp against plain loops (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../oatk_amd/csrc/scan_syncmer_fast.hpp"
using namespace oatk;
__global__ void k(const uint32_t *in, uint32_t *pre, uint32_t *suf, uint64_t *red)
{
    uint32_t lane = threadIdx.x;
    uint32_t p, s;
    wave_prefix_suffix_min_u32(in[lane], lane, p, s);
    pre[lane] = p; suf[lane] = s;
    uint64_t v = (uint64_t) in[lane] << 32 | in[63 - lane];
    uint64_t r = wave_reduce_min_u64(v);
    if (lane == 0) red[0] = r;
}
int main()
{
    uint32_t h[64], hp[64], hs[64]; uint64_t hr;
    uint32_t *d, *dp, *ds; uint64_t *dr;
    (void) hipMalloc(&d, 256); (void) hipMalloc(&dp, 256); (void) hipMalloc(&ds, 256); (void) hipMalloc(&dr, 8);
    int bad = 0;
    for (int t = 0; t < 200; ++t) {
        for (int i = 0; i < 64; ++i) h[i] = (uint32_t) (rand() % 1000) + 5;
        (void) hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dp, ds, dr);
        (void) hipMemcpy(hp, dp, 256, hipMemcpyDeviceToHost); (void) hipMemcpy(hs, ds, 256, hipMemcpyDeviceToHost); (void) hipMemcpy(&hr, dr, 8, hipMemcpyDeviceToHost);
        uint32_t m = 0xffffffffu; uint64_t rm = ~0ULL;
        for (int i = 0; i < 64; ++i) { m = h[i] < m? h[i] : m; if (hp[i] != m) { if (bad < 5) printf("pre[%d] got %u want %u\n", i, hp[i], m); ++bad; } }
        m = 0xffffffffu;
        for (int i = 63; i >= 0; --i) { m = h[i] < m? h[i] : m; if (hs[i] != m) { if (bad < 5) printf("suf[%d] got %u want %u\n", i, hs[i], m); ++bad; } }
        for (int i = 0; i < 64; ++i) { uint64_t v = (uint64_t) h[i] << 32 | h[63 - i]; rm = v < rm? v : rm; }
        if (hr != rm) { if (bad < 5) printf("reduce got %llx want %llx\n", (unsigned long long) hr, (unsigned long long) rm); ++bad; }
    }
    printf("dpp helpers: %d mismatches\n", bad);
    return bad != 0;
}
