// tools/ubench/census.hip -- how many workgroups of a given shape does an MI355X CU really hold?  (development aid)
// Every workgroup notes when it starts and ends (wall clock), and stays for `hold_us`; the host counts the largest number of
// intervals that overlap and divides by the number of CUs.  usage: census <threads> <lds_bytes> <accumulators kept in registers: 0 40 50 58 66 74 82 100> [hold_us]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V>
__global__ void census_kernel(unsigned long long *t, unsigned long long hold, float *sink)
{
    extern __shared__ unsigned int lds[];
    const unsigned long long t0 = wall_clock64();
    float acc[V > 0? V : 1];
    for (int i = 0; i < (V > 0? V : 1); ++i) acc[i] = (float) (threadIdx.x + i);
    lds[threadIdx.x] = threadIdx.x;
    while (wall_clock64() - t0 < hold) {
#pragma unroll
        for (int i = 0; i < (V > 0? V : 1); ++i) acc[i] = acc[i] * 1.0001f + (float) lds[(threadIdx.x + i) & 63];
    }
    float s = 0;
    for (int i = 0; i < (V > 0? V : 1); ++i) s += acc[i];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) t[2 * blockIdx.x] = t0, t[2 * blockIdx.x + 1] = wall_clock64();
}

int main(int argc, char **argv)
{
    const int threads = argc > 1? atoi(argv[1]) : 64, lds = argc > 2? atoi(argv[2]) : 0, v = argc > 3? atoi(argv[3]) : 0;
    const double hold_us = argc > 4? atof(argv[4]) : 300.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, grid = n_cu * 48;
    int rate_khz = 100000;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    unsigned long long *d_t; float *d_s;
    hipMalloc(&d_t, 16ULL * grid); hipMalloc(&d_s, 4);
    const unsigned long long hold = (unsigned long long) (hold_us * 1e-6 * rate_khz * 1e3);
    for (int rep = 0; rep < 2; ++rep) {
#define CASE(V) if (v == V) hipLaunchKernelGGL(census_kernel<V>, dim3(grid), dim3(threads), lds, 0, d_t, hold, d_s)
        CASE(0); CASE(40); CASE(50); CASE(58); CASE(66); CASE(74); CASE(82); CASE(100);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> t(2 * grid);
    hipMemcpy(t.data(), d_t, 16ULL * grid, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < grid; ++i) ev.push_back({t[2 * i], 1}), ev.push_back({t[2 * i + 1], -1});
    std::sort(ev.begin(), ev.end());
    int cur = 0, best = 0;
    for (auto &e : ev) { cur += e.second; best = std::max(best, cur); }
    hipFuncAttributes fa;
#define ATTR(V) if (v == V) hipFuncGetAttributes(&fa, (const void *) census_kernel<V>)
    ATTR(0); ATTR(40); ATTR(50); ATTR(58); ATTR(66); ATTR(74); ATTR(82); ATTR(100);
    printf("threads %4d  lds %6d B  vgpr %3d  -> %5.2f workgroups per CU resident (%d CUs, %d launched), %5.2f waves per CU\n", threads, lds, fa.numRegs, (double) best / n_cu, n_cu, grid,
           (double) best / n_cu * (threads / 64));
    return 0;
}
