#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define ECF_PROF
#include "/root/repo/oatk_amd/csrc/ec_fused.hpp"
using namespace oatk;
int main() {
    const int L = 6000, bw = 120;
    std::vector<uint32_t> tw(ecw_words(L)), qw(ecw_words(L));
    srand(5);
    for (auto &w : tw) w = (uint32_t) rand() * 2654435761u;
    for (auto &w : qw) w = (uint32_t) rand() * 40503u + 17;
    uint64_t two[2] = {0, tw.size()}, qwo[2] = {0, qw.size()}, so[2] = {0, 1};
    int32_t tl = L, band = bw, ql = L;
    uint32_t *d_tw, *d_qw; uint64_t *d_two, *d_qwo, *d_so; int32_t *d_tl, *d_bw, *d_ql, *d_out;
    hipMalloc(&d_tw, tw.size() * 4); hipMalloc(&d_qw, qw.size() * 4); hipMalloc(&d_two, 16); hipMalloc(&d_qwo, 16); hipMalloc(&d_so, 16);
    hipMalloc(&d_tl, 4); hipMalloc(&d_bw, 4); hipMalloc(&d_ql, 4); hipMalloc(&d_out, 12);
    hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_qw, qw.data(), qw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_two, two, 16, hipMemcpyHostToDevice); hipMemcpy(d_qwo, qwo, 16, hipMemcpyHostToDevice); hipMemcpy(d_so, so, 16, hipMemcpyHostToDevice);
    hipMemcpy(d_tl, &tl, 4, hipMemcpyHostToDevice); hipMemcpy(d_bw, &band, 4, hipMemcpyHostToDevice); hipMemcpy(d_ql, &ql, 4, hipMemcpyHostToDevice);
    const unsigned cap = (unsigned) tw.size();
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ecf_wf_ed_kernel<ECF_NW>), dim3(1), dim3(64 * ECF_NW), (ecf_misc_words(ECF_NW) + 2 * cap) * 4, 0, d_tw, d_two, d_tl, d_qw, d_qwo, d_bw, d_ql, d_so, d_out, (int32_t) cap);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int32_t out[3]; hipMemcpy(out, d_out, 12, hipMemcpyDeviceToHost);
        printf("kernel %.3f ms; cycles in steps %d, in exchanges %d, chunks %d (%s)\n", ms, out[0], out[1], out[2], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
