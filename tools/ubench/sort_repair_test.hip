// development aid: the 40-bit sort + repair of count.hpp on random keys against std::stable_sort (hipcc -O2 -I../../oatk_amd/csrc sort_repair_test.hip)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "count.hpp"
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    using namespace oatk;
    const uint32_t n = argc > 1? atoi(argv[1]) : 2000;
    const int tops = argc > 2? atoi(argv[2]) : 64, lows = argc > 3? atoi(argv[3]) : 5;
    std::vector<uint64_t> key(n); std::vector<uint32_t> iota(n);
    srand(7);
    for (uint32_t i = 0; i < n; ++i) key[i] = (uint64_t) (rand() % tops) << 54 | (uint64_t) (rand() % lows) * 0x10001ULL, iota[i] = i;
    uint64_t *d_k, *d_ks; uint32_t *d_i, *d_p, *d_owner, *d_flags; void *tmp; size_t tb = 0;
    CKH(hipMalloc(&d_k, n * 8)); CKH(hipMalloc(&d_ks, n * 8)); CKH(hipMalloc(&d_i, n * 4)); CKH(hipMalloc(&d_p, n * 4)); CKH(hipMalloc(&d_owner, n * 4)); CKH(hipMalloc(&d_flags, 64));
    CKH(hipMemcpy(d_k, key.data(), n * 8, hipMemcpyHostToDevice)); CKH(hipMemcpy(d_i, iota.data(), n * 4, hipMemcpyHostToDevice));
    CKH(hipMemset(d_flags, 0, 64)); CKH(hipMemset(d_owner, 0xFF, n * 4));
    CKH(rocprim::radix_sort_pairs(nullptr, tb, d_k, d_ks, d_i, d_p, n, OATK_SORT_LOW_BITS, 64, 0));
    CKH(hipMalloc(&tmp, tb));
    CKH(rocprim::radix_sort_pairs(tmp, tb, d_k, d_ks, d_i, d_p, n, OATK_SORT_LOW_BITS, 64, 0));
    std::vector<uint64_t> mid(n); CKH(hipMemcpy(mid.data(), d_ks, n * 8, hipMemcpyDeviceToHost));
    const unsigned nb = (n + 255) / 256;
    hipLaunchKernelGGL(sort_repair_find_kernel, dim3(nb), dim3(256), 0, 0, d_ks, n, d_owner, d_flags);
    hipLaunchKernelGGL(sort_repair_kernel, dim3(nb), dim3(256), 0, 0, d_ks, d_p, n, d_owner, d_flags);
    CKH(hipDeviceSynchronize());
    std::vector<uint64_t> got(n); std::vector<uint32_t> gp(n), own(n); uint32_t fl[16];
    CKH(hipMemcpy(got.data(), d_ks, n * 8, hipMemcpyDeviceToHost)); CKH(hipMemcpy(gp.data(), d_p, n * 4, hipMemcpyDeviceToHost));
    CKH(hipMemcpy(own.data(), d_owner, n * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(fl, d_flags, 64, hipMemcpyDeviceToHost));
    std::vector<uint32_t> want(iota);
    std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    uint32_t bad = 0, badmid = 0, owners = 0, badperm = 0;
    for (uint32_t i = 0; i < n; ++i) { bad += got[i] != key[want[i]] || gp[i] != want[i]; badperm += gp[i] >= n; owners += own[i] != 0xFFFFFFFFu; if (i && (mid[i] >> 24) < (mid[i - 1] >> 24)) ++badmid; }
    printf("n %u: %u positions differ from a stable sort, %u inversions of the top bits after the radix sort, %u runs with an owner, %u values out of range, overflow flag %u\n", n, bad, badmid, owners, badperm, fl[3]);
    return bad != 0;
}
