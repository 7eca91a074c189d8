// Do the two directions of the host link add up?  (round 4: sr_read moves 30 GB up and 37 GB down and takes what the two take one after the other.)
// H2D and D2H as hipMemcpyAsync on two streams, and D2H as a KERNEL that stores into page-locked host memory while hipMemcpyAsync goes up.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void copy_kernel(const uint4 *src, uint4 *dst, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}
int main(int argc, char **argv)
{
    const size_t n = (size_t) (argc > 1? atoll(argv[1]) : 2) << 30;
    const int blocks = argc > 2? atoi(argv[2]) : 512;
    uint8_t *d_up, *d_down, *h_up, *h_down;
    CHECK(hipSetDevice(0));
    CHECK(hipMalloc(&d_up, n)); CHECK(hipMalloc(&d_down, n));
    CHECK(hipMemset(d_down, 5, n));
    CHECK(hipHostMalloc(&h_up, n, hipHostMallocDefault));
    memset(h_up, 3, n);
    h_down = (uint8_t *) mmap(0, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(h_down, n, MADV_HUGEPAGE);
    memset(h_down, 0, n);
    CHECK(hipHostRegister(h_down, n, hipHostRegisterMapped));
    uint8_t *h_down_dev = 0;
    CHECK(hipHostGetDevicePointer((void **) &h_down_dev, h_down, 0));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const double g = (double) n / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CHECK(hipMemcpyAsync(d_up, h_up, n, hipMemcpyHostToDevice, s1)); CHECK(hipStreamSynchronize(s1));
        printf("H2D memcpy alone:                 %6.1f GB/s\n", g / (now() - t0));
        t0 = now();
        CHECK(hipMemcpyAsync(h_down, d_down, n, hipMemcpyDeviceToHost, s2)); CHECK(hipStreamSynchronize(s2));
        printf("D2H memcpy alone:                 %6.1f GB/s\n", g / (now() - t0));
        t0 = now();
        CHECK(hipMemcpyAsync(d_up, h_up, n, hipMemcpyHostToDevice, s1));
        CHECK(hipMemcpyAsync(h_down, d_down, n, hipMemcpyDeviceToHost, s2));
        CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
        printf("H2D memcpy + D2H memcpy together: %6.1f GB/s in all (%.3f s)\n", 2 * g / (now() - t0), now() - t0);
        t0 = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *) d_down, (uint4 *) h_down_dev, n / 16);
        CHECK(hipStreamSynchronize(s2));
        printf("D2H kernel (%d blocks) alone:     %6.1f GB/s\n", blocks, g / (now() - t0));
        t0 = now();
        CHECK(hipMemcpyAsync(d_up, h_up, n, hipMemcpyHostToDevice, s1));
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *) d_down, (uint4 *) h_down_dev, n / 16);
        CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
        printf("H2D memcpy + D2H kernel together: %6.1f GB/s in all (%.3f s)\n", 2 * g / (now() - t0), now() - t0);
        t0 = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const uint4 *) h_up, (uint4 *) d_up, n / 16);
        CHECK(hipStreamSynchronize(s1));
        printf("H2D kernel alone:                 %6.1f GB/s\n", g / (now() - t0));
    }
    if (h_down[n - 1] != 5 || h_down[0] != 5) printf("D2H kernel wrote nothing?\n");
    return 0;
}
