// One process: context, then hipMalloc of <gb> GB timed, a memset, optionally keep and write <dirty> GB more before giving everything back at exit.
//   alloc_first <gb> [dirty_gb]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t GB = 1ULL << 30, gb = argc > 1? atoi(argv[1]) : 40, dg = argc > 2? atoi(argv[2]) : 0;
    double t0 = now();
    CHECK(hipSetDevice(0)); CHECK(hipFree(0));
    const double tc = now() - t0;
    void *p = 0, *q = 0;
    t0 = now();
    CHECK(hipMalloc(&p, gb * GB));
    const double ta = now() - t0;
    t0 = now();
    CHECK(hipMemset(p, 1, gb * GB)); CHECK(hipDeviceSynchronize());
    const double tm = now() - t0;
    t0 = now();
    if (dg) { CHECK(hipMalloc(&q, dg * GB)); CHECK(hipMemset(q, 1, dg * GB)); CHECK(hipDeviceSynchronize()); }
    printf("context %.3f s, hipMalloc %zu GB %.3f s, memset %.3f s%s", tc, gb, ta, tm, dg? "" : "\n");
    if (dg) printf(", %zu GB more taken and written in %.3f s\n", dg, now() - t0);
    return 0;
}
