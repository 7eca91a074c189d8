// What does device memory cost to GET?  (round 6: `room for the batch` is 1.0 s of sr_read at 2 M reads in a process that has the device to itself -- ~50 GB in
// fifteen hipMalloc calls -- and 0.03 s in a process that starts right after another one has ended, whose context then takes 2.1 s to come up.)
// Measures in ONE fresh process: context creation; hipMalloc of 1 / 8 / 32 GB blocks (first time, and again after hipFree); the same 32 GB as a reserved
// address range backed chunk by chunk (hipMemCreate + hipMemMap + hipMemSetAccess, 1 GB and 256 MB chunks); a memset over fresh memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    const size_t GB = 1ULL << 30;
    double t0 = now();
    CHECK(hipSetDevice(0));
    CHECK(hipFree(0));
    printf("context: %.3f s\n", now() - t0);
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const size_t sizes[] = {1, 8, 32, 32, 64};
    for (size_t s : sizes) {
        void *p = 0;
        t0 = now();
        CHECK(hipMalloc(&p, s * GB));
        const double ta = now() - t0;
        t0 = now();
        CHECK(hipMemsetAsync(p, 1, s * GB, st));
        CHECK(hipStreamSynchronize(st));
        const double tm = now() - t0;
        t0 = now();
        CHECK(hipFree(p));
        printf("hipMalloc %3zu GB: %.3f s (%.1f GB/s)   first memset %.3f s (%.0f GB/s)   hipFree %.3f s\n", s, ta, s / ta, tm, s / tm, now() - t0);
    }
    {   // many blocks in a row, as the batch's arrays are
        std::vector<void *> v;
        t0 = now();
        for (int i = 0; i < 16; ++i) { void *p = 0; CHECK(hipMalloc(&p, 3 * GB)); v.push_back(p); }
        printf("16 x hipMalloc 3 GB: %.3f s\n", now() - t0);
        t0 = now();
        for (void *p : v) CHECK(hipFree(p));
        printf("16 x hipFree: %.3f s\n", now() - t0);
    }
    for (size_t chunk_mb : {1024, 256}) {
        const size_t total = 32 * GB, chunk = chunk_mb << 20;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        void *va = 0;
        t0 = now();
        CHECK(hipMemAddressReserve(&va, total, gran, 0, 0));
        const double tr = now() - t0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        double tc = 0, tm = 0, ts = 0, worst = 0;
        for (size_t o = 0; o < total; o += chunk) {
            hipMemGenericAllocationHandle_t h;
            double a = now();
            CHECK(hipMemCreate(&h, chunk, &prop, 0));
            double b = now();
            CHECK(hipMemMap((char *) va + o, chunk, 0, h, 0));
            double c = now();
            CHECK(hipMemSetAccess((char *) va + o, chunk, &acc, 1));
            double d = now();
            tc += b - a, tm += c - b, ts += d - c;
            if (d - a > worst) worst = d - a;
            hs.push_back(h);
        }
        printf("32 GB behind one address range in %zu MB chunks (granularity %zu KB): reserve %.4f s, create %.3f s, map %.3f s, set access %.3f s; the slowest chunk %.4f s\n",
               chunk_mb, gran >> 10, tr, tc, tm, ts, worst);
        t0 = now();
        CHECK(hipMemsetAsync(va, 1, total, st));
        CHECK(hipStreamSynchronize(st));
        printf("   memset over the range: %.3f s (%.0f GB/s)\n", now() - t0, 32 / (now() - t0));
        t0 = now();
        for (size_t i = 0; i < hs.size(); ++i) { CHECK(hipMemUnmap((char *) va + i * chunk, chunk)); CHECK(hipMemRelease(hs[i])); }
        CHECK(hipMemAddressFree(va, total));
        printf("   unmap + release + free: %.3f s\n", now() - t0);
    }
    return 0;
}
