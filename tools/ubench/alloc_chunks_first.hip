// The first thing a process does on the device: <gb> GB behind one address range, backed in <mb> MB chunks, every chunk timed (create + map + set access),
// then a memset over the range.   alloc_chunks_first <gb> <mb>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t total = (size_t) (argc > 1? atoi(argv[1]) : 40) << 30, chunk = (size_t) (argc > 2? atoi(argv[2]) : 1024) << 20;
    CHECK(hipSetDevice(0)); CHECK(hipFree(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void *va = 0;
    CHECK(hipMemAddressReserve(&va, total, 2 << 20, 0, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<double> t;
    double t00 = now();
    for (size_t o = 0; o < total; o += chunk) {
        hipMemGenericAllocationHandle_t h;
        double a = now();
        CHECK(hipMemCreate(&h, chunk, &prop, 0));
        CHECK(hipMemMap((char *) va + o, chunk, 0, h, 0));
        CHECK(hipMemSetAccess((char *) va + o, chunk, &acc, 1));
        t.push_back(now() - a);
    }
    printf("%zu chunks of %zu MB in %.3f s; ms per chunk:", t.size(), chunk >> 20, now() - t00);
    for (double x : t) printf(" %.1f", x * 1e3);
    double t0 = now();
    CHECK(hipMemset(va, 1, total)); CHECK(hipDeviceSynchronize());
    printf("\n   memset %.3f s\n", now() - t0);
    return 0;
}
