// Is device memory that came back to the driver cleared when it is handed out again, and does waiting help?  One process: take 8 x 28 GB and write to them, give
// them back, then time hipMalloc of 40 GB at once / after a pause / as an address range backed in 1 GB chunks; then the same once more.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const size_t GB = 1ULL << 30;
static hipStream_t st;

static void dirty(int blocks, size_t gb)
{
    std::vector<void *> v;
    double t0 = now();
    for (int i = 0; i < blocks; ++i) { void *p = 0; CHECK(hipMalloc(&p, gb * GB)); CHECK(hipMemsetAsync(p, 0x5a, gb * GB, st)); v.push_back(p); }
    CHECK(hipStreamSynchronize(st));
    printf("took and wrote %d x %zu GB: %.3f s", blocks, gb, now() - t0);
    t0 = now();
    for (void *p : v) CHECK(hipFree(p));
    printf(", gave them back: %.3f s\n", now() - t0);
}
static void one(size_t gb, const char *what)
{
    void *p = 0;
    double t0 = now();
    CHECK(hipMalloc(&p, gb * GB));
    double ta = now() - t0;
    t0 = now();
    CHECK(hipMemsetAsync(p, 1, gb * GB, st));
    CHECK(hipStreamSynchronize(st));
    double tm = now() - t0;
    t0 = now();
    CHECK(hipFree(p));
    printf("hipMalloc %zu GB %s: %.3f s, first memset %.3f s (free %.3f s)\n", gb, what, ta, tm, now() - t0);
}
static void chunks(size_t gb, size_t chunk_mb)
{
    const size_t total = gb * GB, chunk = chunk_mb << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void *va = 0;
    CHECK(hipMemAddressReserve(&va, total, 2 << 20, 0, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double tc = 0, tm = 0, ts = 0, worst = 0, t00 = now();
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
    printf("%zu GB behind one address range in %zu MB chunks: %.3f s (create %.3f, map %.3f, set access %.3f; the slowest chunk %.4f s)\n", gb, chunk_mb, now() - t00, tc, tm, ts, worst);
    double t0 = now();
    CHECK(hipMemsetAsync(va, 1, total, st));
    CHECK(hipStreamSynchronize(st));
    printf("   first memset over the range: %.3f s\n", now() - t0);
    t0 = now();
    for (size_t i = 0; i < hs.size(); ++i) { CHECK(hipMemUnmap((char *) va + i * chunk, chunk)); CHECK(hipMemRelease(hs[i])); }
    CHECK(hipMemAddressFree(va, total));
    printf("   unmap + release + free: %.3f s\n", now() - t0);
}

int main()
{
    CHECK(hipSetDevice(0));
    CHECK(hipFree(0));
    CHECK(hipStreamCreate(&st));
    size_t fr = 0, tot = 0;
    CHECK(hipMemGetInfo(&fr, &tot));
    printf("device memory: %.1f GB free of %.1f\n", fr / 1e9, tot / 1e9);
    one(40, "in a process that has given nothing back");
    chunks(40, 1024);
    dirty(8, 28);
    one(40, "right after");
    dirty(8, 28);
    chunks(40, 1024);
    dirty(8, 28);
    one(40, "right after");
    dirty(8, 28);
    chunks(40, 64);
    dirty(8, 28);
    sleep(12);
    one(40, "twelve seconds after");
    chunks(40, 1024);
    return 0;
}
