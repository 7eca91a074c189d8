// What does it cost to make gigabytes of host memory the DESTINATION of device-to-host copies?  (round 4: the reads' arrays handed to the
// reference are 29 GB at 2 M reads; round 3 copied them twice on the host -- pinned staging, then memcpy into the arena.)
// Measures, per block size: first touch of a fresh anonymous mapping (4 KiB pages / transparent huge pages, 1 and N threads),
// hipHostRegister of that mapping (untouched and touched), hipHostMalloc, D2H into registered / hipHostMalloc'ed / pageable memory,
// hipHostUnregister and munmap.  Prints GB/s (or GB per second of call time).
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { uint8_t *p; size_t n; int tid, nt; } tj_t;
static void *toucher(void *a)
{
    tj_t *j = (tj_t *) a;
    size_t lo = j->n * j->tid / j->nt, hi = j->n * (j->tid + 1) / j->nt, i;
    lo &= ~(size_t) 4095;
    for (i = lo; i < hi; i += 4096) j->p[i] = 1;
    return 0;
}
static double touch(uint8_t *p, size_t n, int nt)
{
    pthread_t th[256]; tj_t jb[256];
    double t0 = now();
    for (int t = 0; t < nt; ++t) { jb[t].p = p, jb[t].n = n, jb[t].tid = t, jb[t].nt = nt; pthread_create(&th[t], 0, toucher, &jb[t]); }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], 0);
    return now() - t0;
}
static uint8_t *fresh(size_t n, int thp)
{
    uint8_t *p = (uint8_t *) mmap(0, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    uint8_t *q = (uint8_t *) (((uintptr_t) p + (2 << 20) - 1) & ~(uintptr_t) ((2 << 20) - 1));
    if (thp) madvise(q, n, MADV_HUGEPAGE); else madvise(q, n, MADV_NOHUGEPAGE);
    return q;
}

int main(int argc, char **argv)
{
    const size_t GB = (size_t) 1 << 30;
    size_t n = (argc > 1? (size_t) atoll(argv[1]) : 4) * GB;
    int nt = argc > 2? atoi(argv[2]) : 32;
    uint8_t *d;
    CHECK(hipSetDevice(0));
    CHECK(hipMalloc(&d, n));
    CHECK(hipMemset(d, 7, n));
    CHECK(hipDeviceSynchronize());
    const double g = (double) n / 1e9;
    printf("block %.1f GB, %d touching threads\n", g, nt);
    for (int thp = 0; thp < 2; ++thp) {
        double t;
        uint8_t *p = fresh(n, thp);
        t = touch(p, n, 1); printf("thp=%d first touch, 1 thread:   %6.2f GB/s\n", thp, g / t);
        munmap(p, n);
        p = fresh(n, thp);
        t = touch(p, n, nt); printf("thp=%d first touch, %d threads: %6.2f GB/s\n", thp, nt, g / t);
        double t0 = now();
        CHECK(hipHostRegister(p, n, hipHostRegisterDefault));
        printf("thp=%d hipHostRegister (touched): %6.2f GB/s (%.3f s)\n", thp, g / (now() - t0), now() - t0);
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            CHECK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost));
            printf("thp=%d D2H into registered:       %6.2f GB/s\n", thp, g / (now() - t0));
        }
        t0 = now();
        CHECK(hipHostUnregister(p));
        printf("thp=%d hipHostUnregister:         %6.2f GB/s (%.3f s)\n", thp, g / (now() - t0), now() - t0);
        t0 = now();
        munmap(p, n);
        printf("thp=%d munmap:                    %6.2f GB/s\n", thp, g / (now() - t0));
        p = fresh(n, thp);
        t0 = now();
        CHECK(hipHostRegister(p, n, hipHostRegisterDefault));
        printf("thp=%d hipHostRegister (UNtouched): %6.2f GB/s (%.3f s)\n", thp, g / (now() - t0), now() - t0);
        t0 = now();
        CHECK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost));
        printf("thp=%d D2H into registered:       %6.2f GB/s\n", thp, g / (now() - t0));
        CHECK(hipHostUnregister(p));
        munmap(p, n);
        // registered in pieces of 256 MB by several threads at once: does the driver serialise?
        p = fresh(n, thp);
        touch(p, n, nt);
        {
            const size_t piece = (size_t) 256 << 20;
            const int np = (int) (n / piece);
            t0 = now();
            for (int i = 0; i < np; ++i) CHECK(hipHostRegister(p + i * piece, piece, hipHostRegisterDefault));
            printf("thp=%d hipHostRegister in %d pieces of 256 MB (touched, one thread): %6.2f GB/s\n", thp, np, g / (now() - t0));
            t0 = now();
            for (int i = 0; i < np; ++i) CHECK(hipMemcpyAsync(p + i * piece, d + i * piece, piece, hipMemcpyDeviceToHost, 0));
            CHECK(hipDeviceSynchronize());
            printf("thp=%d D2H into the pieces:       %6.2f GB/s\n", thp, g / (now() - t0));
            for (int i = 0; i < np; ++i) CHECK(hipHostUnregister(p + i * piece));
        }
        munmap(p, n);
        // pageable destination
        p = fresh(n, thp);
        touch(p, n, nt);
        t0 = now();
        CHECK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost));
        printf("thp=%d D2H into pageable (touched): %6.2f GB/s\n", thp, g / (now() - t0));
        munmap(p, n);
    }
    {
        void *h;
        double t0 = now();
        CHECK(hipHostMalloc(&h, n, hipHostMallocDefault));
        printf("hipHostMalloc:                   %6.2f GB/s (%.3f s)\n", g / (now() - t0), now() - t0);
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            CHECK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
            printf("D2H into hipHostMalloc:          %6.2f GB/s\n", g / (now() - t0));
        }
        // host memcpy out of it by nt threads into fresh THP memory (what round 3 did)
        uint8_t *p = fresh(n, 1);
        t0 = now();
        {
            pthread_t th[256];
            struct cp { uint8_t *d; const uint8_t *s; size_t n; } c[256];
            auto fn = [](void *a) -> void * { cp *c = (cp *) a; memcpy(c->d, c->s, c->n); return (void *) 0; };
            for (int t = 0; t < nt; ++t) { size_t lo = n * t / nt, hi = n * (t + 1) / nt; c[t].d = p + lo, c[t].s = (uint8_t *) h + lo, c[t].n = hi - lo; pthread_create(&th[t], 0, fn, &c[t]); }
            for (int t = 0; t < nt; ++t) pthread_join(th[t], 0);
        }
        printf("memcpy pinned -> fresh THP, %d threads: %6.2f GB/s\n", nt, g / (now() - t0));
        munmap(p, n);
        t0 = now();
        CHECK(hipHostFree(h));
        printf("hipHostFree:                     %6.2f GB/s (%.3f s)\n", g / (now() - t0), now() - t0);
    }
    return 0;
}
