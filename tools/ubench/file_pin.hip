// Can a FILE's pages (page cache, or tmpfs) be page-locked and uploaded as they lie?  (round 4: the uploader of host/ingest_host.c copies the
// input text once on the host -- pread into page-locked staging, 30 GB at 2 M reads -- before it goes over PCIe.)
// usage: file_pin <dir> [GB]     writes <dir>/oatk_file_pin.tmp, maps it, registers the mapping, copies it to the device.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    const char *dir = argc > 1? argv[1] : "/tmp";
    const size_t n = (size_t) (argc > 2? atoi(argv[2]) : 4) << 30;
    char path[4096];
    snprintf(path, sizeof(path), "%s/oatk_file_pin.tmp", dir);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) { perror("open"); return 1; }
    {
        char *blk = (char *) malloc(1 << 24);
        memset(blk, 'A', 1 << 24);
        for (size_t o = 0; o < n; o += 1 << 24) if (write(fd, blk, 1 << 24) != (1 << 24)) { perror("write"); return 1; }
        free(blk);
    }
    uint8_t *d;
    if (hipMalloc(&d, n) != hipSuccess) return 1;
    const double g = (double) n / 1e9;
    for (int shared = 0; shared < 2; ++shared) {
        void *p = mmap(0, n, PROT_READ, (shared? MAP_SHARED : MAP_PRIVATE) | MAP_POPULATE, fd, 0);
        if (p == MAP_FAILED) { perror("mmap"); continue; }
        double t0 = now();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        printf("%s %s: hipHostRegister %s", dir, shared? "MAP_SHARED" : "MAP_PRIVATE", e == hipSuccess? "ok" : hipGetErrorString(e));
        if (e == hipSuccess) {
            printf(" %.2f GB/s", g / (now() - t0));
            t0 = now();
            e = hipMemcpy(d, p, n, hipMemcpyHostToDevice);
            printf("; H2D from the mapping %s %.2f GB/s", e == hipSuccess? "ok" : hipGetErrorString(e), g / (now() - t0));
            t0 = now();
            hipHostUnregister(p);
            printf("; unregister %.3f s", now() - t0);
        } else (void) hipGetLastError();
        printf("\n");
        // read-only registration flag
        if (e != hipSuccess) {
            t0 = now();
            e = hipHostRegister(p, n, hipHostRegisterReadOnly);
            printf("   with hipHostRegisterReadOnly: %s\n", e == hipSuccess? "ok" : hipGetErrorString(e));
            if (e == hipSuccess) {
                t0 = now();
                e = hipMemcpy(d, p, n, hipMemcpyHostToDevice);
                printf("   H2D %s %.2f GB/s\n", e == hipSuccess? "ok" : hipGetErrorString(e), g / (now() - t0));
                hipHostUnregister(p);
            } else (void) hipGetLastError();
        }
        munmap(p, n);
    }
    {   // what the uploader does today: pread into pinned staging by 16 threads is ~15 GB/s; one plain hipMemcpy from the (unregistered) mapping:
        void *p = mmap(0, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        double t0 = now();
        hipError_t e = hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        printf("%s: hipMemcpy straight from the unregistered mapping %s %.2f GB/s\n", dir, e == hipSuccess? "ok" : hipGetErrorString(e), g / (now() - t0));
        munmap(p, n);
    }
    close(fd);
    unlink(path);
    return 0;
}
