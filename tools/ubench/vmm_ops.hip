// Do memset / memcpy at OFFSETS inside an address range backed by several pieces land where they should?  (round 6: results changed when small buffers became such ranges)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fill(uint32_t *p, size_t n, uint32_t v) { size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + (uint32_t) i; }
__global__ void sum(const uint32_t *p, size_t n, unsigned long long *out) { size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] != 0) atomicAdd(out, 1ULL); }
int main()
{
    const size_t C = 64ull << 20, N = 3;
    CHECK(hipSetDevice(0)); CHECK(hipFree(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    char *va = 0;
    CHECK(hipMemAddressReserve((void **) &va, 4 * N * C, 2 << 20, 0, 0));
    for (size_t i = 0; i < N; ++i) { hipMemGenericAllocationHandle_t h; CHECK(hipMemCreate(&h, C, &prop, 0)); CHECK(hipMemMap(va + i * C, C, 0, h, 0)); CHECK(hipMemSetAccess(va + i * C, C, &acc, 1)); }
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t words = N * C / 4;
    unsigned long long *d_cnt; CHECK(hipMalloc(&d_cnt, 8));
    std::vector<uint32_t> h(1024);
    int bad = 0;
    auto nonzero = [&]() -> unsigned long long { unsigned long long c = 0; CHECK(hipMemsetAsync(d_cnt, 0, 8, st)); sum<<<(unsigned) ((words + 255) / 256), 256, 0, st>>>((uint32_t *) va, words, d_cnt); CHECK(hipMemcpyAsync(&c, d_cnt, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st)); return c; };
    auto refill = [&]() { fill<<<(unsigned) ((words + 255) / 256), 256, 0, st>>>((uint32_t *) va, words, 1u); CHECK(hipStreamSynchronize(st)); };
    struct { size_t off, len; const char *what; } cases[] = {
        {8, 40, "40 bytes at offset 8"}, {4096, 1 << 20, "1 MB at offset 4096"}, {C - 4096, 8192, "8 KB across the first piece boundary"},
        {C + 8, 40, "40 bytes at offset 8 of the second piece"}, {2 * C - 256, 512, "512 bytes across the second boundary"}, {C / 2, 2 * C, "128 MB from the middle of the first piece"},
    };
    for (auto &c : cases) {
        refill();                                          // every word nonzero (1 + i never 0 for i < 2^32 - 1)
        CHECK(hipMemsetAsync(va + c.off, 0, c.len, st));
        const unsigned long long nz = nonzero(), want = words - c.len / 4;
        // and exactly the right words: read the edges back
        uint32_t e[4];
        CHECK(hipMemcpyAsync(e, va + c.off - 4, 8, hipMemcpyDeviceToHost, st)); CHECK(hipMemcpyAsync(e + 2, va + c.off + c.len - 4, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
        const bool ok = nz == want && e[0] != 0 && e[1] == 0 && e[2] == 0 && e[3] != 0;
        printf("memset %-48s: %llu words nonzero, expected %llu; edges %s -> %s\n", c.what, nz, want, (e[0] && !e[1] && !e[2] && e[3])? "right" : "WRONG", ok? "ok" : "BAD");
        bad += !ok;
    }
    {   // device-to-device copy inside the range, source and destination at offsets, across boundaries
        refill();
        const size_t so = C - 1000 * 4, dof = 2 * C - 500 * 4, n = 2000 * 4;
        CHECK(hipMemcpyAsync(va + dof, va + so, n, hipMemcpyDeviceToDevice, st));
        CHECK(hipMemcpyAsync(h.data(), va + dof, 1024 * 4, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
        int ok = 1; for (int i = 0; i < 1024; ++i) ok &= h[i] == 1u + (uint32_t) (so / 4 + i);
        printf("copy inside the range across boundaries: %s\n", ok? "ok" : "BAD"); bad += !ok;
    }
    {   // host to device at an offset in the second piece, and back
        for (int i = 0; i < 1024; ++i) h[i] = 0xC0000000u + i;
        CHECK(hipMemcpyAsync(va + C + 12, h.data(), 4096, hipMemcpyHostToDevice, st));
        std::vector<uint32_t> g(1024);
        CHECK(hipMemcpyAsync(g.data(), va + C + 12, 4096, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
        int ok = memcmp(g.data(), h.data(), 4096) == 0;
        printf("host -> device -> host at offset 12 of the second piece: %s\n", ok? "ok" : "BAD"); bad += !ok;
    }
    printf("%s\n", bad? "SOMETHING IS WRONG" : "all right");
    return bad;
}
