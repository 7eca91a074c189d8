// oatk_amd/csrc/api.hip -- C ABI (include/oatk_hip.h) over the gfx950 kernels.
//
// Owns the device buffers of one resident batch of reads, launches the scan (kernel A + kernel B) and the
// count pipeline on one HIP stream, and times phases with HIP events recorded on that stream.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_transform.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <execinfo.h>
#include <sys/mman.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/oatk_hip.h"
#include "common.hpp"
#include "scan_hpc.hpp"
#include "scan_syncmer.hpp"
#include "scan_syncmer_fast.hpp"
#include "kmer_hash.hpp"
#include "count.hpp"

namespace {

// OATK_DEBUG_ALLOC_LOG=1: every call into the driver for device memory on stderr, with its size and what it took (development aid)
static int dev_alloc_log()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("OATK_DEBUG_ALLOC_LOG"); on = e && e[0] && e[0] != '0'; }
    return on;
}
static double dev_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

// ---- Device memory in pieces (oatk_hip_mem_pool, include/oatk_hip.h) ----
// What the driver does with device memory (tools/ubench/alloc_*.hip, profiles/r08l_alloc_rates.txt): memory nobody has had since the GPU was reset is cleared when it
// is handed out -- 30 ms per GB INSIDE hipMalloc / hipMemCreate --, memory a process gives back is cleared behind its back at ~33 GB/s, and the next call that wants
// memory (of any size, from any process) waits until that is done.  A process that takes 50 GB for its batch at once therefore stands still for up to 1.5 s on a
// fresh GPU, and one that gives a slab back and takes another stands still for the clearing of the first.  With a pool switched on the larger buffers of this process
// are address ranges backed by 64 MB pieces (hipMemAddressReserve / hipMemCreate / hipMemMap): a buffer grows by mapping more pieces where it is (no copy, no
// slack for growth), a buffer that is released hands its pieces to the next one (nothing goes back to the driver before the process ends), and a thread of the
// pool's own takes pieces from the driver AHEAD of the need -- beside the host's work on the reads, which is what a reader that fills structs is bound by.
constexpr size_t DM_CHUNK = 64ull << 20;          // a piece
static size_t dm_min()                             // buffers below this stay hipMalloc's (32 MB; OATK_DEBUG_POOL_MIN: tests put small buffers into pieces too)
{
    static size_t v = 0;
    if (!v) { const char *e = getenv("OATK_DEBUG_POOL_MIN"); v = e && atoll(e) > 0? (size_t) atoll(e) : (32ull << 20); }
    return v;
}
struct ChunkPool {
    int device = -1;
    bool on = false;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<hipMemGenericAllocationHandle_t> ready;       // pieces nobody has mapped ...
    std::vector<char> used;                                   // ... and whether a buffer has had them (what the driver hands out is zero, and so is what a buffer gets: vm_grow)
    size_t created = 0, target = 0;                           // pieces taken from the driver so far; what the thread works towards
    bool warming = false, stop = false, failed = false;
    std::thread th;
    hipMemAllocationProp prop;
    hipMemAccessDesc acc;
    double t_wait = 0;                                        // seconds callers stood waiting for a piece

    bool create(hipMemGenericAllocationHandle_t *h)
    {
        const double t0 = dev_alloc_log()? dev_now() : 0;
        const hipError_t e = hipMemCreate(h, DM_CHUNK, &prop, 0);
        if (dev_alloc_log() && (e != hipSuccess || dev_now() - t0 > 0.01)) fprintf(stderr, "[oatk alloc] %.3f hipMemCreate %zu MB: %.4f s%s\n", dev_now(), DM_CHUNK >> 20, dev_now() - t0, e == hipSuccess? "" : " FAILED");
        if (e != hipSuccess) (void) hipGetLastError();
        return e == hipSuccess;
    }
    void run()
    {
        (void) hipSetDevice(device);
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                if (stop || created >= target) { warming = false; cv.notify_all(); return; }
            }
            hipMemGenericAllocationHandle_t h;
            const bool ok = create(&h);
            std::unique_lock<std::mutex> lk(mu);
            if (!ok) { failed = true, warming = false; cv.notify_all(); return; }
            ready.push_back(h), used.push_back(0), ++created;
            cv.notify_all();
        }
    }
    void warm(size_t bytes)
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && bytes > fr / 10 * 7) bytes = fr / 10 * 7;        // (never more than most of what is free now)
        std::unique_lock<std::mutex> lk(mu);
        const size_t want = created + bytes / DM_CHUNK;                        // on top of what the process holds already
        if (want > target) target = want;
        if (!warming && !failed && created < target) {
            if (th.joinable()) th.join();
            warming = true;
            th = std::thread([this] { run(); });
        }
    }
    bool take(hipMemGenericAllocationHandle_t *h, bool *was_used)
    {
        *was_used = false;
        {
            std::unique_lock<std::mutex> lk(mu);
            const double t0 = dev_now();
            for (;;) {
                if (!ready.empty()) { *h = ready.back(), *was_used = used.back() != 0; ready.pop_back(), used.pop_back(); t_wait += dev_now() - t0; return true; }
                if (warming && !failed && created < target) { cv.wait(lk); continue; }     // the thread is at it: two callers inside the driver would only take turns
                break;
            }
            t_wait += dev_now() - t0;
        }
        if (!create(h)) return false;
        std::unique_lock<std::mutex> lk(mu);
        ++created;
        return true;
    }
    void give(hipMemGenericAllocationHandle_t h)
    {
        std::unique_lock<std::mutex> lk(mu);
        ready.push_back(h), used.push_back(1);
    }
    // what nobody has mapped goes back to the driver (a hipMalloc failed: the pool must not be the reason)
    size_t trim()
    {
        std::unique_lock<std::mutex> lk(mu);
        const size_t n = ready.size();
        for (auto h : ready) (void) hipMemRelease(h);
        ready.clear(), used.clear();
        created -= n, target = created;
        return n;
    }
    void end()                                                 // at exit, before the runtime's own handlers: no thread of ours inside the driver when they run
    {
        { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};
static ChunkPool *g_pool[64];                                  // by device; made by oatk_hip_mem_pool, never destroyed
static std::mutex g_pool_mu;
static void pools_end() { for (ChunkPool *p : g_pool) if (p) p->end(); }
static ChunkPool *pool_of_current_device()
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    ChunkPool *p = g_pool[d];
    return p && p->on? p : nullptr;
}
// (development aid) OATK_DEBUG_POOL_SEQ="lo:hi": only the lo-th .. (hi-1)-th decisions of a buffer to live in pieces are taken; the others stay hipMalloc's
static ChunkPool *pool_for_new_buffer()
{
    ChunkPool *p = pool_of_current_device();
    if (!p) return nullptr;
    static long lo = -1, hi = -1, seq = 0;
    if (lo < 0) { const char *e = getenv("OATK_DEBUG_POOL_SEQ"); lo = 0, hi = 1L << 60; if (e) sscanf(e, "%ld:%ld", &lo, &hi); }
    const long k = seq++;
    if (dev_alloc_log()) {
        void *bt[6];
        const int nb = backtrace(bt, 6);
        char **sy = backtrace_symbols(bt, nb);
        fprintf(stderr, "[oatk alloc] decision %ld%s  <- %s <- %s <- %s\n", k, k >= lo && k < hi? "" : " (hipMalloc)", nb > 2? sy[2] : "", nb > 3? sy[3] : "", nb > 4? sy[4] : "");
        free(sy);
    }
    return k >= lo && k < hi? p : nullptr;
}

static hipError_t dev_malloc(void **p, size_t bytes)
{
    const double t0 = dev_alloc_log()? dev_now() : 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        ChunkPool *pl = pool_of_current_device();
        if (pl && pl->trim()) { (void) hipGetLastError(); e = hipMalloc(p, bytes); }
    }
    if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] %.3f hipMalloc %10.3f MB: %.4f s%s\n", dev_now(), (double) bytes / 1e6, dev_now() - t0, e == hipSuccess? "" : " FAILED");
    {   // OATK_DEBUG_POISON=1 (tests): new memory is 0xA5 all over instead of the driver's zeros -- whatever relies on zeros it did not write shows
        static int poison = -1;
        if (poison < 0) { const char *ev = getenv("OATK_DEBUG_POISON"); poison = ev && ev[0] == '1'; }
        if (poison && e == hipSuccess) { (void) hipMemset(*p, 0xA5, bytes); (void) hipDeviceSynchronize(); }
    }
    return e;
}
static void dev_free(void *p, size_t bytes)
{
    const double t0 = dev_alloc_log()? dev_now() : 0;
    (void) hipFree(p);
    if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] %.3f hipFree   %10.3f MB: %.4f s\n", dev_now(), (double) bytes / 1e6, dev_now() - t0);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // the form in pieces: p is an address range of `va` bytes, its first ch.size() * DM_CHUNK bytes backed
    size_t va = 0;
    ChunkPool *pool = nullptr;
    bool decided = false;                     // whether this buffer lives in pieces was settled (at its first request of the threshold's size)
    std::vector<hipMemGenericAllocationHandle_t> ch;

    static size_t up(size_t b) { return (b + DM_CHUNK - 1) / DM_CHUNK * DM_CHUNK; }
    // an address range of at least `bytes`; what is mapped moves along (no copy).  false: nothing changed
    bool vm_range(size_t bytes, hipStream_t st)
    {
        if (bytes <= va) return true;
        const size_t nva = up(bytes < (1ull << 30)? 4 * bytes + (256ull << 20) : bytes + bytes / 2 + (2ull << 30));     // room to grow in place
        void *np = nullptr;
        { const hipError_t er = hipMemAddressReserve(&np, nva, 2 << 20, nullptr, 0);
          if (er != hipSuccess) { if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] hipMemAddressReserve of %.1f MB FAILED: %s\n", (double) nva / 1e6, hipGetErrorString(er)); (void) hipGetLastError(); return false; } }
        if (!ch.empty()) {
            (void) hipStreamSynchronize(st);
            (void) hipDeviceSynchronize();
            for (size_t i = 0; i < ch.size(); ++i) {
                (void) hipMemUnmap((char *) p + i * DM_CHUNK, DM_CHUNK);
                if (hipMemMap((char *) np + i * DM_CHUNK, DM_CHUNK, 0, ch[i], 0) != hipSuccess) return false;       // (cannot happen on a range just reserved)
            }
            if (hipMemSetAccess(np, ch.size() * DM_CHUNK, &pool->acc, 1) != hipSuccess) return false;
        }
        // An address range, once reserved, is never given back while the process lives -- not the one the pieces have just moved out of, not a released buffer's.
        // With hipMemAddressFree in either place a range reserved LATER (at the same addresses, presumably) showed other contents than were written to it: the
        // correction's results changed in 8 - 14 of 158 cases of tests/test_gpu_ec.py + levdist + light_graph + overlap run over pieces, every run, and in none with the
        // ranges kept (ROCm 7.0.2; translations of the old mapping that outlive it is the guess, not looked into further).  Address space is what this costs: a buffer's
        // range is a few times its size, a process of the CLI has some hundreds of such buffers in its life -- a terabyte of a 47-bit space at the outside.
        if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] %.3f address range %10.3f MB (%zu pieces moved)\n", dev_now(), (double) nva / 1e6, ch.size());
        p = np, va = nva;
        return true;
    }
    bool vm_grow(size_t bytes, hipStream_t st)
    {
        const size_t want = up(bytes);
        if (!vm_range(want, st)) return false;
        const size_t have = ch.size() * DM_CHUNK;
        const double t0 = dev_alloc_log()? dev_now() : 0;
        bool any_used = false;
        while (ch.size() * DM_CHUNK < want) {
            hipMemGenericAllocationHandle_t h;
            bool was_used;
            if (!pool->take(&h, &was_used)) { if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] no piece to be had\n"); break; }
            { const hipError_t er = hipMemMap((char *) p + ch.size() * DM_CHUNK, DM_CHUNK, 0, h, 0);
              if (er != hipSuccess) { if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] hipMemMap FAILED: %s\n", hipGetErrorString(er)); (void) hipGetLastError(); pool->give(h); break; } }
            ch.push_back(h);
            any_used |= was_used;
        }
        const size_t now_b = ch.size() * DM_CHUNK;
        if (now_b > have) { const hipError_t er = hipMemSetAccess((char *) p + have, now_b - have, &pool->acc, 1);
                            if (er != hipSuccess) { if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] hipMemSetAccess FAILED: %s\n", hipGetErrorString(er)); (void) hipGetLastError(); return false; } }
        // memory from hipMalloc is zero, always (the driver clears what it hands out): pieces that served another buffer are made so (5 TB/s: 13 us a piece)
        // -- and waited for: the buffer's first user may be a kernel on another stream than `st`
        if (any_used && now_b > have && (hipMemsetAsync((char *) p + have, 0, now_b - have, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) return false;
        if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] %.3f pieces    %10.3f MB -> %10.3f MB: %.4f s\n", dev_now(), (double) have / 1e6, (double) now_b / 1e6, dev_now() - t0);
        cap = now_b;
        return now_b >= want;
    }
    bool ensure(size_t bytes, hipStream_t st, bool zero_new = false)
    {
        if (bytes <= cap) return true;
        if (!va && !decided) pool = bytes >= dm_min()? pool_for_new_buffer() : nullptr, decided = bytes >= dm_min();
        if (pool && p && !va) { (void) hipStreamSynchronize(st); dev_free(p, cap); p = nullptr; cap = 0; }      // (a small hipMalloc'ed buffer that has outgrown the threshold)
        if (pool) {
            if (vm_grow(bytes + bytes / 16, st)) {
                if (zero_new) (void) hipMemsetAsync(p, 0, cap, st);
                return true;
            }
            if (!ch.empty()) return false;          // (out of memory with pieces in place)
            pool = nullptr, p = nullptr, cap = 0, va = 0;      // no range or no piece to start with: this buffer is hipMalloc's (a range that was reserved stays reserved)
        }
        if (p) { (void) hipStreamSynchronize(st); dev_free(p, cap); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (dev_malloc(&p, want) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        if (zero_new) (void) hipMemsetAsync(p, 0, want, st);
        return true;
    }
    // grow without losing the first `used` bytes (appending to a resident batch)
    bool grow_keep(size_t bytes, size_t used, hipStream_t st)
    {
        if (bytes <= cap) return true;
        if (!va && !decided) pool = bytes >= dm_min()? pool_for_new_buffer() : nullptr, decided = bytes >= dm_min();
        if (pool && p && !va) {                                     // from a hipMalloc'ed buffer to pieces: the one copy of this buffer's life
            void *old = p;
            const size_t old_cap = cap;
            p = nullptr, cap = 0;
            if (!vm_grow(bytes, st)) { release(); p = old, cap = old_cap, pool = nullptr; }
            else {
                if (used && hipMemcpyAsync(p, old, used, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
                (void) hipStreamSynchronize(st);
                dev_free(old, old_cap);
                return true;
            }
        }
        if (pool) return vm_grow(bytes, st);
        void *np = nullptr;
        size_t want = bytes + bytes / 4 + 256;
        if (dev_malloc(&np, want) != hipSuccess) return false;
        if (p && used && hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, st) != hipSuccess) { dev_free(np, want); return false; }
        (void) hipStreamSynchronize(st);
        if (p) dev_free(p, cap);
        p = np, cap = want;
        return true;
    }
    // room for `bytes` LATER: in pieces that is an address range and nothing else (the pieces come as the buffer fills); otherwise the memory itself, now
    bool reserve(size_t bytes, size_t used, hipStream_t st)
    {
        if (bytes <= cap) return true;
        if (!va && !p && !decided) pool = bytes >= dm_min()? pool_for_new_buffer() : nullptr, decided = bytes >= dm_min();
        if (pool && (va || !p)) return vm_range(up(bytes), st);
        return grow_keep(bytes, used, st);
    }
    void release()
    {
        if (va) {
            (void) hipDeviceSynchronize();
            for (size_t i = 0; i < ch.size(); ++i) { (void) hipMemUnmap((char *) p + i * DM_CHUNK, DM_CHUNK); pool->give(ch[i]); }
            ch.clear();
            // (the address range is NOT given back: see vm_range)
            if (dev_alloc_log()) fprintf(stderr, "[oatk alloc] %.3f pieces    %10.3f MB back to the pool\n", dev_now(), (double) cap / 1e6);
            p = nullptr, cap = 0, va = 0, decided = false;
            return;
        }
        if (p) dev_free(p, cap);
        p = nullptr; cap = 0, decided = false;
    }
    template <class T> T *as() const { return (T *) p; }
};

}  // namespace

struct oatk_hip_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    std::string err;
    bool timing = false;
    hipEvent_t ev[OATK_T_COUNT_ + 1][2];
    bool ev_used[OATK_T_COUNT_ + 1];
    float ms[OATK_T_COUNT_];
    uint64_t hash_mask = ~0ULL;
    bool force_general = false;   // test hook: run the general syncmer kernel even where the fast one applies
    void *staging = nullptr;      // page-locked host memory lent to callers (oatk_hip_staging)
    uint64_t staging_cap = 0;
    void *staging_raw = nullptr;  // the anonymous mapping `staging` lies in (nullptr: it came from hipHostMalloc)
    uint64_t staging_raw_size = 0;
    bool ra_two_pass = false;     // test hook: the read alignment counts, scans and runs again instead of writing into its pool
    int list_cap = 0;             // test hook: syncmers the fast kernel collects per read before writing records (0 = default)
    uint64_t import_reserve = 1u << 20;  // bytes kept free behind the hoco strings for k-mers imported from other shards (api_ec.inc)
    int ec_cap_t0 = 0, ec_cap_t1 = 0;   // test hook: block-length limits of the first two EC solver tiers (0 = default)

    // input (device view; owned only when uploaded through scan_host)
    const uint8_t *d_seq = nullptr;
    const uint64_t *d_off = nullptr;
    const uint32_t *d_len = nullptr;
    DevBuf in_seq, in_off, in_len;
    uint64_t n_reads = 0, seq_bytes = 0, sid0 = 0;
    int K = 0, S = 0;
    bool scanned = false, counted = false;

    // scan outputs
    DevBuf hoco_l, n_scm, n_nn, n_lrl, ho_rl, hoco_s, nbits;
    DevBuf nn_key, lrl_key, lrl_val, nn_key2, lrl_key2, lrl_val2;
    DevBuf rec_hash, rec_lo, rec_smer, rec_mpos;
    DevBuf raw_lo, raw_smer, raw_mpos, shard_cnt, shard_prefix;   // sharded append regions (scan_syncmer.hpp)
    uint32_t region_cap = 0;
    DevBuf counters;      // u32[4]
    uint32_t nn_cap = 0, lrl_cap = 0;
    uint64_t n_occ = 0, tot_nn = 0, tot_lrl = 0, n_scm_total = 0;
    uint32_t retries = 0, collisions = 0;
    bool nn_sorted_in_2 = false, lrl_sorted_in_2 = false;

    // count outputs / scratch
    DevBuf n_scm64, scm_off, pos_hash, pos_lo, pos_smer, pos_mpos, pos_kid;
    DevBuf key_hash, key_sorted, iota, perm, head, head_idx, newclus, clus_id, bad_head, tag, tmp_perm, flags, kloc, smer_sorted, slot_rec, scm_loc;
    DevBuf scm_h, scm_s, scm_cov, scm_occ_off, scm_occ;
    DevBuf tmp;           // rocprim temporary storage
    struct EcState *ec = nullptr;   // error-correction buffers (api_ec.inc)
    struct ConsState *cons = nullptr;   // consensus buffers (api_cons.inc)
    struct IngState *ing = nullptr;     // record scan buffers (api_ingest.inc)
    struct StatState *stat = nullptr;   // scan statistics buffers (api_stat.inc)
    struct AgState *ag = nullptr;       // assembly graph buffers (api_graph.inc)
    struct OvlState *ovl = nullptr;     // pair-distance tables (api_ovl.inc)
    struct RaState *ra = nullptr;       // read alignment buffers (api_align.inc)
    struct MultiState *multi = nullptr; // merged table / sharded correction (api_multi.inc)
};

#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return OATK_E_NODEV;                                                                   \
        }                                                                                          \
    } while (0)
#define ENSURE(buf, bytes, ...)                                                                    \
    do {                                                                                           \
        if (!ctx->buf.ensure((bytes), ctx->stream, ##__VA_ARGS__)) {                               \
            ctx->err = "hipMalloc failed for " #buf;                                               \
            return OATK_E_NOMEM;                                                                   \
        }                                                                                          \
    } while (0)

// Test hook (tests/test_gpu_cli.py): OATK_DEBUG_REFUSE=<bit mask> makes an entry point decline with OATK_E_SPLIT as if its input were one of the
// rare shapes it refuses -- 1 the EC graph (duplicate arcs), 2 the assembly graph, 4 the count (oversized hash group), 8 the pair-distance tables --
// so that a caller's fallback to the original routine can be exercised on inputs that do not produce those shapes.  Unset in production.
static bool debug_refuse(oatk_hip_ctx *ctx, int bit, const char *what)
{
    const char *e = getenv("OATK_DEBUG_REFUSE");
    if (!e || !(atoi(e) & bit)) return false;
    ctx->err = std::string(what) + ": refused on request (OATK_DEBUG_REFUSE)";
    return true;
}

static void t_begin(oatk_hip_ctx *ctx, int which)
{
    if (!ctx->timing) return;
    (void) hipEventRecord(ctx->ev[which][0], ctx->stream);
}
static void t_end(oatk_hip_ctx *ctx, int which)
{
    if (!ctx->timing) return;
    (void) hipEventRecord(ctx->ev[which][1], ctx->stream);
    ctx->ev_used[which] = true;
}
static void t_collect(oatk_hip_ctx *ctx, int first, int last)
{
    if (!ctx->timing) return;
    (void) hipStreamSynchronize(ctx->stream);
    for (int i = first; i <= last; ++i) {
        ctx->ms[i] = 0.f;
        if (ctx->ev_used[i]) (void) hipEventElapsedTime(&ctx->ms[i], ctx->ev[i][0], ctx->ev[i][1]);
        ctx->ev_used[i] = false;
    }
}

#include "api_ec.inc"
#include "api_graph.inc"
#include "api_cons.inc"
#include "api_ovl.inc"
#include "api_align.inc"
#include "api_ingest.inc"
#include "api_stat.inc"
#include "api_multi.inc"
#include "api_multi_tail.inc"

static void staging_free(oatk_hip_ctx *ctx)
{
    if (!ctx->staging) return;
    if (ctx->staging_raw) { (void) hipHostUnregister(ctx->staging); munmap(ctx->staging_raw, ctx->staging_raw_size); }
    else (void) hipHostFree(ctx->staging);
    ctx->staging = ctx->staging_raw = nullptr, ctx->staging_cap = ctx->staging_raw_size = 0;
}

extern "C" {

int oatk_hip_abi_version(void) { return OATK_HIP_ABI_VERSION; }

int oatk_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int oatk_hip_max_k(void) { return 8192 - 16 * oatk::SYN_NT - 16 - 64; }

oatk_hip_ctx *oatk_hip_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    oatk_hip_ctx *ctx = new oatk_hip_ctx();
    ctx->device = device;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) ctx->n_cu = cu; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
    for (int i = 0; i <= OATK_T_COUNT_; ++i) {
        (void) hipEventCreate(&ctx->ev[i][0]);
        (void) hipEventCreate(&ctx->ev[i][1]);
        ctx->ev_used[i] = false;
    }
    memset(ctx->ms, 0, sizeof(ctx->ms));
    return ctx;
}

void oatk_hip_destroy(oatk_hip_ctx *ctx)
{
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    DevBuf *all[] = {&ctx->in_seq, &ctx->in_off, &ctx->in_len, &ctx->hoco_l, &ctx->n_scm, &ctx->n_nn, &ctx->n_lrl, &ctx->ho_rl,
                     &ctx->hoco_s, &ctx->nbits, &ctx->nn_key, &ctx->lrl_key, &ctx->lrl_val, &ctx->nn_key2, &ctx->lrl_key2,
                     &ctx->lrl_val2, &ctx->rec_hash, &ctx->rec_lo, &ctx->rec_smer, &ctx->rec_mpos, &ctx->raw_lo, &ctx->raw_smer, &ctx->raw_mpos, &ctx->shard_cnt, &ctx->shard_prefix, &ctx->counters, &ctx->n_scm64,
                     &ctx->scm_off, &ctx->pos_hash, &ctx->pos_lo, &ctx->pos_smer, &ctx->pos_mpos, &ctx->pos_kid, &ctx->key_hash,
                     &ctx->key_sorted, &ctx->iota, &ctx->perm, &ctx->head, &ctx->head_idx, &ctx->newclus, &ctx->clus_id, &ctx->kloc, &ctx->smer_sorted, &ctx->slot_rec, &ctx->scm_loc,
                     &ctx->bad_head, &ctx->tag, &ctx->tmp_perm, &ctx->flags, &ctx->scm_h, &ctx->scm_s, &ctx->scm_cov,
                     &ctx->scm_occ_off, &ctx->scm_occ, &ctx->tmp};
    for (DevBuf *b : all) b->release();
    ec_state_free(ctx);
    cons_state_free(ctx);
    ing_state_free(ctx);
    stat_state_free(ctx);
    ag_state_free(ctx);
    ovl_state_free(ctx);
    ra_state_free(ctx);
    multi_state_free(ctx);
    for (int i = 0; i <= OATK_T_COUNT_; ++i) {
        (void) hipEventDestroy(ctx->ev[i][0]);
        (void) hipEventDestroy(ctx->ev[i][1]);
    }
    staging_free(ctx);
    (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int oatk_hip_mem_pool(oatk_hip_ctx *ctx, uint64_t warm_bytes)
{
    if (!ctx) return OATK_E_NODEV;
    { const char *e = getenv("OATK_POOL"); if (e && e[0] == '0') return OATK_OK; }      // (switch for comparisons: everything stays hipMalloc's)
    CK(hipSetDevice(ctx->device));
    if (ctx->device < 0 || ctx->device >= 64) return OATK_E_ARG;
    ChunkPool *pl;
    {
        std::unique_lock<std::mutex> lk(g_pool_mu);
        pl = g_pool[ctx->device];
        if (!pl) {
            int vmm = 0;
            if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess || !vmm) { (void) hipGetLastError(); return OATK_OK; }
            pl = new ChunkPool();
            pl->device = ctx->device;
            memset(&pl->prop, 0, sizeof(pl->prop));
            pl->prop.type = hipMemAllocationTypePinned;
            pl->prop.location.type = hipMemLocationTypeDevice;
            pl->prop.location.id = ctx->device;
            memset(&pl->acc, 0, sizeof(pl->acc));
            pl->acc.location = pl->prop.location;
            pl->acc.flags = hipMemAccessFlagsProtReadWrite;
            static bool at_exit = false;
            if (!at_exit) { at_exit = true; atexit(pools_end); }
            pl->on = true;
            g_pool[ctx->device] = pl;
        }
    }
    if (warm_bytes) pl->warm((size_t) warm_bytes);
    return OATK_OK;
}

const char *oatk_hip_last_error(oatk_hip_ctx *ctx) { return ctx? ctx->err.c_str() : "no device context"; }
void *oatk_hip_stream(oatk_hip_ctx *ctx) { return ctx? (void *) ctx->stream : nullptr; }
int oatk_hip_sync(oatk_hip_ctx *ctx)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipStreamSynchronize(ctx->stream));
    return OATK_OK;
}
int oatk_hip_set_timing(oatk_hip_ctx *ctx, int enable)
{
    if (!ctx) return OATK_E_NODEV;
    ctx->timing = enable != 0;
    return OATK_OK;
}
int oatk_hip_get_timing(oatk_hip_ctx *ctx, float *ms, int n)
{
    if (!ctx) return OATK_E_NODEV;
    for (int i = 0; i < n && i < OATK_T_COUNT_; ++i) ms[i] = ctx->ms[i];
    return OATK_OK;
}
int oatk_hip_debug_force_general(oatk_hip_ctx *ctx, int on)
{
    if (!ctx) return OATK_E_NODEV;
    ctx->force_general = on != 0;
    return OATK_OK;
}
int oatk_hip_debug_align_two_pass(oatk_hip_ctx *ctx, int on)
{
    if (!ctx) return OATK_E_NODEV;
    ctx->ra_two_pass = on != 0;
    return OATK_OK;
}
int oatk_hip_debug_list_cap(oatk_hip_ctx *ctx, int cap)
{
    if (!ctx) return OATK_E_NODEV;
    if (cap < 0 || cap > oatk::SYF_LIST) { ctx->err = "oatk_hip_debug_list_cap: 0 <= cap <= 128"; return OATK_E_ARG; }
    ctx->list_cap = cap;
    return OATK_OK;
}
int oatk_hip_debug_hash_mask(oatk_hip_ctx *ctx, uint64_t mask)
{
    if (!ctx) return OATK_E_NODEV;
    ctx->hash_mask = mask;
    return OATK_OK;
}

__global__ void widen_kernel(const uint32_t *in, uint64_t *out, uint64_t n)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
    if (i == n) out[i] = 0;
}

static void scan_reset_downstream(oatk_hip_ctx *ctx);

static int launch_scan_kernels(oatk_hip_ctx *ctx)
{
    using namespace oatk;
    const uint64_t n = ctx->n_reads;
    CK(hipMemsetAsync(ctx->counters.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    CK(hipMemsetAsync(ctx->shard_cnt.p, 0, oatk::OATK_REC_SHARDS * sizeof(uint32_t), ctx->stream));
    HpcArgs h;
    h.seq = ctx->d_seq, h.off = ctx->d_off, h.len = ctx->d_len, h.sid0 = ctx->sid0;
    h.ho_rl = ctx->ho_rl.as<uint8_t>(), h.hoco_s = ctx->hoco_s.as<uint8_t>(), h.nbits = ctx->nbits.as<uint32_t>();
    h.hoco_l = ctx->hoco_l.as<uint32_t>(), h.n_nn = ctx->n_nn.as<uint32_t>(), h.n_lrl = ctx->n_lrl.as<uint32_t>();
    h.nn_key = ctx->nn_key.as<uint64_t>(), h.lrl_key = ctx->lrl_key.as<uint64_t>(), h.lrl_val = ctx->lrl_val.as<uint32_t>();
    h.nn_cap = ctx->nn_cap, h.lrl_cap = ctx->lrl_cap, h.counters = ctx->counters.as<uint32_t>();
    t_begin(ctx, OATK_T_HPC);
    h.n_reads = (uint32_t) n;
    constexpr int HPC_NW = 4;                                 // waves per workgroup, each on reads of its own (16.3 KB of LDS: eight workgroups = 32 waves per CU)
    uint64_t hpc_grid = (uint64_t) ctx->n_cu * 8 * 8;         // the waves stride over the reads: 32 resident per CU, eight rounds of them to even out read lengths
                                                              // (400 k reads: 4096 workgroups 3.50 ms, 8192 3.10, 16384 2.94, 32768 2.91, 65536 2.94)
    { const char *ev = getenv("OATK_DEBUG_HPC_GRID"); if (ev && atoi(ev) > 0) hpc_grid = (uint64_t) atoi(ev); }
    if (hpc_grid * HPC_NW > n) hpc_grid = (n + HPC_NW - 1) / HPC_NW;
    unsigned hpc_dyn = 0;                          // development aid: unused LDS on top of the kernel's own lowers its residency
    { const char *ev = getenv("OATK_DEBUG_HPC_LDS"); if (ev && atoi(ev) > 0) hpc_dyn = (unsigned) atoi(ev); }
    hipLaunchKernelGGL((hpc_pack_kernel<HPC_NW>), dim3((unsigned) (hpc_grid? hpc_grid : 1)), dim3(HPC_NW * OATK_WAVE), hpc_dyn, ctx->stream, h);
    t_end(ctx, OATK_T_HPC);

    SynArgs s;
    s.hoco_s = ctx->hoco_s.as<uint8_t>(), s.nbits = ctx->nbits.as<uint32_t>(), s.off = ctx->d_off;
    s.hoco_l = ctx->hoco_l.as<uint32_t>(), s.n_nn = ctx->n_nn.as<uint32_t>(), s.sid0 = ctx->sid0;
    s.K = ctx->K, s.S = ctx->S, s.want_n = 0, s.n_scm = ctx->n_scm.as<uint32_t>(), s.list_cap = ctx->list_cap? ctx->list_cap : SYF_LIST;
    s.rec_hash = nullptr, s.rec_lo = ctx->raw_lo.as<uint64_t>(), s.rec_smer = ctx->raw_smer.as<uint64_t>();
    s.rec_mpos = ctx->raw_mpos.as<uint32_t>(), s.region_cap = ctx->region_cap, s.shard_cnt = ctx->shard_cnt.as<uint32_t>();
    const bool small = ctx->K + 8 * SYN_NT + 8 + 64 <= 4096;
    const int fast_ring = ctx->force_general? 0 : syncmer_fast_ring(ctx->K, ctx->S);
    t_begin(ctx, OATK_T_SYNCMER);
    if (fast_ring == 4096) {
        const dim3 g((unsigned) n), b(SYN_NT);
        unsigned dyn = 0;                          // development aid: unused LDS on top of the kernel's own lowers its residency
        { const char *ev = getenv("OATK_DEBUG_SYNCMER_LDS"); if (ev && atoi(ev) > 0) dyn = (unsigned) atoi(ev); }
        // one instantiation per alignment of the window start against the chunks of 8 (the offsets of the decision's ring reads are constants)
#define OATK_SYF_SWITCH(S31) switch ((-(ctx->K - ctx->S)) & 7) { \
            case 0: OATK_SYF_LAUNCH(S31, 0); break; case 1: OATK_SYF_LAUNCH(S31, 1); break; case 2: OATK_SYF_LAUNCH(S31, 2); break; \
            case 3: OATK_SYF_LAUNCH(S31, 3); break; case 4: OATK_SYF_LAUNCH(S31, 4); break; case 5: OATK_SYF_LAUNCH(S31, 5); break; \
            case 6: OATK_SYF_LAUNCH(S31, 6); break; default: OATK_SYF_LAUNCH(S31, 7); break; }
        // Two forms: two waves per workgroup on a 2048-slot ring (tiles of 1024 positions) wherever the window fits it (K - S <= 1023), four waves on 4096
        // slots otherwise.  A read's last tile costs a tile's time however little of it lies inside the read -- half a tile per read on average, 8 % of
        // the kernel with tiles of 2048 --, and two waves meet at a barrier sooner than four: 8.93 -> 8.37 ms at 400 k reads (r03p).
        const bool two_waves = ctx->K - ctx->S < 1024 && getenv("OATK_DEBUG_SYNCMER_NT256") == nullptr;
#define OATK_SYF_LAUNCH(S31, SH) do { if (two_waves) hipLaunchKernelGGL((syncmer_fast_kernel<2048, S31, 128, SH>), g, dim3(128), dyn, ctx->stream, s); \
                                      else hipLaunchKernelGGL((syncmer_fast_kernel<4096, S31, SYN_NT, SH>), g, b, dyn, ctx->stream, s); } while (0)
        if (ctx->S == 31) OATK_SYF_SWITCH(true) else OATK_SYF_SWITCH(false)
#undef OATK_SYF_SWITCH
#undef OATK_SYF_LAUNCH
    }
    else if (small) hipLaunchKernelGGL((syncmer_kernel<8, 4096, false>), dim3((unsigned) n), dim3(SYN_NT), 0, ctx->stream, s);
    else hipLaunchKernelGGL((syncmer_kernel<16, 8192, false>), dim3((unsigned) n), dim3(SYN_NT), 0, ctx->stream, s);
    t_end(ctx, OATK_T_SYNCMER);
    // reads with ambiguous bases take the general kernel -- if there are any: kernel A has counted them, and the caller waits for these
    // kernels anyway before it reads the record counts, so asking here costs nothing (a launch over 200 k reads that all return at once: 0.14 ms)
    uint32_t n_amb = 0;
    CK(hipMemcpyAsync(&n_amb, ctx->counters.as<uint32_t>(), 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    s.want_n = 1;
    t_begin(ctx, OATK_T_SYNCMER_N);
    if (n_amb) {
        if (small) hipLaunchKernelGGL((syncmer_kernel<8, 4096, true>), dim3((unsigned) n), dim3(SYN_NT), 0, ctx->stream, s);
        else hipLaunchKernelGGL((syncmer_kernel<16, 8192, true>), dim3((unsigned) n), dim3(SYN_NT), 0, ctx->stream, s);
    }
    t_end(ctx, OATK_T_SYNCMER_N);
    CK(hipGetLastError());
    return OATK_OK;
}

int oatk_hip_scan(oatk_hip_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint32_t *d_len,
                  uint64_t n_reads, uint64_t seq_bytes, uint64_t sid0, int k, int s)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    if (!(s > 0 && s < 32 && k > s) || k > oatk_hip_max_k()) { ctx->err = "k/s out of range for the device scan"; return OATK_E_ARG; }
    if (seq_bytes % OATK_READ_ALIGN || n_reads >= 0xFFFFFFFFULL || sid0 + n_reads > 0xFFFFFFFFULL) { ctx->err = "bad batch geometry"; return OATK_E_ARG; }
    ctx->d_seq = d_seq, ctx->d_off = d_off, ctx->d_len = d_len;
    ctx->n_reads = n_reads, ctx->seq_bytes = seq_bytes, ctx->sid0 = sid0, ctx->K = k, ctx->S = s;
    scan_reset_downstream(ctx);                                           // results of the previous batch
    ctx->retries = 0, ctx->collisions = 0;
    ctx->n_occ = ctx->tot_nn = ctx->tot_lrl = ctx->n_scm_total = 0;
    if (n_reads == 0) { ctx->scanned = true; return OATK_OK; }

    ENSURE(hoco_l, n_reads * 4); ENSURE(n_scm, n_reads * 4); ENSURE(n_nn, n_reads * 4); ENSURE(n_lrl, n_reads * 4);
    ENSURE(ho_rl, seq_bytes + 64);
    ENSURE(hoco_s, seq_bytes / 4 + 128 + 16 + ctx->import_reserve);
    ENSURE(nbits, seq_bytes / 8 + 128, true);          // all-zero invariant between scans; kernel B hands it back clean
    ENSURE(counters, 64);
    if (ctx->nn_cap == 0) ctx->nn_cap = 1u << 14;
    if (ctx->lrl_cap == 0) ctx->lrl_cap = 1u << 14;
    const uint32_t NSH = oatk::OATK_REC_SHARDS;
    {   // ~2.5x the expected density at k=1001, split over the shards (reads are dealt round-robin, so shards stay balanced)
        uint64_t guess = (seq_bytes / 256 + 1024) / NSH + 256;
        if (ctx->region_cap < guess) ctx->region_cap = (uint32_t) guess;
    }
    ENSURE(shard_cnt, NSH * 4); ENSURE(shard_prefix, NSH * 8);
    uint32_t sc[oatk::OATK_REC_SHARDS];
    uint64_t spfx[oatk::OATK_REC_SHARDS];

    for (;;) {
        const size_t raw_n = (size_t) ctx->region_cap * NSH;
        ENSURE(raw_lo, raw_n * 8); ENSURE(raw_smer, raw_n * 8); ENSURE(raw_mpos, raw_n * 4);
        ENSURE(nn_key, (size_t) ctx->nn_cap * 8); ENSURE(nn_key2, (size_t) ctx->nn_cap * 8);
        ENSURE(lrl_key, (size_t) ctx->lrl_cap * 8); ENSURE(lrl_key2, (size_t) ctx->lrl_cap * 8);
        ENSURE(lrl_val, (size_t) ctx->lrl_cap * 4); ENSURE(lrl_val2, (size_t) ctx->lrl_cap * 4);
        int rc = launch_scan_kernels(ctx);
        if (rc) return rc;
        uint32_t c[4];
        CK(hipMemcpyAsync(c, ctx->counters.p, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
        CK(hipMemcpyAsync(sc, ctx->shard_cnt.p, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        bool again = false;
        if (c[0] > ctx->nn_cap) ctx->nn_cap = c[0] + c[0] / 4, again = true;
        if (c[1] > ctx->lrl_cap) ctx->lrl_cap = c[1] + c[1] / 4, again = true;
        uint64_t tot = 0;
        uint32_t mx = 0;
        for (uint32_t i = 0; i < NSH; ++i) { spfx[i] = tot; tot += sc[i]; if (sc[i] > mx) mx = sc[i]; }
        if (mx > ctx->region_cap) ctx->region_cap = mx + mx / 4 + 16, again = true;
        if (tot > 0xFFFFFFF0ULL) { ctx->err = "more than 2^32 syncmer occurrences in one batch"; return OATK_E_ARG; }
        ctx->tot_nn = c[0], ctx->tot_lrl = c[1], ctx->n_occ = tot;
        if (!again) break;
        ++ctx->retries;
        // a partial pass may have left ambiguity bits behind: restore the all-zero invariant
        CK(hipMemsetAsync(ctx->nbits.p, 0, ctx->nbits.cap, ctx->stream));
    }
    // shard regions -> dense record arrays
    if (ctx->n_occ) {
        ENSURE(rec_hash, (size_t) ctx->n_occ * 8); ENSURE(rec_lo, (size_t) ctx->n_occ * 8);
        ENSURE(rec_smer, (size_t) ctx->n_occ * 8); ENSURE(rec_mpos, (size_t) ctx->n_occ * 4);
        CK(hipMemcpyAsync(ctx->shard_prefix.p, spfx, sizeof(spfx), hipMemcpyHostToDevice, ctx->stream));
        oatk::CompactArgs ca;
        ca.raw_lo = ctx->raw_lo.as<uint64_t>(), ca.raw_smer = ctx->raw_smer.as<uint64_t>(), ca.raw_mpos = ctx->raw_mpos.as<uint32_t>();
        ca.shard_cnt = ctx->shard_cnt.as<uint32_t>(), ca.shard_prefix = ctx->shard_prefix.as<uint64_t>(), ca.region_cap = ctx->region_cap;
        ca.rec_lo = ctx->rec_lo.as<uint64_t>(), ca.rec_smer = ctx->rec_smer.as<uint64_t>(), ca.rec_mpos = ctx->rec_mpos.as<uint32_t>();
        hipLaunchKernelGGL(oatk::compact_records_kernel, dim3(NSH), dim3(256), 0, ctx->stream, ca);
        CK(hipStreamSynchronize(ctx->stream));     // spfx lives on this stack frame
    }
    if (ctx->timing) t_collect(ctx, OATK_T_HPC, OATK_T_SYNCMER_N);

    // ---- k-mer hashes of all records, one lane per syncmer (kmer_hash.hpp) ----
    if (ctx->n_occ) {
        oatk::KmerHashArgs kh;
        kh.hoco_s = ctx->hoco_s.as<uint8_t>(), kh.off = ctx->d_off, kh.sid0 = ctx->sid0;
        kh.rec_lo = ctx->rec_lo.as<uint64_t>(), kh.rec_mpos = ctx->rec_mpos.as<uint32_t>(), kh.rec_hash = ctx->rec_hash.as<uint64_t>();
        kh.n_rec = (uint32_t) ctx->n_occ, kh.K = ctx->K;
        const int nw = ((ctx->K - 1) / 4 + 1 + 7) / 8;
        t_begin(ctx, OATK_T_KMER_HASH);
        hipLaunchKernelGGL(oatk::kmer_hash_kernel, dim3((unsigned) ((ctx->n_occ + KMH_REC - 1) / KMH_REC)), dim3(64), (size_t) KMH_REC * (nw + 1) * 8, ctx->stream, kh);
        t_end(ctx, OATK_T_KMER_HASH);
    }

    // ---- post: order the rare-event lists, per-read slot offsets ----
    t_begin(ctx, OATK_T_SCAN_POST);
    ctx->nn_sorted_in_2 = ctx->lrl_sorted_in_2 = false;
    if (ctx->tot_nn > 1) {
        size_t tb = 0;
        CK(rocprim::radix_sort_keys(nullptr, tb, ctx->nn_key.as<uint64_t>(), ctx->nn_key2.as<uint64_t>(), ctx->tot_nn, 0, 64, ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::radix_sort_keys(ctx->tmp.p, tb, ctx->nn_key.as<uint64_t>(), ctx->nn_key2.as<uint64_t>(), ctx->tot_nn, 0, 64, ctx->stream));
        ctx->nn_sorted_in_2 = true;
    }
    if (ctx->tot_lrl > 1) {
        size_t tb = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb, ctx->lrl_key.as<uint64_t>(), ctx->lrl_key2.as<uint64_t>(), ctx->lrl_val.as<uint32_t>(),
                                     ctx->lrl_val2.as<uint32_t>(), ctx->tot_lrl, 0, 64, ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::radix_sort_pairs(ctx->tmp.p, tb, ctx->lrl_key.as<uint64_t>(), ctx->lrl_key2.as<uint64_t>(), ctx->lrl_val.as<uint32_t>(),
                                     ctx->lrl_val2.as<uint32_t>(), ctx->tot_lrl, 0, 64, ctx->stream));
        ctx->lrl_sorted_in_2 = true;
    }
    {
        ENSURE(n_scm64, (n_reads + 1) * 8); ENSURE(scm_off, (n_reads + 1) * 8);
        unsigned nb = (unsigned) ((n_reads + 1 + 255) / 256);
        hipLaunchKernelGGL(widen_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->n_scm.as<uint32_t>(), ctx->n_scm64.as<uint64_t>(), n_reads);
        size_t tb = 0;
        CK(rocprim::exclusive_scan(nullptr, tb, ctx->n_scm64.as<uint64_t>(), ctx->scm_off.as<uint64_t>(), (uint64_t) 0, n_reads + 1,
                                   rocprim::plus<uint64_t>(), ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::exclusive_scan(ctx->tmp.p, tb, ctx->n_scm64.as<uint64_t>(), ctx->scm_off.as<uint64_t>(), (uint64_t) 0, n_reads + 1,
                                   rocprim::plus<uint64_t>(), ctx->stream));
    }
    // records -> per-read slots (also the low-key order the count needs)
    {
        using namespace oatk;
        size_t n = ctx->n_occ? ctx->n_occ : 1;
        ENSURE(pos_hash, n * 8); ENSURE(pos_lo, n * 8); ENSURE(pos_smer, n * 8); ENSURE(pos_mpos, n * 4);
        ENSURE(key_hash, n * 8); ENSURE(iota, n * 4);
        if (ctx->n_occ) {
            PlaceArgs p;
            p.rec_hash = ctx->rec_hash.as<uint64_t>(), p.rec_lo = ctx->rec_lo.as<uint64_t>(), p.rec_smer = ctx->rec_smer.as<uint64_t>();
            p.rec_mpos = ctx->rec_mpos.as<uint32_t>(), p.n_rec = (uint32_t) ctx->n_occ, p.sid0 = ctx->sid0;
            p.scm_off = ctx->scm_off.as<uint64_t>(), p.hash_mask = ctx->hash_mask;
            p.pos_hash = ctx->pos_hash.as<uint64_t>(), p.pos_lo = ctx->pos_lo.as<uint64_t>(), p.pos_smer = ctx->pos_smer.as<uint64_t>();
            p.pos_mpos = ctx->pos_mpos.as<uint32_t>(), p.key_hash = ctx->key_hash.as<uint64_t>(), p.iota = ctx->iota.as<uint32_t>();
            t_begin(ctx, OATK_T_COUNT_PLACE);
            hipLaunchKernelGGL(place_records_kernel, dim3((unsigned) ((ctx->n_occ + 255) / 256)), dim3(256), 0, ctx->stream, p);
            t_end(ctx, OATK_T_COUNT_PLACE);
        }
    }
    t_end(ctx, OATK_T_SCAN_POST);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->timing) { t_collect(ctx, OATK_T_SCAN_POST, OATK_T_COUNT_PLACE); t_collect(ctx, OATK_T_KMER_HASH, OATK_T_KMER_HASH); }
    ctx->scanned = true;
    return OATK_OK;
}

// ---- a batch assembled from scanned pieces (include/oatk_hip.h) ----
__global__ void append_rebase_kernel(const uint64_t *src, uint64_t *dst, uint64_t n, uint64_t add)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}
__global__ void append_iota_kernel(uint32_t *iota, uint64_t first, uint64_t n)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) iota[first + i] = (uint32_t) (first + i);
}

static void scan_reset_downstream(oatk_hip_ctx *ctx)
{
    ctx->scanned = ctx->counted = false;
    if (ctx->ec) ctx->ec->done = ctx->ec->marked = ctx->ec->graph_resident = ctx->ec->global = false;
    if (ctx->cons) ctx->cons->done = false;
    if (ctx->ag) ctx->ag->done = false;
    if (ctx->ovl) ctx->ovl->done = false;
    if (ctx->ra) ctx->ra->done = false;
    if (ctx->multi) ctx->multi->merged = ctx->multi->ec_done = false;      // a merged table belongs to the batch it was merged for
}

int oatk_hip_device(oatk_hip_ctx *ctx) { return ctx? ctx->device : -1; }

int oatk_hip_d2d(oatk_hip_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (bytes == 0) return OATK_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return OATK_OK;
}

int oatk_hip_scan_begin(oatk_hip_ctx *ctx, uint64_t sid0, int k, int s)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    if (!(s > 0 && s < 32 && k > s) || k > oatk_hip_max_k()) { ctx->err = "k/s out of range for the device scan"; return OATK_E_ARG; }
    scan_reset_downstream(ctx);
    ctx->d_seq = nullptr, ctx->d_off = nullptr, ctx->d_len = nullptr;
    ctx->n_reads = 0, ctx->seq_bytes = 0, ctx->sid0 = sid0, ctx->K = k, ctx->S = s;
    ctx->retries = 0, ctx->collisions = 0;
    ctx->n_occ = ctx->tot_nn = ctx->tot_lrl = ctx->n_scm_total = 0;
    ctx->nn_sorted_in_2 = ctx->lrl_sorted_in_2 = false;
    ctx->scanned = true;                       // an empty batch is a valid batch
    return OATK_OK;
}

#define KEEP(buf, bytes, used)                                                                     \
    do {                                                                                           \
        if (!ctx->buf.grow_keep((bytes), (used), ctx->stream)) {                                   \
            ctx->err = "hipMalloc failed for " #buf " (append)";                                   \
            return OATK_E_NOMEM;                                                                   \
        }                                                                                          \
    } while (0)

static int append_ensure(oatk_hip_ctx *ctx, uint64_t nr, uint64_t sb, uint64_t occ, uint64_t nn, uint64_t lrl)
{
    const uint64_t n0 = ctx->n_reads, b0 = ctx->seq_bytes, o0 = ctx->n_occ;
    KEEP(in_off, (nr + 1) * 8, n0 * 8);
    KEEP(hoco_l, nr * 4 + 4, n0 * 4); KEEP(n_scm, nr * 4 + 4, n0 * 4); KEEP(n_nn, nr * 4 + 4, n0 * 4); KEEP(n_lrl, nr * 4 + 4, n0 * 4);
    KEEP(ho_rl, sb + 64, b0);
    KEEP(hoco_s, sb / 4 + 128 + 16 + ctx->import_reserve, b0 / 4 + 64);
    KEEP(scm_off, (nr + 1) * 8, (n0 + 1) * 8);
    KEEP(pos_hash, occ * 8 + 8, o0 * 8); KEEP(pos_lo, occ * 8 + 8, o0 * 8); KEEP(pos_smer, occ * 8 + 8, o0 * 8); KEEP(pos_mpos, occ * 4 + 4, o0 * 4);
    KEEP(key_hash, occ * 8 + 8, o0 * 8); KEEP(iota, occ * 4 + 4, o0 * 4);
    // the rare-event lists live in the "2" buffers of a finished scan when they were sorted; an assembled batch keeps them in the plain ones
    KEEP(nn_key, nn * 8 + 8, ctx->tot_nn * 8); KEEP(lrl_key, lrl * 8 + 8, ctx->tot_lrl * 8); KEEP(lrl_val, lrl * 4 + 4, ctx->tot_lrl * 4);
    return OATK_OK;
}

int oatk_hip_scan_reserve(oatk_hip_ctx *ctx, uint64_t seq_bytes, uint64_t n_reads, uint64_t n_occ)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    if (!ctx->scanned || ctx->d_seq) { ctx->err = "oatk_hip_scan_reserve: call oatk_hip_scan_begin first"; return OATK_E_STATE; }
    seq_bytes = (seq_bytes + 63) & ~63ULL;
    if (pool_of_current_device()) {
        // in pieces (oatk_hip_mem_pool) the two large arrays only get their address ranges here; the pieces are mapped append by append, as the pool's thread delivers them
        const uint64_t sb = ctx->seq_bytes + seq_bytes;
        if (!ctx->ho_rl.reserve(sb + 64, ctx->seq_bytes, ctx->stream) || !ctx->hoco_s.reserve(sb / 4 + 128 + 16 + ctx->import_reserve, ctx->seq_bytes / 4 + 64, ctx->stream)) {
            ctx->err = "oatk_hip_scan_reserve: no address range for the batch";
            return OATK_E_NOMEM;
        }
        return OATK_OK;
    }
    return append_ensure(ctx, ctx->n_reads + n_reads, ctx->seq_bytes + seq_bytes, ctx->n_occ + n_occ, ctx->tot_nn, ctx->tot_lrl);
}

int oatk_hip_scan_append(oatk_hip_ctx *ctx, oatk_hip_ctx *src)
{
    if (!ctx || !src) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    if (ctx == src || ctx->device != src->device) { ctx->err = "oatk_hip_scan_append: the piece must be another handle on the same device"; return OATK_E_ARG; }
    if (!ctx->scanned || ctx->d_seq) { ctx->err = "oatk_hip_scan_append: call oatk_hip_scan_begin first"; return OATK_E_STATE; }
    if (!src->scanned) { ctx->err = "oatk_hip_scan_append: the piece holds no scan"; return OATK_E_STATE; }
    const uint64_t n0 = ctx->n_reads, n1 = src->n_reads, b0 = ctx->seq_bytes, b1 = src->seq_bytes, o0 = ctx->n_occ, o1 = src->n_occ;
    if (n1 == 0) return OATK_OK;
    if (src->K != ctx->K || src->S != ctx->S) { ctx->err = "oatk_hip_scan_append: the piece was scanned with another k / s"; return OATK_E_ARG; }
    if (src->sid0 != ctx->sid0 + n0) { ctx->err = "oatk_hip_scan_append: the piece's first read id must continue the batch"; return OATK_E_ARG; }
    if (n0 + n1 >= 0xFFFFFFFFULL || o0 + o1 > 0xFFFFFFF0ULL) { ctx->err = "oatk_hip_scan_append: batch too large"; return OATK_E_ARG; }
    scan_reset_downstream(ctx);
    ctx->scanned = true;
    { int rc = append_ensure(ctx, n0 + n1, b0 + b1, o0 + o1, ctx->tot_nn + src->tot_nn, ctx->tot_lrl + src->tot_lrl); if (rc) return rc; }
    if (hipStreamSynchronize(src->stream) != hipSuccess) { ctx->err = "oatk_hip_scan_append: the piece's stream failed"; return OATK_E_NODEV; }
    hipStream_t st = ctx->stream;
    auto blocks = [](uint64_t n) { return dim3((unsigned) ((n + 255) / 256 > 0? (n + 255) / 256 : 1)); };
    auto d2d = [&](void *dst, const void *from, size_t bytes) { return bytes? hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, st) : hipSuccess; };
    CK(d2d(ctx->ho_rl.as<uint8_t>() + b0, src->ho_rl.p, b1));
    CK(d2d(ctx->hoco_s.as<uint8_t>() + b0 / 4, src->hoco_s.p, b1 / 4 + 64));
    CK(d2d(ctx->hoco_l.as<uint32_t>() + n0, src->hoco_l.p, n1 * 4)); CK(d2d(ctx->n_scm.as<uint32_t>() + n0, src->n_scm.p, n1 * 4));
    CK(d2d(ctx->n_nn.as<uint32_t>() + n0, src->n_nn.p, n1 * 4)); CK(d2d(ctx->n_lrl.as<uint32_t>() + n0, src->n_lrl.p, n1 * 4));
    hipLaunchKernelGGL(append_rebase_kernel, blocks(n1), dim3(256), 0, st, src->d_off, ctx->in_off.as<uint64_t>() + n0, n1, b0);
    hipLaunchKernelGGL(append_rebase_kernel, blocks(n1 + 1), dim3(256), 0, st, src->scm_off.as<uint64_t>(), ctx->scm_off.as<uint64_t>() + n0, n1 + 1, o0);
    if (o1) {
        CK(d2d(ctx->pos_hash.as<uint64_t>() + o0, src->pos_hash.p, o1 * 8)); CK(d2d(ctx->pos_lo.as<uint64_t>() + o0, src->pos_lo.p, o1 * 8));
        CK(d2d(ctx->pos_smer.as<uint64_t>() + o0, src->pos_smer.p, o1 * 8)); CK(d2d(ctx->pos_mpos.as<uint32_t>() + o0, src->pos_mpos.p, o1 * 4));
        CK(d2d(ctx->key_hash.as<uint64_t>() + o0, src->key_hash.p, o1 * 8));
        hipLaunchKernelGGL(append_iota_kernel, blocks(o1), dim3(256), 0, st, ctx->iota.as<uint32_t>(), o0, o1);
    }
    // keys carry global read ids and the pieces come in read order: the concatenation is sorted
    CK(d2d(ctx->nn_key.as<uint64_t>() + ctx->tot_nn, src->nn_sorted_in_2? src->nn_key2.p : src->nn_key.p, src->tot_nn * 8));
    CK(d2d(ctx->lrl_key.as<uint64_t>() + ctx->tot_lrl, src->lrl_sorted_in_2? src->lrl_key2.p : src->lrl_key.p, src->tot_lrl * 8));
    CK(d2d(ctx->lrl_val.as<uint32_t>() + ctx->tot_lrl, src->lrl_sorted_in_2? src->lrl_val2.p : src->lrl_val.p, src->tot_lrl * 4));
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    ctx->d_off = ctx->in_off.as<uint64_t>();
    ctx->n_reads = n0 + n1, ctx->seq_bytes = b0 + b1, ctx->n_occ = o0 + o1;
    ctx->tot_nn += src->tot_nn, ctx->tot_lrl += src->tot_lrl;
    ctx->retries += src->retries;
    return OATK_OK;
}

int oatk_hip_scan_host(oatk_hip_ctx *ctx, const uint8_t *h_seq, const uint64_t *h_off, const uint32_t *h_len,
                       uint64_t n_reads, uint64_t seq_bytes, uint64_t sid0, int k, int s)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    ENSURE(in_seq, seq_bytes + 64); ENSURE(in_off, (n_reads + 1) * 8); ENSURE(in_len, (n_reads + 1) * 4);
    if (seq_bytes) CK(hipMemcpyAsync(ctx->in_seq.p, h_seq, seq_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (n_reads) {
        CK(hipMemcpyAsync(ctx->in_off.p, h_off, n_reads * 8, hipMemcpyHostToDevice, ctx->stream));
        CK(hipMemcpyAsync(ctx->in_len.p, h_len, n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return oatk_hip_scan(ctx, ctx->in_seq.as<uint8_t>(), ctx->in_off.as<uint64_t>(), ctx->in_len.as<uint32_t>(), n_reads, seq_bytes, sid0, k, s);
}

int oatk_hip_count(oatk_hip_ctx *ctx)
{
    using namespace oatk;
    if (!ctx) return OATK_E_NODEV;
    if (!ctx->scanned) { ctx->err = "count before scan"; return OATK_E_STATE; }
    CK(hipSetDevice(ctx->device));
    ctx->counted = false;
    ctx->n_scm_total = 0;
    if (ctx->multi) ctx->multi->merged = ctx->multi->ec_done = false;
    const uint64_t n = ctx->n_occ;
    if (n == 0) { ctx->counted = true; return OATK_OK; }
    const unsigned nb = (unsigned) ((n + 255) / 256);

    ENSURE(key_sorted, n * 8); ENSURE(perm, n * 4); ENSURE(head, n * 4); ENSURE(head_idx, n * 4); ENSURE(newclus, n * 4); ENSURE(kloc, n * 8);
    if (ctx->seq_bytes >> 4 >= 0xFFFFFFFFULL) { ctx->err = "batch too large for 32-bit hoco word indices (64 GB of bases)"; return OATK_E_ARG; }
    ENSURE(clus_id, n * 4); ENSURE(bad_head, n * 4); ENSURE(tag, n * 4); ENSURE(tmp_perm, n * 4); ENSURE(flags, 64);
    ENSURE(pos_kid, n * 8); ENSURE(scm_occ, n * 8);
    CK(hipMemsetAsync(ctx->flags.p, 0, 64, ctx->stream));

    // 2. stable sort by hash; the input is already in (sid, idx) order.  Five radix passes over the top 40 bits and a repair of the few runs in
    //    which two hashes share them (count.hpp: sort_repair_kernel); all 64 bits when the debug switch asks for it or when a mixed run turned
    //    out too long for the repair (the second time round; tests reach both through oatk_hip_debug_hash_mask)
    // (batches under 4 M records go straight to the sort on all 64 bits: rocPRIM sorts small inputs by its merge / block paths, and those were seen to return
    //  keys out of order AND values that are no permutation for a bit range that does not start at bit 0 -- tools/ubench/sort_repair_test.hip: wrong up to
    //  1 M keys, right from 1.5 M on, where the radix passes take over; the repair pass still checks the order it is handed)
    bool full_sort = getenv("OATK_DEBUG_FULL_SORT") != nullptr || n < (1ull << 22);
sort_again:
    t_begin(ctx, OATK_T_COUNT_SORT);
    {
        const unsigned lo_bit = full_sort? 0u : (unsigned) OATK_SORT_LOW_BITS;
        size_t tb = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb, ctx->key_hash.as<uint64_t>(), ctx->key_sorted.as<uint64_t>(), ctx->iota.as<uint32_t>(),
                                     ctx->perm.as<uint32_t>(), n, lo_bit, 64, ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::radix_sort_pairs(ctx->tmp.p, tb, ctx->key_hash.as<uint64_t>(), ctx->key_sorted.as<uint64_t>(), ctx->iota.as<uint32_t>(),
                                     ctx->perm.as<uint32_t>(), n, lo_bit, 64, ctx->stream));
        if (!full_sort) {
            uint32_t *owner = ctx->head.as<uint32_t>();              // (free until mark_heads_kernel)
            CK(hipMemsetAsync(owner, 0xFF, n * 4, ctx->stream));
            hipLaunchKernelGGL(sort_repair_find_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->key_sorted.as<uint64_t>(), (uint32_t) n, owner, ctx->flags.as<uint32_t>());
            hipLaunchKernelGGL(sort_repair_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->key_sorted.as<uint64_t>(), ctx->perm.as<uint32_t>(), (uint32_t) n, owner,
                               ctx->flags.as<uint32_t>());
        }
    }
    t_end(ctx, OATK_T_COUNT_SORT);

    t_begin(ctx, OATK_T_COUNT_GROUP);
    GroupArgs g;
    g.sorted_key = ctx->key_sorted.as<uint64_t>(), g.perm = ctx->perm.as<uint32_t>(), g.n_rec = (uint32_t) n;
    g.pos_lo = ctx->pos_lo.as<uint64_t>(), g.pos_mpos = ctx->pos_mpos.as<uint32_t>(), g.hoco_s = ctx->hoco_s.as<uint8_t>();
    g.off = ctx->d_off, g.sid0 = ctx->sid0, g.K = ctx->K;
    g.head = ctx->head.as<uint32_t>(), g.head_idx = ctx->head_idx.as<uint32_t>(), g.newclus = ctx->newclus.as<uint32_t>();
    g.loc = ctx->kloc.as<uint64_t>();
    ENSURE(smer_sorted, n * 8);
    g.pos_smer = ctx->pos_smer.as<uint64_t>(), g.occ_sorted = ctx->scm_occ.as<uint64_t>(), g.smer_sorted = ctx->smer_sorted.as<uint64_t>();
    g.flags = ctx->flags.as<uint32_t>();
    ENSURE(slot_rec, n * 32);
    g.slot_rec = ctx->slot_rec.as<uint4>();
    hipLaunchKernelGGL(pack_slots_kernel, dim3(nb), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL(pair_sum_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->key_hash.as<uint64_t>(), (const uint32_t *) nullptr, (uint32_t) n, ctx->flags.as<uint32_t>(), 8);
    hipLaunchKernelGGL(pair_sum_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->key_sorted.as<uint64_t>(), ctx->perm.as<uint32_t>(), (uint32_t) n, ctx->flags.as<uint32_t>(), 12);
    // Optimistic order (round 4): heads -> ids -> ONE pass through the permutation (the verification's gathers + the table's and the reads' ids) -> verification.
    // A hash group with two k-mers (never seen outside the forced-collision tests) takes the pessimistic order afterwards: split, gather again, ids again, write again.
    hipLaunchKernelGGL(heads_only_kernel, dim3(nb), dim3(256), 0, ctx->stream, g);
    {   // head_idx := index of the latest head at or before i
        size_t tb = 0;
        CK(rocprim::inclusive_scan(nullptr, tb, ctx->head_idx.as<uint32_t>(), ctx->head_idx.as<uint32_t>(), n, rocprim::maximum<uint32_t>(), ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::inclusive_scan(ctx->tmp.p, tb, ctx->head_idx.as<uint32_t>(), ctx->head_idx.as<uint32_t>(), n, rocprim::maximum<uint32_t>(), ctx->stream));
    }
    uint32_t n_scm = 0;
    FinishArgs f;
    auto ids = [&]() -> int {               // clus_id := number of heads at or before i (= id + 1); n_scm; room for the table
        size_t tb = 0;
        CK(rocprim::inclusive_scan(nullptr, tb, ctx->newclus.as<uint32_t>(), ctx->clus_id.as<uint32_t>(), n, rocprim::plus<uint32_t>(), ctx->stream));
        ENSURE(tmp, tb);
        CK(rocprim::inclusive_scan(ctx->tmp.p, tb, ctx->newclus.as<uint32_t>(), ctx->clus_id.as<uint32_t>(), n, rocprim::plus<uint32_t>(), ctx->stream));
        uint32_t last_id = 0;
        CK(hipMemcpyAsync(&last_id, ctx->clus_id.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        n_scm = last_id;                    // inclusive scan: last value = number of clusters
        ENSURE(scm_h, (size_t) n_scm * 8); ENSURE(scm_s, (size_t) n_scm * 8); ENSURE(scm_cov, (size_t) n_scm * 4); ENSURE(scm_loc, (size_t) n_scm * 8);
        ENSURE(scm_occ_off, ((size_t) n_scm + 1) * 8);
        f.perm = ctx->perm.as<uint32_t>(), f.newclus = ctx->newclus.as<uint32_t>(), f.clus_id = ctx->clus_id.as<uint32_t>(), f.n_rec = (uint32_t) n;
        f.sorted_key = ctx->key_sorted.as<uint64_t>(), f.smer_sorted = ctx->smer_sorted.as<uint64_t>();
        f.loc = ctx->kloc.as<uint64_t>(), f.scm_loc = ctx->scm_loc.as<uint64_t>();
        f.scm_h = ctx->scm_h.as<uint64_t>(), f.scm_s = ctx->scm_s.as<uint64_t>(), f.scm_occ_off = ctx->scm_occ_off.as<uint64_t>();
        f.scm_occ = ctx->scm_occ.as<uint64_t>(), f.pos_kid = ctx->pos_kid.as<uint64_t>(), f.flags = ctx->flags.as<uint32_t>();
        return OATK_OK;
    };
    { int rc = ids(); if (rc) return rc; }
    hipLaunchKernelGGL(gather_finish_kernel, dim3(nb), dim3(256), 0, ctx->stream, g, f, ctx->clus_id.as<uint32_t>(), n_scm);
    CK(hipMemsetAsync(ctx->bad_head.p, 0, n * 4, ctx->stream));
    hipLaunchKernelGGL(verify_group_kernel, dim3((unsigned) ((n + 8 * OATK_VG_STRIP - 1) / (8 * OATK_VG_STRIP))), dim3(256), 0, ctx->stream, g, ctx->bad_head.as<uint32_t>());
    // clus_id holds id + 1: shift in place (check_smer_kernel and the collision path read ids)
    CK(rocprim::transform(ctx->clus_id.as<uint32_t>(), ctx->clus_id.as<uint32_t>(), n, [] __device__(uint32_t v) { return v - 1u; }, ctx->stream));
    uint32_t fl[16];
    CK(hipMemcpyAsync(fl, ctx->flags.p, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    // the sort's output is the sort's input, permuted and in order (count.hpp: pair_sum_kernel)?  The sort on a bit range gets a second chance on all 64 bits
    const bool sort_bad = fl[4] != 0 || memcmp(fl + 8, fl + 12, 16) != 0 || (!full_sort && getenv("OATK_DEBUG_SORT_DISTRUST") != nullptr);
    if (sort_bad && full_sort) {
        t_end(ctx, OATK_T_COUNT_GROUP);
        ctx->err = "the device sort returned something that is not its input in order (rocPRIM radix_sort_pairs): refusing to build the syncmer table from it";
        return OATK_E_NODEV;
    }
    if ((fl[3] || sort_bad) && !full_sort) {                  // a run of equal top bits too long for the repair: everything from the sort on again, on all 64 bits
        t_end(ctx, OATK_T_COUNT_GROUP);
        CK(hipMemsetAsync(ctx->flags.p, 0, 64, ctx->stream));
        full_sort = true;
        goto sort_again;
    }
    if (fl[0]) {
        ctx->collisions = 1;
        hipLaunchKernelGGL(split_collisions_kernel, dim3(nb), dim3(256), 0, ctx->stream, g, ctx->bad_head.as<uint32_t>(), ctx->perm.as<uint32_t>(),
                           ctx->tag.as<uint32_t>(), ctx->tmp_perm.as<uint32_t>());
        hipLaunchKernelGGL(regather_kernel, dim3(nb), dim3(256), 0, ctx->stream, g);
        { int rc = ids(); if (rc) return rc; }
        CK(rocprim::transform(ctx->clus_id.as<uint32_t>(), ctx->clus_id.as<uint32_t>(), n, [] __device__(uint32_t v) { return v - 1u; }, ctx->stream));
        hipLaunchKernelGGL(finish_heads_kernel, dim3(nb), dim3(256), 0, ctx->stream, f, n_scm);
    }
    hipLaunchKernelGGL(check_smer_kernel, dim3(nb), dim3(256), 0, ctx->stream, f);
    hipLaunchKernelGGL(cov_kernel, dim3((n_scm + 255) / 256), dim3(256), 0, ctx->stream, ctx->scm_occ_off.as<uint64_t>(), ctx->scm_cov.as<uint32_t>(), n_scm);
    t_end(ctx, OATK_T_COUNT_GROUP);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(fl, ctx->flags.p, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->timing) t_collect(ctx, OATK_T_COUNT_SORT, OATK_T_COUNT_GROUP);
    if (fl[2]) { ctx->err = "hash group with too many distinct k-mers"; return OATK_E_SPLIT; }
    if (debug_refuse(ctx, 4, "oatk_hip_count")) return OATK_E_SPLIT;
    if (fl[1]) { ctx->err = "identical kmers have different smers"; return OATK_E_SMER; }
    ctx->n_scm_total = n_scm;
    ctx->counted = true;
    return OATK_OK;
}

int oatk_hip_info(oatk_hip_ctx *ctx, oatk_hip_info_t *out)
{
    if (!ctx) return OATK_E_NODEV;
    out->n_reads = ctx->n_reads, out->seq_bytes = ctx->seq_bytes, out->sid0 = ctx->sid0;
    out->k = ctx->K, out->s = ctx->S;
    out->n_occ = ctx->n_occ, out->n_nn = ctx->tot_nn, out->n_lrl = ctx->tot_lrl;
    out->n_scm = ctx->n_scm_total;
    out->scan_retries = ctx->retries, out->collisions = ctx->collisions;
    return OATK_OK;
}

int oatk_hip_buffer(oatk_hip_ctx *ctx, int which, const void **d_ptr, uint64_t *bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (which >= OATK_BUF_INGEST_SEQ && which <= OATK_BUF_INGEST_HDR) return ing_buffer(ctx, which, d_ptr, bytes);      // precedes any scan
    if (which >= OATK_BUF_MG_H && which <= OATK_BUF_MG_LCOV) return multi_buffer(ctx, which, d_ptr, bytes);
    if (which >= OATK_BUF_MG_G_H && which <= OATK_BUF_MG_POS_GKID) return multi_tail_buffer(ctx, which, d_ptr, bytes);
    if (!ctx->scanned) { ctx->err = "no resident scan"; return OATK_E_STATE; }
    const uint64_t n = ctx->n_reads, occ = ctx->n_occ, ns = ctx->n_scm_total;
    const void *p = nullptr;
    uint64_t b = 0;
    switch (which) {
        case OATK_BUF_HOCO_L: p = ctx->hoco_l.p, b = n * 4; break;
        case OATK_BUF_N_SCM: p = ctx->n_scm.p, b = n * 4; break;
        case OATK_BUF_N_NN: p = ctx->n_nn.p, b = n * 4; break;
        case OATK_BUF_N_LRL: p = ctx->n_lrl.p, b = n * 4; break;
        case OATK_BUF_HO_RL: p = ctx->ho_rl.p, b = ctx->seq_bytes; break;
        case OATK_BUF_HOCO_S: p = ctx->hoco_s.p, b = ctx->seq_bytes / 4 + 64; break;
        case OATK_BUF_NN_KEY: p = ctx->nn_sorted_in_2? ctx->nn_key2.p : ctx->nn_key.p, b = ctx->tot_nn * 8; break;
        case OATK_BUF_LRL_KEY: p = ctx->lrl_sorted_in_2? ctx->lrl_key2.p : ctx->lrl_key.p, b = ctx->tot_lrl * 8; break;
        case OATK_BUF_LRL_VAL: p = ctx->lrl_sorted_in_2? ctx->lrl_val2.p : ctx->lrl_val.p, b = ctx->tot_lrl * 4; break;
        case OATK_BUF_SCM_OFF: p = ctx->scm_off.p, b = (n + 1) * 8; break;
        case OATK_BUF_POS_MPOS: p = ctx->pos_mpos.p, b = occ * 4; break;
        case OATK_BUF_POS_SMER: p = ctx->pos_smer.p, b = occ * 8; break;
        case OATK_BUF_POS_HASH: p = ctx->pos_hash.p, b = occ * 8; break;
        default:
            if (!ctx->counted) { ctx->err = "count results requested before oatk_hip_count"; return OATK_E_STATE; }
            switch (which) {
                case OATK_BUF_POS_KID: p = ctx->pos_kid.p, b = occ * 8; break;
                case OATK_BUF_SCM_H: p = ctx->scm_h.p, b = ns * 8; break;
                case OATK_BUF_SCM_S: p = ctx->scm_s.p, b = ns * 8; break;
                case OATK_BUF_SCM_COV: p = ctx->scm_cov.p, b = ns * 4; break;
                case OATK_BUF_SCM_OCC_OFF: p = ctx->scm_occ_off.p, b = (ns + 1) * 8; break;
                case OATK_BUF_SCM_OCC: p = ctx->scm_occ.p, b = occ * 8; break;
                default:
                    if (which >= OATK_BUF_CONS_SEL && which <= OATK_BUF_CONS_TOT) return cons_buffer(ctx, which, d_ptr, bytes);
                    if (which >= OATK_BUF_AG_SCM_DEL && which <= OATK_BUF_AG_ARC_LINK) return ag_buffer(ctx, which, d_ptr, bytes);
                    if (which >= OATK_BUF_OVL_KEY && which <= OATK_BUF_OVL_TAIL) return ovl_buffer(ctx, which, d_ptr, bytes);
                    if (which >= OATK_BUF_RA_ALN_SID && which <= OATK_BUF_RA_SKIPPED) return ra_buffer(ctx, which, d_ptr, bytes);
                    if (which >= OATK_BUF_INGEST_SEQ && which <= OATK_BUF_INGEST_HDR) return ing_buffer(ctx, which, d_ptr, bytes);
                    return ec_buffer(ctx, which, d_ptr, bytes);     // error-correction results (api_ec.inc)
            }
    }
    *d_ptr = p, *bytes = b;
    return OATK_OK;
}

void *oatk_hip_staging(oatk_hip_ctx *ctx, uint64_t bytes)
{
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (bytes <= ctx->staging_cap) return ctx->staging;
    staging_free(ctx);
    // A fresh anonymous mapping on transparent huge pages, touched by a few threads and then page-locked: 8 GB in 0.05 s on the bench box, where
    // hipHostMalloc takes 1.5 s (tools/ubench/pin_rates.hip, profiles/r04a_pin_rates.txt).  hipHostMalloc remains the fallback.
    const uint64_t HP = 2ull << 20, want = (bytes + HP - 1) & ~(HP - 1);
    void *raw = mmap(nullptr, want + HP, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw != MAP_FAILED) {
        uint8_t *base = (uint8_t *) (((uintptr_t) raw + HP - 1) & ~(uintptr_t) (HP - 1));
        (void) madvise(base, want, MADV_HUGEPAGE);
        const int nt = want >= (256ull << 20)? 8 : 1;
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back([=] { for (uint64_t o = want / nt * t; o < (t + 1 == nt? want : want / nt * (t + 1)); o += 4096) *(volatile uint8_t *) (base + o) = 0; });
        for (uint64_t o = 0; o < want / nt; o += 4096) *(volatile uint8_t *) (base + o) = 0;
        for (auto &x : th) x.join();
        if (hipHostRegister(base, want, hipHostRegisterDefault) == hipSuccess) {
            ctx->staging = base, ctx->staging_cap = want, ctx->staging_raw = raw, ctx->staging_raw_size = want + HP;
            return ctx->staging;
        }
        (void) hipGetLastError();
        munmap(raw, want + HP);
    }
    if (hipHostMalloc(&ctx->staging, bytes, hipHostMallocDefault) != hipSuccess) { ctx->staging = nullptr; ctx->err = "hipHostMalloc failed (staging)"; return nullptr; }
    ctx->staging_cap = bytes;
    return ctx->staging;
}

int oatk_hip_host_register(oatk_hip_ctx *ctx, void *p, uint64_t bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (!p || bytes == 0) return OATK_E_ARG;
    CK(hipSetDevice(ctx->device));
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return OATK_OK;
}

int oatk_hip_host_unregister(oatk_hip_ctx *ctx, void *p)
{
    if (!ctx) return OATK_E_NODEV;
    CK(hipSetDevice(ctx->device));
    CK(hipHostUnregister(p));
    return OATK_OK;
}

int oatk_hip_d2h(oatk_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (bytes == 0) return OATK_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return OATK_OK;
}

int oatk_hip_d2h_async(oatk_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (bytes == 0) return OATK_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return OATK_OK;
}

int oatk_hip_h2d_async(oatk_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx) return OATK_E_NODEV;
    if (bytes == 0) return OATK_OK;
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OATK_OK;
}

}  // extern "C"
