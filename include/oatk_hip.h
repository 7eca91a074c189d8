/*
 * include/oatk_hip.h -- C ABI of the MI355X (gfx950) device path for oatk's syncasm hot path.
 *
 * This is the drop-in boundary: plain C, opaque handle, raw pointers and sizes, int return codes
 * (0 = ok).  The reference has no FFI layer; its seam is a set of C functions over in-memory structs
 * (SURVEY.md 8b).  A maintainer keeps those signatures and calls the entry points below from inside
 * them (INTEGRATION.md shows the stubs):
 *
 *   reference symbol (file:line)                         device entry points used
 *   ---------------------------------------------------  -----------------------------------------------
 *   sr_read / sr_read_analysis_thread                    oatk_hip_scan_host / oatk_hip_scan  (+ _buffer)
 *       syncmer.c:487 / syncmer.c:243-421
 *   kmer_hash64 + MurmurHash64A  syncmer.c:175, :131     (inside the scan)
 *   collect_syncmer_from_reads   syncmer.c:1397          oatk_hip_count
 *   process_kmer_cluster         syncmer.c:1270          (inside the count)
 *   read_error_correction        syncerr.c:819           oatk_hip_ec_*   (oatk_hip_ec.h)
 *
 * All device results stay resident in the handle until the next scan; `oatk_hip_buffer` exposes them
 * as (device pointer, byte size) pairs and `oatk_hip_d2h` copies any range to the host.
 *
 * The library fails loudly: there is no CPU fallback.  Without a gfx950 device `oatk_hip_create`
 * returns NULL and every other call returns OATK_E_NODEV.
 */
#ifndef OATK_HIP_H
#define OATK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OATK_HIP_ABI_VERSION 1

/* Packed read stream: read r occupies seq[off[r], off[r] + len[r]); every off[r] is a multiple of
 * OATK_READ_ALIGN and seq_bytes (the allocation) is too.  Bases are ASCII exactly as the FASTA/FASTQ
 * parser delivers them (any case, U, ambiguity codes). */
#define OATK_READ_ALIGN 64

enum {
    OATK_OK = 0,
    OATK_E_NODEV = 1,      /* no usable gfx950 device / HIP runtime error (see oatk_hip_last_error) */
    OATK_E_ARG = 2,        /* bad argument (alignment, k/s range, ...) */
    OATK_E_STATE = 3,      /* call order (e.g. count before scan) */
    OATK_E_SMER = 4,       /* identical k-mers carry different s-mers (fatal in the reference, syncmer.c:1370) */
    OATK_E_SPLIT = 5,      /* a hash group holds more distinct k-mers than the device splitter supports */
    OATK_E_NOMEM = 6
};

typedef struct oatk_hip_ctx oatk_hip_ctx;

int oatk_hip_abi_version(void);
int oatk_hip_device_count(void);
oatk_hip_ctx *oatk_hip_create(int device);
void oatk_hip_destroy(oatk_hip_ctx *ctx);
const char *oatk_hip_last_error(oatk_hip_ctx *ctx);
/* hipStream_t the handle launches on (created by the handle) */
void *oatk_hip_stream(oatk_hip_ctx *ctx);
int oatk_hip_sync(oatk_hip_ctx *ctx);

/* Device memory in pieces, for a process that streams a file through the device (the drop-in CLI calls this before sr_read; nothing in the reference corresponds:
 * it malloc's as it goes, syncmer.c:505-533).  From this call on the larger buffers of every handle of this process on ctx's device are address ranges backed by
 * 64 MB pieces: they grow by mapping more pieces (no copy), a released buffer's pieces serve the next buffer (nothing returns to the driver before the process
 * ends), and a thread of the pool's own takes `warm_bytes` of pieces from the driver ahead of the need.  Why: the driver clears memory when it hands it out for
 * the first time since the GPU was reset (30 ms per GB inside the call) and clears what comes back behind the process's back while the next request for memory
 * waits -- seconds at 2 M reads (tools/ubench/alloc_*.hip).  Results are the same bytes either way; OATK_POOL=0 makes the call do nothing.  Returns OATK_OK
 * also where the device has no virtual-memory management (the buffers then stay hipMalloc's). */
int oatk_hip_mem_pool(oatk_hip_ctx *ctx, uint64_t warm_bytes);

/* largest k the device scan supports for a given s (LDS ring geometry); the reference asserts
 * 0 < s < 32 < ... < k (syncmer.c:251) */
int oatk_hip_max_k(void);

/* ---- scan: replaces sr_read_analysis_thread (syncmer.c:243-421) for a whole batch of reads ----
 * d_* pointers are device memory.  sid0 is the global id of read 0 (reads are numbered in input
 * order, syncmer.c:525).  Results stay resident in ctx. */
int oatk_hip_scan(oatk_hip_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint32_t *d_len,
                  uint64_t n_reads, uint64_t seq_bytes, uint64_t sid0, int k, int s);
/* same, from host memory (uploads into buffers owned by ctx) */
int oatk_hip_scan_host(oatk_hip_ctx *ctx, const uint8_t *h_seq, const uint64_t *h_off, const uint32_t *h_len,
                       uint64_t n_reads, uint64_t seq_bytes, uint64_t sid0, int k, int s);

/* ---- a batch built from pieces: sr_read hands its reads to the analysis threads in batches too (10 000 per thread, syncmer.c:505-533) ----
 * oatk_hip_scan_begin empties ctx and declares the geometry of the batch that is going to be assembled; oatk_hip_scan_append moves the scan
 * that is resident in `piece` (another handle on the same device, scanned with sid0 = ctx's sid0 + the reads ctx already holds) behind the
 * reads of ctx: hoco strings, run lengths, per-read arrays, syncmer slots and rare-event lists, with offsets rebased -- device-to-device, no
 * recomputation.  Afterwards ctx is exactly what ONE scan of all the reads would have left (count, error correction, ... follow on ctx);
 * `piece` can be scanned again.  oatk_hip_scan_reserve sizes ctx's buffers ahead (totals, may be rough: buffers grow when exceeded). */
int oatk_hip_scan_begin(oatk_hip_ctx *ctx, uint64_t sid0, int k, int s);
int oatk_hip_scan_reserve(oatk_hip_ctx *ctx, uint64_t seq_bytes, uint64_t n_reads, uint64_t n_occ);
int oatk_hip_scan_append(oatk_hip_ctx *ctx, oatk_hip_ctx *piece);
/* the device ordinal the handle was created on */
int oatk_hip_device(oatk_hip_ctx *ctx);
/* device-to-device copy on the handle's stream, completed on return */
int oatk_hip_d2d(oatk_hip_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes);

/* ---- count: replaces collect_syncmer_from_reads (syncmer.c:1397-1451) on the resident scan ---- */
int oatk_hip_count(oatk_hip_ctx *ctx);

typedef struct {
    uint64_t n_reads, seq_bytes, sid0;
    int32_t k, s;
    uint64_t n_occ;        /* syncmer occurrences over all reads (sum of sr_t.n)          */
    uint64_t n_nn;         /* ambiguous bases over all reads                              */
    uint64_t n_lrl;        /* homopolymer runs longer than 255                            */
    uint64_t n_scm;        /* distinct syncmers (valid after oatk_hip_count)              */
    uint32_t scan_retries; /* times the scan was re-run because a device list overflowed  */
    uint32_t collisions;   /* 1 if some 64-bit hash group held different k-mers           */
} oatk_hip_info_t;
int oatk_hip_info(oatk_hip_ctx *ctx, oatk_hip_info_t *out);

/* Resident device buffers.  Layouts (r = read index in the batch, o = off[r]):
 *   HOCO_L, N_SCM, N_NN, N_LRL  u32[n_reads]      sr_t.hoco_l, sr_t.n, entries of n_nucl / ho_l_rl
 *   HO_RL    u8 [seq_bytes]      read r at o, hoco_l[r] valid bytes                      (sr_t.ho_rl)
 *   HOCO_S   u8 [seq_bytes/4+64] read r at o/4, ceil(hoco_l[r]/4) valid bytes            (sr_t.hoco_s)
 *   NN_KEY   u64[n_nn]           sid<<32 | raw position, ascending                       (sr_t.n_nucl)
 *   LRL_KEY  u64[n_lrl]          sid<<32 | hoco position, ascending;  LRL_VAL u32: run-1 (sr_t.ho_l_rl)
 *   SCM_OFF  u64[n_reads+1]      exclusive prefix of N_SCM: read r owns slots [SCM_OFF[r], SCM_OFF[r+1])
 *   POS_MPOS u32[n_occ]  POS_SMER u64[n_occ]  POS_HASH u64[n_occ]                        (sr_t.m_pos/s_mer/k_mer)
 *   POS_KID  u64[n_occ]          syncmer id << 1, after count                            (sr_t.k_mer rewritten)
 *   SCM_H, SCM_S u64[n_scm], SCM_COV u32[n_scm]                                          (syncmer_t.h/s/cov)
 *   SCM_OCC_OFF u64[n_scm+1], SCM_OCC u64[n_occ]  sid<<32 | idx<<1 | rev                 (syncmer_t.m_pos)
 */
enum {
    OATK_BUF_HOCO_L = 0, OATK_BUF_N_SCM, OATK_BUF_N_NN, OATK_BUF_N_LRL,
    OATK_BUF_HO_RL, OATK_BUF_HOCO_S,
    OATK_BUF_NN_KEY, OATK_BUF_LRL_KEY, OATK_BUF_LRL_VAL,
    OATK_BUF_SCM_OFF, OATK_BUF_POS_MPOS, OATK_BUF_POS_SMER, OATK_BUF_POS_HASH, OATK_BUF_POS_KID,
    OATK_BUF_SCM_H, OATK_BUF_SCM_S, OATK_BUF_SCM_COV, OATK_BUF_SCM_OCC_OFF, OATK_BUF_SCM_OCC,
    OATK_BUF_COUNT_
};
int oatk_hip_buffer(oatk_hip_ctx *ctx, int which, const void **d_ptr, uint64_t *bytes);
int oatk_hip_d2h(oatk_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
/* the same without waiting: the copy is ordered on the handle's stream (oatk_hip_sync waits for it); the host side should be page-locked
 * (oatk_hip_staging) or the runtime stages it and the call blocks anyway */
int oatk_hip_d2h_async(oatk_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int oatk_hip_h2d_async(oatk_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
/* Page-locked host memory owned by the context (one block, regrown on demand, freed with the context; NULL on failure): oatk_hip_d2h into it
 * runs at PCIe speed, into pageable memory at a fraction of it -- callers that move gigabytes of results stage them through it in pieces. */
void *oatk_hip_staging(oatk_hip_ctx *ctx, uint64_t bytes);
/* The caller's OWN host memory page-locked for a while, so that it can be the destination (or source) of asynchronous copies as it lies -- the reads'
 * arenas of liboatk_host.so are filled that way, without a staging copy (host/srdb.c).  Registering memory that has been touched and sits on
 * transparent huge pages costs ~3 ms per GB on the bench box; hipHostMalloc'ed memory 170 ms per GB (tools/ubench/pin_rates.hip). */
int oatk_hip_host_register(oatk_hip_ctx *ctx, void *p, uint64_t bytes);
int oatk_hip_host_unregister(oatk_hip_ctx *ctx, void *p);

/* ---- measurement: HIP-event timing of the phases, recorded on the handle's stream ---- */
enum {
    OATK_T_HPC = 0,        /* kernel A: homopolymer compression + pack (scan_hpc.hpp)       */
    OATK_T_SYNCMER,        /* kernel B, reads without ambiguous bases (scan_syncmer.hpp)   */
    OATK_T_SYNCMER_N,      /* kernel B, reads with ambiguous bases                         */
    OATK_T_SCAN_POST,      /* list sorts + per-read prefix                                 */
    OATK_T_COUNT_PLACE,    /* place_records                                                */
    OATK_T_COUNT_SORT,     /* radix sort by hash                                           */
    OATK_T_COUNT_GROUP,    /* heads + collision verification + ids + finish                */
    OATK_T_KMER_HASH,      /* MurmurHash64A of every syncmer's k-mer (kmer_hash.hpp)       */
    OATK_T_EC_GRAPH,       /* oatk_hip_ec_graph: adjacent pairs, sort, arcs, overlaps      */
    OATK_T_EC_MARK,        /* find_error_syncmers + live arcs + block lists                */
    OATK_T_EC_SOLVE,       /* the path search of every error block, all tiers (ec_wave.hpp) */
    OATK_T_EC_REFRESH,     /* corrected chains + update_syncmer_db                         */
    OATK_T_COUNT_
};
int oatk_hip_set_timing(oatk_hip_ctx *ctx, int enable);
/* milliseconds of the most recent scan / count / error-correction round, one entry per OATK_T_* */
int oatk_hip_get_timing(oatk_hip_ctx *ctx, float *ms, int n);

/* test hook: AND every k-mer hash with `mask` before grouping, to force "hash collisions" through the
 * sequence-comparison path.  ~0 (default) in production. */
int oatk_hip_debug_hash_mask(oatk_hip_ctx *ctx, uint64_t mask);
/* test hook: 1 = always use the general syncmer kernel (scan_syncmer.hpp), 0 = use the fast path where it applies */
int oatk_hip_debug_force_general(oatk_hip_ctx *ctx, int on);
/* Test hook: syncmers the fast scan kernel collects per read in LDS before it writes their records (1..512; 0 = default 512).
 * Results never depend on it. */
int oatk_hip_debug_list_cap(oatk_hip_ctx *ctx, int cap);

#ifdef __cplusplus
}
#endif
#endif
