/*
 * oatk_amd/csrc/host/srdb.c -- host side of the drop-in boundary: device results -> the reference's structs.
 *
 * oatk_sr_read_packed          is the body of sr_read (syncmer.c:487-556) once the reads are in memory,
 * oatk_collect_syncmer_from_reads is collect_syncmer_from_reads (syncmer.c:1397-1451),
 * both computed on the MI355X through the C ABI of include/oatk_hip.h.  What remains on the host is what the struct
 * layout forces: one malloc + memcpy per member array per read (sr_destroy frees each of them, syncmer.c:1047-1058).
 */
#define _GNU_SOURCE
#include <malloc.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/types.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#include "oatk_syncasm.h"
#include "host_internal.h"

static double host_now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}
static int host_log(void)
{
    const char *e = getenv("OATK_DROPIN_LOG");
    return e && e[0] && e[0] != '0';
}

static void *xmalloc(size_t n)
{
    void *p = malloc(n? n : 1);
    if (!p) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    return p;
}

/* copy one resident device buffer to a fresh host array */
static void *fetch(oatk_hip_ctx *ctx, int which, uint64_t *bytes, int *rc)
{
    const void *d = 0;
    *bytes = 0;
    *rc = oatk_hip_buffer(ctx, which, &d, bytes);
    if (*rc) return 0;
    void *h = xmalloc(*bytes);
    *rc = oatk_hip_d2h(ctx, h, d, *bytes);
    if (*rc) { free(h); return 0; }
    return h;
}

/* Reads already in memory, in pieces (round 4): piece p + 1 goes up and is scanned on a handle of its own while piece p's arrays come down into the reads'
 * structs and piece p - 1 has been moved behind the batch assembled in ctx (oatk_hip_scan_append) -- the bus carries text one way and results the other
 * at the same time, where one H2D - scan - D2H pass used it one direction after the other. */
#define PACKED_PIECE ((uint64_t) 256 << 20)

typedef struct {
    oatk_hip_ctx *piece[2];
    const uint8_t *seq; const uint64_t *off; const uint32_t *len;
    uint64_t n_reads, seq_bytes;
    int k, s;
    pthread_mutex_t mu; pthread_cond_t cv;
    int state[2];                  /* 0 free, 1 scanned */
    uint64_t i0[2], i1[2];
    uint64_t *rel[2];              /* the piece's offsets, relative to its first read */
    int failed, stop;
} packed_t;

static uint64_t packed_end(const packed_t *P, uint64_t i0)
{
    uint64_t i1 = i0 + 1;
    while (i1 < P->n_reads && P->off[i1] - P->off[i0] < PACKED_PIECE) ++i1;
    return i1;
}

static void *packed_producer(void *arg)
{
    packed_t *P = (packed_t *) arg;
    uint64_t i0 = 0, w;
    for (w = 0; i0 < P->n_reads; ++w) {
        const int sl = (int) (w & 1);
        const uint64_t i1 = packed_end(P, i0), n = i1 - i0, bytes = (i1 < P->n_reads? P->off[i1] : P->seq_bytes) - P->off[i0];
        uint64_t i;
        pthread_mutex_lock(&P->mu);
        while (P->state[sl] != 0 && !P->stop) pthread_cond_wait(&P->cv, &P->mu);
        const int stop = P->stop;
        pthread_mutex_unlock(&P->mu);
        if (stop) break;
        uint64_t *rel = (uint64_t *) realloc(P->rel[sl], 8 * n);
        int rc = rel? OATK_OK : OATK_E_NOMEM;
        if (rel) {
            P->rel[sl] = rel;
            for (i = 0; i < n; ++i) rel[i] = P->off[i0 + i] - P->off[i0];
            rc = oatk_hip_scan_host(P->piece[sl], P->seq + P->off[i0], rel, P->len + i0, n, bytes, i0, P->k, P->s);
        }
        pthread_mutex_lock(&P->mu);
        if (rc) P->failed = rc;
        else P->state[sl] = 1, P->i0[sl] = i0, P->i1[sl] = i1;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        if (rc) break;
        i0 = i1;
    }
    return 0;
}

/* the two piece handles outlive the call (their buffers are a gigabyte each, and allocating them is most of what a call of this size takes): kept per device,
 * handed to the next call on that device */
static pthread_mutex_t g_pp_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_pp_dev = -1;
static oatk_hip_ctx *g_pp[2];

static void packed_pieces_take(int dev, oatk_hip_ctx *piece[2])
{
    pthread_mutex_lock(&g_pp_mu);
    if (g_pp[0] && g_pp_dev == dev) piece[0] = g_pp[0], piece[1] = g_pp[1], g_pp[0] = g_pp[1] = 0;
    else piece[0] = piece[1] = 0;
    pthread_mutex_unlock(&g_pp_mu);
    if (!piece[0]) piece[0] = oatk_hip_create(dev), piece[1] = oatk_hip_create(dev);
}

static void packed_pieces_give(int dev, oatk_hip_ctx *piece[2], int failed)
{
    oatk_hip_ctx *old[2] = {0, 0};
    if (!failed && piece[0] && piece[1]) {
        pthread_mutex_lock(&g_pp_mu);
        old[0] = g_pp[0], old[1] = g_pp[1];
        g_pp[0] = piece[0], g_pp[1] = piece[1], g_pp_dev = dev;
        pthread_mutex_unlock(&g_pp_mu);
    } else old[0] = piece[0], old[1] = piece[1];
    if (old[0]) oatk_hip_destroy(old[0]);
    if (old[1]) oatk_hip_destroy(old[1]);
}

static int sr_read_packed_pipelined(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint8_t *seq, const uint64_t *off, const uint32_t *len,
                                    uint64_t n_reads, uint64_t seq_bytes, char **names)
{
    packed_t P;
    pthread_t th;
    int rc, started = 0;
    uint64_t done = 0, w;
    memset(&P, 0, sizeof(P));
    P.seq = seq, P.off = off, P.len = len, P.n_reads = n_reads, P.seq_bytes = seq_bytes, P.k = sr_db->k, P.s = sr_db->s;
    pthread_mutex_init(&P.mu, 0);
    pthread_cond_init(&P.cv, 0);
    sr_db->a = (oatk_sr_t *) calloc(n_reads, sizeof(oatk_sr_t));
    if (!sr_db->a) return OATK_E_NOMEM;
    sr_db->n = 0, sr_db->m = n_reads;
    rc = oatk_hip_scan_begin(ctx, 0, sr_db->k, sr_db->s);
    if (!rc) rc = oatk_hip_scan_reserve(ctx, seq_bytes + (1 << 20), n_reads + 1024, seq_bytes / 500 + 4096);
    packed_pieces_take(oatk_hip_device(ctx), P.piece);
    if (!rc && (!P.piece[0] || !P.piece[1])) rc = OATK_E_NODEV;
    if (!rc && pthread_create(&th, 0, packed_producer, &P) != 0) rc = OATK_E_NOMEM;
    else if (!rc) started = 1;
    for (w = 0; !rc && done < n_reads; ++w) {
        const int sl = (int) (w & 1);
        pthread_mutex_lock(&P.mu);
        while (P.state[sl] != 1 && !P.failed) pthread_cond_wait(&P.cv, &P.mu);
        rc = P.failed;
        const uint64_t i0 = P.i0[sl], i1 = P.i1[sl];
        pthread_mutex_unlock(&P.mu);
        if (rc) break;
        rc = oatk_sr_db_fill_range(P.piece[sl], sr_db, i0, P.rel[sl], i1 - i0, names? names + i0 : 0);
        if (!rc) rc = oatk_hip_scan_append(ctx, P.piece[sl]);
        pthread_mutex_lock(&P.mu);
        P.state[sl] = 0;
        pthread_cond_broadcast(&P.cv);
        pthread_mutex_unlock(&P.mu);
        done = i1;
    }
    if (started) {
        pthread_mutex_lock(&P.mu);
        P.stop = 1;
        pthread_cond_broadcast(&P.cv);
        pthread_mutex_unlock(&P.mu);
        pthread_join(th, 0);
    }
    packed_pieces_give(oatk_hip_device(ctx), P.piece, rc);
    free(P.rel[0]); free(P.rel[1]);
    pthread_mutex_destroy(&P.mu);
    pthread_cond_destroy(&P.cv);
    return rc;
}

int oatk_sr_read_packed(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint8_t *seq, const uint64_t *off, const uint32_t *len,
                        uint64_t n_reads, uint64_t seq_bytes, char **names)
{
    /* (opt-in: with the piece handles created per call it was slower than one pass -- 135.7 against 99.6 ms at 1.5 GB, profiles/r04m_bench.json `results_back` --,
     *  with the handles kept between calls it is as fast and no faster: 67.9 against 69.1 ms with arenas, 99.8 against 109.1 without, profiles/r04n_packed_ab.txt) */
    const char *e = getenv("OATK_HOST_PACKED_PIECES");
    if (seq_bytes > 2 * PACKED_PIECE && n_reads > 1 && e && e[0] == '1') return sr_read_packed_pipelined(ctx, sr_db, seq, off, len, n_reads, seq_bytes, names);
    int rc = oatk_hip_scan_host(ctx, seq, off, len, n_reads, seq_bytes, 0, sr_db->k, sr_db->s);
    if (rc) return rc;
    return oatk_sr_db_fill_resident(ctx, sr_db, off, n_reads, names);
}

/* ---- sr_db->a[0 .. n_reads) from the scan resident in ctx ----
 * The per-base arrays (ho_rl, hoco_s: 1 + 1/4 byte per hoco base) and the per-syncmer arrays (m_pos, s_mer, k_mer hash: 20 bytes each) come
 * over in PIECES of consecutive reads through two page-locked buffers: while the host threads cut piece p into the reads' own malloc'ed
 * blocks, piece p + 1 is already on its way (one stream, copies queue behind each other at PCIe speed). */
#define FILL_RL_BYTES ((uint64_t) 64 << 20)          /* packed ho_rl bytes per piece */
#define FILL_SCM ((uint64_t) 1 << 20)                 /* syncmers per piece */

typedef struct {
    uint8_t *rl, *hs;                                 /* page-locked: one piece */
    uint32_t *m_pos;
    uint64_t *s_mer, *k_hash;
} fill_buf_t;

typedef struct {
    oatk_sr_db_t *sr_db;
    uint64_t first;                                   /* read 0 of the resident scan is sr_db->a[first] */
    const uint64_t *off, *scm_off;
    const uint32_t *hoco_l, *n_nn, *n_lrl, *lrl_val;
    const uint64_t *nn_key, *o_nn, *o_lrl;
    char **names;
    /* the piece being cut */
    const fill_buf_t *buf;
    uint64_t i0, i1, rl0, scm0;
    /* the piece whose blocks are allocated meanwhile (thread 0) */
    uint64_t a0, a1;
    double t_alloc;
    /* the stretch of heap those blocks will come from: touched by the other threads first (see fill_prefault) */
    uint8_t *pre0, *pre1;
} fill_job_t;

static uint8_t *g_heap_top = 0;                       /* where the last large block ended: survives from one call to the next */
static int heap_tune_wanted(void)
{
    const char *e = getenv("OATK_HOST_HEAP_TUNE");
    return e && e[0] == '1';
}

/* ---- arenas (opt-in: oatk_host_set_arena) --------------------------------------------------------------------------------------------------
 * The reference frees every member array of every read with free() (sr_destroy, syncmer.c:1047-1058) and reallocs the chains in
 * read_error_correction (syncerr.c:604-608), so by default every array handed out is its own malloc'ed block: 7 per read, 14 M at 2 M reads,
 * all from one thread (above), and as many free() calls at the end.  A program that also owns the destroy functions -- the drop-in binary
 * does, include/oatk_dropin.h -- can ask for ARENAS instead: one block per piece of reads (tens of MB: an anonymous mapping, fresh zero pages
 * that the copying threads touch first, in parallel), the member pointers carved from it; a registry of the blocks tells arena memory from
 * malloc'ed memory wherever a member is freed or replaced. */
typedef struct { uint8_t *base; size_t size; const void *owner; uint8_t *raw; size_t raw_size; } host_arena_t;      /* raw != NULL: an anonymous mapping (munmap), else malloc'ed */
static host_arena_t *g_ar = 0;
static size_t g_nar = 0, g_mar = 0;
static int g_use_arena = 0;
static pthread_mutex_t g_ar_mu = PTHREAD_MUTEX_INITIALIZER;

void oatk_host_set_arena(int on) { g_use_arena = on != 0; }
int oatk_host_arena(void) { return g_use_arena; }

/* index of the arena that holds p, or -1; *hint (may be NULL) = where the previous lookup ended: members of consecutive reads are neighbours */
static long arena_of(const void *p, long *hint)
{
    const uint8_t *q = (const uint8_t *) p;
    if (!g_nar || !p) return -1;
    if (hint && *hint >= 0 && (size_t) *hint < g_nar && q >= g_ar[*hint].base && q < g_ar[*hint].base + g_ar[*hint].size) return *hint;
    size_t lo = 0, hi = g_nar;                       /* sorted by base */
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (g_ar[mid].base + g_ar[mid].size <= q) lo = mid + 1; else hi = mid; }
    if (lo < g_nar && q >= g_ar[lo].base) { if (hint) *hint = (long) lo; return (long) lo; }
    return -1;
}
static void arena_register(uint8_t *b, size_t size, const void *owner, uint8_t *raw, size_t raw_size);
static uint8_t *arena_new(size_t size, const void *owner)
{
    uint8_t *b = (uint8_t *) xmalloc(size? size : 1);
    arena_register(b, size, owner, 0, 0);
    static int thp = -1;
    if (thp < 0) { const char *e = getenv("OATK_HOST_THP"); thp = !(e && e[0] == '0'); }
    if (thp && size >= ((size_t) 4 << 20)) {        /* huge pages for the bulk of it: 512 times fewer first-touch faults */
        const uintptr_t h0 = ((uintptr_t) b + (((uintptr_t) 2 << 20) - 1)) & ~(((uintptr_t) 2 << 20) - 1), h1 = ((uintptr_t) b + size) & ~(((uintptr_t) 2 << 20) - 1);
        if (h1 > h0) (void) madvise((void *) h0, h1 - h0, MADV_HUGEPAGE);
    }
    return b;
}
static void arena_register(uint8_t *b, size_t size, const void *owner, uint8_t *raw, size_t raw_size)
{
    pthread_mutex_lock(&g_ar_mu);
    if (g_nar == g_mar) { g_mar = g_mar? 2 * g_mar : 256; g_ar = (host_arena_t *) realloc(g_ar, g_mar * sizeof(host_arena_t)); if (!g_ar) abort(); }
    size_t at = g_nar;
    while (at && g_ar[at - 1].base > b) { g_ar[at] = g_ar[at - 1]; --at; }
    g_ar[at].base = b, g_ar[at].size = size? size : 1, g_ar[at].owner = owner, g_ar[at].raw = raw, g_ar[at].raw_size = raw_size;
    ++g_nar;
    pthread_mutex_unlock(&g_ar_mu);
}
/* a malloc'ed block the caller already holds (an array fetched from the device) becomes an arena: its parts are handed out as they lie */
void oatk_host_arena_adopt(void *block, size_t bytes, const void *owner) { if (block) arena_register((uint8_t *) block, bytes + 1, owner, 0, 0); }      /* (+ 1: a pointer one past the end belongs to it too) */
/* The blocks of one owner go back to the system -- on several threads: unmapping 29 GB (the reads of 2 M x 15 kb) from one thread took 1.4 s of the
 * CLI's 7.3 (21 GB/s, tools/ubench/pin_rates.hip); the kernel frees the pages of different mappings side by side. */
typedef struct { host_arena_t *blk; size_t n; } release_job_t;
static void release_worker(void *arg, int tid, int n_threads)
{
    const release_job_t *j = (const release_job_t *) arg;
    size_t i;
    for (i = (size_t) tid; i < j->n; i += (size_t) n_threads) {
        if (j->blk[i].raw) {
            /* (the pages go first, under the mapping lock held for READING -- threads do that side by side; munmap takes it for writing, and what is left
             *  for it to do is the empty mapping.  OATK_HOST_RELEASE_PLAIN=1: munmap alone, as until r04) */
            static int plain = -1;
            if (plain < 0) { const char *e = getenv("OATK_HOST_RELEASE_PLAIN"); plain = e && e[0] == '1'; }
            if (!plain) (void) madvise(j->blk[i].raw, j->blk[i].raw_size, MADV_DONTNEED);
            munmap(j->blk[i].raw, j->blk[i].raw_size);
        } else free(j->blk[i].base);
    }
}
static void arena_release(const void *owner)
{
    release_job_t job = {0, 0};
    pthread_mutex_lock(&g_ar_mu);
    size_t i, k = 0, n = 0;
    for (i = 0; i < g_nar; ++i) n += g_ar[i].owner == owner;
    if (n) job.blk = (host_arena_t *) malloc(n * sizeof(host_arena_t));
    for (i = 0; i < g_nar; ++i) {
        if (g_ar[i].owner == owner) {
            if (job.blk) job.blk[job.n++] = g_ar[i];
            else { if (g_ar[i].raw) munmap(g_ar[i].raw, g_ar[i].raw_size); else free(g_ar[i].base); }
        } else g_ar[k++] = g_ar[i];
    }
    g_nar = k;
    pthread_mutex_unlock(&g_ar_mu);
    if (job.n) oatk_par_run(release_worker, &job);
    free(job.blk);
}
/* free() for a member array that may live in an arena (then it goes with its arena) */
void oatk_sr_member_free(void *p) { if (p && arena_of(p, 0) < 0) free(p); }
void *oatk_host_arena_alloc(size_t bytes, const void *owner) { return arena_new(bytes, owner); }

/* A read's name (sr_t.sname): a block of its own that sr_destroy may free() -- or, with arenas, a piece of a 1 MiB block that goes with the reads
 * (2 M mallocs and as many frees less at 2 M reads).  `b`: the calling thread's current block. */
char *oatk_host_name_dup(const uint8_t *src, size_t len, oatk_name_bump_t *b, const void *owner)
{
    char *nm;
    if (!g_use_arena || !b) {
        nm = (char *) malloc(len + 1);
        if (!nm) return 0;
    } else {
        if (!b->p || (size_t) (b->end - b->p) < len + 1) {
            const size_t sz = len + 1 > ((size_t) 1 << 20)? len + 1 : (size_t) 1 << 20;
            b->p = arena_new(sz, owner), b->end = b->p + sz;
        }
        nm = (char *) b->p, b->p += len + 1;
    }
    memcpy(nm, src, len);
    nm[len] = 0;
    return nm;
}

/* the occurrence list of every syncmer as a block of its own again: what update_syncmer_db frees and mallocs (syncerr.c:789-790) */
void oatk_syncmer_db_own_mpos(oatk_syncmer_db_t *db)
{
    size_t i;
    long hint = -1;
    if (!db || !g_nar) return;
    for (i = 0; i < db->n; ++i) {
        oatk_syncmer_t *m = &db->a[i];
        if (m->m_pos && arena_of(m->m_pos, &hint) >= 0) m->m_pos = (uint64_t *) memcpy(xmalloc(8 * (size_t) m->cov + 8), m->m_pos, 8 * (size_t) m->cov);
    }
}

/* the chains of every read as blocks of their own again: what the reference's read_error_correction reallocs (syncerr.c:604-608) */
void oatk_sr_db_own_chains(oatk_sr_db_t *sr_db)
{
    size_t i;
    long hint = -1;
    if (!sr_db || !g_nar) return;
    for (i = 0; i < sr_db->n; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        const size_t n = r->n;
        if (r->k_mer && arena_of(r->k_mer, &hint) >= 0) r->k_mer = (uint64_t *) memcpy(xmalloc(8 * n + 8), r->k_mer, 8 * n);
        if (r->m_pos && arena_of(r->m_pos, &hint) >= 0) r->m_pos = (uint32_t *) memcpy(xmalloc(4 * n + 8), r->m_pos, 4 * n);
        if (r->s_mer && arena_of(r->s_mer, &hint) >= 0) r->s_mer = (uint64_t *) memcpy(xmalloc(8 * n + 8), r->s_mer, 8 * n);
    }
}

/* the blocks of reads [i0, i1), allocated by ONE thread: glibc grows a thread arena a few pages at a time under the address-space lock, so
 * many threads allocating gigabytes get in each other's way; the main heap grows in large steps (M_TOP_PAD below) and costs ~40 ns per block */
static void fill_alloc(fill_job_t *j)
{
    uint64_t i;
    const double t0 = host_now();
    uint8_t *first = 0, *last = 0;
    if (g_use_arena) {
        /* one block for the piece; every member 8-byte aligned inside it */
#define A8(x) (((size_t) (x) + 7) & ~(size_t) 7)
        size_t tot = 0;
        for (i = j->a0; i < j->a1; ++i) {
            const size_t hl = j->hoco_l[i], ns = j->scm_off[i + 1] - j->scm_off[i];
            tot += A8((hl + 3) / 4) + A8(hl) + A8(4 * (size_t) j->n_lrl[i]) + A8(4 * (size_t) j->n_nn[i]) + A8(4 * ns) + 16 * ns;
        }
        uint8_t *p = arena_new(tot, j->sr_db);
        for (i = j->a0; i < j->a1; ++i) {
            oatk_sr_t *r = &j->sr_db->a[j->first + i];
            const size_t hl = j->hoco_l[i], ns = j->scm_off[i + 1] - j->scm_off[i];
            r->sid = j->first + i;
            r->sname = j->names? j->names[i] : 0;
            r->hoco_l = (uint32_t) hl;
            r->hoco_s = hl? p : 0, p += A8((hl + 3) / 4);
            r->ho_rl = hl? p : 0, p += A8(hl);
            r->ho_l_rl = j->n_lrl[i]? (uint32_t *) p : 0, p += A8(4 * (size_t) j->n_lrl[i]);
            r->n_nucl = j->n_nn[i]? (uint32_t *) p : 0, p += A8(4 * (size_t) j->n_nn[i]);
            r->n = (uint32_t) ns;
            r->m_pos = ns? (uint32_t *) p : 0, p += A8(4 * ns);
            r->s_mer = ns? (uint64_t *) p : 0, p += 8 * ns;
            r->k_mer = ns? (uint64_t *) p : 0, p += 8 * ns;
        }
#undef A8
        j->t_alloc += host_now() - t0;
        return;
    }
    for (i = j->a0; i < j->a1; ++i) {
        oatk_sr_t *r = &j->sr_db->a[j->first + i];
        const uint32_t hl = j->hoco_l[i];
        const uint64_t ns = j->scm_off[i + 1] - j->scm_off[i];
        r->sid = j->first + i;                             /* reads are numbered in input order, syncmer.c:525 */
        r->sname = j->names? j->names[i] : 0;
        r->hoco_l = hl;
        /* empty arrays are NULL in the reference (kvec never allocated), syncmer.c:396-412 */
        r->hoco_s = hl? (uint8_t *) xmalloc(((size_t) hl + 3) / 4) : 0;
        r->ho_rl = hl? (uint8_t *) xmalloc(hl) : 0;
        r->ho_l_rl = j->n_lrl[i]? (uint32_t *) xmalloc(4 * (size_t) j->n_lrl[i]) : 0;
        r->n_nucl = j->n_nn[i]? (uint32_t *) xmalloc(4 * (size_t) j->n_nn[i]) : 0;
        r->n = (uint32_t) ns;
        r->m_pos = ns? (uint32_t *) xmalloc(4 * (size_t) ns) : 0;
        r->s_mer = ns? (uint64_t *) xmalloc(8 * (size_t) ns) : 0;
        r->k_mer = ns? (uint64_t *) xmalloc(8 * (size_t) ns) : 0;
        if (!first) first = r->hoco_s;
        if (r->ho_rl) last = r->ho_rl + hl;
    }
    (void) first;
    if (last) g_heap_top = last;                           /* large blocks are carved off the top of the heap one after the other */
    j->t_alloc += host_now() - t0;
}

/* Fresh heap costs a page fault per 4 KiB, and the thread that allocates would take one for nearly every block it heads (a read's arrays are
 * larger than a page): 1.2 us per read, the whole pipeline's bottleneck.  So the stretch the NEXT blocks will be carved from -- from the top
 * of the heap on, as far as the break already reaches (M_TOP_PAD keeps it far ahead) -- is declared huge-page territory and touched by the
 * copying threads first, in parallel: an atomic OR of zero changes nothing (fresh heap must stay zero for calloc; a header the allocator has
 * just written there survives) but brings the page in. */
static void fill_prefault_plan(fill_job_t *j)
{
    j->pre0 = j->pre1 = 0;
    if (g_use_arena || !g_heap_top || j->a1 <= j->a0 || !heap_tune_wanted()) return;          /* (an arena is a fresh mapping: its pages are touched by the copies themselves) */
    if ((pid_t) syscall(SYS_gettid) != getpid()) return;       /* the main heap is the main thread's arena: only there do the next blocks come from its top */
    uint64_t need = 0, i;
    for (i = j->a0; i < j->a1; ++i) need += (uint64_t) j->hoco_l[i] + ((uint64_t) j->hoco_l[i] + 3) / 4 + 20 * (j->scm_off[i + 1] - j->scm_off[i]) + 160;
    uint8_t *brk_now = (uint8_t *) sbrk(0), *lo = g_heap_top + 4096, *hi = g_heap_top + need + (need >> 6);
    if (brk_now == (uint8_t *) -1 || lo >= brk_now || g_heap_top + ((uint64_t) 8 << 30) < brk_now) return;       /* not the main heap */
    if (hi > brk_now) hi = brk_now;
    lo = (uint8_t *) (((uintptr_t) lo + 4095) & ~(uintptr_t) 4095), hi = (uint8_t *) ((uintptr_t) hi & ~(uintptr_t) 4095);
    if (hi <= lo) return;
    const uintptr_t h0 = ((uintptr_t) lo + (((uintptr_t) 2 << 20) - 1)) & ~(((uintptr_t) 2 << 20) - 1), h1 = (uintptr_t) hi & ~(((uintptr_t) 2 << 20) - 1);
    if (h1 > h0) (void) madvise((void *) h0, h1 - h0, MADV_HUGEPAGE);
    j->pre0 = lo, j->pre1 = hi;
}

static void fill_prefault_worker(void *arg, int tid, int n_threads);

static void fill_prefault(const fill_job_t *j, int tid, int n_threads)
{
    if (j->pre1 <= j->pre0) return;
    const uint64_t pages = (uint64_t) (j->pre1 - j->pre0) >> 12, a = pages * (uint64_t) tid / (uint64_t) n_threads, b = pages * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    uint64_t p;
    for (p = a; p < b; ++p) (void) __atomic_fetch_or((uint64_t *) (j->pre0 + (p << 12)), 0, __ATOMIC_RELAXED);
}

static void fill_prefault_worker(void *arg, int tid, int n_threads) { fill_prefault((const fill_job_t *) arg, tid, n_threads); }

/* ... and filled by all of them (the first touch of every page happens here, in parallel) */
static void fill_worker(void *arg, int tid, int n_threads)
{
    fill_job_t *j = (fill_job_t *) arg;
    /* thread 0 allocates the NEXT piece's blocks while the others copy this one (alone, it does both) */
    if (tid == 0 && j->a1 > j->a0) fill_alloc(j);
    if (n_threads > 1) {
        if (tid == 0) return;
        --tid, --n_threads;
    }
    fill_prefault(j, tid, n_threads);                      /* ahead of the allocating thread, before the copying */
    const uint64_t n = j->i1 - j->i0, a = j->i0 + n * (uint64_t) tid / (uint64_t) n_threads, b = j->i0 + n * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    uint64_t i;
    for (i = a; i < b; ++i) {
        oatk_sr_t *r = &j->sr_db->a[j->first + i];
        const uint32_t hl = r->hoco_l;
        const uint64_t ns = r->n, os = j->scm_off[i] - j->scm0;
        if (hl) {
            memcpy(r->hoco_s, j->buf->hs + (j->off[i] - j->rl0) / 4, ((size_t) hl + 3) / 4);
            memcpy(r->ho_rl, j->buf->rl + (j->off[i] - j->rl0), hl);
        }
        if (j->n_lrl[i]) memcpy(r->ho_l_rl, j->lrl_val + j->o_lrl[i], 4 * (size_t) j->n_lrl[i]);
        if (j->n_nn[i]) {
            uint32_t t;
            for (t = 0; t < j->n_nn[i]; ++t) r->n_nucl[t] = (uint32_t) j->nn_key[j->o_nn[i] + t];     /* low word = raw coordinate */
        }
        if (ns) {
            memcpy(r->m_pos, j->buf->m_pos + os, 4 * (size_t) ns);
            memcpy(r->s_mer, j->buf->s_mer + os, 8 * (size_t) ns);
            memcpy(r->k_mer, j->buf->k_hash + os, 8 * (size_t) ns);
        }
    }
}


/* ---- arenas, filled WITHOUT a host copy (round 4) ------------------------------------------------------------------------------------------
 * Until round 3 a piece's arrays came over PCIe into a page-locked staging buffer and were then copied into the arena by the host threads: every
 * byte of the reads' structs (29 GB at 2 M reads) crossed the host's memory twice, 2.3 of the CLI's 8.7 s.  Member pointers into an arena may point
 * anywhere, so the arena of a piece is now laid out exactly as the piece lies on the DEVICE -- the run-length slab with the reads at their 64-byte
 * aligned offsets, the packed-base slab at a quarter of them, the three per-syncmer arrays back to back -- and is itself the destination of the
 * copies: a fresh anonymous mapping on transparent huge pages, touched by the host threads while the piece before is on the bus (294 GB/s with 32
 * threads on the bench box, tools/ubench/pin_rates.hip), page-locked for the duration of the copy (hipHostRegister of touched huge pages: 0.02 s per
 * 8 GB; hipHostMalloc would be 5.8 GB/s) and unlocked when it has landed.  What the host still does per read is set eight pointers. */
#define ZC_RL_BYTES ((uint64_t) 256 << 20)            /* run-length slab bytes per piece */

typedef struct {
    uint8_t *raw; size_t raw_size;                    /* the mapping */
    uint8_t *base; size_t size;                       /* 2 MiB aligned part that is used */
    uint8_t *rl, *hs; uint64_t *s_mer, *k_hash; uint32_t *m_pos, *lrl, *nn;
    uint64_t i0, i1, rl_bytes, ns;
} zc_piece_t;

typedef struct {
    oatk_sr_db_t *sr_db;
    uint64_t first;
    const uint64_t *off, *scm_off;
    const uint32_t *hoco_l, *n_nn, *n_lrl, *lrl_val;
    const uint64_t *nn_key, *o_nn, *o_lrl;
    char **names;
    const zc_piece_t *cur;                            /* structs of this piece are set ... */
    const zc_piece_t *next;                           /* ... while this one's pages are touched */
} zc_job_t;

static int zc_wanted(void)
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("OATK_HOST_ZERO_COPY"); v = !(e && e[0] == '0'); }
    return v;
}

static int zc_map(zc_piece_t *pc, const zc_job_t *j, uint64_t i0, uint64_t i1)
{
#define A64(x) (((size_t) (x) + 63) & ~(size_t) 63)
    const uint64_t last = i1 - 1;
    pc->i0 = i0, pc->i1 = i1;
    pc->rl_bytes = j->off[last] + j->hoco_l[last] - j->off[i0];
    pc->ns = j->scm_off[i1] - j->scm_off[i0];
    const size_t n_lrl = (size_t) (j->o_lrl[i1] - j->o_lrl[i0]), n_nn = (size_t) (j->o_nn[i1] - j->o_nn[i0]);
    const size_t o_rl = 0, o_hs = A64(pc->rl_bytes + 64), o_sm = o_hs + A64(pc->rl_bytes / 4 + 128), o_kh = o_sm + A64(8 * pc->ns), o_mp = o_kh + A64(8 * pc->ns),
                 o_lrl = o_mp + A64(4 * pc->ns), o_nn = o_lrl + A64(4 * n_lrl), tot = o_nn + A64(4 * n_nn) + 64;
    const size_t HP = (size_t) 2 << 20;
    pc->raw_size = ((tot + HP - 1) & ~(HP - 1)) + HP;
    pc->raw = (uint8_t *) mmap(0, pc->raw_size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (pc->raw == MAP_FAILED) { pc->raw = 0; return OATK_E_NOMEM; }
    pc->base = (uint8_t *) (((uintptr_t) pc->raw + HP - 1) & ~(uintptr_t) (HP - 1));
    pc->size = (tot + HP - 1) & ~(HP - 1);
    {
        static int thp = -1;
        if (thp < 0) { const char *e = getenv("OATK_HOST_THP"); thp = !(e && e[0] == '0'); }
        if (thp) (void) madvise(pc->base, pc->size, MADV_HUGEPAGE);
    }
    pc->rl = pc->base + o_rl, pc->hs = pc->base + o_hs, pc->s_mer = (uint64_t *) (pc->base + o_sm), pc->k_hash = (uint64_t *) (pc->base + o_kh);
    pc->m_pos = (uint32_t *) (pc->base + o_mp), pc->lrl = (uint32_t *) (pc->base + o_lrl), pc->nn = (uint32_t *) (pc->base + o_nn);
    arena_register(pc->base, pc->size, j->sr_db, pc->raw, pc->raw_size);
    return OATK_OK;
#undef A64
}

static void zc_worker(void *arg, int tid, int n_threads)
{
    const zc_job_t *j = (const zc_job_t *) arg;
    if (j->next && j->next->base) {                    /* first touch of the next piece's pages: a huge page per 2 MiB where the kernel grants them */
        const size_t pages = j->next->size >> 12, a = pages * (size_t) tid / (size_t) n_threads, b = pages * (size_t) (tid + 1) / (size_t) n_threads;
        size_t p;
        for (p = a; p < b; ++p) *(volatile uint8_t *) (j->next->base + (p << 12)) = 0;
    }
    if (j->cur) {
        const zc_piece_t *pc = j->cur;
        const uint64_t n = pc->i1 - pc->i0, a = pc->i0 + n * (uint64_t) tid / (uint64_t) n_threads, b = pc->i0 + n * (uint64_t) (tid + 1) / (uint64_t) n_threads;
        const uint64_t rl0 = j->off[pc->i0], scm0 = j->scm_off[pc->i0], lrl0 = j->o_lrl[pc->i0], nn0 = j->o_nn[pc->i0];
        uint64_t i;
        for (i = a; i < b; ++i) {
            oatk_sr_t *r = &j->sr_db->a[j->first + i];
            const uint32_t hl = j->hoco_l[i];
            const uint64_t ns = j->scm_off[i + 1] - j->scm_off[i], os = j->scm_off[i] - scm0;
            r->sid = j->first + i;                         /* reads are numbered in input order, syncmer.c:525 */
            r->sname = j->names? j->names[i] : 0;
            r->hoco_l = hl;
            /* empty arrays are NULL in the reference (kvec never allocated), syncmer.c:396-412 */
            r->hoco_s = hl? pc->hs + (j->off[i] - rl0) / 4 : 0;
            r->ho_rl = hl? pc->rl + (j->off[i] - rl0) : 0;
            r->n = (uint32_t) ns;
            r->m_pos = ns? pc->m_pos + os : 0;
            r->s_mer = ns? pc->s_mer + os : 0;
            r->k_mer = ns? pc->k_hash + os : 0;
            r->ho_l_rl = 0, r->n_nucl = 0;
            if (j->n_lrl[i]) {
                r->ho_l_rl = pc->lrl + (j->o_lrl[i] - lrl0);
                memcpy(r->ho_l_rl, j->lrl_val + j->o_lrl[i], 4 * (size_t) j->n_lrl[i]);
            }
            if (j->n_nn[i]) {
                uint32_t t;
                r->n_nucl = pc->nn + (j->o_nn[i] - nn0);
                for (t = 0; t < j->n_nn[i]; ++t) r->n_nucl[t] = (uint32_t) j->nn_key[j->o_nn[i] + t];     /* low word = raw coordinate */
            }
        }
    }
}

static double g_zc_t[4];          /* (log only) mapping, first touch + pointers, page-locking, unlocking: summed over the pieces of a call */
static int fill_range_zero_copy(oatk_hip_ctx *ctx, zc_job_t *j, uint64_t n_reads, const void *d_rl, const void *d_hs, const void *d_mp, const void *d_sm, const void *d_kh,
                                double *t_prep, double *t_wait)
{
    zc_piece_t pc[2];
    int rc = OATK_OK, cur = 0, have_next;
    uint64_t p0 = 0, p1;
    memset(pc, 0, sizeof(pc));
    /* piece limits as in the copying form; a single read larger than the default makes its own piece */
#define NEXT_END(a, e) do { e = (a) + 1; while (e < n_reads && j->off[e] + j->hoco_l[e] - j->off[a] <= ZC_RL_BYTES) ++e; } while (0)
    NEXT_END(p0, p1);
    double t0 = host_now();
    rc = zc_map(&pc[0], j, p0, p1);
    if (rc) return rc;
    j->cur = 0, j->next = &pc[0];
    oatk_par_run(zc_worker, j);
    *t_prep += host_now() - t0;
    /* Page-locking a piece and releasing the one before it both run while a piece is on the bus (r04: they were done around the copy, ~1.5 ms per 256 MB piece
     * of which the copy itself takes 4.4): lock the first piece, then per piece -- queue its copy, unlock the piece before, map + touch + lock the next, wait. */
    int locked[2] = {0, 0};
    zc_piece_t *prev = 0;
    int prev_locked = 0;
    locked[0] = oatk_hip_host_register(ctx, pc[0].base, pc[0].size) == OATK_OK;      /* (not page-locked the copies still arrive, staged by the runtime) */
    for (;;) {
        zc_piece_t *P = &pc[cur];
        t0 = host_now();
        rc = oatk_hip_d2h_async(ctx, P->rl, (const uint8_t *) d_rl + j->off[P->i0], P->rl_bytes);
        if (!rc) rc = oatk_hip_d2h_async(ctx, P->hs, (const uint8_t *) d_hs + j->off[P->i0] / 4, (P->rl_bytes + 3) / 4 + 1);
        if (!rc) rc = oatk_hip_d2h_async(ctx, P->m_pos, (const uint32_t *) d_mp + j->scm_off[P->i0], P->ns * 4);
        if (!rc) rc = oatk_hip_d2h_async(ctx, P->s_mer, (const uint64_t *) d_sm + j->scm_off[P->i0], P->ns * 8);
        if (!rc) rc = oatk_hip_d2h_async(ctx, P->k_hash, (const uint64_t *) d_kh + j->scm_off[P->i0], P->ns * 8);
        { const double tu = host_now(); if (prev && prev_locked) (void) oatk_hip_host_unregister(ctx, prev->base); g_zc_t[3] += host_now() - tu; }
        prev = 0;
        /* while it is on the bus: the next piece's mapping is made, touched and locked, this piece's structs are set */
        p0 = P->i1, have_next = 0;
        if (!rc && p0 < n_reads) {
            NEXT_END(p0, p1);
            const double tm = host_now();
            rc = zc_map(&pc[cur ^ 1], j, p0, p1);
            g_zc_t[0] += host_now() - tm;
            have_next = !rc;
        }
        j->cur = P, j->next = have_next? &pc[cur ^ 1] : 0;
        { const double tt = host_now(); if (!rc) oatk_par_run(zc_worker, j); g_zc_t[1] += host_now() - tt; }
        { const double tr = host_now(); if (have_next) locked[cur ^ 1] = oatk_hip_host_register(ctx, pc[cur ^ 1].base, pc[cur ^ 1].size) == OATK_OK; g_zc_t[2] += host_now() - tr; }
        *t_prep += host_now() - t0, t0 = host_now();
        {
            const int rs = oatk_hip_sync(ctx);                 /* the piece has landed */
            if (!rc) rc = rs;
        }
        *t_wait += host_now() - t0;
        if (rc || !have_next) {                                /* (nothing is on the bus any more) */
            if (locked[cur]) (void) oatk_hip_host_unregister(ctx, P->base);
            if (have_next && locked[cur ^ 1]) (void) oatk_hip_host_unregister(ctx, pc[cur ^ 1].base);
            if (rc) return rc;
        }
        j->sr_db->n = j->first + P->i1;
        if (!have_next) break;
        prev = P, prev_locked = locked[cur];
        cur ^= 1;
    }
#undef NEXT_END
    return OATK_OK;
}

int oatk_sr_db_fill_resident(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint64_t *off, uint64_t n_reads, char **names)
{
    if (n_reads == 0) return OATK_OK;
    /* zeroed, and counted only as far as it is filled: after a failure half way sr_db_destroy / oatk_sr_db_clean free what exists */
    sr_db->a = (oatk_sr_t *) calloc(n_reads, sizeof(oatk_sr_t));
    if (!sr_db->a) return OATK_E_NOMEM;
    sr_db->n = 0, sr_db->m = n_reads;
    return oatk_sr_db_fill_range(ctx, sr_db, 0, off, n_reads, names);
}

int oatk_sr_db_fill_range(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, uint64_t first, const uint64_t *off, uint64_t n_reads, char **names)
{
    int rc = 0;
    if (n_reads == 0) return OATK_OK;
    if (!sr_db->a || sr_db->m < first + n_reads) return OATK_E_ARG;
    /* One malloc'ed block per member array (no arenas): gigabytes of small blocks are about to be allocated.  Letting the heap grow in 1 GiB steps and
     * touching the fresh pages from the copying threads (fill_prefault) is worth 2x on this path, but it changes the HOST PROGRAM's allocator settings and
     * leans on glibc internals (a contiguous main arena), so it is opt-in: OATK_HOST_HEAP_TUNE=1.  The arena mode needs neither. */
    const int heap_tune = !g_use_arena && heap_tune_wanted();
    if (heap_tune) { (void) mallopt(M_TOP_PAD, 1 << 30); (void) mallopt(M_TRIM_THRESHOLD, 1 << 30); }
    const double t_begin = host_now();
    double t_copy = 0, t_wait = 0, t_setup = 0;
    fill_job_t job;
    memset(&job, 0, sizeof(job));

    uint64_t b, i;
    uint32_t *hoco_l = 0, *n_nn = 0, *n_lrl = 0, *lrl_val = 0;
    uint64_t *nn_key = 0, *scm_off = 0, *o_nn = 0, *o_lrl = 0;
    hoco_l = (uint32_t *) fetch(ctx, OATK_BUF_HOCO_L, &b, &rc); if (rc) goto done;
    n_nn = (uint32_t *) fetch(ctx, OATK_BUF_N_NN, &b, &rc); if (rc) goto done;
    n_lrl = (uint32_t *) fetch(ctx, OATK_BUF_N_LRL, &b, &rc); if (rc) goto done;
    scm_off = (uint64_t *) fetch(ctx, OATK_BUF_SCM_OFF, &b, &rc); if (rc) goto done;
    nn_key = (uint64_t *) fetch(ctx, OATK_BUF_NN_KEY, &b, &rc); if (rc) goto done;
    lrl_val = (uint32_t *) fetch(ctx, OATK_BUF_LRL_VAL, &b, &rc); if (rc) goto done;
    o_nn = (uint64_t *) xmalloc(8 * (n_reads + 1)), o_lrl = (uint64_t *) xmalloc(8 * (n_reads + 1));
    for (i = 0, o_nn[0] = o_lrl[0] = 0; i < n_reads; ++i) o_nn[i + 1] = o_nn[i] + n_nn[i], o_lrl[i + 1] = o_lrl[i] + n_lrl[i];

    const void *d_rl = 0, *d_hs = 0, *d_mp = 0, *d_sm = 0, *d_kh = 0;
    rc = oatk_hip_buffer(ctx, OATK_BUF_HO_RL, &d_rl, &b); if (rc) goto done;
    rc = oatk_hip_buffer(ctx, OATK_BUF_HOCO_S, &d_hs, &b); if (rc) goto done;
    rc = oatk_hip_buffer(ctx, OATK_BUF_POS_MPOS, &d_mp, &b); if (rc) goto done;
    rc = oatk_hip_buffer(ctx, OATK_BUF_POS_SMER, &d_sm, &b); if (rc) goto done;
    rc = oatk_hip_buffer(ctx, OATK_BUF_POS_HASH, &d_kh, &b); if (rc) goto done;

    if (g_use_arena && zc_wanted()) {
        zc_job_t zj;
        double t_prep = 0;
        memset(&zj, 0, sizeof(zj));
        zj.sr_db = sr_db, zj.first = first, zj.off = off, zj.scm_off = scm_off, zj.hoco_l = hoco_l, zj.n_nn = n_nn, zj.n_lrl = n_lrl, zj.lrl_val = lrl_val;
        zj.nn_key = nn_key, zj.o_nn = o_nn, zj.o_lrl = o_lrl, zj.names = names;
        t_setup = host_now() - t_begin;
        rc = fill_range_zero_copy(ctx, &zj, n_reads, d_rl, d_hs, d_mp, d_sm, d_kh, &t_prep, &t_wait);
        t_copy = t_prep;
        goto done;
    }
    /* piece limits: a single read larger than the defaults makes its own (larger) piece */
    uint64_t cap_rl = off[n_reads - 1] + (((uint64_t) hoco_l[n_reads - 1] + 63) & ~63ULL) + 64, cap_scm = scm_off[n_reads] + 1;      /* small inputs: one piece */
    if (cap_rl > FILL_RL_BYTES) cap_rl = FILL_RL_BYTES;
    if (cap_scm > FILL_SCM) cap_scm = FILL_SCM;
    for (i = 0; i < n_reads; ++i) {
        const uint64_t need = (((uint64_t) hoco_l[i] + 63) & ~63ULL) + 64, ns = scm_off[i + 1] - scm_off[i];
        if (need > cap_rl) cap_rl = need;
        if (ns > cap_scm) cap_scm = ns;
    }
    const uint64_t one = ((cap_rl + 63) & ~63ULL) + ((cap_rl / 4 + 128) & ~63ULL) + cap_scm * 20 + 256;
    uint8_t *stage = (uint8_t *) oatk_hip_staging(ctx, 2 * one);
    if (!stage) { rc = OATK_E_NOMEM; goto done; }
    fill_buf_t buf[2];
    for (i = 0; i < 2; ++i) {
        uint8_t *p = stage + i * one;
        buf[i].rl = p, p += (cap_rl + 63) & ~63ULL;
        buf[i].hs = p, p += (cap_rl / 4 + 128) & ~63ULL;
        buf[i].s_mer = (uint64_t *) p, p += cap_scm * 8;
        buf[i].k_hash = (uint64_t *) p, p += cap_scm * 8;
        buf[i].m_pos = (uint32_t *) p;
    }

    job.sr_db = sr_db, job.first = first, job.off = off, job.scm_off = scm_off, job.hoco_l = hoco_l, job.n_nn = n_nn, job.n_lrl = n_lrl, job.lrl_val = lrl_val;
    job.nn_key = nn_key, job.o_nn = o_nn, job.o_lrl = o_lrl, job.names = names;

    t_setup = host_now() - t_begin;
    uint64_t p0 = 0, p1 = 0, q0 = 0, q1 = 0;          /* piece in flight: reads [p0, p1); piece being cut: [q0, q1) */
    int flight = -1, which = 0;
    for (;;) {
        /* queue the next piece */
        p0 = p1;
        if (p0 < n_reads) {
            p1 = p0 + 1;
            while (p1 < n_reads && off[p1] + (((uint64_t) hoco_l[p1] + 63) & ~63ULL) - off[p0] <= cap_rl && scm_off[p1 + 1] - scm_off[p0] <= cap_scm) ++p1;
            const uint64_t rl_bytes = off[p1 - 1] + hoco_l[p1 - 1] - off[p0], ns = scm_off[p1] - scm_off[p0];
            const fill_buf_t *B = &buf[which];
            rc = oatk_hip_d2h_async(ctx, B->rl, (const uint8_t *) d_rl + off[p0], rl_bytes); if (rc) goto done;
            rc = oatk_hip_d2h_async(ctx, B->hs, (const uint8_t *) d_hs + off[p0] / 4, (rl_bytes + 3) / 4 + 1); if (rc) goto done;
            rc = oatk_hip_d2h_async(ctx, B->m_pos, (const uint32_t *) d_mp + scm_off[p0], ns * 4); if (rc) goto done;
            rc = oatk_hip_d2h_async(ctx, B->s_mer, (const uint64_t *) d_sm + scm_off[p0], ns * 8); if (rc) goto done;
            rc = oatk_hip_d2h_async(ctx, B->k_hash, (const uint64_t *) d_kh + scm_off[p0], ns * 8); if (rc) goto done;
        }
        /* cut the piece that arrived before it (its blocks exist already), allocating the queued one's meanwhile */
        job.a0 = p0 < n_reads? p0 : 0, job.a1 = p0 < n_reads? p1 : 0;
        fill_prefault_plan(&job);
        if (flight >= 0) {
            const double tc = host_now();
            job.buf = &buf[flight], job.i0 = q0, job.i1 = q1, job.rl0 = off[q0], job.scm0 = scm_off[q0];
            oatk_par_run(fill_worker, &job);
            sr_db->n = first + q1;
            t_copy += host_now() - tc;
        } else {
            oatk_par_run(fill_prefault_worker, &job);
            fill_alloc(&job);
        }
        if (p0 >= n_reads) break;
        {
            const double tw = host_now();
            rc = oatk_hip_sync(ctx); if (rc) goto done;   /* the queued piece has landed */
            t_wait += host_now() - tw;
        }
        flight = which, which ^= 1, q0 = p0, q1 = p1;
    }
done:
    free(hoco_l); free(n_nn); free(n_lrl); free(nn_key); free(lrl_val); free(scm_off); free(o_nn); free(o_lrl);
    if (heap_tune) { (void) mallopt(M_TOP_PAD, 128 * 1024); (void) mallopt(M_TRIM_THRESHOLD, 128 * 1024); }      /* glibc's defaults back: the host program's heap is its own again */
    if (host_log() && g_use_arena && zc_wanted()) {
        fprintf(stderr, "[M::%s] ... of the preparation: mapping %.4f, first touch + pointers %.4f, page-locking %.4f, unlocking %.4f s\n", __func__, g_zc_t[0], g_zc_t[1], g_zc_t[2], g_zc_t[3]);
        g_zc_t[0] = g_zc_t[1] = g_zc_t[2] = g_zc_t[3] = 0;
    }
    if (host_log()) fprintf(stderr, "[M::%s] %lu reads into sr_db_t: %.3f s on %d host threads (setup %.3f, %s %.3f beside %s %.3f, waiting for PCIe %.3f)\n",
                            __func__, (unsigned long) n_reads, host_now() - t_begin, oatk_host_threads(), t_setup, g_use_arena && zc_wanted()? "no host copy: block allocation" : "block allocation", job.t_alloc,
                            g_use_arena && zc_wanted()? "mapping + first touch + page-locking + pointers" : "copying", t_copy, t_wait);
    return rc;
}

typedef struct {
    oatk_syncmer_db_t *db;
    oatk_sr_db_t *sr_db;
    const uint64_t *h, *s;
    const uint32_t *cov;
    const uint64_t *occ_off, *occ, *kid;
    uint64_t *kid_off;
    int adopt;
} collect_job_t;

static void collect_worker(void *arg, int tid, int n_threads)
{
    const collect_job_t *j = (const collect_job_t *) arg;
    uint64_t i, t;
    const uint64_t ns = j->db->n, a = ns * (uint64_t) tid / (uint64_t) n_threads, b = ns * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    for (i = a; i < b; ++i) {
        oatk_syncmer_t *m = &j->db->a[i];
        m->h = j->h[i], m->s = j->s[i], m->cov = j->cov[i], m->del = 0;
        m->m_pos = j->adopt? (j->cov[i]? (uint64_t *) j->occ + j->occ_off[i] : 0)          /* arenas: the fetched array IS the table's storage */
                           : (uint64_t *) memcpy(xmalloc(8 * (size_t) j->cov[i]), j->occ + j->occ_off[i], 8 * (size_t) j->cov[i]);
        j->db->c[i] = 1;                                   /* syncmer.c:1443-1444 */
    }
    /* reads: k-mer hash -> syncmer id << 1 (syncmer.c:1378) */
    const uint64_t nr = j->sr_db->n, ra = nr * (uint64_t) tid / (uint64_t) n_threads, rb = nr * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    for (i = ra; i < rb; ++i) {
        oatk_sr_t *r = &j->sr_db->a[i];
        const uint64_t *k = j->kid + j->kid_off[i];
        for (t = 0; t < r->n; ++t) r->k_mer[t] = k[t];
    }
}

oatk_syncmer_db_t *oatk_collect_syncmer_from_reads(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, int *rc_out)
{
    int rc = oatk_hip_count(ctx);
    if (rc_out) *rc_out = rc;
    if (rc == OATK_E_SMER) {                               /* fatal in the reference, syncmer.c:1370-1375 */
        fprintf(stderr, "[E::%s] identical kmers have different smers\n", __func__);
        exit(EXIT_FAILURE);
    }
    if (rc) return 0;
    oatk_hip_info_t inf;
    oatk_hip_info(ctx, &inf);
    if (inf.n_occ == 0) return 0;                          /* syncmer.c:1414-1417 */

    uint64_t b;
    uint64_t *h = 0, *s = 0, *occ_off = 0, *occ = 0, *kid = 0;
    uint32_t *cov = 0;
    oatk_syncmer_db_t *db = 0;
    h = (uint64_t *) fetch(ctx, OATK_BUF_SCM_H, &b, &rc); if (rc) goto fail;
    s = (uint64_t *) fetch(ctx, OATK_BUF_SCM_S, &b, &rc); if (rc) goto fail;
    cov = (uint32_t *) fetch(ctx, OATK_BUF_SCM_COV, &b, &rc); if (rc) goto fail;
    occ_off = (uint64_t *) fetch(ctx, OATK_BUF_SCM_OCC_OFF, &b, &rc); if (rc) goto fail;
    occ = (uint64_t *) fetch(ctx, OATK_BUF_SCM_OCC, &b, &rc); if (rc) goto fail;
    kid = (uint64_t *) fetch(ctx, OATK_BUF_POS_KID, &b, &rc); if (rc) goto fail;
    db = oatk_host_build_syncmer_db(sr_db, inf.n_scm, inf.n_occ, h, s, cov, occ_off, &occ, kid);
fail:
    free(h); free(s); free(cov); free(occ_off); free(occ); free(kid);
    if (rc_out) *rc_out = rc;
    return db;
}

oatk_syncmer_db_t *oatk_host_build_syncmer_db(oatk_sr_db_t *sr_db, uint64_t n_scm, uint64_t n_occ, const uint64_t *h, const uint64_t *s, const uint32_t *cov,
                                              const uint64_t *occ_off, uint64_t **occ, const uint64_t *kid)
{
    oatk_syncmer_db_t *db = (oatk_syncmer_db_t *) xmalloc(sizeof(oatk_syncmer_db_t));
    db->n = db->m = n_scm;
    db->a = (oatk_syncmer_t *) xmalloc(sizeof(oatk_syncmer_t) * n_scm);
    db->c = (uint16_t *) xmalloc(sizeof(uint16_t) * n_scm);
    db->h = 0;
    collect_job_t job = {db, sr_db, h, s, cov, occ_off, *occ, kid, 0, g_use_arena};
    /* where each read's ids start in the id array: the chain lengths have not changed since the scan */
    job.kid_off = (uint64_t *) xmalloc(8 * (sr_db->n + 1));
    {
        uint64_t i;
        for (i = 0, job.kid_off[0] = 0; i < sr_db->n; ++i) job.kid_off[i + 1] = job.kid_off[i] + sr_db->a[i].n;
    }
    oatk_par_run(collect_worker, &job);
    free(job.kid_off);
    if (job.adopt) oatk_host_arena_adopt(*occ, 8 * (size_t) n_occ, db), *occ = 0;
    return db;
}

static void clean_worker(void *arg, int tid, int n_threads)
{
    oatk_sr_db_t *sr_db = (oatk_sr_db_t *) arg;
    const size_t a = sr_db->n * (size_t) tid / (size_t) n_threads, b = sr_db->n * (size_t) (tid + 1) / (size_t) n_threads;
    size_t i;
    long hint = -1;
    for (i = a; i < b; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        void *m[8] = {r->sname, r->hoco_s, r->ho_rl, r->ho_l_rl, r->n_nucl, r->m_pos, r->s_mer, r->k_mer};
        int k;
        for (k = 0; k < 8; ++k) if (m[k] && arena_of(m[k], &hint) < 0) free(m[k]);
    }
}

void oatk_sr_db_clean(oatk_sr_db_t *sr_db)
{
    size_t i;
    if (!sr_db) return;
    if (g_nar) {
        oatk_par_run(clean_worker, sr_db);          /* members outside the arenas (chains the reference realloc'ed, names of a malloc'ing caller) are blocks of their own */
        arena_release(sr_db);
    } else {
    for (i = 0; i < sr_db->n; ++i) {
        oatk_sr_t *r = &sr_db->a[i];
        free(r->sname); free(r->hoco_s); free(r->ho_rl); free(r->ho_l_rl); free(r->n_nucl);
        free(r->m_pos); free(r->s_mer); free(r->k_mer);
    }
    }
    free(sr_db->a);
    free(sr_db->stats);
    sr_db->a = 0, sr_db->n = sr_db->m = 0, sr_db->stats = 0;
}

void oatk_sr_destroy(oatk_sr_t *r)
{
    if (!r) return;
    void *m[8] = {r->sname, r->hoco_s, r->ho_rl, r->ho_l_rl, r->n_nucl, r->m_pos, r->s_mer, r->k_mer};
    int k;
    for (k = 0; k < 8; ++k) oatk_sr_member_free(m[k]);
}

void oatk_syncmer_db_clean(oatk_syncmer_db_t *db)      /* syncmer.c:1094-1103 */
{
    size_t i;
    if (!db) return;
    if (g_nar) {
        long hint = -1;
        for (i = 0; i < db->n; ++i) if (db->a[i].m_pos && arena_of(db->a[i].m_pos, &hint) < 0) free(db->a[i].m_pos);
        arena_release(db);
    } else {
        for (i = 0; i < db->n; ++i) free(db->a[i].m_pos);
    }
    free(db->a); free(db->c); free(db->h);
    db->a = 0, db->n = db->m = 0, db->c = 0, db->h = 0;
}

void oatk_syncmer_db_destroy(oatk_syncmer_db_t *db)
{
    if (!db) return;
    oatk_syncmer_db_clean(db);
    free(db);
}

/* malloc'ed, initialised like sr_db_init (syncmer.c:1060-1067); freed by the reference's sr_db_destroy or oatk_sr_db_clean + free */
oatk_sr_db_t *oatk_sr_db_new(int k, int s)
{
    oatk_sr_db_t *db = (oatk_sr_db_t *) xmalloc(sizeof(oatk_sr_db_t));
    db->n = db->m = 0, db->a = 0, db->k = k, db->s = s, db->stats = 0;
    return db;
}
