/*
 * oatk_amd/csrc/host/ingest_host.c -- files to the device reader (include/oatk_hip_ingest.h), in place of sstream_open / sstream_read
 * (sstream.c:70-103): several files are read one after the other as one stream of records; plain or gzip'ed (zlib inflates, exactly as the
 * reference's gzdopen does, sstream.c:50).  The host only moves bytes, and moves them once: a plain file goes from the page cache straight
 * into page-locked memory (pread by the host threads, piece by piece) and from there over PCIe while the next piece is being read; a
 * gzip'ed one is inflated AS A STREAM (gzsrc.c: BGZF members in parallel, plain members one after the other) straight into the page-locked
 * window it is uploaded from, while the windows before it are parsed and scanned on the device (round 4; until then it was inflated whole,
 * in front of everything).  The records are found on the device.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include "oatk_hip_ingest.h"
#include "oatk_syncasm.h"
#include "host_internal.h"
#include "ingest_estimate.h"

#define UP_CHUNK ((uint64_t) 96 << 20)                 /* bytes per upload piece */

/* one input file as a segment of the text: either a descriptor to pread from or inflated bytes in memory */
/* set by sr_read_stream when what it returns OATK_E_NOMEM for is a record that did not fit a window (a genuine allocation failure must not be answered with windows
 * eight times the size, and a pipe cannot be read again) */
static __thread int g_record_too_long;

typedef struct {
    int fd;                        /* >= 0: plain file */
    uint8_t *mem;                  /* inflated content (gzip'ed file), malloc'ed */
    uint64_t size;                 /* bytes of content */
    uint64_t base;                 /* where the content starts in the text */
    int add_nl;                    /* the file does not end in a newline: one is put behind it (a fresh kseq starts every file at a header
                                    * line, sstream.c:91-97 -- its last line must not glue to the next file's first header) */
    uint8_t *map;                  /* lazily mapped (no populate) for the header lines only */
    int borrowed;                  /* mem belongs to the caller */
    int pinned;                    /* ... and is page-locked: it goes over PCIe as it lies */
    oatk_gzsrc_t *gz;              /* a gzip'ed file read as a stream (sr_read path): size and base are unknown */
    uint64_t fsize;                /* bytes of the file itself */
    int no_map;                    /* its pages could not be page-locked where they lie: windows of it are staged */
} seg_t;

static void seg_close(seg_t *s, int n)
{
    int i;
    for (i = 0; i < n; ++i) {
        if (s[i].map) munmap(s[i].map, (size_t) s[i].size);
        if (s[i].fd >= 0) close(s[i].fd);
        if (s[i].gz) oatk_gzsrc_close(s[i].gz);
        if (!s[i].borrowed) free(s[i].mem);
    }
    free(s);
}

static int inflate_file(const char *path, seg_t *s)
{
    size_t cap = (size_t) 1 << 26, len = 0;
    struct stat sb;
    if (stat(path, &sb) == 0 && sb.st_size > 0) cap += (size_t) sb.st_size * 4;       /* HiFi FASTA compresses about fourfold */
    uint8_t *buf = (uint8_t *) malloc(cap);
    if (!buf) return OATK_E_NOMEM;
    gzFile fp = gzopen(path, "r");
    if (!fp) { free(buf); return OATK_E_ARG; }
    (void) gzbuffer(fp, 1 << 20);
    for (;;) {
        if (cap - len < ((size_t) 1 << 24)) {
            cap += cap / 2;
            uint8_t *nb = (uint8_t *) realloc(buf, cap);
            if (!nb) { gzclose(fp); free(buf); return OATK_E_NOMEM; }
            buf = nb;
        }
        const size_t want = cap - len > ((size_t) 1 << 30)? (size_t) 1 << 30 : cap - len;
        const int got = gzread(fp, buf + len, (unsigned) want);
        if (got < 0) { gzclose(fp); free(buf); return OATK_E_ARG; }
        if (got == 0) break;
        len += (size_t) got;
    }
    gzclose(fp);
    s->fd = -1, s->mem = buf, s->size = len;
    return OATK_OK;
}

static seg_t *open_segments(char **files, int n_files, uint64_t *total, int *rc, int stream_gz, int *any_gz)
{
    seg_t *s = (seg_t *) calloc((size_t) n_files, sizeof(seg_t));
    uint64_t base = 0;
    int i;
    *rc = OATK_OK;
    for (i = 0; i < n_files; ++i) s[i].fd = -1;
    for (i = 0; i < n_files; ++i) {
        unsigned char mg[2] = {0, 0};
        struct stat sb;
        const int fd = open(files[i], O_RDONLY);
        if (fd < 0 || fstat(fd, &sb) != 0) {
            fprintf(stderr, "[E::%s] fail to open file \"%s\"\n", __func__, files[i]);       /* sstream.c:46-49 */
            if (fd >= 0) close(fd);
            *rc = OATK_E_ARG;
            break;
        }
        const ssize_t nm = pread(fd, mg, 2, 0);
        s[i].fsize = S_ISREG(sb.st_mode)? (uint64_t) sb.st_size : 0;
        if ((nm == 2 && mg[0] == 0x1f && mg[1] == 0x8b) || !S_ISREG(sb.st_mode)) {
            /* gzip magic: inflate; anything else is read as it lies (gzread would do the same).  A pipe or device: no pread; zlib's transparent mode streams it */
            close(fd);
            if (any_gz) *any_gz = 1;
            if (stream_gz) {
                s[i].gz = oatk_gzsrc_open(files[i], oatk_host_threads_granted(), rc);
                if (!s[i].gz) { fprintf(stderr, "[E::%s] fail to open file \"%s\"\n", __func__, files[i]); break; }
                continue;                                          /* size, last byte and base are known when the stream has been read */
            }
            if ((*rc = inflate_file(files[i], &s[i])) != OATK_OK) break;
        } else {
            s[i].fd = fd, s[i].size = (uint64_t) sb.st_size;
        }
        uint8_t last = '\n';
        if (s[i].size) {
            if (s[i].fd >= 0) { if (pread(s[i].fd, &last, 1, (off_t) s[i].size - 1) != 1) last = '\n'; }
            else last = s[i].mem[s[i].size - 1];
        }
        s[i].add_nl = s[i].size && last != '\n';
        s[i].base = base;
        base += s[i].size + (uint64_t) s[i].add_nl;
    }
    if (*rc) { seg_close(s, n_files); return 0; }
    *total = base;
    return s;
}

/* ---- upload: text range [g0, g1) into a page-locked buffer by the host threads, then one DMA ---- */
typedef struct { const seg_t *seg; int n_seg; uint8_t *dst; uint64_t g0, g1; int failed; } up_job_t;

static void up_worker(void *arg, int tid, int n_threads)
{
    up_job_t *j = (up_job_t *) arg;
    const uint64_t n = j->g1 - j->g0;
    uint64_t a = j->g0 + n * (uint64_t) tid / (uint64_t) n_threads, b = j->g0 + n * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    int i;
    for (i = 0; i < j->n_seg && a < b; ++i) {
        const seg_t *s = &j->seg[i];
        const uint64_t end = s->base + s->size + (uint64_t) s->add_nl;
        if (a >= end) continue;
        uint64_t lo = a, hi = b < end? b : end;                    /* the part of [a, b) inside this segment */
        uint8_t *d = j->dst + (lo - j->g0);
        if (hi > s->base + s->size) { d[s->base + s->size - lo] = '\n'; hi = s->base + s->size; }       /* the added newline */
        if (hi > lo) {
            if (s->fd >= 0) {
                uint64_t done = 0;
                while (done < hi - lo) {
                    const ssize_t got = pread(s->fd, d + done, (size_t) (hi - lo - done), (off_t) (lo - s->base + done));
                    if (got <= 0) { j->failed = 1; return; }
                    done += (uint64_t) got;
                }
            } else memcpy(d, s->mem + (lo - s->base), (size_t) (hi - lo));
        }
        a = b < end? b : end;
    }
}

static int upload_text(oatk_hip_ctx *ctx, const seg_t *seg, int n_seg, uint64_t total, uint8_t **d_text_out)
{
    uint8_t *d_text = 0;
    struct timespec t_in;
    clock_gettime(CLOCK_MONOTONIC, &t_in);
    int rc = oatk_hip_ingest_text_buffer(ctx, total, &d_text);
    if (rc) return rc;
    *d_text_out = d_text;
    if (total == 0) return OATK_OK;
    const uint64_t chunk = total < UP_CHUNK? ((total + 63) & ~63ULL) : UP_CHUNK;
    /* a large input asks at once for what filling the reads' structs will want afterwards (srdb.c), so the block is pinned only once */
    struct timespec ta, tb;
    ta = t_in;
    uint8_t *stage = (uint8_t *) oatk_hip_staging(ctx, total > ((uint64_t) 1 << 30)? (uint64_t) 640 << 20 : 2 * chunk);
    if (!stage) return OATK_E_NOMEM;
    clock_gettime(CLOCK_MONOTONIC, &tb);
    {
        const char *lg = getenv("OATK_DROPIN_LOG");
        if (lg && lg[0] && lg[0] != '0') fprintf(stderr, "[M::oatk_%s] device text buffer + page-locked staging: %.3f s\n", __func__,
                                                 (double) (tb.tv_sec - ta.tv_sec) + 1e-9 * (double) (tb.tv_nsec - ta.tv_nsec));
    }
    up_job_t job = {seg, n_seg, stage, 0, total < chunk? total : chunk, 0};
    oatk_par_run(up_worker, &job);
    uint64_t g0 = 0;
    int which = 0;
    while (g0 < total && !job.failed) {
        const uint64_t g1 = g0 + chunk < total? g0 + chunk : total;
        rc = oatk_hip_h2d_async(ctx, d_text + g0, stage + (uint64_t) which * chunk, g1 - g0);
        if (rc) return rc;
        if (g1 < total) {                                          /* read the next piece while this one is on the bus */
            job.dst = stage + (uint64_t) (which ^ 1) * chunk, job.g0 = g1, job.g1 = g1 + chunk < total? g1 + chunk : total;
            oatk_par_run(up_worker, &job);
        }
        rc = oatk_hip_sync(ctx);
        if (rc) return rc;
        g0 = g1, which ^= 1;
    }
    return job.failed? OATK_E_ARG : OATK_OK;
}

static int ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads, seg_t **seg_out)
{
    uint64_t total = 0, used = 0;
    int rc = OATK_OK;
    seg_t *seg = open_segments(files, n_files, &total, &rc, 0, 0);
    if (!seg) return rc;
    uint8_t *d_text = 0;
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    rc = upload_text(ctx, seg, n_files, total, &d_text);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!rc) rc = oatk_hip_ingest(ctx, d_text, total, OATK_FMT_AUTO, 1, n_reads, &used);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    {
        const char *lg = getenv("OATK_DROPIN_LOG");
        if (lg && lg[0] && lg[0] != '0') fprintf(stderr, "[M::oatk_%s] %.2f GB of text: upload %.3f s, record scan %.3f s\n", __func__, (double) total / 1e9,
                                                 (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec),
                                                 (double) (t2.tv_sec - t1.tv_sec) + 1e-9 * (double) (t2.tv_nsec - t1.tv_nsec));
    }
    if (rc || !seg_out) seg_close(seg, n_files);
    else *seg_out = seg;
    return rc;
}

int oatk_ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads)
{
    return ingest_files(ctx, files, n_files, n_reads, 0);
}

/* ---- read names (kseq's name: the header up to the first white space), cut out of the text on the host threads ---- */
typedef struct { seg_t *seg; int n_seg; const uint64_t *hdr; uint64_t hdr_base, n; char **names; const void *owner; } name_job_t;

/* (Until round 4 the files were mapped and the header lines read through the mapping: a minor page fault per read, 1.2 s of the 2 M-read CLI run.  A
 * 128-byte pread from the page cache costs a quarter of that and needs no mapping.) */
static void name_worker(void *arg, int tid, int n_threads)
{
    name_job_t *j = (name_job_t *) arg;
    const uint64_t a = j->n * (uint64_t) tid / (uint64_t) n_threads, b = j->n * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    oatk_name_bump_t bump = {0, 0};
    uint8_t small[128], *big = 0;
    size_t big_cap = 0;
    uint64_t i;
    int si = 0;
    for (i = a; i < b; ++i) {
        const uint64_t g = j->hdr_base + j->hdr[i] + 1;            /* behind '>' / '@'; headers ascend, so the segment index only moves forward */
        while (si + 1 < j->n_seg && g >= j->seg[si + 1].base) ++si;
        const seg_t *s = &j->seg[si];
        const uint64_t p = g - s->base;
        const uint8_t *t;
        uint64_t avail, e = 0;
        if (s->fd >= 0) {                                           /* the line's first bytes from the page cache; a longer name is read again, whole */
            size_t want = sizeof(small);
            uint8_t *buf = small;
            for (;;) {
                const uint64_t left = s->size - p;
                const size_t n = left < want? (size_t) left : want;
                size_t done = 0;
                while (done < n) { const ssize_t got = pread(s->fd, buf + done, n - done, (off_t) (p + done)); if (got <= 0) break; done += (size_t) got; }
                for (e = 0; e < done && buf[e] != ' ' && buf[e] != '\t' && buf[e] != '\n' && buf[e] != '\r'; ++e) {}
                if (e < done || done < want || done == left) break;
                want *= 8;
                if (want > big_cap) { free(big); big = (uint8_t *) malloc(want); big_cap = big? want : 0; if (!big) { e = 0; break; } }
                buf = big;
            }
            t = buf;
        } else {
            t = s->mem + p, avail = s->size - p;
            while (e < avail && t[e] != ' ' && t[e] != '\t' && t[e] != '\n' && t[e] != '\r') ++e;
        }
        j->names[i] = oatk_host_name_dup(t, (size_t) e, &bump, j->owner);
    }
    free(big);
}

/* ---- sr_read (syncmer.c:487) for files, streamed ----
 * The text moves through the device in windows.  An uploader thread keeps reading the next window into page-locked memory and sending it
 * (its own handle: stream + staging) while this thread takes the window before it through the record scan and the syncmer scan on a piece
 * handle, fills the reads' structs from the piece (D2H behind the copying threads, srdb.c) and moves the piece behind the batch that is
 * being assembled in the caller's handle (oatk_hip_scan_append).  A record cut by a window's end is carried over: the unconsumed tail is
 * copied, on the device, in front of the next window.  Two slots alternate; nothing larger than a window (plus the assembled results) is
 * ever allocated, and what the reference does one after the other -- read, analyse, store -- runs side by side. */
#define WIN_DEFAULT ((uint64_t) 768 << 20)
#define CARRY_CAP ((uint64_t) 32 << 20)

static uint64_t g_window = 0;
void oatk_host_debug_window(uint64_t bytes) { g_window = bytes; }

/* what the stream keeps on one DEVICE: with the reads spread over several handles (include/oatk_multi.h) a window is uploaded to, parsed and scanned on
 * the device of the handle its reads will be assembled in */
typedef struct {
    int dev;
    oatk_hip_ctx *up;              /* the uploader's handle */
    oatk_hip_ctx *piece[2];        /* record scan + syncmer scan of one window */
    uint8_t *d_win[2];             /* where a slot's window goes on the device */
    uint8_t *stage;                /* the uploader's page-locked pieces */
    uint8_t *h_win[2];             /* streamed input: a slot's window on the host (page-locked; the names are cut out of it), CARRY_CAP bytes of room in front */
} stream_dev_t;

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int state[2];                  /* 0 free, 1 window uploaded */
    uint64_t g0[2], g1[2];         /* text range of the window in the slot */
    int failed, stop;
    stream_dev_t *res;             /* one per device in use */
    int n_res;
    const int *res_of_rank;        /* [n_rank] */
    int n_rank;
    const seg_t *seg;
    int n_seg, n_up;
    uint64_t total, win;
    /* where a window lies in the FILES' bytes (their sizes summed: known in advance, unlike the length of a gzip'ed file's text): decides the handle its
     * reads go to and how much room to reserve.  For plain files it is the text's own offset. */
    uint64_t f0[2], f1[2], ftotal;
    int streamed;                  /* the input is pulled through sources (a gzip'ed file among them), window by window; the last window says so */
    int final[2];
} stream_t;

/* the handle a window's reads go to: by where the window starts in the input, so the ranks hold contiguous ranges of reads of about equal text */
static int rank_of_window(const stream_t *st, uint64_t f0)
{
    if (st->n_rank <= 1 || st->ftotal == 0) return 0;
    const uint64_t r = (uint64_t) (((unsigned __int128) f0 * (uint64_t) st->n_rank) / st->ftotal);
    return r >= (uint64_t) st->n_rank? st->n_rank - 1 : (int) r;
}

static int map_upload_wanted(void)
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("OATK_HOST_MAP_UPLOAD"); v = !(e && e[0] == '0'); }
    return v;
}

typedef struct { const uint8_t *p; uint64_t n; } touch_job_t;
static void touch_worker(void *arg, int tid, int n_threads)
{
    const touch_job_t *j = (const touch_job_t *) arg;
    const uint64_t pages = (j->n + 4095) >> 12, a = pages * (uint64_t) tid / (uint64_t) n_threads, b = pages * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    uint64_t i;
    unsigned acc = 0;
    for (i = a; i < b; ++i) acc += *(volatile const uint8_t *) (j->p + (i << 12));
    (void) acc;
}

/* text [g0, g1) of ONE plain file from its mapping to the device; anything else than OATK_OK: the caller stages the window instead (nothing has been sent) */

/* the upload straight from the file's page-locked mapping is given up for this file (its windows are read and staged instead): said once, so that a run whose reader got
 * slower can be told from one that did not.  Such a mapping also means a file truncated WHILE it is read ends the process with SIGBUS where pread would have
 * returned short -- OATK_HOST_MAP_UPLOAD=0 is the opt-out */
static void map_abandoned(const char *why)
{
    static int said;
    if (!__atomic_exchange_n(&said, 1, __ATOMIC_RELAXED)) fprintf(stderr, "[oatk host] upload from the mapped file abandoned (%s): windows are read and staged from here on\n", why);
}

static int upload_mapped(stream_t *st, stream_dev_t *D, int slot, uint64_t g0, uint64_t g1)
{
    int i;
    seg_t *sg = 0;
    for (i = 0; i < st->n_seg; ++i) {
        seg_t *c = (seg_t *) &st->seg[i];
        if (g0 >= c->base && g1 <= c->base + c->size && c->fd >= 0) { sg = c; break; }
    }
    if (!sg || sg->no_map) return OATK_E_ARG;
    if (!sg->map) {
        sg->map = (uint8_t *) mmap(0, (size_t) sg->size, PROT_READ, MAP_SHARED, sg->fd, 0);
        if (sg->map == MAP_FAILED) { sg->map = 0, sg->no_map = 1; map_abandoned("mmap of the file failed"); return OATK_E_ARG; }
    }
    const uint64_t a = (g0 - sg->base) & ~(uint64_t) 4095, b = g1 - sg->base;           /* the mapping covers whole pages */
    const uint64_t blen = ((b - a) + 4095) & ~(uint64_t) 4095;
    touch_job_t tj = {sg->map + a, b - a};
    oatk_par_run_n(touch_worker, &tj, st->n_up);
    if (oatk_hip_host_register(D->up, sg->map + a, blen) != OATK_OK) { sg->no_map = 1; map_abandoned("hipHostRegister refused the read-only mapping"); return OATK_E_ARG; }
    int rc = oatk_hip_h2d_async(D->up, D->d_win[slot], sg->map + (g0 - sg->base), g1 - g0);
    if (!rc) rc = oatk_hip_sync(D->up);
    (void) oatk_hip_host_unregister(D->up, sg->map + a);
    if (rc) sg->no_map = 1, map_abandoned("the copy from the mapping failed");
    return rc;
}

static void *uploader(void *arg)
{
    stream_t *st = (stream_t *) arg;
    const uint64_t chunk = st->win < UP_CHUNK? ((st->win + 63) & ~63ULL) : UP_CHUNK;
    uint64_t g0, w;
    int rc = OATK_OK;
    for (w = 0, g0 = 0; !rc && g0 < st->total; ++w) {
        const int s = (int) (w & 1);
        const uint64_t g1 = g0 + st->win < st->total? g0 + st->win : st->total;
        stream_dev_t *D = &st->res[st->res_of_rank[rank_of_window(st, g0)]];
        pthread_mutex_lock(&st->mu);
        while (st->state[s] != 0 && !st->stop) pthread_cond_wait(&st->cv, &st->mu);
        const int stop = st->stop;
        pthread_mutex_unlock(&st->mu);
        if (stop) break;
        if (st->n_seg == 1 && st->seg[0].pinned) {                   /* page-locked text: one copy, no staging */
            rc = oatk_hip_h2d_async(D->up, D->d_win[s], st->seg[0].mem + g0, g1 - g0);
            if (!rc) rc = oatk_hip_sync(D->up);
            pthread_mutex_lock(&st->mu);
            if (rc) st->failed = rc;
            else st->state[s] = 1, st->g0[s] = st->f0[s] = g0, st->g1[s] = st->f1[s] = g1, st->final[s] = g1 == st->total;
            pthread_cond_broadcast(&st->cv);
            pthread_mutex_unlock(&st->mu);
            g0 = g1;
            continue;
        }
        if (map_upload_wanted()) {
            /* The window as it lies in the PAGE CACHE (round 4): the file is mapped, the window's pages are touched by the host threads (minor faults: the
             * page-cache pages enter this process's page table), the range is page-locked and goes over PCIe from where it lies -- no copy on the host.
             * Registering a populated file mapping runs at hundreds of GB/s on the bench box and the upload from it at the bus's 57 GB/s, where the
             * pread into page-locked staging below managed ~15 GB/s (tools/ubench/file_pin.hip, profiles/r04a_file_pin.txt).  A window that spans two
             * files, or a file system whose pages cannot be locked, takes the staging road. */
            const int rm = upload_mapped(st, D, s, g0, g1);
            if (rm == OATK_OK) {
                pthread_mutex_lock(&st->mu);
                st->state[s] = 1, st->g0[s] = st->f0[s] = g0, st->g1[s] = st->f1[s] = g1, st->final[s] = g1 == st->total;
                pthread_cond_broadcast(&st->cv);
                pthread_mutex_unlock(&st->mu);
                g0 = g1;
                continue;
            }
        }
        /* the window in pieces: read piece p + 1 while piece p is on the bus */
        if (!D->stage) D->stage = (uint8_t *) oatk_hip_staging(D->up, 2 * chunk);
        uint8_t *stage = D->stage;
        if (!stage) { rc = OATK_E_NOMEM; break; }
        up_job_t job = {st->seg, st->n_seg, stage, g0, g0 + chunk < g1? g0 + chunk : g1, 0};
        oatk_par_run_n(up_worker, &job, st->n_up);
        uint64_t p0 = g0;
        int which = 0;
        while (!rc && p0 < g1 && !job.failed) {
            const uint64_t p1 = p0 + chunk < g1? p0 + chunk : g1;
            rc = oatk_hip_h2d_async(D->up, D->d_win[s] + (p0 - g0), stage + (uint64_t) which * chunk, p1 - p0);
            if (!rc && p1 < g1) {
                job.dst = stage + (uint64_t) (which ^ 1) * chunk, job.g0 = p1, job.g1 = p1 + chunk < g1? p1 + chunk : g1;
                oatk_par_run_n(up_worker, &job, st->n_up);
            }
            if (!rc) rc = oatk_hip_sync(D->up);
            p0 = p1, which ^= 1;
        }
        if (job.failed) rc = OATK_E_ARG;
        pthread_mutex_lock(&st->mu);
        if (rc) st->failed = rc;
        else st->state[s] = 1, st->g0[s] = st->f0[s] = g0, st->g1[s] = st->f1[s] = g1, st->final[s] = g1 == st->total;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
        g0 = g1;
    }
    if (rc) {
        pthread_mutex_lock(&st->mu);
        st->failed = rc;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
    }
    return 0;
}


/* ---- streamed input: the files pulled through sources, one after the other ---- */
typedef struct {
    const seg_t *seg; int n_seg, cur;
    uint64_t in_seg;               /* plain file: bytes delivered of the current one */
    uint64_t out_seg;              /* text delivered of the current file */
    uint8_t last;
    uint64_t f_before;             /* bytes of the files before the current one */
} src_t;

static uint64_t src_fpos(const src_t *q)
{
    if (q->cur >= q->n_seg) return q->f_before;
    const seg_t *s = &q->seg[q->cur];
    return q->f_before + (s->gz? oatk_gzsrc_tell_in(s->gz) : q->in_seg);
}

/* the next bytes of the input (at most cap, at least one unless the input is spent: 0), < 0 on a damaged file */
static int64_t src_read(src_t *q, uint8_t *dst, uint64_t cap)
{
    while (q->cur < q->n_seg) {
        const seg_t *s = &q->seg[q->cur];
        int64_t n = 0;
        if (s->gz) {
            n = oatk_gzsrc_read(s->gz, dst, cap);
            if (n < 0) { fprintf(stderr, "[E::%s] input file %d is damaged (gzip stream)\n", __func__, q->cur + 1); return -1; }
        } else if (s->fd >= 0) {
            const uint64_t left = s->size - q->in_seg, want = left < cap? left : cap;
            uint64_t done = 0;
            while (done < want) {
                const ssize_t got = pread(s->fd, dst + done, (size_t) (want - done), (off_t) (q->in_seg + done));
                if (got <= 0) return -1;
                done += (uint64_t) got;
            }
            n = (int64_t) want, q->in_seg += want;
        } else {
            const uint64_t left = s->size - q->in_seg, want = left < cap? left : cap;
            memcpy(dst, s->mem + q->in_seg, (size_t) want);
            n = (int64_t) want, q->in_seg += want;
        }
        if (n > 0) { q->last = dst[n - 1], q->out_seg += (uint64_t) n; return n; }
        /* the file is spent: a fresh kseq starts every file at a header line (sstream.c:91-97), so a last line without a newline gets one */
        const int add = q->out_seg && q->last != '\n';
        q->f_before += s->fsize, q->in_seg = q->out_seg = 0, ++q->cur;
        if (add) { dst[0] = '\n'; q->last = '\n'; return 1; }
    }
    return 0;
}

static void *uploader_src(void *arg)
{
    stream_t *st = (stream_t *) arg;
    const uint64_t chunk = st->win < ((uint64_t) 32 << 20)? st->win : (uint64_t) 32 << 20;
    src_t q;
    uint64_t g0 = 0, w;
    int rc = OATK_OK, final = 0;
    memset(&q, 0, sizeof(q));
    q.seg = st->seg, q.n_seg = st->n_seg, q.last = '\n';
    for (w = 0; !rc && !final; ++w) {
        const int s = (int) (w & 1);
        const uint64_t f0 = src_fpos(&q);
        stream_dev_t *D = &st->res[st->res_of_rank[rank_of_window(st, f0)]];
        if (!D->stage) {
            D->stage = (uint8_t *) oatk_hip_staging(D->up, 2 * (CARRY_CAP + st->win));
            if (!D->stage) { rc = OATK_E_NOMEM; break; }
            D->h_win[0] = D->stage + CARRY_CAP, D->h_win[1] = D->stage + CARRY_CAP + st->win + CARRY_CAP;
        }
        pthread_mutex_lock(&st->mu);
        while (st->state[s] != 0 && !st->stop) pthread_cond_wait(&st->cv, &st->mu);
        const int stop = st->stop;
        pthread_mutex_unlock(&st->mu);
        if (stop) break;
        uint64_t filled = 0;
        while (!rc && filled < st->win) {                            /* a piece is on the bus while the next one is inflated / read */
            const uint64_t want = st->win - filled < chunk? st->win - filled : chunk;
            const int64_t n = src_read(&q, D->h_win[s] + filled, want);
            if (n < 0) { rc = OATK_E_ARG; break; }
            if (n == 0) { final = 1; break; }
            rc = oatk_hip_h2d_async(D->up, D->d_win[s] + filled, D->h_win[s] + filled, (uint64_t) n);
            filled += (uint64_t) n;
        }
        if (!rc) rc = oatk_hip_sync(D->up);
        pthread_mutex_lock(&st->mu);
        if (rc) st->failed = rc;
        else st->state[s] = 1, st->g0[s] = g0, st->g1[s] = g0 + filled, st->f0[s] = f0, st->f1[s] = src_fpos(&q), st->final[s] = final;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
        g0 += filled;
    }
    if (rc) {
        pthread_mutex_lock(&st->mu);
        st->failed = rc;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
    }
    return 0;
}

/* read names out of a window that lies in host memory (streamed input) */
typedef struct { const uint8_t *text; uint64_t len; const uint64_t *hdr; uint64_t n; char **names; const void *owner; } hname_job_t;

static void hname_worker(void *arg, int tid, int n_threads)
{
    hname_job_t *j = (hname_job_t *) arg;
    const uint64_t a = j->n * (uint64_t) tid / (uint64_t) n_threads, b = j->n * (uint64_t) (tid + 1) / (uint64_t) n_threads;
    oatk_name_bump_t bump = {0, 0};
    uint64_t i;
    for (i = a; i < b; ++i) {
        const uint64_t p = j->hdr[i] + 1;                           /* behind '>' / '@' */
        uint64_t e = p;
        while (e < j->len && j->text[e] != ' ' && j->text[e] != '\t' && j->text[e] != '\n' && j->text[e] != '\r') ++e;
        j->names[i] = oatk_host_name_dup(j->text + p, (size_t) (e - p), &bump, j->owner);
    }
}

/* the first character of the text that is not white space decides the format, as kseq and oatk_hip_ingest(AUTO) decide it */
static int sniff_format(const seg_t *seg, int n_seg, uint64_t total)
{
    uint8_t head[4096];
    uint64_t n = total < sizeof(head)? total : sizeof(head), i;
    up_job_t job = {seg, n_seg, head, 0, n, 0};
    up_worker(&job, 0, 1);
    for (i = 0; i < n && (head[i] == '\n' || head[i] == '\r' || head[i] == ' ' || head[i] == '\t'); ++i) {}
    return i < n && head[i] == '@'? OATK_FMT_FASTQ : OATK_FMT_FASTA;
}

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

/* ctxs[0 .. n_ctx): the handles the reads are assembled in -- one, or one per GPU: then handle r receives the reads of the r-th part of the input
 * (rank_of_window) and first[0 .. n_ctx] their ranges; sid0 of handle r = first[r] */
/* m_data: sr_read's data cap (syncmer.c:537-541; 0 = none): the read that takes the total of raw bases to the cap is the last one taken */
static int sr_read_stream(oatk_hip_ctx **ctxs, int n_ctx, uint64_t *first, oatk_sr_db_t *sr_db, int K, int S, seg_t *seg, int n_files, uint64_t total, uint64_t win,
                          uint64_t m_data, int streamed)
{
    const char *lg = getenv("OATK_DROPIN_LOG");
    const int log = lg && lg[0] && lg[0] != '0';
    const int threads = oatk_host_threads();
    const double t_begin = now_s();
    double t_wait = 0, t_dev = 0, t_fill = 0, t_app = 0, t_room = 0, t_names = 0;
    stream_t st;
    stream_dev_t res[64];
    int res_of_rank[64];
    pthread_t th;
    int rc = OATK_OK, started = 0, i, r, cur = 0, prev_res = -1, prev_slot = 0;
    uint64_t n_done = 0, carry = 0, w, text_done = 0;
    uint64_t *off = 0, *hdr = 0;
    uint8_t *hop = 0;                                               /* a carried record tail on its way from one device to the next */
    uint8_t *h_carry = 0;                                           /* streamed input: the same tail on the host (the names are cut out of the host's copy of the text) */
    char **names = 0, **early_names = 0;
    uint64_t early_n = 0, names_n = 0;                /* entries of the two arrays that hold names not yet handed to the reads (freed one by one on the way out when they are blocks of their own: ADVICE r04) */
    uint64_t ftotal = 0;
    int all_done = 0;
    memset(&st, 0, sizeof(st));
    memset(res, 0, sizeof(res));
    if (n_ctx < 1 || n_ctx > 64) return OATK_E_ARG;
    pthread_mutex_init(&st.mu, 0);
    pthread_cond_init(&st.cv, 0);
    st.seg = seg, st.n_seg = n_files, st.total = total, st.win = win, st.res = res, st.res_of_rank = res_of_rank, st.n_rank = n_ctx;
    st.streamed = streamed;
    if (streamed) { for (i = 0; i < n_files; ++i) ftotal += seg[i].fsize; } else ftotal = total;
    st.ftotal = ftotal;
    st.n_up = threads > 1? threads : 1;            /* readers of the file beside the threads that fill the structs: with the reads in arenas the file is what sr_read waits for */
    { const char *e = getenv("OATK_HOST_UP_THREADS"); if (e && atoi(e) > 0) st.n_up = atoi(e); }
    int fmt = streamed? -1 : sniff_format(seg, n_files, total);     /* (streamed: from the first window, below) */
    uint64_t n_bases = 0;                                           /* raw bases taken so far */
    int capped = 0;
    for (r = 0; r < n_ctx; ++r) {                                   /* one set of working handles per device */
        const int dev = oatk_hip_device(ctxs[r]);
        for (i = 0; i < st.n_res && res[i].dev != dev; ++i) {}
        if (i == st.n_res) res[st.n_res++].dev = dev;
        res_of_rank[r] = i;
    }
    rc = oatk_hip_scan_begin(ctxs[0], 0, K, S);
    if (first) first[0] = 0;
    for (r = 0; !rc && r < st.n_res; ++r) {
        res[r].up = oatk_hip_create(res[r].dev);
        for (i = 0; !rc && i < 2; ++i) {
            uint8_t *d = 0;
            res[r].piece[i] = oatk_hip_create(res[r].dev);
            if (!res[r].piece[i] || !res[r].up) { rc = OATK_E_NODEV; break; }
            if (i == 1 && !streamed && total <= win) break;           /* one window: one slot */
            rc = oatk_hip_ingest_text_buffer(res[r].piece[i], CARRY_CAP + (streamed || win < total? win : total) + 64, &d);
            res[r].d_win[i] = d + CARRY_CAP;
        }
    }
    if (rc) goto done;
    {
        int fill_threads = threads > 1? threads / 2 : 1;            /* the other half reads the file */
        const char *e = getenv("OATK_HOST_FILL_THREADS");           /* (experiments: the split between the two halves) */
        if (e && atoi(e) > 0) fill_threads = atoi(e);
        oatk_host_set_threads_internal(fill_threads);
    }
    if (streamed && !(h_carry = (uint8_t *) malloc(CARRY_CAP))) { rc = OATK_E_NOMEM; goto done; }
    if (pthread_create(&th, 0, streamed? uploader_src : uploader, &st) != 0) { rc = OATK_E_NOMEM; goto done; }
    started = 1;

    for (w = 0; !all_done; ++w) {
        const int s = (int) (w & 1);
        double t0 = now_s();
        pthread_mutex_lock(&st.mu);
        while (st.state[s] != 1 && !st.failed) pthread_cond_wait(&st.cv, &st.mu);
        rc = st.failed;
        const uint64_t g0 = st.g0[s], g1 = st.g1[s], wf0 = st.f0[s], wf1 = st.f1[s];
        const int final = st.final[s];
        pthread_mutex_unlock(&st.mu);
        if (rc) break;
        t_wait += now_s() - t0, t0 = now_s();
        const int rank = rank_of_window(&st, wf0), ri = res_of_rank[rank];
        stream_dev_t *D = &res[ri];
        while (cur < rank) {                                         /* the reads from here on belong to the next handle(s) */
            ++cur;
            rc = oatk_hip_scan_begin(ctxs[cur], n_done, K, S);
            if (rc) break;
            if (first) first[cur] = n_done;
        }
        if (rc) break;
        if (carry && prev_res >= 0 && prev_res != ri) {
            /* the record cut by the previous window's end was parked on another device: bring it over through the host (once per change of device) */
            if (!hop) hop = (uint8_t *) malloc(CARRY_CAP);
            if (!hop) { rc = OATK_E_NOMEM; break; }
            rc = oatk_hip_d2h(res[prev_res].piece[prev_slot], hop, res[prev_res].d_win[s] - carry, carry);
            if (!rc) rc = oatk_hip_h2d_async(D->piece[s], D->d_win[s] - carry, hop, carry);
            if (!rc) rc = oatk_hip_sync(D->piece[s]);
            if (rc) break;
        }
        uint8_t *d_text = D->d_win[s] - carry;
        const uint64_t len = carry + (g1 - g0);
        uint64_t n = 0, used = 0, b = 0;
        const uint8_t *h_text = 0;
        if (streamed) {                                              /* the host's copy of the same text: the carried tail in front of the window */
            if (carry) memcpy(D->h_win[s] - carry, h_carry, carry);
            h_text = D->h_win[s] - carry;
            if (fmt < 0) {
                uint64_t i2;
                for (i2 = 0; i2 < len && (h_text[i2] == '\n' || h_text[i2] == '\r' || h_text[i2] == ' ' || h_text[i2] == '\t'); ++i2) {}
                if (i2 < len || final) fmt = i2 < len && h_text[i2] == '@'? OATK_FMT_FASTQ : OATK_FMT_FASTA;
            }
            if (len == 0 || fmt < 0) {                               /* an empty last window, or nothing but white space so far (dropped, as kseq skips it) */
                pthread_mutex_lock(&st.mu);
                st.state[s] = 0;
                pthread_cond_broadcast(&st.cv);
                pthread_mutex_unlock(&st.mu);
                carry = 0, text_done = g1, prev_res = ri, prev_slot = s, all_done = final;
                continue;
            }
        }
        rc = oatk_hip_ingest(D->piece[s], d_text, len, fmt, final, &n, &used);
        if (rc == OATK_E_SPLIT && fmt != OATK_FMT_KSEQ) {
            /* not what the two device-only formats take (wrapped FASTQ; FASTQ records among FASTA ones): kseq's own reading from here on, line by line */
            fmt = OATK_FMT_KSEQ;
            rc = oatk_hip_ingest(D->piece[s], d_text, len, fmt, final, &n, &used);
        }
        if (rc) break;
        if (m_data && n) {                                           /* the cap: count the raw bases of this piece's reads */
            uint32_t *ln = (uint32_t *) malloc(4 * n);
            const void *dl = 0;
            uint64_t bb = 0, i2;
            if (!ln) { rc = OATK_E_NOMEM; break; }
            rc = oatk_hip_buffer(D->piece[s], OATK_BUF_INGEST_LEN, &dl, &bb);
            if (!rc) rc = oatk_hip_d2h(D->piece[s], ln, dl, 4 * n);
            for (i2 = 0; !rc && i2 < n; ++i2) {
                n_bases += ln[i2];
                if (n_bases >= m_data) { capped = 1; break; }
            }
            free(ln);
            if (rc) break;
            if (capped) { n = i2 + 1; rc = oatk_hip_ingest_truncate(D->piece[s], n); if (rc) break; }
        }
        const uint64_t next_carry = len - used;
        if (!final && !capped) {
            if (next_carry > CARRY_CAP || used == 0) { rc = OATK_E_NOMEM, g_record_too_long = 1; break; }       /* a record longer than a window: the caller retries in one piece */
            rc = oatk_hip_d2d(D->piece[s], D->d_win[s ^ 1] - next_carry, d_text + used, next_carry);      /* (moved on above if the next window lands on another device) */
            if (rc) break;
        }
        if (streamed) {                                             /* before the slot goes back: the names and the carried tail, from the host's copy */
            if (sr_db && n) {
                const void *dh = 0;
                uint64_t bb = 0;
                hdr = (uint64_t *) malloc(8 * n), early_names = (char **) calloc(n, sizeof(char *));
                if (!hdr || !early_names) { rc = OATK_E_NOMEM; break; }
                rc = oatk_hip_buffer(D->piece[s], OATK_BUF_INGEST_HDR, &dh, &bb);
                if (!rc) rc = oatk_hip_d2h(D->piece[s], hdr, dh, 8 * n);
                if (rc) break;
                hname_job_t hj = {h_text, len, hdr, n, early_names, sr_db};
                early_n = n;
                oatk_par_run(hname_worker, &hj);
                free(hdr), hdr = 0;
            }
            if (!final && !capped && next_carry) memcpy(h_carry, h_text + used, next_carry);
        }
        pthread_mutex_lock(&st.mu);                                 /* the window's text is spent: the uploader may have the slot back */
        st.state[s] = 0;
        pthread_cond_broadcast(&st.cv);
        pthread_mutex_unlock(&st.mu);
        const uint64_t text0 = g0 - carry;                          /* where this piece's text starts in the whole text */
        carry = next_carry, text_done = g1, prev_res = ri, prev_slot = s, all_done = final || capped;
        if (n == 0) continue;
        oatk_hip_ctx *ctx = ctxs[cur];
        rc = oatk_hip_scan_ingested(D->piece[s], n_done, K, S);
        if (rc) break;
        t_dev += now_s() - t0, t0 = now_s();
        /* room for the reads: from the first piece's density, generously; grown when a later piece needs more */
        /* (how much of the input this window took, as the estimates below see it: a source that cannot say -- or says something no deflate stream can mean, less than a
         *  fortieth of the window's text -- is not extrapolated from: the arrays then grow window by window) */
        const int trust_in = oatk_est_trust(wf0, wf1, g1 - text0);
        if (sr_db && sr_db->m < n_done + n) {
            /* (an estimate is a guess: one that asks for more than a quarter of the machine's memory is not believed -- the array is zeroed below, i.e. touched) */
            const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
            const uint64_t phys = pages > 0 && psz > 0? (uint64_t) pages * (uint64_t) psz : 0;
            const uint64_t m = oatk_est_reads(n_done, n, sr_db->m, wf0, wf1, ftotal, trust_in && used, final || capped, phys / 4 / sizeof(oatk_sr_t));
            oatk_sr_t *na = (oatk_sr_t *) realloc(sr_db->a, sizeof(oatk_sr_t) * m);
            if (!na) { rc = OATK_E_NOMEM; break; }
            memset(na + sr_db->m, 0, sizeof(oatk_sr_t) * (m - sr_db->m));
            sr_db->a = na, sr_db->m = m;
        }
        {   /* ... and for the batch assembled in this handle, at its first piece */
            oatk_hip_info_t have;
            oatk_hip_info(ctx, &have);
            if (have.n_reads == 0 && !final && !capped && used && trust_in) {
                oatk_hip_info_t inf;
                oatk_hip_info(D->piece[s], &inf);
                const double scale = oatk_est_scale(wf0, wf1, ftotal, n_ctx);
                rc = oatk_hip_scan_reserve(ctx, (uint64_t) ((double) inf.seq_bytes * scale) + (1 << 20), (uint64_t) ((double) n * scale) + 1024,
                                           (uint64_t) ((double) inf.n_occ * scale * 1.1) + 4096);
                if (rc) break;
            }
        }
        t_room += now_s() - t0, t0 = now_s();
        if (!sr_db) {                                               /* the scan only: no structs to fill */
            rc = oatk_hip_scan_append(ctx, D->piece[s]);
            if (rc) break;
            t_app += now_s() - t0;
            n_done += n;
            continue;
        }
        const void *d = 0;
        off = (uint64_t *) malloc(8 * n);
        if (streamed) names = early_names, early_names = 0, names_n = early_n, early_n = 0;           /* (cut out of the host's window above) */
        else hdr = (uint64_t *) malloc(8 * n), names = (char **) calloc(n, sizeof(char *));
        if (!off || (!streamed && !hdr) || !names) { rc = OATK_E_NOMEM; break; }
        rc = oatk_hip_buffer(D->piece[s], OATK_BUF_INGEST_OFF, &d, &b);
        if (!rc) rc = oatk_hip_d2h(D->piece[s], off, d, 8 * n);
        if (!rc && !streamed) {
            rc = oatk_hip_buffer(D->piece[s], OATK_BUF_INGEST_HDR, &d, &b);
            if (!rc) rc = oatk_hip_d2h(D->piece[s], hdr, d, 8 * n);
        }
        if (rc) break;
        if (!streamed) {                                            /* (cutting the names on threads of their own beside the download was tried in r04: the download slows down by what the names took) */
            name_job_t nj = {seg, n_files, hdr, text0, n, names, sr_db};
            oatk_par_run(name_worker, &nj);
        }
        t_names += now_s() - t0, t0 = now_s();
        rc = oatk_sr_db_fill_range(D->piece[s], sr_db, n_done, off, n, names);
        free(off); free(hdr); free(names);
        off = hdr = 0, names = 0;
        if (rc) break;
        t_fill += now_s() - t0, t0 = now_s();
        rc = oatk_hip_scan_append(ctx, D->piece[s]);
        if (rc) break;
        t_app += now_s() - t0;
        n_done += n;
    }
    while (!rc && cur < n_ctx - 1) {                                /* handles the input did not reach hold no reads */
        ++cur;
        rc = oatk_hip_scan_begin(ctxs[cur], n_done, K, S);
        if (first) first[cur] = n_done;
    }
    if (first) first[n_ctx] = n_done;
    if (capped && !rc) fprintf(stderr, "[M::%s] data limit (%lu) reached. Discard the remaining sequences...\n", "sr_read", (unsigned long) m_data);       /* syncmer.c:539 */
done:
    if (started) {
        pthread_mutex_lock(&st.mu);
        st.stop = 1;
        pthread_cond_broadcast(&st.cv);
        pthread_mutex_unlock(&st.mu);
        pthread_join(th, 0);
    }
    oatk_host_set_threads_internal(threads);
    if (rc && !oatk_host_arena()) {                                 /* names that never reached a read: blocks of their own without arenas */
        uint64_t q;
        for (q = 0; names && q < names_n; ++q) free(names[q]);
        for (q = 0; early_names && q < early_n; ++q) free(early_names[q]);
    }
    free(off); free(hdr); free(names); free(hop); free(h_carry);
    free(early_names);
    for (r = 0; r < st.n_res; ++r) {
        for (i = 0; i < 2; ++i) if (res[r].piece[i]) oatk_hip_destroy(res[r].piece[i]);
        if (res[r].up) oatk_hip_destroy(res[r].up);
    }
    pthread_mutex_destroy(&st.mu);
    pthread_cond_destroy(&st.cv);
    if (!rc && sr_db && sr_db->m > sr_db->n) {                      /* give back what the estimate left over */
        oatk_sr_t *na = (oatk_sr_t *) realloc(sr_db->a, sizeof(oatk_sr_t) * (sr_db->n? sr_db->n : 1));
        if (na) sr_db->a = na, sr_db->m = sr_db->n;
    }
    if (log) fprintf(stderr, "[M::oatk_sr_read_files] %.2f GB of text in %lu windows, %lu reads into %d handle(s): %.3f s (waiting for the uploader %.3f, record + syncmer scan %.3f, "
                             "room for the batch %.3f, names %.3f, structs %.3f, append %.3f)\n", (double) text_done / 1e9, (unsigned long) w, (unsigned long) n_done, n_ctx, now_s() - t_begin, t_wait, t_dev, t_room, t_names, t_fill, t_app);
    return rc;
}

int oatk_sr_read_files(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, char **files, int n_files)
{
    return oatk_host_sr_read_files_n(&ctx, 1, sr_db, files, n_files, 0, 0);
}

int oatk_sr_read_files_capped(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t m_data)
{
    return oatk_host_sr_read_files_n(&ctx, 1, sr_db, files, n_files, 0, m_data);
}

int oatk_host_sr_read_files_n(oatk_hip_ctx **ctxs, int n_ctx, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t *first, uint64_t m_data)
{
    uint64_t total = 0;
    int i, rc = OATK_OK, any_gz = 0;
    seg_t *seg = open_segments(files, n_files, &total, &rc, 1, &any_gz);
    if (!seg) return rc;
    const char *ew = getenv("OATK_DEBUG_WINDOW");                  /* test hook, like oatk_host_debug_window */
    const uint64_t forced = g_window? g_window : (ew && atoll(ew) > 0? (uint64_t) atoll(ew) : 0);
    if (any_gz) {
        /* A gzip'ed file among the inputs: everything is pulled through sources, window by window (uploader_src).  The windows are smaller than for
         * plain files -- the device waits for the inflating host anyway, and two of them lie page-locked on the host. */
        uint64_t win = forced? forced : (uint64_t) 256 << 20;
        int attempt;
        if (win < 4096) win = 4096;
        for (attempt = 0;; ++attempt) {
            int rewindable = 1;
            for (i = 0; i < n_files; ++i) if (seg[i].gz && oatk_gzsrc_kind(seg[i].gz) == 3) rewindable = 0;       /* a pipe: what has been read is gone */
            g_record_too_long = 0;
            rc = sr_read_stream(ctxs, n_ctx, first, sr_db, sr_db->k, sr_db->s, seg, n_files, 0, win, m_data, 1);
            if (rc != OATK_E_NOMEM || !g_record_too_long || !rewindable || attempt == 2) break;
            /* a record longer than a window: the stream cannot be rewound, so the files are opened again and read with windows eight times the size */
            seg_close(seg, n_files);
            oatk_sr_db_clean(sr_db);
            win *= 8;
            seg = open_segments(files, n_files, &total, &rc, 1, &any_gz);
            if (!seg) return rc;
        }
        seg_close(seg, n_files);
        return rc;
    }
    if (!rc && total == 0) {
        for (i = 0; !rc && i < n_ctx; ++i) rc = oatk_hip_scan_begin(ctxs[i], 0, sr_db->k, sr_db->s);
        for (i = 0; first && i <= n_ctx; ++i) first[i] = 0;
    } else if (!rc) {
        uint64_t win = forced? forced : WIN_DEFAULT;
        if (n_ctx > 1 && !forced) {                                 /* several handles: at least four windows each, so the parts come out even */
            const uint64_t even = total / (4 * (uint64_t) n_ctx);
            if (even < win) win = even > ((uint64_t) 64 << 20)? even : (uint64_t) 64 << 20;
        }
        if (win < 4096) win = 4096;
        rc = sr_read_stream(ctxs, n_ctx, first, sr_db, sr_db->k, sr_db->s, seg, n_files, total, win, m_data, 0);
        if (rc == OATK_E_NOMEM && win < total) {                    /* a record longer than a window: once more, in one piece */
            oatk_sr_db_clean(sr_db);
            rc = sr_read_stream(ctxs, n_ctx, first, sr_db, sr_db->k, sr_db->s, seg, n_files, total, total, m_data, 0);
        }
    }
    seg_close(seg, n_files);
    return rc;
}

/* The same stream for text that is already in memory (a caller with its own reader; bench.py's PCIe-inclusive leg): windows of the text go over
 * PCIe while the window before is parsed and scanned on the device, the pieces are assembled in ctx; no structs are filled.  `pinned`: the
 * text is page-locked (hipHostMalloc / hipHostRegister) and is copied as it lies; otherwise it is staged through page-locked pieces. */
int oatk_scan_text(oatk_hip_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int pinned, int k, int s, uint64_t window, uint64_t *n_reads)
{
    seg_t seg;
    memset(&seg, 0, sizeof(seg));
    seg.fd = -1, seg.mem = (uint8_t *) text, seg.size = n_bytes, seg.borrowed = 1, seg.pinned = pinned;
    seg.add_nl = n_bytes && text[n_bytes - 1] != '\n';
    const uint64_t total = n_bytes + (uint64_t) seg.add_nl;
    if (seg.add_nl) seg.pinned = 0;                                  /* (the added newline is not in the caller's memory) */
    int rc;
    if (total == 0) rc = oatk_hip_scan_begin(ctx, 0, k, s);
    else {
        uint64_t win = window? window : WIN_DEFAULT;
        if (win < 4096) win = 4096;
        rc = sr_read_stream(&ctx, 1, 0, 0, k, s, &seg, 1, total, win, 0, 0);
        if (rc == OATK_E_NOMEM && win < total) rc = sr_read_stream(&ctx, 1, 0, 0, k, s, &seg, 1, total, total, 0, 0);
    }
    if (!rc && n_reads) { oatk_hip_info_t inf; oatk_hip_info(ctx, &inf); *n_reads = inf.n_reads; }
    return rc;
}
