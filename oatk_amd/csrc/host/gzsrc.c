/*
 * oatk_amd/csrc/host/gzsrc.c -- a gzip'ed input file as a STREAM of inflated bytes (round 4), for the reader of host/ingest_host.c.
 *
 * The reference reads its input through zlib's gzread (sstream.c:39-54, kseq.h:192-235): one member or many, and what follows the last member
 * is ignored.  Until round 3 this build inflated a .gz file WHOLE, on one thread, into memory before the first byte went to the device; at the
 * size of BASELINE.json's configs[0] that is a minute in front of a pipeline of seconds and the file's whole text in RAM.  Here the caller asks for
 * the next piece of text, straight into the (page-locked) buffer it uploads from, so inflating overlaps the upload, the record scan and the
 * syncmer scan, and nothing larger than the caller's buffers is ever held.  Three kinds of file:
 *
 *   BGZF (bgzip, htslib: every member carries its compressed size in a `BC` extra field, RFC 1952 2.3.1.1 / SAM spec 4.1) -- the members that
 *        fit the caller's buffer are found by hopping from header to header, their inflated sizes are read from their trailers, and they are
 *        inflated IN PARALLEL, each straight to its place in the buffer;
 *   one member (gzip, pigz) -- serial by nature: inflated as the caller asks for more;
 *   several plain members (cat a.gz b.gz) -- one after the other like gzread does; a member that turns out to be BGZF switches to the parallel form.
 *
 * A file that is not a regular file (a pipe) goes through zlib's own gzread.  CRC and length of every member are checked, as gzread checks them.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "oatk_hip.h"
#include "host_internal.h"

#define GZ_MAX_THREADS 64

typedef struct { uint64_t in_off; uint32_t in_len, hdr_len, out_len; uint64_t out_off; } bg_block_t;

struct oatk_gzsrc {
    int fd;
    const uint8_t *map; uint64_t size;      /* the compressed file (regular files) */
    uint64_t pos;                            /* next compressed byte */
    int eof, failed;
    /* serial member decoding */
    z_stream z; int z_live, in_member;
    uint32_t crc; uint64_t member_out;
    oatk_gzpar_t *par;                       /* the member at hand is inflated on many threads (host/gzpar.c) */
    int par_kind;                            /* (statistics: 1 once a member went that way) */
    int small_members;                       /* a member that went the many-thread way turned out smaller than the gate: the members behind it are read by zlib */
    uint64_t par_least;                      /* the gate that member passed */
    uint64_t n_par_opened;                   /* (statistics, tests) members the many-thread reader was opened for */
    /* pipe fallback */
    gzFile gzf;
    /* BGZF */
    int bgzf;
    bg_block_t *blk; uint64_t n_blk, m_blk;
    /* worker pool */
    int n_threads, started;
    pthread_t th[GZ_MAX_THREADS];
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    uint64_t gen; int busy, quit;
    uint64_t next_blk;                       /* claimed with an atomic add */
    uint8_t *dst;
    struct { struct oatk_gzsrc *g; uint64_t seen0; } targ[GZ_MAX_THREADS];     /* what a worker starts with: the generation that was current when it was created */
};

/* length of the gzip header at p (RFC 1952), or 0 if there is none / it is cut; *bsize = the BGZF block size if the member carries a BC field */
static uint32_t gzs_header(const uint8_t *p, uint64_t n, uint32_t *bsize)
{
    if (bsize) *bsize = 0;
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
    const int flg = p[3];
    uint64_t o = 10;
    if (flg & 4) {
        if (o + 2 > n) return 0;
        const uint32_t xlen = p[o] | (uint32_t) p[o + 1] << 8;
        o += 2;
        if (o + xlen > n) return 0;
        uint64_t q = o;
        while (q + 4 <= o + xlen) {
            const uint32_t slen = p[q + 2] | (uint32_t) p[q + 3] << 8;
            if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= o + xlen && bsize) *bsize = (p[q + 4] | (uint32_t) p[q + 5] << 8) + 1;
            q += 4 + slen;
        }
        o += xlen;
    }
    if (flg & 8) { while (o < n && p[o]) ++o; ++o; }
    if (flg & 16) { while (o < n && p[o]) ++o; ++o; }
    if (flg & 2) o += 2;
    return o + 8 <= n? (uint32_t) o : 0;
}

static void *gz_worker(void *arg)
{
    /* A worker created in a LATER call of bgzf_read (an earlier one had fewer blocks than threads, or pthread_create failed part of the way) must
     * not take the generations that went before it for work it has not done: it would run a phantom round and count `busy` down once too often
     * (the caller then hung, or went on before every block was inflated).  It starts from the generation current at its creation. */
    oatk_gzsrc_t *g = ((oatk_gzsrc_t **) arg)[0];
    z_stream z;
    uint64_t seen = ((uint64_t *) arg)[1];
    int live = 0;
    memset(&z, 0, sizeof(z));
    for (;;) {
        pthread_mutex_lock(&g->mu);
        while (g->gen == seen && !g->quit) pthread_cond_wait(&g->cv_go, &g->mu);
        if (g->quit) { pthread_mutex_unlock(&g->mu); break; }
        seen = g->gen;
        pthread_mutex_unlock(&g->mu);
        for (;;) {
            const uint64_t b = __atomic_fetch_add(&g->next_blk, 1, __ATOMIC_RELAXED);
            if (b >= g->n_blk) break;
            const bg_block_t *B = &g->blk[b];
            int ok = 1;
            {   /* (a block that says it holds no text -- bgzip's end marker -- is inflated like any other, into a few bytes of its own: a damaged block whose length field
                 *  happens to read 0 must not pass for an empty one; found by tests/c/gzsrc_fuzz.c) */
                uint8_t none[8];
                uint8_t *to = B->out_len? g->dst + B->out_off : none;
                if (!live) { live = inflateInit2(&z, -15) == Z_OK; if (!live) ok = 0; }
                else inflateReset(&z);
                if (ok) {
                    z.next_in = (Bytef *) (g->map + B->in_off + B->hdr_len), z.avail_in = B->in_len - B->hdr_len - 8;
                    z.next_out = to, z.avail_out = B->out_len? B->out_len : (uInt) sizeof(none);
                    ok = inflate(&z, Z_FINISH) == Z_STREAM_END && z.total_out == B->out_len && z.avail_in == 0;
                }
                if (ok) {
                    uint32_t want;
                    memcpy(&want, g->map + B->in_off + B->in_len - 8, 4);
                    ok = (uint32_t) crc32(crc32(0L, Z_NULL, 0), to, B->out_len) == want;
                }
            }
            if (!ok) __atomic_store_n(&g->failed, 1, __ATOMIC_RELAXED);
        }
        pthread_mutex_lock(&g->mu);
        if (--g->busy == 0) pthread_cond_signal(&g->cv_done);
        pthread_mutex_unlock(&g->mu);
    }
    if (live) inflateEnd(&z);
    return 0;
}

oatk_gzsrc_t *oatk_gzsrc_open(const char *path, int n_threads, int *rc)
{
    struct stat sb;
    oatk_gzsrc_t *g = (oatk_gzsrc_t *) calloc(1, sizeof(*g));
    if (rc) *rc = OATK_OK;
    if (!g) { if (rc) *rc = OATK_E_NOMEM; return 0; }
    g->fd = open(path, O_RDONLY);
    if (g->fd < 0 || fstat(g->fd, &sb) != 0) { if (g->fd >= 0) close(g->fd); free(g); if (rc) *rc = OATK_E_ARG; return 0; }
    pthread_mutex_init(&g->mu, 0);
    pthread_cond_init(&g->cv_go, 0);
    pthread_cond_init(&g->cv_done, 0);
    if (!S_ISREG(sb.st_mode)) {                      /* a pipe: zlib's own reader (it also passes text that is not gzip'ed through) */
        g->gzf = gzdopen(g->fd, "r");
        if (!g->gzf) { close(g->fd); free(g); if (rc) *rc = OATK_E_ARG; return 0; }
        (void) gzbuffer(g->gzf, 1 << 20);
        return g;
    }
    g->size = (uint64_t) sb.st_size;
    if (g->size) {
        g->map = (const uint8_t *) mmap(0, (size_t) g->size, PROT_READ, MAP_PRIVATE, g->fd, 0);
        if (g->map == MAP_FAILED) { g->map = 0; close(g->fd); free(g); if (rc) *rc = OATK_E_NOMEM; return 0; }
        (void) madvise((void *) g->map, (size_t) g->size, MADV_SEQUENTIAL);
    }
    uint32_t bsize = 0;
    g->bgzf = g->size && gzs_header(g->map, g->size, &bsize) && bsize;
    g->n_threads = n_threads < 1? 1 : (n_threads > GZ_MAX_THREADS? GZ_MAX_THREADS : n_threads);
    return g;
}

void oatk_gzsrc_close(oatk_gzsrc_t *g)
{
    int i;
    if (!g) return;
    if (g->started) {
        pthread_mutex_lock(&g->mu);
        g->quit = 1;
        pthread_cond_broadcast(&g->cv_go);
        pthread_mutex_unlock(&g->mu);
        for (i = 0; i < g->started; ++i) pthread_join(g->th[i], 0);
    }
    if (g->z_live) inflateEnd(&g->z);
    if (g->par) oatk_gzpar_close(g->par);
    if (g->gzf) gzclose(g->gzf);                     /* (closes the descriptor) */
    else {
        if (g->map) munmap((void *) g->map, (size_t) g->size);
        close(g->fd);
    }
    pthread_mutex_destroy(&g->mu);
    pthread_cond_destroy(&g->cv_go);
    pthread_cond_destroy(&g->cv_done);
    free(g->blk);
    free(g);
}

/* compressed bytes consumed so far.  Inside a member that is inflated on many threads (host/gzpar.c) that is the boundary its delivered chunks have reached -- the reader of
 * host/ingest_host.c sizes its arrays and the device's batch by "text of this window per compressed byte it took": a position that stood still through a member made that
 * estimate a ten-million-fold over-estimate (r05: a box lost to the memset of what realloc had promised) */
uint64_t oatk_gzsrc_tell_in(const oatk_gzsrc_t *g) { return g->gzf? 0 : g->pos + (g->par? oatk_gzpar_in_used(g->par) : 0); }
uint64_t oatk_gzsrc_size_in(const oatk_gzsrc_t *g) { return g->size; }
int oatk_gzsrc_kind(const oatk_gzsrc_t *g) { return g->gzf? 3 : (g->bgzf? 2 : 1); }
uint64_t oatk_gzsrc_members_on_many_threads(const oatk_gzsrc_t *g) { return g->n_par_opened; }

/* the BGZF members from pos on that fit `cap` bytes of text, inflated on the pool; 0: the next member is not BGZF (or nothing fits) */
static int64_t bgzf_read(oatk_gzsrc_t *g, uint8_t *dst, uint64_t cap)
{
    uint64_t out = 0, p = g->pos;
    g->n_blk = 0;
    while (p < g->size) {
        uint32_t bsize = 0;
        const uint32_t hl = gzs_header(g->map + p, g->size - p, &bsize);
        if (!hl || !bsize || p + bsize > g->size || bsize < hl + 8) break;
        uint32_t isize;
        memcpy(&isize, g->map + p + bsize - 4, 4);
        if (isize > 0x10000) break;                  /* not what bgzip writes */
        if (out + isize > cap) break;
        if (g->n_blk == g->m_blk) {
            g->m_blk = g->m_blk? 2 * g->m_blk : 4096;
            bg_block_t *nb = (bg_block_t *) realloc(g->blk, g->m_blk * sizeof(bg_block_t));
            if (!nb) return -1;
            g->blk = nb;
        }
        bg_block_t *B = &g->blk[g->n_blk++];
        B->in_off = p, B->in_len = bsize, B->hdr_len = hl, B->out_len = isize, B->out_off = out;
        out += isize, p += bsize;
    }
    if (g->n_blk == 0) return 0;
    g->dst = dst;
    __atomic_store_n(&g->next_blk, 0, __ATOMIC_RELAXED);
    const int want = g->n_blk < (uint64_t) g->n_threads? (int) g->n_blk : g->n_threads;
    while (g->started < want) {
        g->targ[g->started].g = g, g->targ[g->started].seen0 = g->gen;       /* (gen is written by this thread only, under the mutex, below) */
        if (pthread_create(&g->th[g->started], 0, gz_worker, &g->targ[g->started]) != 0) break;
        ++g->started;
    }
    if (g->started == 0) return -1;
    pthread_mutex_lock(&g->mu);
    g->busy = g->started, ++g->gen;
    pthread_cond_broadcast(&g->cv_go);
    while (g->busy) pthread_cond_wait(&g->cv_done, &g->mu);
    pthread_mutex_unlock(&g->mu);
    if (g->failed) return -1;
    g->pos = p;
    return (int64_t) out;
}

/* the member at pos (or the one being decoded) inflated until `cap` bytes are out or the member ends; 0 with eof set when no member follows */
static int64_t serial_read(oatk_gzsrc_t *g, uint8_t *dst, uint64_t cap)
{
    uint64_t out = 0;
    if (!g->in_member) {
        uint32_t bg_size = 0;                        /* (a BGZF member says how long it is: at most 64 KiB, never worth many threads) */
        const uint32_t hl = g->pos < g->size? gzs_header(g->map + g->pos, g->size - g->pos, &bg_size) : 0;
        if (!hl) {                                   /* the end, or bytes that are no gzip member: ignored like gzread ignores them (gzread.c: "trailing garbage") */
            if (g->pos == 0 && g->size) return -1;
            g->eof = 1;
            return 0;
        }
        if (!g->z_live) { if (inflateInit2(&g->z, -15) != Z_OK) return -1; g->z_live = 1; }
        else inflateReset(&g->z);
        g->pos += hl, g->in_member = 1, g->crc = (uint32_t) crc32(0L, Z_NULL, 0), g->member_out = 0;
        /* a large member and threads to spare: many threads enter it at block boundaries (host/gzpar.c); OATK_HOST_GZ_PARALLEL=0: zlib on one thread */
        {
            const char *e = getenv("OATK_HOST_GZ_PARALLEL");
            const uint64_t least = e && atoi(e) > 1? (uint64_t) atoi(e) : (8u << 20);
            /* (round 6: the gate looked at the rest of the FILE, not at the member -- a BGZF member cut by the end of the caller's buffer, one per 32 MB of text, opened the
             *  many-thread reader on everything behind it: threads started, boundaries searched and megabytes decoded in LATER members, all thrown away when the member ended
             *  64 KiB on; 1.7 s of the config-1 surrogate's 2.0 s sr_read from BGZF.  A member that says it is BGZF is inflated by zlib; and a file whose members turn out
             *  small -- concatenated lanes, per-block gzip writers: ADVICE r05 -- is read by zlib from the first such member on.) */
            if (!(e && e[0] == '0' && !e[1]) && g->n_threads >= 4 && g->size - g->pos >= least && !bg_size && !g->small_members)
                g->par = oatk_gzpar_open(g->map + g->pos, g->size - g->pos, g->n_threads), g->par_least = least, ++g->n_par_opened;
        }
    }
    while (g->par && out < cap) {
        const int64_t n = oatk_gzpar_read(g->par, dst + out, cap - out);
        if (n == -2) { oatk_gzpar_close(g->par); g->par = 0; break; }          /* nothing was taken: this is not text one can enter in the middle -- zlib from the member's start */
        if (n < 0) return -1;
        out += (uint64_t) n, g->member_out += (uint64_t) n, g->par_kind = 1;
        if (oatk_gzpar_done(g->par)) {
            uint32_t t[2];
            if (oatk_gzpar_in_used(g->par) < g->par_least) g->small_members = 1;
            g->pos += oatk_gzpar_in_used(g->par);
            if (g->size - g->pos < 8) return -1;
            memcpy(t, g->map + g->pos, 8);
            if (t[0] != oatk_gzpar_crc(g->par) || t[1] != (uint32_t) oatk_gzpar_total(g->par)) return -1;
            g->pos += 8, g->in_member = 0;
            oatk_gzpar_close(g->par), g->par = 0;
            return (int64_t) out;
        }
    }
    if (g->par) return (int64_t) out;
    while (out < cap && g->in_member) {
        const uint64_t in_left = g->size - g->pos, want = cap - out;
        g->z.next_in = (Bytef *) (g->map + g->pos), g->z.avail_in = in_left > (1u << 30)? (1u << 30) : (uInt) in_left;
        g->z.next_out = dst + out, g->z.avail_out = want > (1u << 30)? (1u << 30) : (uInt) want;
        const uInt in0 = g->z.avail_in, out0 = g->z.avail_out;
        const int r = inflate(&g->z, Z_NO_FLUSH);
        const uint64_t used = in0 - g->z.avail_in, made = out0 - g->z.avail_out;
        g->crc = (uint32_t) crc32(g->crc, dst + out, (uInt) made);
        g->pos += used, out += made, g->member_out += made;
        if (r == Z_STREAM_END) {
            uint32_t t[2];
            if (g->size - g->pos < 8) return -1;
            memcpy(t, g->map + g->pos, 8);
            if (t[0] != g->crc || t[1] != (uint32_t) g->member_out) return -1;
            g->pos += 8, g->in_member = 0;
        } else if (r != Z_OK && r != Z_BUF_ERROR) return -1;                         /* corrupt */
        else if (!used && !made) return -1;                                          /* the file ends inside a member */
    }
    return (int64_t) out;
}

/* the next bytes of the text, at most cap; fewer than cap does not mean the end: 0 does.  < 0 when the file is damaged */
int64_t oatk_gzsrc_read(oatk_gzsrc_t *g, uint8_t *dst, uint64_t cap)
{
    if (!g || g->failed) return -1;
    if (g->eof || cap == 0) return 0;
    uint64_t out = 0;
    if (g->gzf) {
        while (out < cap) {
            const uint64_t want = cap - out > (1u << 30)? (1u << 30) : cap - out;
            const int got = gzread(g->gzf, dst + out, (unsigned) want);
            if (got < 0) { g->failed = 1; return -1; }
            if (got == 0) { g->eof = 1; break; }
            out += (uint64_t) got;
        }
        return (int64_t) out;
    }
    while (out < cap && !g->eof) {
        int64_t n;
        if (!g->in_member) {
            if (g->pos >= g->size) { g->eof = 1; break; }
            n = bgzf_read(g, dst + out, cap - out);      /* as many whole BGZF members as fit, in parallel */
            if (n < 0) { g->failed = 1; return -1; }
            if (n > 0) { out += (uint64_t) n; continue; }
        }
        n = serial_read(g, dst + out, cap - out);        /* a plain member, an empty one (bgzip's end marker), or a BGZF member cut by the end of the buffer */
        if (n < 0) { g->failed = 1; return -1; }
        out += (uint64_t) n;
    }
    return (int64_t) out;
}
