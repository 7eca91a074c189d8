/*
 * oatk_amd/csrc/host/fasta_out.c -- a packed read stream written as a FASTA file, plain or gzip'ed (include/oatk_host.h: oatk_write_fasta).
 *
 * Test and bench inputs only: BASELINE.json's configs[0] is a `.hifi.fa.gz`, and the reference reads such a file through zlib's gzread
 * (sstream.c:39-54), which accepts ONE member as well as many (BGZF, `cat a.gz b.gz`).  The three gzip'ed forms are what the reader of
 * host/ingest_host.c has to take apart, so the writer makes all three -- on every host thread, because a bench input is gigabytes:
 * the text is deflated in independent blocks that are either members of their own (BGZF; plain members) or, for the single-member form,
 * raw deflate pieces that end on a byte boundary (Z_FULL_FLUSH) and are laid behind each other under one header, one CRC (crc32_combine)
 * and one length -- pigz's construction.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "oatk_host.h"

#define BATCH_TEXT ((uint64_t) 256 << 20)
#define BGZF_IN 0xff00u                         /* text bytes per BGZF block, as bgzip cuts it */

typedef struct {
    /* text of the batch */
    const uint8_t *seq; const uint64_t *off; const uint32_t *len; const uint64_t *tpos;     /* tpos[i]: where read i's record starts in the batch's text */
    uint64_t r0, r1, first_id;
    uint8_t *text; uint64_t n_text;
    /* blocks of the batch */
    int mode, level, last_batch;
    uint64_t blk_bytes, n_blk;
    uint8_t *out; uint64_t out_stride; uint32_t *out_len; uint32_t *blk_crc;
    int failed;
    int tid, nthr;
} wjob_t;

static int digits(uint64_t v) { int d = 1; while (v >= 10) v /= 10, ++d; return d; }

static void *text_worker(void *arg)
{
    wjob_t *j = (wjob_t *) arg;
    uint64_t i;
    for (i = j->r0 + (uint64_t) j->tid; i < j->r1; i += (uint64_t) j->nthr) {
        uint8_t *p = j->text + j->tpos[i - j->r0];
        p += sprintf((char *) p, ">r%lu\n", (unsigned long) (j->first_id + i));
        memcpy(p, j->seq + j->off[i], j->len[i]);
        p[j->len[i]] = '\n';
    }
    return 0;
}

static void *deflate_worker(void *arg)
{
    wjob_t *j = (wjob_t *) arg;
    uint64_t b;
    for (b = (uint64_t) j->tid; b < j->n_blk; b += (uint64_t) j->nthr) {
        const uint64_t t0 = b * j->blk_bytes, t1 = t0 + j->blk_bytes < j->n_text? t0 + j->blk_bytes : j->n_text;
        uint8_t *o = j->out + b * j->out_stride;
        z_stream z;
        memset(&z, 0, sizeof(z));
        /* mode 1, 2: raw deflate (the wrapper is written by hand); mode 3: zlib writes the gzip wrapper */
        if (deflateInit2(&z, j->level, Z_DEFLATED, j->mode == 3? 31 : -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { j->failed = 1; return 0; }
        uint32_t head = 0;
        if (j->mode == 2) head = 18;
        z.next_in = (Bytef *) (j->text + t0), z.avail_in = (uInt) (t1 - t0);
        z.next_out = o + head, z.avail_out = (uInt) (j->out_stride - head - 8);
        const int last = j->last_batch && b + 1 == j->n_blk;
        const int rc = deflate(&z, j->mode == 1 && !last? Z_FULL_FLUSH : Z_FINISH);
        if ((j->mode == 1 && !last? rc != Z_OK : rc != Z_STREAM_END) || z.avail_in) { deflateEnd(&z); j->failed = 1; return 0; }
        uint32_t n = head + (uint32_t) z.total_out;
        deflateEnd(&z);
        const uint32_t crc = (uint32_t) crc32(crc32(0L, Z_NULL, 0), j->text + t0, (uInt) (t1 - t0));
        j->blk_crc[b] = crc;
        if (j->mode == 2) {
            static const uint8_t H[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            const uint32_t isz = (uint32_t) (t1 - t0), bsize = n + 8 - 1;
            memcpy(o, H, 16);
            o[16] = (uint8_t) (bsize & 0xff), o[17] = (uint8_t) (bsize >> 8);
            memcpy(o + n, &crc, 4), memcpy(o + n + 4, &isz, 4);
            n += 8;
            if (bsize > 0xffff) { j->failed = 1; return 0; }
        }
        j->out_len[b] = n;
    }
    return 0;
}

static void run_threads(void *(*fn)(void *), wjob_t *proto, int nthr)
{
    pthread_t th[256];
    wjob_t jb[256];
    int t, started[256];
    for (t = 0; t < nthr; ++t) {
        jb[t] = *proto, jb[t].tid = t, jb[t].nthr = nthr;
        started[t] = t > 0 && pthread_create(&th[t], 0, fn, &jb[t]) == 0;
    }
    fn(&jb[0]);
    for (t = 1; t < nthr; ++t) { if (started[t]) pthread_join(th[t], 0); else fn(&jb[t]); }
    for (t = 0; t < nthr; ++t) if (jb[t].failed) proto->failed = 1;
}

int oatk_write_fasta(const char *path, const uint8_t *seq, const uint64_t *off, const uint32_t *len, uint64_t n_reads, uint64_t first_id,
                     int mode, int level, uint64_t member_bytes, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (mode < 0 || mode > 3) return -1;
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    int rc = 0;
    uint64_t r0 = 0, total_text = 0;
    uint32_t crc_all = (uint32_t) crc32(0L, Z_NULL, 0);
    if (mode == 1) {
        static const uint8_t H[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        if (fwrite(H, 1, 10, fp) != 10) rc = -1;
    }
    uint64_t *tpos = 0;
    uint8_t *text = 0, *out = 0;
    uint32_t *out_len = 0, *blk_crc = 0;
    uint64_t cap_text = 0, cap_out = 0, cap_blk = 0, cap_reads = 0;
    const uint64_t blk_bytes = mode == 2? BGZF_IN : (mode == 3? (member_bytes? member_bytes : (uint64_t) 64 << 20) : (uint64_t) 4 << 20);
    while (!rc && (r0 < n_reads || (n_reads == 0 && r0 == 0))) {
        /* a batch of whole records */
        uint64_t r1 = r0, n_text = 0;
        while (r1 < n_reads && (n_text < BATCH_TEXT || r1 == r0)) { n_text += 4 + (uint64_t) digits(first_id + r1) + len[r1]; ++r1; }
        if (r1 - r0 + 1 > cap_reads) { cap_reads = r1 - r0 + 1; free(tpos); tpos = (uint64_t *) malloc(8 * cap_reads); }
        if (n_text + 64 > cap_text) { cap_text = n_text + 64; free(text); text = (uint8_t *) malloc(cap_text); }
        if (!tpos || !text) { rc = -1; break; }
        uint64_t i, t = 0;
        for (i = r0; i < r1; ++i) { tpos[i - r0] = t; t += 4 + (uint64_t) digits(first_id + i) + len[i]; }
        wjob_t job;
        memset(&job, 0, sizeof(job));
        job.seq = seq, job.off = off, job.len = len, job.tpos = tpos, job.r0 = r0, job.r1 = r1, job.first_id = first_id, job.text = text, job.n_text = n_text;
        run_threads(text_worker, &job, n_threads);
        const int last_batch = r1 >= n_reads;
        if (mode == 0) {
            if (n_text && fwrite(text, 1, n_text, fp) != n_text) rc = -1;
        } else {
            const uint64_t n_blk = n_text? (n_text + blk_bytes - 1) / blk_bytes : (mode == 1 && last_batch? 1 : 0);
            const uint64_t stride = deflateBound(0, (uLong) blk_bytes) + 64 + (blk_bytes >> 6);
            if (n_blk * stride > cap_out) { cap_out = n_blk * stride; free(out); out = (uint8_t *) malloc(cap_out? cap_out : 1); }
            if (n_blk > cap_blk) { cap_blk = n_blk; free(out_len); free(blk_crc); out_len = (uint32_t *) malloc(4 * cap_blk), blk_crc = (uint32_t *) malloc(4 * cap_blk); }
            if (n_blk && (!out || !out_len || !blk_crc)) { rc = -1; break; }
            job.mode = mode, job.level = level, job.last_batch = last_batch, job.blk_bytes = blk_bytes, job.n_blk = n_blk;
            job.out = out, job.out_stride = stride, job.out_len = out_len, job.blk_crc = blk_crc;
            run_threads(deflate_worker, &job, n_threads);
            if (job.failed) { rc = -1; break; }
            uint64_t b;
            for (b = 0; b < n_blk && !rc; ++b) {
                if (fwrite(out + b * stride, 1, out_len[b], fp) != out_len[b]) rc = -1;
                const uint64_t t0 = b * blk_bytes, t1 = t0 + blk_bytes < n_text? t0 + blk_bytes : n_text;
                crc_all = (uint32_t) crc32_combine(crc_all, blk_crc[b], (z_off_t) (t1 - t0));
            }
        }
        total_text += n_text;
        r0 = r1;
        if (n_reads == 0) break;
    }
    if (!rc && mode == 1) {
        const uint32_t isz = (uint32_t) total_text;
        if (fwrite(&crc_all, 4, 1, fp) != 1 || fwrite(&isz, 4, 1, fp) != 1) rc = -1;
    }
    if (!rc && mode == 2) {                      /* bgzip's end-of-file marker: an empty block */
        static const uint8_t E[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (fwrite(E, 1, 28, fp) != 28) rc = -1;
    }
    free(tpos); free(text); free(out); free(out_len); free(blk_crc);
    if (fclose(fp) != 0) rc = -1;
    return rc;
}
