/*
 * tests/c/gzsrc_fuzz.c -- host/gzsrc.c (+ gzpar.c) against zlib's gzread on FILES made here, under AddressSanitizer + UBSan (tests/test_host_gzpar_fuzz.py).
 * A file is a sequence of members of three kinds -- a plain gzip member (optionally with FEXTRA / FNAME / FCOMMENT / FHCRC in its header, as gzip -N and others write),
 * a run of BGZF blocks (the BC extra field, SAM spec 4.1), an empty member -- optionally followed by bytes that are no member; then the same file damaged.
 * What the reference sees is what gzread hands out (sstream.c:39-54, kseq.h:192-235).  Required: where gzread reads the file to its end without an error, gzsrc
 * delivers the same bytes and no error; where gzsrc reports no error, gzread must have delivered the same bytes; no access outside its own memory, no hang.
 * Test infrastructure: nothing in the product links this.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "host_internal.h"

static uint64_t rng_s = 0x2545F4914F6CDD1DULL;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

static size_t text_of(uint8_t *t, size_t n)
{
    size_t i = 0;
    while (i < n) {
        i += (size_t) snprintf((char *) t + i, n - i, ">r%lu\n", (unsigned long) (rnd() % 1000000));
        size_t l = 200 + rnd() % 9000, j;
        const uint64_t g = rnd() % 30000;
        for (j = 0; j < l && i < n; ++j, ++i) t[i] = (uint8_t) "ACGT"[((g + j) * 2654435761u >> 9 & 3) ^ (rnd() % 300 == 0)];
        if (i < n) t[i++] = '\n';
    }
    return n;
}

static size_t plain_member(const uint8_t *t, size_t n, uint8_t *out, size_t cap, int fancy)
{
    z_stream z;
    gz_header h;
    static uint8_t extra[40];
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 1 + (int) (rnd() % 9), Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY) != Z_OK) abort();
    if (fancy) {
        memset(&h, 0, sizeof(h));
        if (fancy & 1) h.name = (Bytef *) "reads.fa";
        if (fancy & 2) h.comment = (Bytef *) "a comment";
        if (fancy & 4) { size_t q; for (q = 0; q < sizeof(extra); ++q) extra[q] = (uint8_t) rnd(); extra[0] = 'X', extra[1] = 'Y', extra[2] = 36, extra[3] = 0; h.extra = extra, h.extra_len = 40; }
        if (fancy & 8) h.hcrc = 1;
        h.os = 3;
        deflateSetHeader(&z, &h);
    }
    z.next_in = (Bytef *) t, z.avail_in = (uInt) n, z.next_out = out, z.avail_out = (uInt) cap;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    const size_t len = cap - z.avail_out;
    deflateEnd(&z);
    return len;
}

static size_t bgzf_block(const uint8_t *t, size_t n, uint8_t *out, size_t cap)      /* n <= 65280 */
{
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, 1 + (int) (rnd() % 9), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) abort();
    static const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    memcpy(out, hdr, 12);
    out[12] = 'B', out[13] = 'C', out[14] = 2, out[15] = 0;
    z.next_in = (Bytef *) t, z.avail_in = (uInt) n, z.next_out = out + 18, z.avail_out = (uInt) (cap - 26);
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    const size_t dl = cap - 26 - z.avail_out, tot = 18 + dl + 8;
    deflateEnd(&z);
    if (tot > 65536) abort();
    out[16] = (uint8_t) ((tot - 1) & 255), out[17] = (uint8_t) ((tot - 1) >> 8);
    const uint32_t crc = (uint32_t) crc32(crc32(0L, Z_NULL, 0), t, (uInt) n);
    uint8_t *tr = out + 18 + dl;
    tr[0] = (uint8_t) crc, tr[1] = (uint8_t) (crc >> 8), tr[2] = (uint8_t) (crc >> 16), tr[3] = (uint8_t) (crc >> 24);
    tr[4] = (uint8_t) n, tr[5] = (uint8_t) (n >> 8), tr[6] = (uint8_t) (n >> 16), tr[7] = (uint8_t) (n >> 24);
    return tot;
}

static void put(const char *path, const uint8_t *d, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(d, 1, n, f) != n) abort();
    fclose(f);
}

/* gzread's view: the bytes, and whether it ended without an error */
static char g_zmsg[256];
static int by_gzread(const char *path, uint8_t *out, size_t cap, size_t *n_out)
{
    g_zmsg[0] = 0;
    gzFile f = gzopen(path, "rb");
    size_t tot = 0;
    int ok = 1;
    if (!f) abort();
    for (;;) {
        const int got = gzread(f, out + tot, (unsigned) (cap - tot > (1u << 20)? (1u << 20) : cap - tot));
        if (got < 0) { int e; snprintf(g_zmsg, sizeof(g_zmsg), "%s", gzerror(f, &e)); ok = 0; break; }
        if (got == 0) { int e; const char *m = gzerror(f, &e); if (e != Z_OK && e != Z_STREAM_END) { snprintf(g_zmsg, sizeof(g_zmsg), "%s", m); ok = 0; } break; }
        tot += (size_t) got;
        if (tot == cap) { ok = 0; break; }
    }
    gzclose(f);
    *n_out = tot;
    return ok;
}

static int by_gzsrc(const char *path, int threads, size_t bufsz, uint8_t *out, size_t cap, size_t *n_out)
{
    int rc = 0, ok = 1;
    oatk_gzsrc_t *g = oatk_gzsrc_open(path, threads, &rc);
    size_t tot = 0;
    if (!g) { *n_out = 0; return 0; }
    for (;;) {
        const size_t want = bufsz < cap - tot? bufsz : cap - tot;
        if (want == 0) { ok = 0; break; }
        const int64_t got = oatk_gzsrc_read(g, out + tot, want);
        if (got < 0) { ok = 0; break; }
        if (got == 0) break;
        tot += (size_t) got;
    }
    oatk_gzsrc_close(g);
    *n_out = tot;
    return ok;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1? atoi(argv[1]) : 30;
    const char *dir = argc > 3? argv[3] : "/tmp";
    const size_t N = 4u << 20, FCAP = 6u << 20;
    uint8_t *t = malloc(N), *f = malloc(FCAP), *bad = malloc(FCAP), *o1 = malloc(N + 65536), *o2 = malloc(N + 65536);
    char path[512];
    int r, n_same = 0, n_both_err = 0, n_strict = 0;
    if (argc > 2) rng_s ^= (uint64_t) atoll(argv[2]) * 0x9E3779B97F4A7C15ULL;
    snprintf(path, sizeof(path), "%s/gzsrc_fuzz_%d.gz", dir, (int) getpid());
    setenv("OATK_HOST_GZ_PARALLEL", "60000", 1);                  /* members of 60 kB and more go to gzpar.c */
    for (r = 0; r < rounds; ++r) {
        const size_t n = (size_t) (1000 + rnd() % (N - 1000));
        size_t at = 0, fn = 0;
        text_of(t, n);
        while (at < n) {
            const int kind = (int) (rnd() % 8);
            if (kind < 3) {                                    /* a plain member of any size */
                size_t m = 1 + rnd() % (n - at);
                if (rnd() & 1) m = m > 300000? 300000 : m;
                fn += plain_member(t + at, m, f + fn, FCAP - fn, rnd() % 3? 0 : (int) (rnd() & 15));
                at += m;
            } else if (kind < 7) {                             /* a run of BGZF blocks */
                int q, nb = 1 + (int) (rnd() % 40);
                for (q = 0; q < nb && at < n; ++q) {
                    size_t m = rnd() % 5? 65280 : rnd() % 65280;
                    if (m > n - at) m = n - at;
                    fn += bgzf_block(t + at, m, f + fn, FCAP - fn);
                    at += m;
                }
            } else fn += rnd() & 1? bgzf_block(t, 0, f + fn, FCAP - fn) : plain_member(t, 0, f + fn, FCAP - fn, 0);      /* an empty member in between */
        }
        if (rnd() % 3 == 0) fn += bgzf_block(t, 0, f + fn, FCAP - fn);                                                       /* bgzip's end marker */
        if (rnd() % 4 == 0) { size_t q, g = 1 + rnd() % 300; const int zeros = (int) (rnd() & 1); for (q = 0; q < g; ++q) f[fn++] = zeros? 0 : (uint8_t) rnd(); }     /* what follows the last member */
        char env[32];
        snprintf(env, sizeof(env), "%d", 16 << (rnd() % 5));
        setenv("OATK_HOST_GZ_CHUNK_KB", env, 1);
        const int threads = 1 + (int) (rnd() % 9);
        const size_t bufsz = rnd() & 1? 1 + (size_t) (rnd() % 200000) : N + 65536;
        int k;
        for (k = 0; k < 5; ++k) {
            size_t bn = fn, n1, n2, where = 0;
            int what = -1;
            memcpy(bad, f, fn);
            if (k > 0) {
                what = (int) (rnd() % 4);
                if (what == 0) { int q, m = 1 + (int) (rnd() % 3); for (q = 0; q < m; ++q) bad[rnd() % fn] ^= (uint8_t) (1u << (rnd() & 7)); }
                else if (what == 1) { const size_t a = rnd() % fn, cut = 1 + rnd() % 40; where = a; if (a + cut < fn) { memmove(bad + a, bad + a + cut, fn - a - cut); bn = fn - cut; } }
                else if (what == 2) bn = 1 + rnd() % fn;
                else { const size_t a = rnd() % fn; where = a; bad[a] = (uint8_t) rnd(); if (a + 1 < fn) bad[a + 1] = (uint8_t) rnd(); }
            }
            put(path, bad, bn);
            const int ok2 = by_gzread(path, o2, N + 65536, &n2);
            const int ok1 = by_gzsrc(path, threads, bufsz, o1, N + 65536, &n1);
            if (k == 0 && !(ok1 && ok2 && n1 == n && n2 == n && !memcmp(o1, t, n))) { fprintf(stderr, "round %d: the undamaged file: gzsrc ok %d %zu bytes, gzread ok %d %zu bytes, text %zu\n", r, ok1, n1, ok2, n2, n); return 1; }
            if (ok2 && !(ok1 && n1 == n2 && !memcmp(o1, o2, n1))) { fprintf(stderr, "round %d damage %d: gzread reads the file (%zu bytes), gzsrc: ok %d, %zu bytes\n", r, k, n2, ok1, n1); return 1; }
            if (ok1 && !(n1 == n2 && !memcmp(o1, o2, n1))) { fprintf(stderr, "round %d damage %d (kind %d at %zu, file %zu of %zu): gzsrc reports no error (%zu bytes), gzread: ok %d, %zu bytes (%s)\n", r, k, what, where, bn, fn, n1, ok2, n2, g_zmsg); return 1; }
            if (ok1 && ok2) ++n_same; else if (!ok1 && !ok2) ++n_both_err; else ++n_strict;
        }
    }
    unlink(path);
    printf("%d files read alike, %d refused by both, %d with the same bytes where only gzread ends on an error\n", n_same, n_both_err, n_strict);
    free(t), free(f), free(bad), free(o1), free(o2);
    return 0;
}
