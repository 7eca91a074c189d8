/*
 * tests/c/gzpar_fuzz.c -- host/gzpar.c against zlib on streams it has not seen, under AddressSanitizer + UBSan (tests/test_host_gzpar_fuzz.py builds and runs it).
 * Part 1: texts of several kinds deflated by zlib with every strategy / level / window / memLevel and random flushes: the bytes must be zlib's, whatever the
 * thread count, chunk size and buffer size.  Part 2: the same streams with bits flipped, bytes dropped or the tail cut: whatever gzpar.c answers (bytes, "corrupt",
 * "not for me") it must answer without touching memory that is not its own and without hanging; where it delivers a whole member, zlib must have delivered the same.
 * Test infrastructure: nothing in the product links this.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "host_internal.h"

static uint64_t rng_s = 88172645463325252ULL;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

static size_t make_text(uint8_t *t, size_t n, int kind)
{
    size_t i = 0;
    static const char acgt[] = "ACGT";
    while (i < n) {
        if (kind == 0 || kind == 3) {                /* FASTA / FASTQ records, reads cut from a short genome (long matches) or random (short ones) */
            i += (size_t) snprintf((char *) t + i, n - i, "%cread%lu some words\n", kind == 3? '@' : '>', (unsigned long) (rnd() % 100000));
            size_t l = 500 + rnd() % 6000, j;
            uint64_t g = rnd() % 50000;
            for (j = 0; j < l && i < n; ++j, ++i) t[i] = (uint8_t) acgt[kind == 0? ((g + j) * 2654435761u >> 7 & 3) ^ (rnd() % 400 == 0) : rnd() & 3];
            if (i < n) t[i++] = '\n';
            if (kind == 3 && i + l + 3 < n) { t[i++] = '+'; t[i++] = '\n'; for (j = 0; j < l; ++j) t[i++] = (uint8_t) ('!' + rnd() % 40); t[i++] = '\n'; }
        } else if (kind == 1) {                      /* runs and short periods: distances of 1, 2, 7; matches of 258 */
            size_t l = 1 + rnd() % 3000, j, per = 1 + rnd() % 9;
            for (j = 0; j < l && i < n; ++j, ++i) t[i] = (uint8_t) acgt[(j % per) & 3];
            if (i < n && rnd() % 3 == 0) t[i++] = '\n';
        } else {                                     /* wrapped lines of text with few symbols */
            size_t j;
            for (j = 0; j < 60 && i < n; ++j, ++i) t[i] = (uint8_t) ("ACGTNacgtn"[rnd() % (rnd() % 50? 4 : 10)]);
            if (i < n) t[i++] = '\n';
        }
    }
    return n;
}

/* raw deflate of t with random flushes; returns the length */
static size_t deflate_it(const uint8_t *t, size_t n, uint8_t *out, size_t cap, int level, int strategy, int wbits, int memlevel, int flushes)
{
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level, Z_DEFLATED, -wbits, memlevel, strategy) != Z_OK) abort();
    z.next_out = out, z.avail_out = (uInt) cap;
    size_t at = 0;
    while (at < n) {
        size_t m = flushes? 1 + rnd() % (n / (size_t) flushes + 1) : n;
        if (m > n - at) m = n - at;
        z.next_in = (Bytef *) (t + at), z.avail_in = (uInt) m;
        at += m;
        static const int fl[4] = {Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_NO_FLUSH, Z_PARTIAL_FLUSH};
        if (deflate(&z, at == n? Z_FINISH : fl[rnd() & 3]) == Z_STREAM_ERROR) abort();
        if (z.avail_in) abort();
    }
    const size_t len = cap - z.avail_out;
    deflateEnd(&z);
    return len;
}

/* zlib's answer: 0 and the text, or -1 */
static int inflate_it(const uint8_t *d, size_t n, uint8_t *out, size_t cap, size_t *n_out, size_t *used)
{
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) abort();
    z.next_in = (Bytef *) d, z.avail_in = (uInt) n, z.next_out = out, z.avail_out = (uInt) cap;
    const int rc = inflate(&z, Z_FINISH);
    *n_out = cap - z.avail_out, *used = n - z.avail_in;
    inflateEnd(&z);
    return rc == Z_STREAM_END? 0 : -1;
}

/* gzpar's answer: 0 and the text; -1 corrupt; -2 declined */
static int gzpar_it(const uint8_t *d, size_t n, int threads, size_t bufsz, uint8_t *out, size_t cap, size_t *n_out, size_t *used, uint32_t *crc)
{
    oatk_gzpar_t *p = oatk_gzpar_open(d, n, threads);
    if (!p) abort();
    size_t tot = 0;
    int rc = 0, spins = 0;
    while (!oatk_gzpar_done(p)) {
        const size_t want = bufsz < cap - tot? bufsz : cap - tot;
        if (want == 0) { rc = -3; break; }               /* more text than the original: cannot be right */
        const int64_t got = oatk_gzpar_read(p, out + tot, want);
        if (got < 0) { rc = (int) got; break; }
        if (got == 0 && ++spins > 1000000) { fprintf(stderr, "no progress\n"); abort(); }
        tot += (size_t) got;
    }
    *n_out = tot, *used = (size_t) oatk_gzpar_in_used(p), *crc = oatk_gzpar_crc(p);
    oatk_gzpar_close(p);
    return rc;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1? atoi(argv[1]) : 40;
    const size_t N = 3u << 20;
    uint8_t *t = malloc(N), *d = malloc(N + N / 2 + 1024), *bad = malloc(N + N / 2 + 1024), *o1 = malloc(N + 4096), *o2 = malloc(N + 4096);
    int r, n_equal = 0, n_declined = 0, n_bad_err = 0, n_bad_same = 0, n_bad_declined = 0;
    if (argc > 2) rng_s ^= (uint64_t) atoll(argv[2]) * 0x9E3779B97F4A7C15ULL;
    for (r = 0; r < rounds; ++r) {
        const size_t n = (size_t) (200000 + rnd() % (N - 200000));
        make_text(t, n, r & 3);
        static const int strat[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        const int level = 1 + (int) (rnd() % 9), strategy = strat[rnd() % (r % 5 == 4? 5 : 2)], wbits = 9 + (int) (rnd() % 7), memlevel = 1 + (int) (rnd() % 9);
        const size_t dn = deflate_it(t, n, d, N + N / 2 + 1024, level, strategy, wbits, memlevel, r % 3 == 0? (int) (rnd() % 200) : 0);
        char env[32];
        snprintf(env, sizeof(env), "%d", 16 << (rnd() % 6));
        setenv("OATK_HOST_GZ_CHUNK_KB", env, 1);
        snprintf(env, sizeof(env), "%d", 1 + (int) (rnd() % 4));
        setenv("OATK_HOST_GZ_SLOTS", env, 1);
        const int threads = 2 + (int) (rnd() % 7);
        const size_t bufsz = rnd() & 1? 1 + (size_t) (rnd() % 70000) : N;
        size_t n1, u1, n2, u2;
        uint32_t crc;
        int rc = gzpar_it(d, dn, threads, bufsz, o1, N + 4096, &n1, &u1, &crc);
        if (rc == -2) ++n_declined;
        else {
            if (rc != 0 || n1 != n || memcmp(o1, t, n) || u1 != dn || crc != (uint32_t) crc32(crc32(0L, Z_NULL, 0), t, (uInt) n)) {
                fprintf(stderr, "round %d: level %d strategy %d wbits %d memlevel %d chunk %s threads %d buf %zu: rc %d, %zu bytes of %zu, used %zu of %zu\n", r, level, strategy, wbits, memlevel, getenv("OATK_HOST_GZ_CHUNK_KB"), threads, bufsz, rc, n1, n, u1, dn);
                return 1;
            }
            ++n_equal;
        }
        /* damage */
        int k;
        for (k = 0; k < 6; ++k) {
            size_t bn = dn;
            memcpy(bad, d, dn);
            const int what = (int) (rnd() % 4);
            if (what == 0) { int q, m = 1 + (int) (rnd() % 4); for (q = 0; q < m; ++q) bad[rnd() % dn] ^= (uint8_t) (1u << (rnd() & 7)); }
            else if (what == 1) { const size_t at = rnd() % dn, cut = 1 + rnd() % 64; if (at + cut < dn) { memmove(bad + at, bad + at + cut, dn - at - cut); bn = dn - cut; } }
            else if (what == 2) bn = 1 + rnd() % dn;
            else { const size_t at = rnd() % dn, len = 1 + rnd() % 3000; size_t q; for (q = at; q < dn && q < at + len; ++q) bad[q] = (uint8_t) rnd(); }
            const int zr = inflate_it(bad, bn, o2, N + 4096, &n2, &u2);
            rc = gzpar_it(bad, bn, threads, bufsz, o1, N + 4096, &n1, &u1, &crc);
            if (rc == -2) ++n_bad_declined;
            else if (rc != 0) ++n_bad_err;
            else {
                /* a whole member delivered: zlib must agree, byte for byte and about where it ended */
                if (zr != 0 || n1 != n2 || memcmp(o1, o2, n1) || u1 != u2) { fprintf(stderr, "round %d damage %d (kind %d): gzpar delivered %zu bytes (used %zu), zlib rc %d %zu bytes (used %zu)\n", r, k, what, n1, u1, zr, n2, u2); return 1; }
                ++n_bad_same;
            }
            if (rc != 0 && rc != -2 && zr == 0) { fprintf(stderr, "round %d damage %d (kind %d): zlib inflates (%zu bytes) what gzpar calls corrupt (rc %d after %zu bytes)\n", r, k, what, n2, rc, n1); return 1; }
        }
    }
    printf("%d streams equal to zlib's text, %d declined; damaged: %d reported, %d declined, %d delivered as zlib delivers them\n", n_equal, n_declined, n_bad_err, n_bad_declined, n_bad_same);
    free(t), free(d), free(bad), free(o1), free(o2);
    return 0;
}
