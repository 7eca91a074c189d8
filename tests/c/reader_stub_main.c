/*
 * tests/c/reader_stub_main.c -- the streamed reader end to end on the CPU: oatk_sr_read_files (host/ingest_host.c, gzsrc.c, gzpar.c, srdb.c) over the STUB device
 * (tests/c/stub_device.c), on a .fa.gz written here, under an address-space limit set here.  Test infrastructure (tests/test_host_reader_stub.py).
 *
 *   reader_stub <file.fa.gz> <mode 0 plain | 1 one member | 2 BGZF | 3 several members> <n_reads> <mean_len> <as_limit_gb> <arena 0|1> [threads]
 *
 * Writes the file (synthetic reads: oatk_synth_*), reads it back through the host library, and checks every read of the result: name, compressed length, packed bases and
 * run lengths (recomputed here from the bases that were written), and the fabricated syncmer arrays.  Prints one line of numbers; exit 0 = all equal.
 * A reader that sizes an array from a wrong estimate and touches it runs into RLIMIT_AS (malloc / mmap fail -> OATK_E_NOMEM or abort), here, not on a GPU box.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <time.h>

#include "oatk_host.h"
#include "oatk_syncasm.h"

uint64_t stub_device_peak_bytes(void);
uint64_t stub_device_biggest_request(void);
void oatk_host_set_arena(int on);
void oatk_sr_db_clean(oatk_sr_db_t *sr_db);

static uint64_t mix(uint64_t x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ULL; x ^= x >> 29; return x; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

static long peak_rss_kb(void)
{
    FILE *f = fopen("/proc/self/status", "r");
    char line[256];
    long v = 0;
    while (f && fgets(line, sizeof line, f)) if (!strncmp(line, "VmHWM:", 6)) v = atol(line + 6);
    if (f) fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s file mode n_reads mean_len as_limit_gb arena [threads]\n", argv[0]); return 2; }
    const char *path = argv[1];
    const int mode = atoi(argv[2]);
    const uint64_t n = strtoull(argv[3], 0, 10), mean = strtoull(argv[4], 0, 10);
    const double lim_gb = atof(argv[5]);
    const int arena = atoi(argv[6]), threads = argc > 7? atoi(argv[7]) : 8;
    oatk_host_set_threads(threads);

    /* the reads, as bench.py's workload draws them */
    oatk_synth_t P = {2000000, n, 1001, 31, mean, 500};
    uint8_t *genome = (uint8_t *) malloc(P.genome_len);
    oatk_synth_genome(&P, genome);
    uint32_t *len = (uint32_t *) malloc(4 * n);
    uint64_t *off = (uint64_t *) malloc(8 * (n + 1)), i, total = 0;
    oatk_synth_lengths(&P, 0, n, len);
    for (i = 0; i < n; ++i) off[i] = total, total += ((uint64_t) len[i] + 63) & ~63ULL;
    uint8_t *seq = (uint8_t *) malloc(total + 64);
    oatk_synth_reads(&P, genome, 0, n, off, seq, threads);
    /* a few reads with what the synthetic ones lack: an N run, a long homopolymer, lower case */
    if (n > 8) {
        memset(seq + off[3] + 100, 'N', 7);
        if (len[5] > 1000) memset(seq + off[5] + 200, 'A', 400);
        for (i = 0; i < len[7]; ++i) seq[off[7] + i] |= 0x20;
    }
    double t0 = now();
    if (oatk_write_fasta(path, seq, off, len, n, 0, mode, 1, (uint64_t) 64 << 20, threads) != 0) { fprintf(stderr, "cannot write %s\n", path); return 2; }
    struct stat sb;
    stat(path, &sb);
    const double t_write = now() - t0;

    /* from here on under the limit (the file's text + the generator's copy of the reads are inside it: the reader's own share is what is left) */
    if (lim_gb > 0) {
        struct rlimit rl;
        rl.rlim_cur = rl.rlim_max = (rlim_t) (lim_gb * 1073741824.0);
        if (setrlimit(RLIMIT_AS, &rl) != 0) { perror("setrlimit"); return 2; }
    }
    oatk_host_set_arena(arena);
    oatk_hip_ctx *ctx = oatk_hip_create(0);
    oatk_sr_db_t *db = oatk_sr_db_new(1001, 31);
    char *files[1] = {(char *) path};
    t0 = now();
    const int rc = oatk_sr_read_files(ctx, db, files, 1);
    const double t_read = now() - t0;
    if (rc) { fprintf(stderr, "oatk_sr_read_files: rc %d (%s)\n", rc, oatk_hip_last_error(ctx)); return 1; }
    if (db->n != n) { fprintf(stderr, "reads: got %zu, wrote %lu\n", db->n, (unsigned long) n); return 1; }
    uint64_t bad = 0, n_scm = 0, n_nn = 0, n_lrl = 0;
    for (i = 0; i < n && bad < 5; ++i) {
        const oatk_sr_t *r = &db->a[i];
        char nm[32];
        snprintf(nm, sizeof nm, "r%lu", (unsigned long) i);
        if (r->sid != i || !r->sname || strcmp(r->sname, nm)) { fprintf(stderr, "read %lu: sid %lu name %s\n", (unsigned long) i, (unsigned long) r->sid, r->sname? r->sname : "(null)"); ++bad; continue; }
        const uint8_t *q = seq + off[i];
        uint64_t j = 0, h = 0, nn = 0, lrl = 0;
        int ok = 1;
        while (j < len[i] && ok) {
            const uint8_t ch = q[j] & 0xDF;
            const int cd = ch == 'A'? 0 : ch == 'C'? 1 : ch == 'G'? 2 : ch == 'T'? 3 : 4;
            uint64_t run = 1;
            if (cd < 4) while (j + run < len[i] && (q[j + run] & 0xDF) == ch) ++run;
            if (h >= r->hoco_l) { ok = 0; break; }
            const int got = (r->hoco_s[h >> 2] >> (((h & 3) ^ 3) << 1)) & 3;
            if (got != (cd & 3) || r->ho_rl[h] != (uint8_t) ((run > 256? 256 : run) - 1)) ok = 0;
            if (cd == 4) { if (!r->n_nucl || r->n_nucl[nn] != (uint32_t) j) ok = 0; ++nn; }
            if (run > 255) { if (!r->ho_l_rl || r->ho_l_rl[lrl] != (uint32_t) (run - 1)) ok = 0; ++lrl; }
            ++h, j += run;
        }
        if (!ok || h != r->hoco_l) { fprintf(stderr, "read %lu: compressed bases differ (at %lu of %u)\n", (unsigned long) i, (unsigned long) h, r->hoco_l); ++bad; continue; }
        if (r->n != r->hoco_l / 512) { fprintf(stderr, "read %lu: %u syncmers, the stub made %u\n", (unsigned long) i, r->n, r->hoco_l / 512); ++bad; continue; }
        for (j = 0; j < r->n; ++j) {
            const uint64_t x = mix((i << 20) ^ j);
            if (r->m_pos[j] != (uint32_t) ((j * 512) << 1 | (x & 1)) || r->s_mer[j] != x >> 3 || r->k_mer[j] != mix(x)) { fprintf(stderr, "read %lu: syncmer %lu differs\n", (unsigned long) i, (unsigned long) j); ++bad; break; }
        }
        n_scm += r->n, n_nn += nn, n_lrl += lrl;
    }
    printf("{\"reads\": %lu, \"text_bytes\": %lu, \"file_bytes\": %lu, \"write_s\": %.2f, \"read_s\": %.2f, \"syncmers\": %lu, \"n_bases\": %lu, \"long_runs\": %lu, "
           "\"device_peak_bytes\": %lu, \"device_biggest_request\": %lu, \"host_peak_rss_kb\": %ld, \"sr_db_m\": %zu, \"bad\": %lu}\n",
           (unsigned long) n, (unsigned long) total, (unsigned long) sb.st_size, t_write, t_read, (unsigned long) n_scm, (unsigned long) n_nn, (unsigned long) n_lrl,
           (unsigned long) stub_device_peak_bytes(), (unsigned long) stub_device_biggest_request(), peak_rss_kb(), db->m, (unsigned long) bad);
    return bad? 1 : 0;
}
