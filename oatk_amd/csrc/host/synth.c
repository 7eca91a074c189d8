/*
 * oatk_amd/csrc/host/synth.c -- deterministic synthetic HiFi reads for bench.py and the tests.
 *
 * The workload generator SURVEY.md 8(d) / BASELINE.md describe: a uniform-random circular genome and
 * N reads sampled from it (length ~ N(L, 0.1 L) clipped to [2000, min(G, 2L)], uniform start, either
 * strand with p = 1/2, 0.05 % errors split equally into substitutions, 1-base insertions and 1-base
 * deletions, upper-case ACGT).  Integer-only and counter-based: read i depends only on
 * (reads_seed, i), so any slice of the read set can be regenerated anywhere (GPU box host cores,
 * several ranks) and always comes out identical.  Reads are written straight into the packed read
 * stream of include/oatk_hip.h (64-byte aligned starts).
 *
 * Own code; PRNG = splitmix64 (public-domain constants).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_host.h"

static inline uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* genome block b = 32 bases from the b-th splitmix64 output of genome_seed */
void oatk_synth_genome(const oatk_synth_t *p, uint8_t *genome)
{
    uint64_t i, nblk = (p->genome_len + 31) / 32;
    for (i = 0; i < nblk; ++i) {
        uint64_t st = p->genome_seed + i * 0x9E3779B97F4A7C15ULL, v = sm64(&st);
        uint64_t j, e = (i + 1) * 32 < p->genome_len? 32 : p->genome_len - i * 32;
        for (j = 0; j < e; ++j) genome[i * 32 + j] = (uint8_t) ((v >> (2 * j)) & 3);
    }
}

static inline uint64_t read_state(const oatk_synth_t *p, uint64_t i)
{
    return p->reads_seed ^ ((i + 1) * 0xD1342543DE82EF95ULL);
}

/* length draw: Irwin-Hall(12) approximation of the normal, all integer */
static inline uint32_t draw_len(const oatk_synth_t *p, uint64_t *st)
{
    uint64_t a = sm64(st), b = sm64(st), c = sm64(st);
    int64_t sum = 0;
    int t;
    for (t = 0; t < 4; ++t) sum += (int64_t) ((a >> (16 * t)) & 0xFFFF) + (int64_t) ((b >> (16 * t)) & 0xFFFF) + (int64_t) ((c >> (16 * t)) & 0xFFFF);
    int64_t z = sum - 6 * 65535;                              /* ~ N(0, 65536^2) */
    int64_t sd = (int64_t) (p->mean_len / 10);
    int64_t len = (int64_t) p->mean_len + (sd * z + (z >= 0? 32768 : -32768)) / 65536;
    int64_t lo = 2000, hi = (int64_t) (p->genome_len < 2 * p->mean_len? p->genome_len : 2 * p->mean_len);
    if (lo > hi) lo = hi;
    if (len < lo) len = lo;
    if (len > hi) len = hi;
    return (uint32_t) len;
}

void oatk_synth_lengths(const oatk_synth_t *p, uint64_t first, uint64_t count, uint32_t *len)
{
    uint64_t i;
    for (i = 0; i < count; ++i) {
        uint64_t st = read_state(p, first + i);
        len[i] = draw_len(p, &st);
    }
}

/* L bases sampled from the circular genome g[0 .. G) from `start` on, on either strand, with errors at err_ppm; `st` is the read's own stream */
static void sample_read(const uint8_t *g, uint64_t G, uint64_t *st, uint32_t L, uint64_t start, uint64_t strand, uint64_t err_ppm, uint8_t *out)
{
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint32_t n = 0;
    uint64_t pos = start;                                      /* template cursor; walks backwards on the reverse strand */
    uint64_t thr = (uint64_t) ((double) err_ppm * 18446744073709.551616);   /* err_ppm * 2^64 / 1e6 */
#define ADVANCE() do { if (strand) pos = pos? pos - 1 : G - 1; else pos = pos + 1 == G? 0 : pos + 1; } while (0)
    while (n < L) {
        uint8_t tb = strand? (uint8_t) (3 ^ g[pos]) : g[pos];
        uint64_t r = sm64(st);
        if (r < thr) {
            uint64_t kind = r % 3, r2 = sm64(st);
            if (kind == 0) { out[n++] = (uint8_t) ACGT[(tb + 1 + r2 % 3) & 3]; ADVANCE(); }  /* substitution */
            else if (kind == 1) { out[n++] = (uint8_t) ACGT[r2 & 3]; }                         /* insertion (template not consumed) */
            else { ADVANCE(); }                                                                 /* deletion */
        } else {
            out[n++] = (uint8_t) ACGT[tb];
            ADVANCE();
        }
    }
#undef ADVANCE
}

static void one_read(const oatk_synth_t *p, const uint8_t *g, uint64_t idx, uint8_t *out)
{
    uint64_t st = read_state(p, idx);
    uint32_t L = draw_len(p, &st);
    uint64_t G = p->genome_len, start = sm64(&st) % G, strand = sm64(&st) & 1;
    sample_read(g, G, &st, L, start, strand, p->err_ppm, out);
}

typedef struct {
    const oatk_synth_t *p;
    const uint8_t *g;
    uint64_t first, count;
    const uint64_t *off;
    uint8_t *seq;
    int tid, nthr;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *) arg;
    uint64_t i;
    for (i = (uint64_t) j->tid; i < j->count; i += (uint64_t) j->nthr) one_read(j->p, j->g, j->first + i, j->seq + j->off[i]);
    return 0;
}

/* reads [first, first+count) written at seq + off[i]; genome = 2-bit codes from oatk_synth_genome */
void oatk_synth_reads(const oatk_synth_t *p, const uint8_t *genome, uint64_t first, uint64_t count, const uint64_t *off,
                      uint8_t *seq, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    job_t jb[256];
    int t;
    for (t = 0; t < n_threads; ++t) {
        jb[t].p = p, jb[t].g = genome, jb[t].first = first, jb[t].count = count, jb[t].off = off, jb[t].seq = seq;
        jb[t].tid = t, jb[t].nthr = n_threads;
        if (t > 0) pthread_create(&th[t], 0, worker, &jb[t]);
    }
    worker(&jb[0]);
    for (t = 1; t < n_threads; ++t) pthread_join(th[t], 0);
}

/* ---- the mixture (include/oatk_host.h: oatk_synth_mix_t) ---- */
typedef struct { uint32_t comp, len; int is_short; } mix_head_t;

/* component and length of read idx; leaves *st behind the draws both functions below share */
static mix_head_t mix_head(const oatk_synth_mix_t *p, uint64_t idx, uint64_t *st)
{
    mix_head_t h;
    *st = p->reads_seed ^ ((idx + 1) * 0xD1342543DE82EF95ULL);
    const uint64_t u = sm64(st) % 1000000, v = sm64(st);
    uint32_t c = 0;
    while (c + 1 < p->n_comp && u >= p->cum_ppm[c]) ++c;
    h.comp = c;
    h.is_short = (v % 1000000) < p->short_ppm;
    if (h.is_short) {
        const uint64_t hi = p->short_max < 30? 30 : p->short_max;
        h.len = (uint32_t) (30 + (v >> 20) % (hi - 30 + 1));
        (void) sm64(st); (void) sm64(st); (void) sm64(st);      /* keep the stream position independent of the kind */
    } else {
        oatk_synth_t q;
        q.genome_len = p->genome_len[c], q.mean_len = p->mean_len;
        h.len = draw_len(&q, st);
    }
    return h;
}

void oatk_synth_mix_lengths(const oatk_synth_mix_t *p, uint64_t first, uint64_t count, uint32_t *len)
{
    uint64_t i, st;
    for (i = 0; i < count; ++i) len[i] = mix_head(p, first + i, &st).len;
}

static void mix_one(const oatk_synth_mix_t *p, uint64_t idx, uint8_t *out)
{
    uint64_t st;
    const mix_head_t h = mix_head(p, idx, &st);
    const uint64_t G = p->genome_len[h.comp], start = sm64(&st) % G, strand = sm64(&st) & 1, flags = sm64(&st);
    sample_read(p->genome[h.comp], G, &st, h.len, start, strand, p->err_ppm, out);
    if ((flags % 1000000) < p->n_ppm) {                        /* runs of N over the sampled bases */
        uint64_t runs = 1 + (flags >> 20) % 3, r;
        for (r = 0; r < runs; ++r) {
            const uint64_t a = sm64(&st) % h.len, l = 1 + sm64(&st) % 40;
            uint64_t j;
            for (j = a; j < a + l && j < h.len; ++j) out[j] = 'N';
        }
    }
    if (((flags >> 24) % 1000000) < p->lower_ppm) {
        uint32_t j;
        for (j = 0; j < h.len; ++j) out[j] |= 0x20;
    }
}

typedef struct { const oatk_synth_mix_t *p; uint64_t first, count; const uint64_t *off; uint8_t *seq; int tid, nthr; } mix_job_t;

static void *mix_worker(void *arg)
{
    mix_job_t *j = (mix_job_t *) arg;
    uint64_t i;
    for (i = (uint64_t) j->tid; i < j->count; i += (uint64_t) j->nthr) mix_one(j->p, j->first + i, j->seq + j->off[i]);
    return 0;
}

void oatk_synth_mix_reads(const oatk_synth_mix_t *p, uint64_t first, uint64_t count, const uint64_t *off, uint8_t *seq, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    mix_job_t jb[256];
    int t, started[256];
    for (t = 0; t < n_threads; ++t) {
        jb[t].p = p, jb[t].first = first, jb[t].count = count, jb[t].off = off, jb[t].seq = seq, jb[t].tid = t, jb[t].nthr = n_threads;
        started[t] = t > 0 && pthread_create(&th[t], 0, mix_worker, &jb[t]) == 0;
    }
    mix_worker(&jb[0]);
    for (t = 1; t < n_threads; ++t) { if (started[t]) pthread_join(th[t], 0); else mix_worker(&jb[t]); }
}
