/*
 * tests/c/stub_device.c -- a STUB of liboatk_hip.so for the host reader (oatk_amd/csrc/host/ingest_host.c, srdb.c, gzsrc.c, gzpar.c), CPU only.
 *
 * Test infrastructure, never shipped: tests/test_host_reader_stub.py links the host library's sources against THIS instead of the HIP library and
 * runs oatk_sr_read_files end to end on a >= 100 MB single-member .fa.gz under RLIMIT_AS, so that a reader that over-estimates, over-allocates or
 * touches what realloc merely promised dies in the authoring container and not on a GPU box (exactly that cost round 5 three boxes).
 *
 * "Device memory" is malloc'ed and COUNTED (stub_device_peak_bytes); a request beyond STUB_HBM_BYTES (288 GB, an MI355X) fails like hipMalloc would.
 * The record scan is a plain FASTA reader (one or many lines per record, no '\r', headers begin a line -- what the test writes); the "syncmer scan"
 * does real homopolymer compression (syncmer.c:284-323: 2-bit codes MSB first, run lengths min(rl, 256) - 1, long runs and N lists) and FABRICATES
 * syncmers -- one per 512 compressed bases, fields a function of (sid, index) -- so that every array the reader moves is checked for content.
 * Layouts follow include/oatk_hip.h (oatk_hip_buffer) as srdb.c consumes them.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip.h"
#include "oatk_hip_ingest.h"

#define STUB_HBM_BYTES ((uint64_t) 288 << 30)

static uint64_t g_cur, g_peak, g_biggest;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

typedef struct { uint8_t *p; uint64_t cap; } buf_t;

static int ensure(buf_t *b, uint64_t bytes)
{
    if (bytes <= b->cap) return 1;
    pthread_mutex_lock(&g_mu);
    const uint64_t would = g_cur - b->cap + bytes;
    if (bytes > g_biggest) g_biggest = bytes;
    if (would > STUB_HBM_BYTES) { pthread_mutex_unlock(&g_mu); fprintf(stderr, "[stub device] request of %.1f GB refused: the device has 288 GB\n", (double) bytes / 1e9); return 0; }
    g_cur = would;
    if (g_cur > g_peak) g_peak = g_cur;
    pthread_mutex_unlock(&g_mu);
    free(b->p);
    b->p = (uint8_t *) malloc(bytes? bytes : 1);        /* (not touched: like VRAM, it does not count against the host) */
    b->cap = bytes;
    if (!b->p) { fprintf(stderr, "[stub device] malloc of %.1f GB failed\n", (double) bytes / 1e9); b->cap = 0; return 0; }
    return 1;
}

static void release(buf_t *b)
{
    pthread_mutex_lock(&g_mu);
    g_cur -= b->cap;
    pthread_mutex_unlock(&g_mu);
    free(b->p);
    b->p = 0, b->cap = 0;
}

uint64_t stub_device_peak_bytes(void) { return g_peak; }
uint64_t stub_device_biggest_request(void) { return g_biggest; }

enum { B_TEXT, B_STAGE, B_SEQ, B_OFF, B_LEN, B_HDR, B_HOCO_L, B_N_SCM, B_N_NN, B_N_LRL, B_HO_RL, B_HOCO_S, B_NN_KEY, B_LRL_KEY, B_LRL_VAL, B_SCM_OFF, B_MPOS, B_SMER, B_HASH, B_RESERVE, B_N_ };

struct oatk_hip_ctx {
    int dev;
    buf_t b[B_N_];
    uint64_t n_ing, seq_bytes;             /* the records of the last oatk_hip_ingest */
    int ing_done, scan_done;
    uint64_t n_nn, n_lrl, n_occ;           /* of the last scan */
    /* the batch assembled by scan_begin / scan_append (counts only: the stub keeps no batch) */
    uint64_t bat_reads, bat_seq, bat_occ, sid0;
    uint64_t res_seq, res_reads, res_occ;
    int k, s;
    const char *err;
};

int oatk_hip_abi_version(void) { return OATK_HIP_ABI_VERSION; }
int oatk_hip_device_count(void) { return 1; }
oatk_hip_ctx *oatk_hip_create(int device) { oatk_hip_ctx *c = (oatk_hip_ctx *) calloc(1, sizeof(*c)); if (c) c->dev = device, c->err = ""; return c; }
void oatk_hip_destroy(oatk_hip_ctx *c) { int i; if (!c) return; for (i = 0; i < B_N_; ++i) release(&c->b[i]); free(c); }
const char *oatk_hip_last_error(oatk_hip_ctx *c) { return c? c->err : "no handle"; }
void *oatk_hip_stream(oatk_hip_ctx *c) { (void) c; return 0; }
int oatk_hip_sync(oatk_hip_ctx *c) { (void) c; return OATK_OK; }
int oatk_hip_device(oatk_hip_ctx *c) { return c->dev; }
int oatk_hip_max_k(void) { return 1024; }
int oatk_hip_d2d(oatk_hip_ctx *c, void *d, const void *s, uint64_t n) { (void) c; memmove(d, s, n); return OATK_OK; }
int oatk_hip_d2h(oatk_hip_ctx *c, void *d, const void *s, uint64_t n) { (void) c; memcpy(d, s, n); return OATK_OK; }
int oatk_hip_d2h_async(oatk_hip_ctx *c, void *d, const void *s, uint64_t n) { (void) c; memcpy(d, s, n); return OATK_OK; }
int oatk_hip_h2d_async(oatk_hip_ctx *c, void *d, const void *s, uint64_t n) { (void) c; memcpy(d, s, n); return OATK_OK; }
int oatk_hip_host_register(oatk_hip_ctx *c, void *p, uint64_t n) { (void) c; (void) p; (void) n; return OATK_OK; }
int oatk_hip_host_unregister(oatk_hip_ctx *c, void *p) { (void) c; (void) p; return OATK_OK; }
int oatk_hip_set_timing(oatk_hip_ctx *c, int e) { (void) c; (void) e; return OATK_OK; }

void *oatk_hip_staging(oatk_hip_ctx *c, uint64_t bytes)
{
    /* page-locked HOST memory in the real library: here it is host memory too, and it is touched (it counts against RLIMIT_AS like the real one would against the box) */
    buf_t *b = &c->b[B_STAGE];
    if (bytes > b->cap) {
        free(b->p);
        b->p = (uint8_t *) malloc(bytes);
        b->cap = b->p? bytes : 0;
        if (!b->p) return 0;
    }
    return b->p;
}

int oatk_hip_ingest_text_buffer(oatk_hip_ctx *c, uint64_t n_bytes, uint8_t **d_text)
{
    if (!ensure(&c->b[B_TEXT], n_bytes + 64)) return OATK_E_NOMEM;
    *d_text = c->b[B_TEXT].p;
    return OATK_OK;
}

/* FASTA records of text[0, n): a record is finished by the next header line (or by the end of a final text) */
int oatk_hip_ingest(oatk_hip_ctx *c, const uint8_t *t, uint64_t n, int format, int final, uint64_t *n_reads, uint64_t *consumed)
{
    uint64_t i = 0, n_rec = 0, cap = 1024, used = 0, seq_bytes = 0;
    c->ing_done = c->scan_done = 0, c->n_ing = 0;
    if (n_reads) *n_reads = 0;
    if (consumed) *consumed = 0;
    if (format == OATK_FMT_FASTQ) { c->err = "the stub reads FASTA only"; return OATK_E_ARG; }
    uint64_t *hdr = (uint64_t *) malloc(8 * cap), *s0 = (uint64_t *) malloc(8 * cap), *s1 = (uint64_t *) malloc(8 * cap);      /* header offset, sequence text range */
    while (i < n && t[i] != '>') { while (i < n && t[i] != '\n') ++i; if (i < n) ++i; }       /* junk before the first header */
    used = i;
    while (i < n) {                                     /* at a '>' */
        const uint64_t h = i;
        const uint8_t *e = (const uint8_t *) memchr(t + i, '\n', n - i);
        if (!e) { if (!final) break; e = t + n; }
        uint64_t a = (uint64_t) (e - t) + (e < t + n), b = a;
        for (;;) {                                      /* sequence lines up to the next header line */
            if (b >= n) break;
            if (t[b] == '>') break;
            const uint8_t *f = (const uint8_t *) memchr(t + b, '\n', n - b);
            b = f? (uint64_t) (f - t) + 1 : n;
        }
        if (b >= n && !final) break;                    /* may go on in the next window */
        if (n_rec == cap) { cap *= 2; hdr = (uint64_t *) realloc(hdr, 8 * cap), s0 = (uint64_t *) realloc(s0, 8 * cap), s1 = (uint64_t *) realloc(s1, 8 * cap); }
        hdr[n_rec] = h, s0[n_rec] = a, s1[n_rec] = b, ++n_rec;
        i = used = b;
    }
    if (final) used = n;
    if (!ensure(&c->b[B_OFF], 8 * (n_rec + 1)) || !ensure(&c->b[B_LEN], 4 * (n_rec + 1)) || !ensure(&c->b[B_HDR], 8 * (n_rec + 1))) return OATK_E_NOMEM;
    uint64_t *off = (uint64_t *) c->b[B_OFF].p, *hd = (uint64_t *) c->b[B_HDR].p, r;
    uint32_t *len = (uint32_t *) c->b[B_LEN].p;
    for (r = 0; r < n_rec; ++r) {                       /* (line by line: memchr + memcpy, the text is hundreds of megabytes) */
        uint64_t l = 0, j = s0[r];
        while (j < s1[r]) { const uint8_t *f = (const uint8_t *) memchr(t + j, '\n', s1[r] - j); const uint64_t e = f? (uint64_t) (f - t) : s1[r]; l += e - j, j = e + 1; }
        off[r] = seq_bytes, len[r] = (uint32_t) l, hd[r] = hdr[r];
        seq_bytes += (l + 63) & ~63ULL;
    }
    if (!ensure(&c->b[B_SEQ], seq_bytes + 64)) return OATK_E_NOMEM;
    for (r = 0; r < n_rec; ++r) {
        uint8_t *d = c->b[B_SEQ].p + off[r];
        uint64_t j = s0[r];
        while (j < s1[r]) { const uint8_t *f = (const uint8_t *) memchr(t + j, '\n', s1[r] - j); const uint64_t e = f? (uint64_t) (f - t) : s1[r]; memcpy(d, t + j, e - j); d += e - j, j = e + 1; }
    }
    free(hdr); free(s0); free(s1);
    c->n_ing = n_rec, c->seq_bytes = seq_bytes, c->ing_done = 1;
    if (n_reads) *n_reads = n_rec;
    if (consumed) *consumed = used;
    return OATK_OK;
}

int oatk_hip_ingest_truncate(oatk_hip_ctx *c, uint64_t keep) { if (keep < c->n_ing) c->n_ing = keep; return OATK_OK; }

static inline int code_of(uint8_t ch)
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}

/* what the stub calls a syncmer of read `sid`: one per 512 compressed bases */
static inline uint64_t stub_mix(uint64_t x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ULL; x ^= x >> 29; return x; }

int oatk_hip_scan_ingested(oatk_hip_ctx *c, uint64_t sid0, int k, int s)
{
    if (!c->ing_done) return OATK_E_STATE;
    const uint64_t n = c->n_ing, *off = (const uint64_t *) c->b[B_OFF].p;
    const uint32_t *len = (const uint32_t *) c->b[B_LEN].p;
    const uint8_t *seq = c->b[B_SEQ].p;
    uint64_t r, n_nn = 0, n_lrl = 0, n_occ = 0, m_nn = 1024, m_lrl = 1024;
    c->k = k, c->s = s, c->sid0 = sid0;
    if (!ensure(&c->b[B_HOCO_L], 4 * (n + 1)) || !ensure(&c->b[B_N_SCM], 4 * (n + 1)) || !ensure(&c->b[B_N_NN], 4 * (n + 1)) || !ensure(&c->b[B_N_LRL], 4 * (n + 1)) ||
        !ensure(&c->b[B_SCM_OFF], 8 * (n + 2)) || !ensure(&c->b[B_HO_RL], c->seq_bytes + 64) || !ensure(&c->b[B_HOCO_S], c->seq_bytes / 4 + 64)) return OATK_E_NOMEM;
    uint32_t *hoco_l = (uint32_t *) c->b[B_HOCO_L].p, *nscm = (uint32_t *) c->b[B_N_SCM].p, *nnn = (uint32_t *) c->b[B_N_NN].p, *nlrl = (uint32_t *) c->b[B_N_LRL].p;
    uint64_t *scm_off = (uint64_t *) c->b[B_SCM_OFF].p;
    uint64_t *nn_key = (uint64_t *) malloc(8 * m_nn);
    uint32_t *lrl_val = (uint32_t *) malloc(4 * m_lrl);
    memset(c->b[B_HOCO_S].p, 0, c->seq_bytes / 4 + 64);
    for (r = 0; r < n; ++r) {
        const uint8_t *q = seq + off[r];
        uint8_t *rl = c->b[B_HO_RL].p + off[r], *hs = c->b[B_HOCO_S].p + off[r] / 4;
        uint64_t i = 0, h = 0;
        nnn[r] = nlrl[r] = 0;
        while (i < len[r]) {
            const int cd = code_of(q[i]);
            uint64_t run = 1;
            if (cd < 4) while (i + run < len[r] && code_of(q[i + run]) == cd) ++run;      /* (an ambiguous base is stored as A with run 1 and never merged, syncmer.c:316-320) */
            else {
                if (n_nn == m_nn) m_nn *= 2, nn_key = (uint64_t *) realloc(nn_key, 8 * m_nn);
                nn_key[n_nn++] = (r << 32) | i, ++nnn[r];
            }
            hs[h >> 2] |= (uint8_t) ((cd & 3) << (((h & 3) ^ 3) << 1));
            rl[h] = (uint8_t) ((run > 256? 256 : run) - 1);
            if (run > 255) {
                if (n_lrl == m_lrl) m_lrl *= 2, lrl_val = (uint32_t *) realloc(lrl_val, 4 * m_lrl);
                lrl_val[n_lrl++] = (uint32_t) (run - 1), ++nlrl[r];
            }
            ++h, i += run;
        }
        hoco_l[r] = (uint32_t) h;
        nscm[r] = (uint32_t) (h / 512);
        scm_off[r] = n_occ, n_occ += nscm[r];
    }
    scm_off[n] = n_occ;
    if (!ensure(&c->b[B_NN_KEY], 8 * (n_nn + 1)) || !ensure(&c->b[B_LRL_VAL], 4 * (n_lrl + 1)) || !ensure(&c->b[B_LRL_KEY], 8 * (n_lrl + 1)) ||
        !ensure(&c->b[B_MPOS], 4 * (n_occ + 1)) || !ensure(&c->b[B_SMER], 8 * (n_occ + 1)) || !ensure(&c->b[B_HASH], 8 * (n_occ + 1))) return OATK_E_NOMEM;
    memcpy(c->b[B_NN_KEY].p, nn_key, 8 * n_nn);
    memcpy(c->b[B_LRL_VAL].p, lrl_val, 4 * n_lrl);
    free(nn_key); free(lrl_val);
    for (r = 0; r < n; ++r) {
        uint32_t j;
        for (j = 0; j < nscm[r]; ++j) {
            const uint64_t o = scm_off[r] + j, x = stub_mix(((sid0 + r) << 20) ^ j);
            ((uint32_t *) c->b[B_MPOS].p)[o] = (uint32_t) ((j * 512) << 1 | (x & 1));
            ((uint64_t *) c->b[B_SMER].p)[o] = x >> 3;
            ((uint64_t *) c->b[B_HASH].p)[o] = stub_mix(x);
        }
    }
    c->n_nn = n_nn, c->n_lrl = n_lrl, c->n_occ = n_occ, c->scan_done = 1;
    return OATK_OK;
}

int oatk_hip_buffer(oatk_hip_ctx *c, int which, const void **d, uint64_t *bytes)
{
    const uint64_t n = c->n_ing;
    int b = -1;
    uint64_t sz = 0;
    switch (which) {
        case OATK_BUF_INGEST_SEQ: b = B_SEQ, sz = c->seq_bytes; break;
        case OATK_BUF_INGEST_OFF: b = B_OFF, sz = 8 * n; break;
        case OATK_BUF_INGEST_LEN: b = B_LEN, sz = 4 * n; break;
        case OATK_BUF_INGEST_HDR: b = B_HDR, sz = 8 * n; break;
        case OATK_BUF_HOCO_L: b = B_HOCO_L, sz = 4 * n; break;
        case OATK_BUF_N_SCM: b = B_N_SCM, sz = 4 * n; break;
        case OATK_BUF_N_NN: b = B_N_NN, sz = 4 * n; break;
        case OATK_BUF_N_LRL: b = B_N_LRL, sz = 4 * n; break;
        case OATK_BUF_HO_RL: b = B_HO_RL, sz = c->seq_bytes; break;
        case OATK_BUF_HOCO_S: b = B_HOCO_S, sz = c->seq_bytes / 4; break;
        case OATK_BUF_NN_KEY: b = B_NN_KEY, sz = 8 * c->n_nn; break;
        case OATK_BUF_LRL_KEY: b = B_LRL_KEY, sz = 8 * c->n_lrl; break;
        case OATK_BUF_LRL_VAL: b = B_LRL_VAL, sz = 4 * c->n_lrl; break;
        case OATK_BUF_SCM_OFF: b = B_SCM_OFF, sz = 8 * (n + 1); break;
        case OATK_BUF_POS_MPOS: b = B_MPOS, sz = 4 * c->n_occ; break;
        case OATK_BUF_POS_SMER: b = B_SMER, sz = 8 * c->n_occ; break;
        case OATK_BUF_POS_HASH: b = B_HASH, sz = 8 * c->n_occ; break;
        default: c->err = "the stub has no such buffer"; return OATK_E_ARG;
    }
    if (which < OATK_BUF_INGEST_SEQ && !c->scan_done) return OATK_E_STATE;
    if (!c->ing_done) return OATK_E_STATE;
    *d = c->b[b].p, *bytes = sz;
    return OATK_OK;
}

int oatk_hip_scan_begin(oatk_hip_ctx *c, uint64_t sid0, int k, int s) { c->sid0 = sid0, c->k = k, c->s = s, c->bat_reads = c->bat_seq = c->bat_occ = 0; return OATK_OK; }

int oatk_hip_scan_reserve(oatk_hip_ctx *c, uint64_t seq_bytes, uint64_t n_reads, uint64_t n_occ)
{
    /* what the real library allocates for a batch of that size: the text's bytes 1.25 times over and 28 bytes per occurrence, 16 per read (api.hip) */
    c->res_seq = seq_bytes, c->res_reads = n_reads, c->res_occ = n_occ;
    const unsigned __int128 want = (unsigned __int128) seq_bytes * 5 / 4 + (unsigned __int128) n_occ * 28 + (unsigned __int128) n_reads * 16;
    if (want > STUB_HBM_BYTES) { fprintf(stderr, "[stub device] oatk_hip_scan_reserve(%.3g bytes, %.3g reads, %.3g occurrences) refused: the device has 288 GB\n", (double) seq_bytes, (double) n_reads, (double) n_occ); return OATK_E_NOMEM; }
    return ensure(&c->b[B_RESERVE], (uint64_t) want)? OATK_OK : OATK_E_NOMEM;
}

int oatk_hip_scan_append(oatk_hip_ctx *c, oatk_hip_ctx *piece)
{
    if (!piece->scan_done) return OATK_E_STATE;
    c->bat_reads += piece->n_ing, c->bat_seq += piece->seq_bytes, c->bat_occ += piece->n_occ;
    return OATK_OK;
}

int oatk_hip_info(oatk_hip_ctx *c, oatk_hip_info_t *o)
{
    memset(o, 0, sizeof(*o));
    if (c->bat_reads || !c->scan_done) o->n_reads = c->bat_reads, o->seq_bytes = c->bat_seq, o->n_occ = c->bat_occ;
    else o->n_reads = c->n_ing, o->seq_bytes = c->seq_bytes, o->n_occ = c->n_occ, o->n_nn = c->n_nn, o->n_lrl = c->n_lrl;
    o->sid0 = c->sid0, o->k = c->k, o->s = c->s;
    return OATK_OK;
}
