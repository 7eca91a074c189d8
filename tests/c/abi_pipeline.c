/*
 * tests/c/abi_pipeline.c -- the C ABI used from plain C, the way a maintainer of the reference would (no Python, no torch): synthetic reads
 * -> scan -> count -> sr_db_stat -> EC graph + error correction -> consensus sums -> pair tables -> assembly graph, every step a call of
 * include/oatk_hip*.h on one context.  Prints one line of figures; tests/test_gpu_c_abi.py builds it with gcc, runs it and checks the
 * figures against the same pipeline driven through ctypes.
 *
 *   gcc -O2 -Iinclude tests/c/abi_pipeline.c -Loatk_amd/lib -loatk_host -loatk_hip -Wl,-rpath,oatk_amd/lib -o abi_pipeline
 *   ./abi_pipeline <n_reads> <genome_len> <mean_len> <k> <s> <min_k_cov>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip.h"
#include "oatk_hip_ec.h"
#include "oatk_hip_cons.h"
#include "oatk_hip_graph.h"
#include "oatk_hip_stat.h"
#include "oatk_host.h"

#define DIE(ctx, what, rc) do { fprintf(stderr, "%s failed (%d): %s\n", what, rc, (ctx)? oatk_hip_last_error(ctx) : ""); return 2; } while (0)
#define CALL(ctx, expr) do { int rc_ = (expr); if (rc_) DIE(ctx, #expr, rc_); } while (0)

static uint64_t fnv(const void *p, size_t n)                      /* a checksum of a fetched buffer */
{
    const uint8_t *b = (const uint8_t *) p;
    uint64_t h = 1469598103934665603ULL;
    size_t i;
    for (i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ULL;
    return h;
}
static int fetch_sum(oatk_hip_ctx *ctx, int which, uint64_t *sum, uint64_t *bytes)
{
    const void *d = 0;
    int rc = oatk_hip_buffer(ctx, which, &d, bytes);
    if (rc) return rc;
    void *h = malloc(*bytes? *bytes : 1);
    rc = oatk_hip_d2h(ctx, h, d, *bytes);
    if (!rc) *sum = fnv(h, *bytes);
    free(h);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s n_reads genome_len mean_len k s min_k_cov\n", argv[0]); return 1; }
    const uint64_t n = strtoull(argv[1], 0, 10), G = strtoull(argv[2], 0, 10), L = strtoull(argv[3], 0, 10);
    const int k = atoi(argv[4]), s = atoi(argv[5]);
    const uint32_t c = (uint32_t) atoi(argv[6]);
    if (oatk_hip_device_count() < 1) { fprintf(stderr, "no HIP device: this path has no CPU fallback\n"); return 3; }

    /* synthetic reads, packed the way sr_read would hand them over: ASCII, each read on a 64-byte boundary */
    oatk_synth_t p = {G, n, 1001, 31, L, 500};
    uint8_t *genome = (uint8_t *) malloc(G);
    uint32_t *len = (uint32_t *) malloc(4 * n);
    uint64_t *off = (uint64_t *) malloc(8 * n), total = 0, i;
    oatk_synth_genome(&p, genome);
    oatk_synth_lengths(&p, 0, n, len);
    for (i = 0; i < n; ++i) off[i] = total, total += ((uint64_t) len[i] + OATK_READ_ALIGN - 1) / OATK_READ_ALIGN * OATK_READ_ALIGN;
    uint8_t *seq = (uint8_t *) calloc(total? total : 1, 1);
    oatk_synth_reads(&p, genome, 0, n, off, seq, 4);

    oatk_hip_ctx *ctx = oatk_hip_create(0);
    if (!ctx) { fprintf(stderr, "oatk_hip_create failed\n"); return 3; }
    CALL(ctx, oatk_hip_scan_host(ctx, seq, off, len, n, total, 0, k, s));               /* sr_read */
    CALL(ctx, oatk_hip_count(ctx));                                                      /* collect_syncmer_from_reads */
    oatk_hip_info_t info;
    CALL(ctx, oatk_hip_info(ctx, &info));
    static oatk_stat_raw_t st;
    CALL(ctx, oatk_hip_stat(ctx, &st));                                                  /* sr_db_stat */
    CALL(ctx, oatk_hip_ec_graph(ctx));                                                   /* make_syncmer_graph(0, 0.) + scg_consensus(hoco) */
    CALL(ctx, oatk_hip_ec(ctx, 0, 0.02, c, 10 * c, c, 0.35));                            /* read_error_correction */
    uint64_t ecs[12];
    CALL(ctx, oatk_hip_ec_stats(ctx, ecs));
    CALL(ctx, oatk_hip_consensus(ctx, c));                                               /* scg_syncmer_consensus' sums */
    uint64_t n_pairs = 0, n_entries = 0, n_vtx = 0, n_arc = 0;
    CALL(ctx, oatk_hip_overlap_hist(ctx, &n_pairs, &n_entries));                         /* calc_syncmer_overlap's tables */
    CALL(ctx, oatk_hip_asm_graph(ctx, c, 0.35, &n_vtx, &n_arc));                         /* make_syncmer_graph(c, a) */
    uint64_t s_kmer = 0, s_arc = 0, s_rl = 0, b = 0;
    CALL(ctx, fetch_sum(ctx, OATK_BUF_EC_KMER, &s_kmer, &b));
    CALL(ctx, fetch_sum(ctx, OATK_BUF_AG_ARC_W, &s_arc, &b));
    CALL(ctx, fetch_sum(ctx, OATK_BUF_CONS_RL, &s_rl, &b));
    printf("n_occ=%llu n_scm=%llu kmer_unique=%llu sum_dist=%lld blocks=%llu corrected=%llu pairs=%llu entries=%llu n_vtx=%llu n_arc=%llu "
           "ec_kmer=%016llx ag_arc_w=%016llx cons_rl=%016llx\n",
           (unsigned long long) info.n_occ, (unsigned long long) info.n_scm, (unsigned long long) st.kmer_unique, (long long) st.sum_dist,
           (unsigned long long) (ecs[0] + ecs[5] + ecs[10]), (unsigned long long) (ecs[2] + ecs[7]), (unsigned long long) n_pairs,
           (unsigned long long) n_entries, (unsigned long long) n_vtx, (unsigned long long) n_arc, (unsigned long long) s_kmer,
           (unsigned long long) s_arc, (unsigned long long) s_rl);
    oatk_hip_destroy(ctx);
    free(genome); free(len); free(off); free(seq);
    return 0;
}
