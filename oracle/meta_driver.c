/*
 * oracle/meta_driver.c -- TEST INFRASTRUCTURE (oracle/README.md): what `oatk` does with syncasm()'s `meta` hand-off, as a program of its own.
 *
 * oatk.c:383-456 calls syncasm(..., scg_meta) -- ownership of scg, scm_db, sr_db, ra_db moves to the caller (run_syncasm.c:306-313) --, hands the
 * structures to pathfinder_minicircle, which re-aligns the reads and rebuilds the consensus on them (path_finder.c:811-819), and frees everything
 * with scg_meta_destroy (oatk.c:456, syncasm.c:87-92).  This driver does exactly that sequence, then calls syncasm() a SECOND time into the same
 * meta (run_syncasm.c:307: scg_meta_clean frees the first round's structures while the second round's are live), and writes what it sees to
 * <out>.meta.ra (every alignment) and <out>.meta.gfa (the rebuilt consensus).  It is linked twice by oracle/Makefile (`make ref_meta`): over the
 * reference's own objects, and over the drop-in (include/oatk_dropin.h) exactly like oracle/_ref/syncasm_dropin; tests/test_gpu_meta.py compares the
 * two programs' files byte for byte and runs the drop-in one under the allocator's checks (the members it hands out live in arenas,
 * include/oatk_syncasm.h -- every free() of one through the reference's own destroy functions would be an invalid free).
 *
 *     meta_driver <out> <k> <s> <min_k_cov> <threads> <file> [file ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "misc.h"
#include "syncasm.h"

int VERBOSE = 0;                                   /* every main() of the reference provides it (run_syncasm.c:39) */

int syncasm(char **file_in, int n_file, size_t m_data, int k, int s, int bubble_size, int tip_size, int min_k_cov, double min_a_cov_f,
            double weak_cross, int do_ec, int do_unzip, int n_threads, char *out, scg_meta_t *meta, int VERBOSE);      /* oatk.c:51-52 */

static FILE *out_file(const char *out, const char *suffix)
{
    char *p = (char *) malloc(strlen(out) + strlen(suffix) + 1);
    strcpy(p, out), strcat(p, suffix);
    FILE *f = fopen(p, "w");
    if (!f) { perror(p); exit(EXIT_FAILURE); }
    free(p);
    return f;
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: meta_driver <out> <k> <s> <min_k_cov> <threads> <file> [file ...]\n"); return 2; }
    char *out = argv[1];
    const int k = atoi(argv[2]), s = atoi(argv[3]), c = atoi(argv[4]), t = atoi(argv[5]);
    scg_meta_t *meta;
    int ret, round;
    sys_init();
    meta = (scg_meta_t *) calloc(1, sizeof(scg_meta_t));                          /* oatk.c:385 */
    for (round = 0; round < 2; ++round) {
        /* the defaults of run_syncasm.c:356-367; the second round drops the unzipping so the two rounds differ */
        ret = syncasm(argv + 6, argc - 6, 0, k, s, 100000, 10000, c, .35, 0.3, 1, round == 0? 3 : 0, t, out, meta, 0);
        if (ret) { fprintf(stderr, "[E::meta_driver] syncasm returned %d in round %d\n", ret, round); return 1; }
        if (!meta->scg || !meta->sr_db || !meta->scm_db || !meta->ra_db || meta->k != k || meta->s != s) { fprintf(stderr, "[E::meta_driver] meta not filled\n"); return 1; }
        /* path_finder.c:811-819 */
        asmg_clean_consensus(meta->scg->utg_asmg);
        scg_read_alignment(meta->sr_db, meta->ra_db, meta->scg, t, 0);
        FILE *fo = out_file(out, round == 0? ".meta.ra" : ".meta2.ra");
        scg_rv_print(meta->ra_db, fo);
        fclose(fo);
        fo = out_file(out, round == 0? ".meta.gfa" : ".meta2.gfa");
        scg_consensus(meta->sr_db, meta->scg, 0, 0, fo);
        fclose(fo);
    }
    scg_meta_destroy(meta);                                                        /* oatk.c:456 */
    fprintf(stderr, "[M::meta_driver] done\n");
    return 0;
}
