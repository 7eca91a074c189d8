/*
 * include/oatk_multi.h -- the reference's hot-path entry points with the reads spread over SEVERAL MI355X of one node, from one host process.
 *
 * north_star: "Reads shard by record across the 8 GPUs of one node with an RCCL all-reduce over xGMI to merge per-GPU syncmer count tables before
 * graph construction"; SURVEY.md 8e: "Either yields identical syncmer_db_t ... graph construction stays on host rank 0".  This is the host half of
 * that hand-off: an oatk_multi owns one device handle per GPU (include/oatk_hip.h) and one communicator per handle (include/oatk_hip_multi.h); its
 * entry points mirror those of include/oatk_syncasm.h one for one and leave the SAME structs -- one sr_db_t, one syncmer_db_t with every syncmer's
 * occurrences in (sid, idx) order (syncmer.c:1353-1360, syncerr.c:796-805), one asmg_t, one consensus / distance table set, one scg_ra_v -- so the
 * reference's serial tail (unitigging, cleaning, unzipping, GFA) runs on them unchanged.  Results are bit-identical to one handle holding all reads
 * (tests/test_gpu_cli.py::test_cli_over_several_handles: both GFA files byte-identical with 2 and 4 handles).
 *
 * Who talks to whom: handle r holds the reads of the r-th part of the input (contiguous read ids).  Scan and count are local.  The collectives run on
 * one host thread per handle: over RCCL when every handle sits on its own device (librccl is loaded then, not before), over the in-process group
 * (device-to-device copies) when devices repeat -- which is how a single-GPU box runs 2 - 4 handles in the tests.
 */
#ifndef OATK_MULTI_H
#define OATK_MULTI_H

#include "oatk_hip_multi.h"
#include "oatk_syncasm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oatk_multi oatk_multi;

/* devices[0 .. n): the ordinal each handle is made on (1 <= n <= 64); NULL when a device is unusable or the communicators cannot be made */
oatk_multi *oatk_multi_create(const int *devices, int n);
void oatk_multi_destroy(oatk_multi *m);
int oatk_multi_size(const oatk_multi *m);
oatk_hip_ctx *oatk_multi_ctx(oatk_multi *m, int rank);
/* reads [first, first + n) of sr_db live in handle `rank` (valid after oatk_multi_sr_read_files) */
void oatk_multi_range(const oatk_multi *m, int rank, uint64_t *first, uint64_t *n);
const char *oatk_multi_backend(const oatk_multi *m);         /* "rccl" or "local" */
const char *oatk_multi_last_error(oatk_multi *m);

/* sr_read (syncmer.c:487): the files' text streams through the devices in windows, consecutive parts of the input to consecutive handles */
int oatk_multi_sr_read_files(oatk_multi *m, oatk_sr_db_t *sr_db, char **files, int n_files);
int oatk_multi_sr_read_files_capped(oatk_multi *m, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t m_data);      /* with sr_read's data cap (-D), 0 = none */
/* sr_db_stat (syncmer.c:867), after the read and after the correction */
int oatk_multi_sr_db_stat(oatk_multi *m, oatk_sr_db_t *sr_db, FILE *fo, int verbose);
/* collect_syncmer_from_reads (syncmer.c:1397): local counts, the table merge, the table gathered into the reference's struct */
oatk_syncmer_db_t *oatk_multi_collect_syncmer_from_reads(oatk_multi *m, oatk_sr_db_t *sr_db, int *rc);
/* read_error_correction (syncerr.c:819) with the EC graph built on the devices (the asmg == NULL form of oatk_read_error_correction) */
int oatk_multi_read_error_correction(oatk_multi *m, oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, double max_edist, uint32_t err_mer_c, uint32_t max_err_c,
                                     uint32_t err_arc_c, double max_arc_f, uint64_t *stats12);
/* make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) of the corrected reads (run_syncasm.c:138) */
oatk_asmg_t *oatk_multi_make_syncmer_asmg(oatk_multi *m, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f, int *rc);
/* the consensus sums and distance tables behind scg_consensus (include/oatk_syncasm.h: oatk_consensus_fetch / oatk_overlap_fetch); the tables cover
 * the pairs whose two syncmers are alive and seen >= min_cov times -- every pair of neighbours in the graph made with that min_k_cov */
oatk_consensus_t *oatk_multi_consensus_fetch(oatk_multi *m, uint32_t min_cov, int k, int *rc);
oatk_overlap_t *oatk_multi_overlap_fetch(oatk_multi *m, uint32_t min_cov, int *rc);
/* scg_read_alignment (alignment.c:596): every handle aligns its own reads against the same graph, no exchange */
int oatk_multi_scg_read_alignment(oatk_multi *m, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip, uint64_t *n_skipped);

#ifdef __cplusplus
}
#endif
#endif
