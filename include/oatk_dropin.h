/*
 * include/oatk_dropin.h -- the link-level drop-in for syncasm: the reference's own hot-path SYMBOLS, served from the MI355X.
 *
 * oatk_amd/csrc/dropin/syncasm_dropin.c (built into oatk_amd/lib/liboatk_dropin.a) defines, with the reference's signatures,
 *
 *     sr_read                      syncmer.c:487      -> oatk_sr_read_files              (text -> records -> scan on the device)
 *     sr_db_stat                   syncmer.c:867      -> oatk_sr_db_stat
 *     collect_syncmer_from_reads   syncmer.c:1397     -> oatk_collect_syncmer_from_reads
 *     make_syncmer_graph           syncasm.c:203      -> oatk_hip_ec_graph (the (0, 0.) call) / oatk_make_syncmer_asmg
 *     read_error_correction        syncerr.c:819      -> oatk_read_error_correction
 *     scg_read_alignment           alignment.c:596    -> oatk_scg_read_alignment
 *     sr_destroy, sr_db_clean, sr_db_destroy, syncmer_db_clean, syncmer_db_destroy   syncmer.c:1047-1110   -> oatk_sr_destroy / oatk_sr_db_clean /
 *                                  oatk_syncmer_db_clean: owning these is what lets sr_read
 *                                  hand out the member arrays of a whole piece of reads as ONE block instead of seven malloc'ed blocks per read
 *                                  (include/oatk_syncasm.h: arenas; OATK_DROPIN_ARENA=0 keeps the reference's one-block-per-array layout)
 *
 * and the two hook pointers below, which scg_syncmer_consensus (syncasm.c:888) and calc_syncmer_overlap (:477) consult first
 * (INTEGRATION.md 3b / 3b').  A maintainer links it in front of the reference's objects after renaming the eleven original
 * definitions to orig_<name> (objcopy --redefine-sym; every caller lives in another translation unit, so the calls bind to the
 * new definitions and the original bodies stay reachable):  run_syncasm.c, the CLI and everything downstream are untouched.
 *
 * Every function falls back to its ORIGINAL body -- the maintainer's code, never a CPU restatement of ours -- when the device
 * path does not apply or refuses: no gfx950 device, OATK_DROPIN=0, k beyond oatk_hip_max_k(), text kseq itself would stop in (a quality string longer than its sequence),
 * OATK_E_SPLIT (duplicate arcs / oversized hash groups), reads beyond the aligner's per-read limits.  After a fallback that
 * changes the reads or the table on the host the device batch is stale and the later calls fall back as well.
 *
 * Environment: OATK_DROPIN=0 (original bodies only), OATK_DEVICE=<ordinal>, OATK_DROPIN_LOG=1 (one line per call with its
 * wall-clock and the path taken, a summary at exit).
 */
#ifndef OATK_DROPIN_H
#define OATK_DROPIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* returns the consensus length, or < 0: run the original body */
extern int64_t (*oatk_hook_cons)(void *sr_db, void *scm, int rev, int64_t beg, void *c_seq, int hoco_seq);
/* returns the number of distinct distances of m1 -> m2 (first-appearance order, counts, "the walk ended on a repeat"), or < 0: run
 * the original walk */
extern int (*oatk_hook_ovl)(void *m1, uint64_t rc1, void *m2, uint64_t rc2, const int32_t **dist, const uint32_t **cnt, int *tail);

/* calls served by the device / by the original bodies so far, per function, in the order of the list above, then the two hooks:
 * out16[2 i] device, out16[2 i + 1] original */
void oatk_dropin_counts(uint64_t *out16);

#ifdef __cplusplus
}
#endif
#endif
