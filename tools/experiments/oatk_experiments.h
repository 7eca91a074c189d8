/* tools/experiments/oatk_experiments.h -- entry points of liboatk_hip_experiments.so (tools/experiments/build.sh): liboatk_hip.so's sources compiled with
 * -DOATK_EXPERIMENTS, which adds what was built to be measured against the product path and lost.  Not part of the C ABI of include/. */
#ifndef OATK_EXPERIMENTS_H
#define OATK_EXPERIMENTS_H
#include "../../include/oatk_hip_ec.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The A/B SURVEY.md 7-5 asks for ("benchmark both"): the same jobs -- whole query, no resumption -- through the wavefront routine above (myers = 0) or
 * through Myers' bit-vector algorithm with one lane per pair and the bit-vectors over the target (myers = 1; north_star's "bit-parallel (Myers) kernel
 * batching reads per wavefront").  out3 holds one (score, t_end, q_end) per JOB -- both give wf_ed's values --, *kernel_ms the duration of the kernel.
 * The correction uses the wavefront routine: its search resumes, saves and restores alignments. */
int oatk_hip_debug_ed_ab(oatk_hip_ctx *ctx, int myers, uint64_t n_jobs, const uint8_t *t_codes, const uint64_t *t_off, const uint8_t *q_codes, const uint64_t *q_off,
                         const int32_t *bw, int32_t *out3, float *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif
