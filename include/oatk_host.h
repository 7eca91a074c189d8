/*
 * include/oatk_host.h -- host-side C API of the MI355X build (liboatk_host.so, plain C99, no HIP types).
 *
 * Part 1 (this file, top): the synthetic HiFi workload generator used by bench.py and the tests.
 * Part 2: the host mirror of the reference's hot-path symbols (sr_read / collect_syncmer_from_reads /
 *         read_error_correction over sr_db_t / syncmer_db_t), see oatk_syncasm.h.
 */
#ifndef OATK_HOST_H
#define OATK_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* genome seed 1001, reads seed 31, err 0.05 % = 500 ppm */
typedef struct {
    uint64_t genome_len, n_reads, genome_seed, reads_seed;
    uint64_t mean_len;     /* L; sd = L/10; clipped to [2000, min(G, 2L)] */
    uint64_t err_ppm;      /* total error rate in parts per million, split sub/ins/del equally */
} oatk_synth_t;

/* genome[i] in {0,1,2,3} */
void oatk_synth_genome(const oatk_synth_t *p, uint8_t *genome);
/* len[i] of reads [first, first+count): cheap, lets the caller lay out the packed stream first */
void oatk_synth_lengths(const oatk_synth_t *p, uint64_t first, uint64_t count, uint32_t *len);
/* ASCII bases of reads [first, first+count) at seq + off[i] (off relative to this slice) */
void oatk_synth_reads(const oatk_synth_t *p, const uint8_t *genome, uint64_t first, uint64_t count, const uint64_t *off,
                      uint8_t *seq, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
