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

/* ---- a read set shaped like oatk's real input (BASELINE.json configs[0]: an organelle HiFi data set) ----
 * Reads are drawn from SEVERAL circular genomes at once -- organelle genomes at thousand-fold coverage beside a large nuclear background at a few
 * fold, so that most distinct syncmers sit below the coverage cutoff -- and a share of them is short (below K), carries runs of N, or is lower case.
 * Low-complexity sequence (homopolymers beyond 256, telomere / microsatellite arrays, inverted repeats) is put into the GENOMES by the caller
 * (oatk_amd/synth.py), so it recurs across reads.  Counter-based like the generator above: read i depends only on (reads_seed, i). */
#define OATK_SYNTH_MAX_COMP 8
typedef struct {
    uint64_t n_comp;
    const uint8_t *genome[OATK_SYNTH_MAX_COMP];   /* codes 0..3 */
    uint64_t genome_len[OATK_SYNTH_MAX_COMP];
    uint64_t cum_ppm[OATK_SYNTH_MAX_COMP];        /* read i comes from the first component c with (draw % 1e6) < cum_ppm[c] */
    uint64_t reads_seed, mean_len, err_ppm;
    uint64_t short_ppm, short_max;                /* this share of the reads has a length uniform in [30, short_max] */
    uint64_t n_ppm;                               /* this share carries 1-3 runs of N (1-40 bases each) */
    uint64_t lower_ppm;                           /* this share is written in lower case */
} oatk_synth_mix_t;
void oatk_synth_mix_lengths(const oatk_synth_mix_t *p, uint64_t first, uint64_t count, uint32_t *len);
void oatk_synth_mix_reads(const oatk_synth_mix_t *p, uint64_t first, uint64_t count, const uint64_t *off, uint8_t *seq, int n_threads);

/* ---- FASTA text of a packed read stream, written plain or gzip'ed (test and bench inputs; the reference reads .gz through zlib, sstream.c:50) ----
 * records ">r<first_id + i>\n<bases>\n".  mode: 0 plain text; 1 ONE gzip member (what `gzip file` writes; deflated in parallel blocks that are
 * joined into a single deflate stream, pigz's construction); 2 BGZF (members of <= 64 KiB with the BC extra field, what bgzip writes);
 * 3 several plain gzip members one after the other (what `cat a.gz b.gz` gives), `member_bytes` of text each.  Returns 0, or -1 on an I/O error. */
int oatk_write_fasta(const char *path, const uint8_t *seq, const uint64_t *off, const uint32_t *len, uint64_t n_reads, uint64_t first_id,
                     int mode, int level, uint64_t member_bytes, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
