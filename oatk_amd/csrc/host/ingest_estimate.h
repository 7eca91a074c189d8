/*
 * oatk_amd/csrc/host/ingest_estimate.h -- how much room the streamed reader (ingest_host.c) asks for ahead of the reads: plain arithmetic, kept apart so that it can be
 * tested without a device (tests/c/ingest_estimate_test.c).  The reference grows its reads' array by doubling as it reads (syncmer.c:487-556, kvec); this build sizes it
 * ONCE from the first window -- reads per compressed (or plain) input byte, times the bytes that are left -- because the array is a gigabyte at 2 M reads and growing it
 * window by window copied it several times.  An estimate from a source that cannot say where it stands must not be made (round 5: a position that stood still made it a
 * ten-million-fold over-estimate, and zeroing what realloc had promised took the machine down).
 */
#ifndef OATK_INGEST_ESTIMATE_H
#define OATK_INGEST_ESTIMATE_H
#include <stdint.h>

/* can "this window took input bytes [in0, in1) and held `text` bytes of text" be extrapolated from?  A source that does not move cannot; nor one that claims more than
 * forty bytes of text per input byte (no FASTA / FASTQ deflates like that: 2 bits a base is four, long runs of one base are what is left) */
static inline int oatk_est_trust(uint64_t in0, uint64_t in1, uint64_t text)
{
    return in1 > in0 && (double) (in1 - in0) * 40.0 >= (double) text;
}

/* room (in reads) for the reads' array when it has to grow to hold n_done + n: the extrapolation if one can be made, twice the array otherwise; never more than
 * `max_new` NEW entries beyond what is needed now (a quarter of the machine's memory: the caller's business), never less than n_done + n */
static inline uint64_t oatk_est_reads(uint64_t n_done, uint64_t n, uint64_t have_m, uint64_t in0, uint64_t in1, uint64_t in_total, int trust, int last, uint64_t max_new)
{
    const uint64_t need = n_done + n;
    uint64_t m = need;
    if (!last && trust) {
        const double left = in_total > in0? (double) (in_total - in0) : (double) (in1 - in0);
        const double e = (double) n * (left / (double) (in1 - in0)) * 1.05 + 1024.0;
        m = e < 1.8e19? n_done + (uint64_t) e : UINT64_MAX;
        if (m < need) m = need;                      /* (overflow, or an estimate below what is already here) */
    } else if (!last && m < 2 * have_m) m = 2 * have_m;
    if (m > need && m > (1ULL << 36)) m = need;      /* (seventy billion reads: whatever said so is wrong, and the array's size in bytes must not wrap) */
    if (max_new && m > need && m - have_m > max_new) {
        m = need > 2 * have_m? need : 2 * have_m;
        if (m - have_m > max_new) m = need;
    }
    return m;
}

/* the factor by which a handle's first piece (seq bytes, reads, occurrences) is multiplied to reserve the handle's whole batch: the smaller of the handle's share of the
 * input and what is left of it, over what the piece took; one more piece where several handles share the input (a handle's part ends on a window boundary) */
static inline double oatk_est_scale(uint64_t in0, uint64_t in1, uint64_t in_total, int n_ctx)
{
    const double share = (double) in_total / (double) n_ctx, left = in_total > in0? (double) (in_total - in0) : (double) (in1 - in0);
    return (share < left? share : left) / (double) (in1 - in0) * 1.03 + (n_ctx > 1? 1.0 : 0.0);
}
#endif
