/*
 * include/oatk_hip_ingest.h -- C ABI of the device-side FASTA / FASTQ record scan (the reader loop of sr_read, syncmer.c:522-543,
 * over sstream_read -> kseq_read, sstream.c:83-103, kseq.h:192-235).
 *
 * The reference parses on one thread (kseq + strdup per read) and that bounds it: 0.4 Gbases/s whether it runs 8 or 128 analysis
 * threads.  Here the TEXT of the file goes to the device as it is (PCIe is ~100x faster than kseq) and the records are found there:
 * newline positions, header lines, per-record sequence lengths by prefix sums, and one copy pass that leaves the packed read stream
 * oatk_hip_scan takes (sequence bytes only, every read on a 64-byte boundary) in HBM -- the host never touches a base.
 *
 * Accepted text (kseq's reading of it, restricted to what sequencing files look like):
 *   FASTA   a header line starts with '>' (kseq also takes '@'); the sequence is every following line up to the next header, line
 *           breaks removed ("\n" or "\r\n"), empty lines skipped; text before the first header is ignored
 *   FASTQ   four lines per record: '@' header, sequence, '+' line, quality of the same length (the form every sequencer writes)
 *   KSEQ    anything kseq reads with headers at line starts: wrapped FASTQ, FASTA and FASTQ records in one stream (OATK_FMT_KSEQ below)
 * Read names are not kept on the device (sr_t.sname is only ever printed): OATK_BUF_INGEST_HDR gives the header-line offsets for a
 * caller that wants them.  gzip'ed input must be inflated by the caller (zlib is serial per stream, on any hardware).
 */
#ifndef OATK_HIP_INGEST_H
#define OATK_HIP_INGEST_H

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OATK_FMT_AUTO 0           /* by the first non-blank character: '>' FASTA, '@' FASTQ */
#define OATK_FMT_FASTA 1
#define OATK_FMT_FASTQ 2
#define OATK_FMT_KSEQ 3           /* kseq_read's own reading, line by line (kseq.h:192-235): every record FASTA or FASTQ by whether a '+' line follows its
                                   * sequence, sequence AND quality over any number of lines (a quality line may start with '@').  The two formats above
                                   * are what sequencing files look like and run entirely on the device; they answer OATK_E_SPLIT for text that needs
                                   * this one (FASTQ that is not four lines per record; a '+' line in FASTA text).  Here the device extracts three numbers per
                                   * line and the classification -- inherently serial: what a line is depends on the lines before it -- is a walk over them on
                                   * the host; lengths, offsets and the copy stay on the device.  Headers must begin a line. */

/* Parse n_bytes of text resident in device memory.  final = 0: the text is a chunk of a longer file -- only records that are certainly
 * complete are taken and *consumed tells how many bytes they span (feed the rest again in front of the next chunk); final = 1: the
 * text ends the file.  The packed stream stays resident (OATK_BUF_INGEST_*) until the next ingest. */
int oatk_hip_ingest(oatk_hip_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, int format, int final, uint64_t *n_reads, uint64_t *consumed);
/* the same from host memory (one hipMemcpy first) */
int oatk_hip_ingest_host(oatk_hip_ctx *ctx, const uint8_t *h_text, uint64_t n_bytes, int format, int final, uint64_t *n_reads, uint64_t *consumed);
/* sr_read's data cap (-D; syncmer.c:537-541: the read that takes the total to the cap is the last one): keep only the first n_keep reads of the
 * resident packed stream */
int oatk_hip_ingest_truncate(oatk_hip_ctx *ctx, uint64_t n_keep);
/* device memory owned by the context for n_bytes of text (valid until the next call or the next oatk_hip_ingest_host): a caller that uploads the
 * text itself -- in pieces, through page-locked memory, while it is still reading the file -- fills it with oatk_hip_h2d_async and then calls
 * oatk_hip_ingest on it */
int oatk_hip_ingest_text_buffer(oatk_hip_ctx *ctx, uint64_t n_bytes, uint8_t **d_text);
/* oatk_hip_scan on the resident packed stream: reads are numbered sid0, sid0 + 1, ... in file order (syncmer.c:525) */
int oatk_hip_scan_ingested(oatk_hip_ctx *ctx, uint64_t sid0, int k, int s);

/* INGEST_SEQ u8[seq_bytes]  INGEST_OFF u64[n_reads]  INGEST_LEN u32[n_reads]  INGEST_HDR u64[n_reads] byte offset of each header line */
enum { OATK_BUF_INGEST_SEQ = 160, OATK_BUF_INGEST_OFF, OATK_BUF_INGEST_LEN, OATK_BUF_INGEST_HDR };

#ifdef __cplusplus
}
#endif
#endif
