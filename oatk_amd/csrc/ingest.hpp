// oatk_amd/csrc/ingest.hpp -- FASTA / FASTQ record scan on the device (include/oatk_hip_ingest.h).
//
// Everything is a map or a prefix sum over bytes or over lines:
//   ing_count_nl / ing_fill_nl     newline positions, 4 KiB of text per workgroup (ballot + popcount, no atomics)
//   ing_line_kernel                per line: header?  sequence bytes (CR stripped)?
//   (exclusive scans)              header rank = record index; sequence bytes before each line
//   ing_record_kernel              per record: length, header offset
//   (exclusive scan)               64-byte aligned offsets of the packed stream
//   ing_copy_kernel                one wave per sequence line: bytes to their place in the packed stream
#pragma once
#include "common.hpp"

namespace oatk {

#define ING_BLOCK 4096

__global__ __launch_bounds__(256) void ing_count_nl_kernel(const uint8_t *text, uint64_t n, uint32_t *cnt)
{
    __shared__ uint32_t w[4];
    const uint64_t base = (uint64_t) blockIdx.x * ING_BLOCK;
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < ING_BLOCK; i += 256) c += base + i < n && text[base + i] == '\n';
    for (int d = 32; d; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(256) void ing_fill_nl_kernel(const uint8_t *text, uint64_t n, const uint64_t *blk_off, uint64_t *nl_pos)
{
    __shared__ uint32_t w[4];
    const uint64_t base = (uint64_t) blockIdx.x * ING_BLOCK;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t out = blk_off[blockIdx.x];
    for (uint32_t i0 = 0; i0 < ING_BLOCK; i0 += 256) {
        const uint64_t p = base + i0 + threadIdx.x;
        const bool is = p < n && text[p] == '\n';
        const uint64_t m = __ballot(is);
        if (lane == 0) w[wid] = (uint32_t) __builtin_popcountll(m);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t j = 0; j < wid; ++j) before += w[j];
        const uint32_t tot = w[0] + w[1] + w[2] + w[3];
        if (is) nl_pos[out + before + (uint32_t) __builtin_popcountll(m & ((1ULL << lane) - 1ULL))] = p;
        out += tot;
        __syncthreads();
    }
}

struct IngLines {
    const uint8_t *text;
    uint64_t n_bytes, n_lines;    // a last line without '\n' counts
    const uint64_t *nl_pos;       // [number of '\n']
    uint64_t n_nl;
};
__device__ __forceinline__ uint64_t ing_line_start(const IngLines &l, uint64_t i) { return i == 0? 0 : l.nl_pos[i - 1] + 1; }
__device__ __forceinline__ uint64_t ing_line_end(const IngLines &l, uint64_t i)     // exclusive, CR stripped
{
    uint64_t e = i < l.n_nl? l.nl_pos[i] : l.n_bytes;
    const uint64_t s = ing_line_start(l, i);
    if (e > s && l.text[e - 1] == '\r') --e;
    return e;
}

// FASTA: is_hdr[i], seq_len[i] (both widened to u64 for the scans; entry n_lines = 0).  A sequence line that starts with '+' is where kseq
// switches to a quality string (kseq.h:207): such text is not plain FASTA and is left to the line-by-line reading below (flags[2]).
__global__ void ing_line_kernel(IngLines l, uint64_t *is_hdr, uint64_t *seq_len, uint32_t *flags)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i > l.n_lines) return;
    if (i == l.n_lines) { is_hdr[i] = 0, seq_len[i] = 0; return; }
    const uint64_t s = ing_line_start(l, i), e = ing_line_end(l, i);
    const bool h = e > s && (l.text[s] == '>' || l.text[s] == '@');
    if (e > s && l.text[s] == '+') flags[2] = 1u;
    is_hdr[i] = h;
    seq_len[i] = h? 0 : e - s;
}
// kseq's reading line by line (kseq.h:192-235), for text that is neither plain FASTA nor four-line FASTQ -- wrapped FASTQ, FASTA and FASTQ records
// in one stream: what a line IS depends on the lines before it (inside a quality string a line that starts with '@' is quality), so the classification
// is a serial walk; it runs on the host over three numbers per line that this kernel extracts: first character (0 for an empty line), length with
// the CR stripped, and whether a '>' or '@' sits anywhere behind the first character (kseq looks for the next header character by character, so
// such a line inside skipped text would start a record in the middle of a line: refused)
struct __attribute__((packed)) IngLineInfo { uint32_t len; uint8_t c0, mid; };
__global__ __launch_bounds__(256) void ing_line_info_kernel(IngLines l, IngLineInfo *info)
{
    const uint64_t i = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (i >= l.n_lines) return;
    const uint64_t s = ing_line_start(l, i), e = ing_line_end(l, i);
    bool m = false;
    for (uint64_t j = 1 + lane; j < e - s; j += 64) { const uint8_t ch = l.text[s + j]; m |= ch == '>' || ch == '@'; }
    const bool mid = e > s && __ballot(m) != 0;
    if (lane == 0) {
        IngLineInfo x;
        x.len = e - s > 0xFFFFFFFFULL? 0xFFFFFFFFu : (uint32_t) (e - s), x.c0 = e > s? l.text[s] : 0, x.mid = mid;
        info[i] = x;
    }
}
// header lines by rank
__global__ void ing_hdr_lines_kernel(uint64_t n_lines, const uint64_t *is_hdr, const uint64_t *hdr_rank, uint64_t *hdr_line)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lines && is_hdr[i]) hdr_line[hdr_rank[i]] = i;
}
// per record: length (u32 + widened, padded to 64), header offset
__global__ void ing_record_kernel(IngLines l, uint64_t n_rec, const uint64_t *hdr_line, const uint64_t *seq_before, uint32_t *len, uint64_t *padded,
                                  uint64_t *hdr_off, uint32_t *flags)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { padded[r] = 0; return; }
    const uint64_t h = hdr_line[r], nxt = hdr_line[r + 1];          // hdr_line[n_rec] = first line that does not belong
    const uint64_t n = seq_before[nxt] - seq_before[h];
    if (n > 0xFFFFFFF0ULL) flags[0] = 1u;                           // a read of 4 G bases
    len[r] = (uint32_t) n;
    padded[r] = (n + 63) & ~63ULL;
    hdr_off[r] = ing_line_start(l, h);
}
// one wave per line of text: copy sequence lines to their place
__global__ __launch_bounds__(256) void ing_copy_fasta_kernel(IngLines l, uint64_t line_end, const uint64_t *is_hdr, const uint64_t *seq_len, const uint64_t *hdr_rank,
                                                             const uint64_t *hdr_line, const uint64_t *seq_before, const uint64_t *off, uint8_t *seq)
{
    const uint64_t i = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (i >= line_end || is_hdr[i]) return;
    const uint64_t rk = hdr_rank[i];                                // headers before this line
    if (rk == 0) return;                                            // text in front of the first header (kseq skips it)
    const uint64_t r = rk - 1, s = ing_line_start(l, i), n = seq_len[i];       // (0 for a line that carries no sequence: '+', quality, skipped text)
    uint8_t *dst = seq + off[r] + (seq_before[i] - seq_before[hdr_line[r]]);
    for (uint64_t j = lane; j < n; j += 64) dst[j] = l.text[s + j];
}

// FASTQ, four lines per record
__global__ void ing_fastq_record_kernel(IngLines l, uint64_t n_rec, uint32_t *len, uint64_t *padded, uint64_t *hdr_off, uint32_t *flags)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { padded[r] = 0; return; }
    const uint64_t h = 4 * r;
    const uint64_t s0 = ing_line_start(l, h), s1 = ing_line_start(l, h + 1), e1 = ing_line_end(l, h + 1), s2 = ing_line_start(l, h + 2);
    const uint64_t s3 = ing_line_start(l, h + 3), e3 = ing_line_end(l, h + 3);
    if (l.text[s0] != '@' || l.text[s2] != '+' || e3 - s3 != e1 - s1) flags[1] = 1u;     // not the four-line form
    const uint64_t n = e1 - s1;
    if (n > 0xFFFFFFF0ULL) flags[0] = 1u;
    len[r] = (uint32_t) n;
    padded[r] = (n + 63) & ~63ULL;
    hdr_off[r] = s0;
}
// lines [first, l.n_lines) of the last chunk that do not make up a whole record: blank is fine, anything else is a truncated record
__global__ void ing_fastq_leftover_kernel(IngLines l, uint64_t first, uint32_t *flags)
{
    const uint64_t i = first + threadIdx.x;
    if (i < l.n_lines && ing_line_end(l, i) > ing_line_start(l, i)) flags[1] = 1u;
}
__global__ __launch_bounds__(256) void ing_copy_fastq_kernel(IngLines l, uint64_t n_rec, const uint64_t *off, uint8_t *seq)
{
    const uint64_t r = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n_rec) return;
    const uint64_t s = ing_line_start(l, 4 * r + 1), e = ing_line_end(l, 4 * r + 1);
    uint8_t *dst = seq + off[r];
    for (uint64_t j = lane; j < e - s; j += 64) dst[j] = l.text[s + j];
}

}  // namespace oatk
