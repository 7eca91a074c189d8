// oatk_amd/csrc/scan_hpc.hpp -- kernel A of the read scan: homopolymer compression + 2-bit pack.
//
// Replaces the first half of the reference's per-read loop (syncmer.c:284-323): every maximal run of
// one ACGT base becomes one "hoco" position with a 2-bit base in hoco_s (MSB-first, 4 per byte,
// syncmer.c:290) and min(run,256)-1 in ho_rl (:303-304); runs > 255 also go to the ho_l_rl overflow
// list (:301-302); every non-ACGT byte is its own position stored as base A with run 1, and its raw
// coordinate goes to n_nucl (:316-321).
//
// MI355X mapping: HBM-bound streaming kernel.  One 256-thread workgroup per read walks 4 KiB tiles;
// each lane loads one aligned 16-byte vector (reads start on 64-byte boundaries of the packed stream),
// a workgroup scan of (run-start count, last run-start position) turns raw coordinates into hoco
// coordinates, finished runs are staged in an LDS ring and leave as full 16-byte stores.
#pragma once
#include "common.hpp"

namespace oatk {

constexpr int HPC_NT = 256;
constexpr int HPC_BPT = 16;
constexpr int HPC_TILE = HPC_NT * HPC_BPT;
constexpr int HPC_RING = 8192;   // staged hoco positions; > one tile + one unflushed 64-group

struct HpcArgs {
    const uint8_t *seq;       // packed read stream, read r at off[r] (64-byte aligned), len[r] bytes
    const uint64_t *off;
    const uint32_t *len;
    uint64_t sid0;            // global id of read 0
    uint8_t *ho_rl;           // read r at off[r]
    uint8_t *hoco_s;          // read r at off[r] / 4
    uint32_t *nbits;          // one bit per hoco position that is an ambiguous base; read r at off[r] / 32 (words)
    uint32_t *hoco_l, *n_nn, *n_lrl;   // per read
    uint64_t *nn_key;         // sid << 32 | raw position           (unordered append)
    uint64_t *lrl_key;        // sid << 32 | hoco position          (unordered append)
    uint32_t *lrl_val;        // run length - 1
    uint32_t nn_cap, lrl_cap;
    uint32_t *counters;       // [0] appended to nn, [1] appended to lrl (may exceed the capacities)
};

__global__ __launch_bounds__(HPC_NT) void hpc_pack_kernel(HpcArgs a)
{
    __shared__ uint4 ring_rl4[HPC_RING / 16];
    __shared__ uint4 ring_hs4[HPC_RING / 64];
    __shared__ uint32_t w_cnt[HPC_NT / OATK_WAVE];
    __shared__ int32_t w_max[HPC_NT / OATK_WAVE];
    __shared__ uint32_t s_nn, s_lrl;

    uint8_t *ring_rl = (uint8_t *) ring_rl4;
    uint32_t *ring_hs = (uint32_t *) ring_hs4;

    const uint32_t r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t o = a.off[r];
    const uint32_t L = a.len[r];
    const uint64_t sid = a.sid0 + r;
    const uint8_t *in = a.seq + o;
    uint8_t *out_rl = a.ho_rl + o;
    uint8_t *out_hs = a.hoco_s + (o >> 2);
    uint32_t *out_nb = a.nbits + (o >> 5);

    for (uint32_t i = tid; i < HPC_RING / 16; i += HPC_NT) ring_hs[i] = 0;
    if (tid == 0) s_nn = 0, s_lrl = 0;
    __syncthreads();

    // a finished run: hoco index h, raw start position p0, length rl, class c of its bases
    auto finish_run = [&](uint32_t h, uint32_t p0, uint32_t rl, uint32_t c) {
        uint32_t code = c < 4u? c : 0u;
        uint32_t hr = h & (HPC_RING - 1);
        ring_rl[hr] = (uint8_t) ((rl > 256u? 256u : rl) - 1u);
        if (code) atomicOr(&ring_hs[hr >> 4], code << (8u * ((h >> 2) & 3u) + (((h & 3u) ^ 3u) << 1)));
        if (rl > 255u) {
            uint32_t idx = atomicAdd(&a.counters[1], 1u);
            if (idx < a.lrl_cap) a.lrl_key[idx] = sid << 32 | h, a.lrl_val[idx] = rl - 1u;
            atomicAdd(&s_lrl, 1u);
        }
        if (c == 4u) {
            atomicOr(&out_nb[h >> 5], 1u << (h & 31u));
            uint32_t idx = atomicAdd(&a.counters[0], 1u);
            if (idx < a.nn_cap) a.nn_key[idx] = sid << 32 | p0;
            atomicAdd(&s_nn, 1u);
        }
    };
    // move finished 64-position groups [g0, g1) from the LDS ring to HBM as 16-byte stores
    auto flush = [&](uint32_t g0, uint32_t g1) {
        uint32_t n_item = (g1 - g0) * 5u;
        for (uint32_t it = tid; it < n_item; it += HPC_NT) {
            uint32_t g = g0 + it / 5u, q = it % 5u;
            uint32_t hr = (g * 64u) & (HPC_RING - 1);
            if (q < 4u) {
                ((uint4 *) out_rl)[g * 4u + q] = ring_rl4[hr / 16u + q];
            } else {
                ((uint4 *) out_hs)[g] = ring_hs4[hr / 64u];
                ring_hs4[hr / 64u] = make_uint4(0, 0, 0, 0);
            }
        }
    };

    uint32_t nstart = 0;      // run starts seen in earlier tiles
    int32_t last_start = -1;  // raw position of the most recent one
    uint32_t flushed = 0;     // 64-groups already in HBM

    for (uint32_t t0 = 0; t0 < L; t0 += HPC_TILE) {
        const uint32_t b0 = t0 + tid * HPC_BPT;
        uint32_t cls[HPC_BPT];
        int nvalid = 0;
        uint32_t up = 5u;   // class of the byte before this lane's first byte (5 = none)
        if (b0 < L) {
            uint4 v = *(const uint4 *) (in + b0);
            nvalid = (int) (L - b0 < (uint32_t) HPC_BPT? L - b0 : (uint32_t) HPC_BPT);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < HPC_BPT; ++b) cls[b] = nt4_code((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
            if (b0 > 0) up = nt4_code(in[b0 - 1]);
        } else {
#pragma unroll
            for (int b = 0; b < HPC_BPT; ++b) cls[b] = 5u;
        }
        // pass 1: run starts in this lane's bytes
        uint32_t smask = 0, cnt = 0, pc = up;
        int32_t lpos = -1;
#pragma unroll
        for (int b = 0; b < HPC_BPT; ++b) {
            if (b < nvalid) {
                uint32_t c = cls[b];
                bool st = (c == 4u) | (c != pc);   // position 0 has pc == 5, so it always starts a run
                if (st) smask |= 1u << b, ++cnt, lpos = (int32_t) (b0 + b);
                pc = c;
            }
        }
        uint32_t icnt = wave_incl_sum(cnt, lane);
        int32_t imax = wave_incl_max(lpos, lane);
        if (lane == 63) w_cnt[wid] = icnt, w_max[wid] = imax;
        __syncthreads();
        uint32_t n = nstart + icnt - cnt;
        int32_t ls = __shfl_up(imax, 1);
        if (lane == 0) ls = -1;
        if (last_start > ls) ls = last_start;
        for (uint32_t w = 0; w < wid; ++w) {
            n += w_cnt[w];
            if (w_max[w] > ls) ls = w_max[w];
        }
        // pass 2: a run is finished when the next one starts
        pc = up;
#pragma unroll
        for (int b = 0; b < HPC_BPT; ++b) {
            if (b < nvalid) {
                if ((smask >> b) & 1u) {
                    uint32_t i = b0 + b;
                    if (i > 0) finish_run(n - 1u, (uint32_t) ls, i - (uint32_t) ls, pc);
                    ++n, ls = (int32_t) i;
                }
                pc = cls[b];
            }
        }
        __syncthreads();
        for (uint32_t w = 0; w < HPC_NT / OATK_WAVE; ++w) {
            nstart += w_cnt[w];
            if (w_max[w] > last_start) last_start = w_max[w];
        }
        uint32_t done = nstart? (nstart - 1u) >> 6 : 0u;   // complete 64-groups among finished runs
        flush(flushed, done);
        flushed = done;
        __syncthreads();
    }
    // the last run ends with the read
    if (tid == 0 && nstart) finish_run(nstart - 1u, (uint32_t) last_start, L - (uint32_t) last_start, nt4_code(in[L - 1]));
    __syncthreads();
    flush(flushed, (nstart + 63u) >> 6);
    if (tid == 0) {
        a.hoco_l[r] = nstart;
        a.n_nn[r] = s_nn;
        a.n_lrl[r] = s_lrl;
    }
}

}  // namespace oatk
