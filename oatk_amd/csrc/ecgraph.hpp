// oatk_amd/csrc/ecgraph.hpp -- the graph reads are corrected against, built on the device (SURVEY.md 8f rows 1-2,
// restricted to what the error correction needs).
//
// Replaces, for the fresh databases of one resident batch:
//   make_syncmer_graph(sr_db, scm_db, 0, 0.)   syncasm.c:203-299  -> egr_pair_keys_kernel, one 64-bit radix sort, a run-length
//                                                                    encode (= the khashl arc counter :242-261), egr_expand /
//                                                                    egr_unpack kernels, a second sort into (v, w) order
//   asmg_arc_index / asmg_arc_fix_symm          graph.c:85-113, :205-233 -> egr_index_kernel (+ the self-complement flag flip)
//   arc overlaps of scg_consensus(hoco)         syncasm.c:793-812, calc_syncmer_overlap :477-582 -> egr_overlap_kernel
//
// The overlap of an arc is K minus the MOST FREQUENT distance between its two syncmers on the reads; ties go to the first
// key in khashl bucket order (syncasm.c:558-571).  Low-coverage arcs tie all the time (two reads, two distances), so the
// kernel carries a faithful miniature of khashl<int,int>: identity hash, Fibonacci bucket mapping (khashl.h:82), linear
// probing, growth at 75 % with the kick-out rehash (khashl.h:150-218).
#pragma once
#include "common.hpp"

namespace oatk {

#define EGR_INVALID 0xFFFFFFFFFFFFFFFFULL

// canonical key of every pair of syncmers adjacent on a read (syncasm.c:242-261); slot 0 of a read carries no pair
__global__ void egr_pair_keys_kernel(uint64_t n_reads, const uint64_t *scm_off, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t *keys)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t o = scm_off[r], n = scm_off[r + 1] - o;
    if (n == 0) return;
    keys[o] = EGR_INVALID;
    uint64_t v0 = (k_mer[o] >> 1) << 1 | (m_pos[o] & 1u);
    for (uint64_t j = 1; j < n; ++j) {
        const uint64_t v1 = (k_mer[o + j] >> 1) << 1 | (m_pos[o + j] & 1u);
        keys[o + j] = v0 <= v1? v0 << 32 | v1 : (v1 ^ 1ULL) << 32 | (v0 ^ 1ULL);
        v0 = v1;
    }
}

// arcs per distinct key: itself, plus its complement unless it is its own (syncasm.c:264-282)
__global__ void egr_expand_count_kernel(uint64_t n_keys, const uint64_t *ukeys, uint32_t *n_out)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t k = ukeys[i];
    if (k == EGR_INVALID) { n_out[i] = 0; return; }
    const uint64_t v0 = k >> 32, v1 = k & 0xFFFFFFFFULL;
    n_out[i] = (v1 ^ 1ULL) != v0? 2u : 1u;
}

__global__ void egr_expand_kernel(uint64_t n_keys, const uint64_t *ukeys, const uint32_t *counts, const uint64_t *out_off, uint64_t *akey, uint64_t *aval)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t k = ukeys[i];
    if (k == EGR_INVALID) return;
    const uint64_t v0 = k >> 32, v1 = k & 0xFFFFFFFFULL, o = out_off[i], c = counts[i];
    akey[o] = k, aval[o] = c << 1;                                        // comp = 0
    if ((v1 ^ 1ULL) != v0) akey[o + 1] = (v1 ^ 1ULL) << 32 | (v0 ^ 1ULL), aval[o + 1] = c << 1 | 1ULL;
}

struct EgrArcs {
    uint64_t n_arc;
    uint64_t *arc_v, *arc_w;
    uint32_t *arc_ls, *arc_cov;
    uint8_t *arc_comp, *arc_del;
    uint64_t *idx_p;
    uint32_t *idx_n;
    uint32_t *flags;              // [0] duplicate (v, w) ("multi-arc"), [1] more distinct distances than the table miniature holds
};

// sorted (key, value) -> arc arrays; an arc that is its own complement ends with comp = 1 (asmg_arc_fix_symm, graph.c:205-233)
__global__ void egr_unpack_kernel(EgrArcs a, const uint64_t *skey, const uint64_t *sval)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_arc) return;
    const uint64_t k = skey[i], v = k >> 32, w = k & 0xFFFFFFFFULL;
    a.arc_v[i] = v, a.arc_w[i] = w;
    a.arc_cov[i] = (uint32_t) (sval[i] >> 1);
    uint8_t comp = (uint8_t) (sval[i] & 1ULL);
    if ((w ^ 1ULL) == v) comp ^= 1;
    a.arc_comp[i] = comp, a.arc_del[i] = 0, a.arc_ls[i] = 0;
    if (i && skey[i - 1] == k) a.flags[0] = 1u;
    if (i == 0 || (skey[i - 1] >> 32) != v) a.idx_p[v] = i;               // asmg_arc_index, graph.c:85-113
    atomicAdd(&a.idx_n[v], 1u);
}

// ---- khashl<int,int> miniature (syncasm.c:63 instantiation), at most 64 buckets ----
struct MiniKh {
    uint32_t bits, count;
    uint64_t used;
    int32_t keys[64], vals[64];
    bool overflow;

    __device__ static uint32_t h2b(uint32_t hash, uint32_t bits) { return (hash * 2654435769U) >> (32 - bits); }
    __device__ void init() { bits = 0, count = 0, used = 0, overflow = false; }
    __device__ uint32_t nb() const { return bits? 1U << bits : 0U; }
    __device__ void resize(uint32_t want)                                  // khashl.h:150-192
    {
        uint32_t j = 0, x = want;
        while ((x >>= 1) != 0) ++j;
        if (want & (want - 1)) ++j;
        const uint32_t nbits = j > 2? j : 2;
        if (nbits > 6) { overflow = true; return; }
        const uint32_t n_old = nb(), n_new = 1U << nbits;
        uint64_t nused = 0;
        for (j = 0; j != n_old; ++j) {
            if (!((used >> j) & 1ULL)) continue;
            int32_t key = keys[j], val = vals[j];
            used &= ~(1ULL << j);
            for (;;) {
                uint32_t i = h2b((uint32_t) key, nbits);
                while ((nused >> i) & 1ULL) i = (i + 1) & (n_new - 1);
                nused |= 1ULL << i;
                if (i < n_old && ((used >> i) & 1ULL)) {
                    const int32_t tk = keys[i], tv = vals[i];
                    keys[i] = key, vals[i] = val, key = tk, val = tv;
                    used &= ~(1ULL << i);
                } else {
                    keys[i] = key, vals[i] = val;
                    break;
                }
            }
        }
        used = nused, bits = nbits;
    }
    __device__ void add1(int32_t key)                                      // add_ovl_count, syncasm.c:465-474
    {
        uint32_t n = nb();
        if (count >= (n >> 1) + (n >> 2)) { resize(n + 1U); if (overflow) return; n = nb(); }
        uint32_t i = h2b((uint32_t) key, bits);
        const uint32_t last = i;
        while (((used >> i) & 1ULL) && keys[i] != key) { i = (i + 1U) & (n - 1); if (i == last) break; }
        if (!((used >> i) & 1ULL)) keys[i] = key, vals[i] = 1, used |= 1ULL << i, ++count;
        else ++vals[i];
    }
    __device__ int32_t mode() const                                        // syncasm.c:558-571: first bucket reaching the maximum
    {
        int32_t movl = 0, mcnt = 0;
        for (uint32_t k = 0; k < nb(); ++k) if (((used >> k) & 1ULL) && vals[k] > mcnt) mcnt = vals[k], movl = keys[k];
        return movl;
    }
};

struct EgrOverlapArgs {
    EgrArcs a;
    int K;
    uint64_t sid0;
    const uint64_t *occ_off, *occ;    // syncmer occurrence lists (resident count)
    const uint64_t *scm_off;          // slots of the per-read chains
    const uint32_t *m_pos;
};

// one lane per non-complement arc: calc_syncmer_overlap (syncasm.c:477-582) and the arc.ls assignment (:793-812)
__global__ void egr_overlap_kernel(EgrOverlapArgs g)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.a.n_arc || g.a.arc_comp[i]) return;
    const uint64_t v = g.a.arc_v[i], w = g.a.arc_w[i];
    const uint64_t m1 = v >> 1, m2 = w >> 1, rc1 = v & 1ULL, rc2 = w & 1ULL;
    const uint64_t *pos1 = g.occ + g.occ_off[m1], *pos2 = g.occ + g.occ_off[m2];
    const uint64_t n1 = g.occ_off[m1 + 1] - g.occ_off[m1], n2 = g.occ_off[m2 + 1] - g.occ_off[m2];
    MiniKh h;
    h.init();
    uint64_t p2 = 0;
    for (uint64_t p1 = 0; p1 < n1; ++p1) {
        const uint64_t r1 = pos1[p1] >> 32, i1 = (pos1[p1] >> 1) & 0x7FFFFFFFULL, c1 = pos1[p1] & 1ULL;
        const uint64_t base = g.scm_off[r1 - g.sid0];
        const int64_t l1 = g.m_pos[base + i1] >> 1;
        while (p2 < n2 && (pos2[p2] >> 32) < r1) ++p2;
        for (uint64_t t = p2; t < n2 && (pos2[t] >> 32) == r1; ++t) {
            const uint64_t i2 = (pos2[t] >> 1) & 0x7FFFFFFFULL, c2 = pos2[t] & 1ULL;
            const int64_t l2 = g.m_pos[base + i2] >> 1;
            if (i1 == i2 + 1 && c1 != rc1 && c2 != rc2) h.add1((int32_t) (l1 - l2));
            else if (i1 + 1 == i2 && c1 == rc1 && c2 == rc2) h.add1((int32_t) (l2 - l1));
        }
    }
    if (h.overflow) { g.a.flags[1] = 1u; return; }
    int64_t l = h.mode();
    if (l < g.K) l = l < 0? g.K : g.K - l;      // scg_syncmer_consensus(beg = l) then MIN with the vertex length K
    else l = 0;
    g.a.arc_ls[i] = (uint32_t) l;
    const uint64_t cv = w ^ 1ULL, cw = v ^ 1ULL, p = g.a.idx_p[cv];
    const uint32_t n = g.a.idx_n[cv];
    for (uint32_t t = 0; t < n; ++t) if (g.a.arc_w[p + t] == cw) { g.a.arc_ls[p + t] = (uint32_t) l; break; }
}

}  // namespace oatk
