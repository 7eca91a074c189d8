// oatk_amd/csrc/asmgraph.hpp -- the assembly graph of the (corrected) reads, built on the device.
//
// Replaces make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299) with asmg_finalize(g, 1) applied
// (graph.c:148-203, :70-113, :205-233, :126-146).  Shares the arc counter of ecgraph.hpp (canonical keys of adjacent pairs,
// one radix sort, a run-length encode); what is new here are the two filters, the squeeze of dropped vertices and the link ids.
#pragma once
#include "common.hpp"
#include "ecgraph.hpp"

namespace oatk {

// canonical key of every adjacent pair, (read, slot) order; slot 0 of a read is a filler (syncasm.c:242-261)
__global__ void agr_pair_keys_kernel(uint64_t n_reads, const uint64_t *scm_off, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t *keys)
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

// syncasm.c:226-233: scm[i].del |= scm[i].cov < min_k_cov; keep = !del
__global__ void agr_vtx_filter_kernel(uint64_t n_scm, const uint32_t *cov, const uint8_t *del_in, uint32_t min_k_cov, uint8_t *del_out, uint32_t *keep)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_scm) return;
    const uint8_t d = (uint8_t) ((del_in? del_in[i] : 0) | (cov[i] < min_k_cov));
    del_out[i] = d, keep[i] = !d;
}

// asmg_cleanup, graph.c:153-173: surviving vertices move up in order
__global__ void agr_vtx_squeeze_kernel(uint64_t n_scm, const uint8_t *del, const uint64_t *vidx, const uint32_t *cov, uint32_t *vtx_scm, uint32_t *vtx_cov)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_scm || del[i]) return;
    vtx_scm[vidx[i]] = (uint32_t) i, vtx_cov[vidx[i]] = cov[i] & 0x3FFFFFFFu;       // asmg_vtx_t.cov is a 30-bit field
}

// the arc filter of syncasm.c:269-272, in the reference's double arithmetic
__device__ inline bool agr_arc_kept(uint64_t k, uint32_t cnt, const uint32_t *cov, const uint8_t *del, double min_a_cov_f)
{
    const uint64_t a = (k >> 32) >> 1, b = (k & 0xFFFFFFFFULL) >> 1;
    const uint32_t ca = cov[a], cb = cov[b];
    if ((double) cnt < min_a_cov_f * (double) (ca < cb? ca : cb)) return false;
    return !(del[a] | del[b]);
}

__global__ void agr_expand_count_kernel(uint64_t n_keys, const uint64_t *ukeys, const uint32_t *counts, const uint32_t *cov, const uint8_t *del,
                                        double min_a_cov_f, uint32_t *n_out)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t k = ukeys[i];
    if (k == EGR_INVALID || !agr_arc_kept(k, counts[i], cov, del, min_a_cov_f)) { n_out[i] = 0; return; }
    const uint64_t v0 = k >> 32, v1 = k & 0xFFFFFFFFULL;
    n_out[i] = (v1 ^ 1ULL) != v0? 2u : 1u;
}

// arcs in the squeezed numbering (graph.c:194-200); payload through the (v, w) sort: coverage << 1 | complement flag
__global__ void agr_expand_kernel(uint64_t n_keys, const uint64_t *ukeys, const uint32_t *counts, const uint32_t *n_out, const uint64_t *out_off,
                                  const uint64_t *vidx, uint64_t *akey, uint64_t *aval)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys || n_out[i] == 0) return;
    const uint64_t k = ukeys[i], v0 = k >> 32, v1 = k & 0xFFFFFFFFULL, o = out_off[i];
    const uint64_t n0 = vidx[v0 >> 1] << 1 | (v0 & 1ULL), n1 = vidx[v1 >> 1] << 1 | (v1 & 1ULL);
    const uint64_t c = (uint64_t) (counts[i] & 0x3FFFFFFFu) << 1;                     // asmg_arc_t.cov is a 30-bit field
    akey[o] = n0 << 32 | n1, aval[o] = c;
    if (n_out[i] == 2) akey[o + 1] = (n1 ^ 1ULL) << 32 | (n0 ^ 1ULL), aval[o + 1] = c | 1ULL;
}

// asmg_shrink_link_id, graph.c:126-146: walking the arcs in order, an arc without an id takes the next one and hands it to its
// complement.  So an arc opens an id iff its complement does not precede it; the id is the number of such arcs before it.
__device__ inline uint64_t agr_find(const uint64_t *skey, uint64_t n, uint64_t key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (skey[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;                                                                        // first position with skey >= key
}
__global__ void agr_link_first_kernel(uint64_t n_arc, const uint64_t *skey, uint32_t *opens, uint64_t *comp_idx)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_arc) return;
    const uint64_t k = skey[i], v = k >> 32, w = k & 0xFFFFFFFFULL;
    const uint64_t j = agr_find(skey, n_arc, (w ^ 1ULL) << 32 | (v ^ 1ULL));
    comp_idx[i] = j, opens[i] = j >= i;
}
__global__ void agr_link_assign_kernel(uint64_t n_arc, const uint32_t *opens, const uint64_t *comp_idx, const uint64_t *open_rank, uint64_t *link)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_arc) return;
    link[i] = opens[i]? open_rank[i] : open_rank[comp_idx[i]];
}

} // namespace oatk
