// oatk_amd/csrc/ecgraph.hpp -- the graph reads are corrected against, built on the device (SURVEY.md 8f rows 1-2,
// restricted to what the error correction needs).
//
// Replaces, for the fresh databases of one resident batch:
//   make_syncmer_graph(sr_db, scm_db, 0, 0.)   syncasm.c:203-299  -> egr_pair_keys_kernel, one 64-bit radix sort, a run-length
//                                                                    encode (= the khashl arc counter :242-261), egr_expand /
//                                                                    egr_unpack kernels, a second sort into (v, w) order
//   asmg_arc_index / asmg_arc_fix_symm          graph.c:85-113, :205-233 -> egr_unpack_kernel (+ the self-complement flag flip)
//   arc overlaps of scg_consensus(hoco)         syncasm.c:793-812, calc_syncmer_overlap :477-582 -> egr_mode_kernel: the distance
//                                                                    of every adjacent pair rides through the key sort, so a run
//                                                                    of equal keys IS the multiset the reference tabulates
//
// The overlap of an arc is K minus the MOST FREQUENT distance between its two syncmers on the reads; ties go to the first
// key in khashl bucket order (syncasm.c:558-571).  Low-coverage arcs tie all the time (two reads, two distances), so the
// kernel carries a faithful miniature of khashl<int,int>: identity hash, Fibonacci bucket mapping (khashl.h:82), linear
// probing, growth at 75 % with the kick-out rehash (khashl.h:150-218).
#pragma once
#include "common.hpp"

namespace oatk {

#define EGR_INVALID 0xFFFFFFFFFFFFFFFFULL

// canonical key of every pair of syncmers adjacent on a read (syncasm.c:242-261) and the distance between the two; slot 0
// of a read carries no pair.  Entries are produced in (read, slot) order, which is the order calc_syncmer_overlap meets
// the pairs of one arc in (it walks the occurrences of the arc's first syncmer, syncasm.c:497-556; the slot of that
// syncmer is the pair's first or second slot, and two pairs of one arc never share it).
// skip_corrected: pairs with an error-corrected member (low bit of k_mer) become fillers, as calc_syncmer_overlap ignores them (syncasm.c:499, :511)
__global__ void egr_pair_keys_kernel(uint64_t n_reads, const uint64_t *scm_off, const uint64_t *k_mer, const uint32_t *m_pos, uint64_t *keys, uint32_t *dist,
                                     int skip_corrected = 0)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t o = scm_off[r], n = scm_off[r + 1] - o;
    if (n == 0) return;
    keys[o] = EGR_INVALID, dist[o] = 0;
    uint64_t v0 = (k_mer[o] >> 1) << 1 | (m_pos[o] & 1u);
    uint32_t p0 = m_pos[o] >> 1;
    for (uint64_t j = 1; j < n; ++j) {
        const uint64_t v1 = (k_mer[o + j] >> 1) << 1 | (m_pos[o + j] & 1u);
        const uint32_t p1 = m_pos[o + j] >> 1;
        keys[o + j] = skip_corrected && ((k_mer[o + j - 1] | k_mer[o + j]) & 1ULL)? EGR_INVALID : (v0 <= v1? v0 << 32 | v1 : (v1 ^ 1ULL) << 32 | (v0 ^ 1ULL));
        dist[o + j] = p1 - p0;
        v0 = v1, p0 = p1;
    }
}

// The LIGHT graph (api_ec.inc, oatk_hip_ec_graph_light): find_error_syncmers (syncerr.c:690-718) deletes every syncmer below err_mer_c, and with it
// every arc that touches one (asmg_vtx_del :748-752); all it ever asks of such an arc is (a) that it EXISTS on its side of a kept candidate
// (b[k] = 0 rather than -1, :699-706) and (b) whether it is "good" -- which it cannot be when err_arc_c >= err_mer_c, because an arc is seen at
// most as often as its rarer end.  So pairs with an end below `c` do not go through the sorts at all: they leave one flag per oriented
// candidate vertex (`other`), and only pairs between two candidates become arcs.  keep[i] = 1 for those.
// Pair keys and the split are made in one pass, a wave per read (lane j holds entry j of the chain and its left neighbour's comes by a lane
// shift; reads with more than 64 syncmers are walked by lane 0): a lane-per-read key kernel that strides through the chains plus a pass over its
// output took 2.4 ms at 2 M reads, this takes 0.85.
__global__ __launch_bounds__(256) void egr_pair_light_wave_kernel(uint64_t n_reads, const uint64_t *scm_off, const uint64_t *k_mer, const uint32_t *m_pos, const uint32_t *cov,
                                                                  uint32_t c, uint64_t *keys, uint32_t *dist, uint8_t *keep, uint8_t *other)
{
    const uint64_t r = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const uint64_t o = scm_off[r];
    const int64_t n = (int64_t) (scm_off[r + 1] - o);
    auto emit = [&](uint64_t at, uint64_t v0, uint32_t p0, uint64_t v1, uint32_t p1) {
        const bool c0 = cov[v0 >> 1] >= c, c1 = cov[v1 >> 1] >= c;
        const uint64_t a = v0 <= v1? v0 : v1 ^ 1ULL, b = v0 <= v1? v1 : v0 ^ 1ULL;      // the key stands for a -> b and (b ^ 1) -> (a ^ 1)
        keys[at] = a << 32 | b, dist[at] = p1 - p0, keep[at] = c0 && c1;
        const bool ca = v0 <= v1? c0 : c1, cb = v0 <= v1? c1 : c0;
        if (ca && !cb) other[a] = 1;
        if (cb && !ca) other[b ^ 1ULL] = 1;
    };
    if (n > 64) {
        if (lane == 0) {
            keys[o] = EGR_INVALID, dist[o] = 0, keep[o] = 0;
            uint64_t v0 = (k_mer[o] >> 1) << 1 | (m_pos[o] & 1u);
            uint32_t p0 = m_pos[o] >> 1;
            for (int64_t j = 1; j < n; ++j) {
                const uint64_t v1 = (k_mer[o + j] >> 1) << 1 | (m_pos[o + j] & 1u);
                const uint32_t p1 = m_pos[o + j] >> 1;
                emit(o + j, v0, p0, v1, p1);
                v0 = v1, p0 = p1;
            }
        }
        return;
    }
    const bool in = lane < n;
    const uint64_t km = in? k_mer[o + lane] : 0;
    const uint32_t mp = in? m_pos[o + lane] : 0;
    const uint64_t v1 = (km >> 1) << 1 | (mp & 1u);
    const uint32_t p1 = mp >> 1;
    const uint64_t v0 = (uint64_t) __shfl_up((long long) v1, 1);
    const uint32_t p0 = (uint32_t) __shfl_up((int) p1, 1);
    if (in && lane == 0) keys[o] = EGR_INVALID, dist[o] = 0, keep[o] = 0;
    if (in && lane > 0) emit(o + lane, v0, p0, v1, p1);
}

// a run of equal (key, distance) in the key-sorted pair list becomes one weighted segment; head[i] = 1 where one starts
__global__ void egr_seg_head_kernel(uint64_t n, const uint64_t *skeys, const uint32_t *sdist, uint8_t *head)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = i == 0 || skeys[i] != skeys[i - 1] || sdist[i] != sdist[i - 1];
}
__global__ void egr_seg_emit_kernel(uint64_t n_seg, uint64_t n, const uint32_t *head_pos, const uint64_t *skeys, const uint32_t *sdist, uint64_t *okey, uint64_t *oval)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg) return;
    const uint64_t p = head_pos[i], q = i + 1 < n_seg? head_pos[i + 1] : n;
    okey[i] = skeys[p], oval[i] = (uint64_t) sdist[p] | (q - p) << 32;       // distance | weight << 32
}
__global__ void egr_seg_split_kernel(uint64_t n, const uint64_t *val, uint32_t *dist, uint32_t *wgt)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dist[i] = (uint32_t) val[i], wgt[i] = (uint32_t) (val[i] >> 32);
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

// payload of an arc through the (v, w) sort: overlap << 44 | coverage << 1 | complement flag
__global__ void egr_expand_kernel(uint64_t n_keys, const uint64_t *ukeys, const uint32_t *counts, const uint32_t *run_ls, const uint64_t *out_off,
                                  uint64_t *akey, uint64_t *aval)       // counts: calls per key (the run length, or the weighted sum of a run of segments)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t k = ukeys[i];
    if (k == EGR_INVALID) return;
    const uint64_t v0 = k >> 32, v1 = k & 0xFFFFFFFFULL, o = out_off[i];
    // An arc that is its own complement (v -> v^1: a syncmer next to its reverse complement on a read) never gets an overlap in the reference:
    // asmg_arc_fix_symm looks up "the complement", finds the arc itself and sets its comp flag (graph.c:221), and scg_consensus skips arcs with
    // that flag (syncasm.c:779), so ls keeps its initial 0.  (Found by the config-1 surrogate in round 4: 16 such arcs among 4.5 M; the earlier
    // read sets had none.)
    const uint64_t ls = (v1 ^ 1ULL) == v0? 0ULL : (uint64_t) run_ls[i];
    const uint64_t c = ls << 44 | (uint64_t) counts[i] << 1;
    akey[o] = k, aval[o] = c;                                             // comp = 0
    if ((v1 ^ 1ULL) != v0) akey[o + 1] = (v1 ^ 1ULL) << 32 | (v0 ^ 1ULL), aval[o + 1] = c | 1ULL;
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
    a.arc_comp[i] = comp, a.arc_del[i] = 0, a.arc_ls[i] = (uint32_t) (sval[i] >> 44);
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
    __device__ void addw(int32_t key, uint32_t w)                          // w consecutive add_ovl_count calls for one distance
    {
        add1(key);
        if (w < 2 || overflow) return;
        add1(key);                                                         // the second call is the one that can find the table due to grow
        if (w < 3 || overflow) return;
        const uint32_t n = nb();
        uint32_t i = h2b((uint32_t) key, bits);
        while (keys[i] != key || !((used >> i) & 1ULL)) i = (i + 1U) & (n - 1);
        vals[i] += (int32_t) (w - 2);
    }
    __device__ int32_t mode() const                                        // syncasm.c:558-571: first bucket reaching the maximum
    {
        int32_t movl = 0, mcnt = 0;
        for (uint32_t k = 0; k < nb(); ++k) if (((used >> k) & 1ULL) && vals[k] > mcnt) mcnt = vals[k], movl = keys[k];
        return movl;
    }
};

// the same table for a whole wave: arrays in LDS, scalars uniform; every lane runs the same code on the same values
struct WaveKh {
    int32_t *keys, *vals;         // [64] in LDS
    uint64_t used;
    uint32_t bits, count;
    bool overflow;

    __device__ void init(int32_t *k, int32_t *v) { keys = k, vals = v, used = 0, bits = 0, count = 0, overflow = false; }
    __device__ uint32_t nb() const { return bits? 1U << bits : 0U; }
    __device__ bool due() const { const uint32_t n = nb(); return count >= (n >> 1) + (n >> 2); }
    __device__ void resize(uint32_t want)
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
                uint32_t i = MiniKh::h2b((uint32_t) key, nbits);
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
    // `cnt` add_ovl_count calls for one distance, the first of them at run position `first`; returns true for a new key
    __device__ bool add(int32_t key, int32_t cnt)
    {
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t n = nb(), i = 0;
            bool found = false;
            if (n) {
                i = MiniKh::h2b((uint32_t) key, bits);
                const uint32_t last = i;
                while (((used >> i) & 1ULL) && keys[i] != key) { i = (i + 1U) & (n - 1); if (i == last) break; }
                found = (used >> i) & 1ULL;
            }
            if (due()) {                               // every call checks first (khashl.h:199): the one after a threshold insert grows the table
                resize(n + 1U);
                if (overflow) return false;
                continue;
            }
            if (found) { vals[i] += cnt; return false; }
            keys[i] = key, vals[i] = cnt, used |= 1ULL << i, ++count;
            return true;
        }
        return false;
    }
    __device__ int32_t mode() const
    {
        int32_t movl = 0, mcnt = 0;
        for (uint32_t k = 0; k < nb(); ++k) if (((used >> k) & 1ULL) && vals[k] > mcnt) mcnt = vals[k], movl = keys[k];
        return movl;
    }
};

#define EGR_SMALL_RUN 48
#define EGR_HUGE_SCRATCH (256ull << 20)     // working memory of egr_mode_huge_kernel (tables of the runs with more than 48 distinct distances)

// overlap of the arc a run of equal keys stands for: K minus the most frequent distance (calc_syncmer_overlap,
// syncasm.c:477-582, and the arc.ls assignment :793-812).  One lane per short run; long runs go on a list and are then
// taken one per wave, sixty-four distances at a time (egr_mode_big_kernel).
// With `swgt` the entries of a run are SEGMENTS -- swgt[t] consecutive calls for distance sdist[t] (the pair lists of several shards, each
// compressed before it travelled) -- and run_cov[i] receives the number of calls, the arc's coverage.
__device__ __forceinline__ uint32_t egr_to_ls(int64_t l, int K)   // scg_syncmer_consensus(beg = l) then MIN with the vertex length K
{
    if (l < K) l = l < 0? K : K - l;
    else l = 0;
    return (uint32_t) l;
}

__global__ __launch_bounds__(64) void egr_mode_kernel(uint64_t n_runs, const uint64_t *ukeys, const uint32_t *counts, const uint64_t *run_off,
                                                      const uint32_t *sdist, int K, uint32_t *run_ls, uint32_t *flags, uint32_t *big_list, uint32_t *huge_list,
                                                      const uint32_t *swgt = nullptr, uint32_t *run_cov = nullptr)
{
    const int lane = threadIdx.x;
    const uint64_t i = (uint64_t) blockIdx.x * 64 + lane;
    const bool valid = i < n_runs && ukeys[i] != EGR_INVALID;
    const uint32_t c = valid? counts[i] : 0u;
    const uint64_t o = valid? run_off[i] : 0;
    if (valid && c <= EGR_SMALL_RUN) {
        MiniKh h;
        h.init();
        if (swgt) {
            uint32_t tot = 0;
            for (uint32_t t = 0; t < c; ++t) h.addw((int32_t) sdist[o + t], swgt[o + t]), tot += swgt[o + t];
            run_cov[i] = tot;
        } else {
            for (uint32_t t = 0; t < c; ++t) h.add1((int32_t) sdist[o + t]);
        }
        if (h.overflow) huge_list[atomicAdd(&flags[5], 1u)] = (uint32_t) i;       // (48 weighted segments and a call behind the last insert)
        else run_ls[i] = egr_to_ls(h.mode(), K);
    }
    const bool is_big = valid && c > EGR_SMALL_RUN;
    const uint64_t big = __ballot(is_big);
    if (big) {
        uint32_t base = 0;
        if (lane == __builtin_ctzll(big)) base = atomicAdd(&flags[4], (uint32_t) __builtin_popcountll(big));
        base = (uint32_t) __shfl((int32_t) base, __builtin_ctzll(big));
        if (is_big) big_list[base + (uint32_t) __builtin_popcountll(big & ((1ULL << lane) - 1ULL))] = (uint32_t) i;
    }
}

// the long runs, one per wave (flags[4] = how many; the grid is fixed and strides over the list)
__global__ __launch_bounds__(64) void egr_mode_big_kernel(const uint32_t *counts, const uint64_t *run_off, const uint32_t *sdist, int K, uint32_t *run_ls,
                                                          uint32_t *flags, const uint32_t *big_list, uint32_t *huge_list, const uint32_t *swgt = nullptr,
                                                          uint32_t *run_cov = nullptr)
{
    __shared__ int32_t tk[64], tv[64];
    const int lane = threadIdx.x;
    const uint32_t n_big = flags[4];
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t i = big_list[b];
        const uint32_t cc = counts[i];
        const uint64_t oo = run_off[i];
        WaveKh h;
        h.init(tk, tv);
        bool tail_new = false;                         // was the very last add call an insert?
        uint32_t tot = 0;
        for (uint32_t t0 = 0; t0 < cc && !h.overflow; t0 += 64) {
            const bool in = t0 + lane < cc;
            const int32_t d = in? (int32_t) sdist[oo + t0 + lane] : 0;
            const uint32_t w = in? (swgt? swgt[oo + t0 + lane] : 1u) : 0u;
            uint64_t rest = __ballot(in);
            while (rest && !h.overflow) {
                const int f = __builtin_ctzll(rest);
                const int32_t x = __builtin_amdgcn_readfirstlane(__shfl(d, f));
                const uint64_t eq = __ballot(in && d == x) & rest;
                rest &= ~eq;
                uint32_t calls = (uint32_t) __builtin_popcountll(eq);
                const uint32_t wf = (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) __shfl(w, f));
                if (swgt) {
                    calls = 0;
                    for (uint64_t q = eq; q; q &= q - 1) calls += (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) __shfl(w, __builtin_ctzll(q)));
                }
                tot += calls;
                const bool fresh = h.add(x, (int32_t) calls);
                tail_new = fresh && eq == (1ULL << f) && t0 + f == cc - 1 && wf == 1u;
            }
        }
        if (!h.overflow && h.due() && !tail_new) h.resize(h.nb() + 1U);       // khashl grows at the call AFTER the insert that filled it
        if (lane == 0) {
            if (swgt && !h.overflow) run_cov[i] = tot;
            if (h.overflow) huge_list[atomicAdd(&flags[5], 1u)] = i;        // more than 48 distinct distances: egr_mode_huge_kernel
            else run_ls[i] = egr_to_ls(h.mode(), K);
        }
        __syncthreads();                               // the table in LDS is reused
    }
}

// ---- the same table without a size limit, in global memory: runs with more than 48 distinct distances (arcs across a tandem array read at
// thousand-fold coverage; none on random sequence).  One lane replays the run call by call.  Until round 4 such an arc was OATK_E_SPLIT. ----
struct BigKh {
    int32_t *keys, *vals;
    uint8_t *used, *nused;
    uint32_t cap, bits, count;
    bool overflow;

    __device__ uint32_t nb() const { return bits? 1U << bits : 0U; }
    __device__ void resize(uint32_t want)                                  // khashl.h:150-192, growing only
    {
        uint32_t j = 0, x = want;
        while ((x >>= 1) != 0) ++j;
        if (want & (want - 1)) ++j;
        const uint32_t nbits = j > 2? j : 2;
        if (nbits > 30 || (1U << nbits) > cap) { overflow = true; return; }
        const uint32_t n_old = nb(), n_new = 1U << nbits;
        for (j = 0; j != n_new; ++j) nused[j] = 0;
        for (j = 0; j != n_old; ++j) {
            if (!used[j]) continue;
            int32_t key = keys[j], val = vals[j];
            used[j] = 0;
            for (;;) {
                uint32_t i = MiniKh::h2b((uint32_t) key, nbits);
                while (nused[i]) i = (i + 1) & (n_new - 1);
                nused[i] = 1;
                if (i < n_old && used[i]) {
                    const int32_t tk = keys[i], tv = vals[i];
                    keys[i] = key, vals[i] = val, key = tk, val = tv;
                    used[i] = 0;
                } else {
                    keys[i] = key, vals[i] = val;
                    break;
                }
            }
        }
        uint8_t *t = used; used = nused, nused = t;
        bits = nbits;
    }
    __device__ void add1(int32_t key)                                      // add_ovl_count, syncasm.c:465-474
    {
        uint32_t n = nb();
        if (count >= (n >> 1) + (n >> 2)) { resize(n + 1U); if (overflow) return; n = nb(); }
        uint32_t i = MiniKh::h2b((uint32_t) key, bits);
        const uint32_t last = i;
        while (used[i] && keys[i] != key) { i = (i + 1U) & (n - 1); if (i == last) break; }
        if (!used[i]) keys[i] = key, vals[i] = 1, used[i] = 1, ++count;
        else ++vals[i];
    }
    __device__ void addw(int32_t key, uint32_t w)                          // w consecutive calls for one distance
    {
        add1(key);
        if (w < 2 || overflow) return;
        add1(key);
        if (w < 3 || overflow) return;
        const uint32_t n = nb();
        uint32_t i = MiniKh::h2b((uint32_t) key, bits);
        while (keys[i] != key || !used[i]) i = (i + 1U) & (n - 1);
        vals[i] += (int32_t) (w - 2);
    }
    __device__ int32_t mode() const
    {
        int32_t movl = 0, mcnt = 0;
        for (uint32_t k = 0; k < nb(); ++k) if (used[k] && vals[k] > mcnt) mcnt = vals[k], movl = keys[k];
        return movl;
    }
};

// buckets a run of c calls can need: the table doubles at three quarters full, and once more when a call follows the insert that filled it
__device__ __forceinline__ uint32_t egr_huge_cap(uint32_t c)
{
    uint32_t n = 4;
    while ((uint64_t) n * 3 < (uint64_t) c * 4 + 4 && n < (1U << 30)) n <<= 1;
    return n < (1U << 30)? n << 1 : n;
}

// flags[5] runs on huge_list; scratch: `scratch_bytes` of working memory handed out with flags[6] as the bump pointer (in units of 16 bytes);
// a run that does not get its table sets flags[1] (the caller answers OATK_E_SPLIT)
__global__ __launch_bounds__(64) void egr_mode_huge_kernel(const uint32_t *counts, const uint64_t *run_off, const uint32_t *sdist, int K, uint32_t *run_ls,
                                                           uint32_t *flags, const uint32_t *huge_list, uint8_t *scratch, uint64_t scratch_bytes,
                                                           const uint32_t *swgt = nullptr, uint32_t *run_cov = nullptr)
{
    if (threadIdx.x != 0) return;
    const uint32_t n_huge = flags[5];
    for (uint32_t b = blockIdx.x; b < n_huge; b += gridDim.x) {
        const uint32_t i = huge_list[b], cc = counts[i];
        const uint64_t oo = run_off[i];
        const uint32_t cap = egr_huge_cap(cc);
        const uint64_t need = ((uint64_t) cap * 10 + 15) >> 4;             // keys + vals + two flag arrays, in 16-byte units
        const uint64_t at = (uint64_t) atomicAdd(&flags[6], (uint32_t) need);
        if (need >= (1ULL << 31) || (at + need) * 16 > scratch_bytes) { flags[1] = 1u; continue; }
        BigKh h;
        uint8_t *base = scratch + at * 16;
        h.keys = (int32_t *) base, h.vals = h.keys + cap, h.used = (uint8_t *) (h.vals + cap), h.nused = h.used + cap;
        h.cap = cap, h.bits = 0, h.count = 0, h.overflow = false;
        uint32_t tot = 0;
        for (uint32_t t = 0; t < cc && !h.overflow; ++t) {
            if (swgt) h.addw((int32_t) sdist[oo + t], swgt[oo + t]), tot += swgt[oo + t];
            else h.add1((int32_t) sdist[oo + t]);
        }
        if (swgt) run_cov[i] = tot;
        if (h.overflow) flags[1] = 1u;
        else run_ls[i] = egr_to_ls(h.mode(), K);
    }
}

// the pairs a light graph keeps, in their order: one exclusive scan of the keep flags places keys and distances together (r03q; two
// rocprim::select calls over 42 M pairs, one per array and each with its own wait on the host, were 1.6 ms at config 3)
__global__ void egr_compact_pairs_kernel(uint64_t n, const uint8_t *keep, const uint32_t *pos, const uint64_t *keys, const uint32_t *dist,
                                         uint64_t *lkeys, uint32_t *ldist, uint64_t *n_out)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keep[i], p = pos[i];
    if (k) lkeys[p] = keys[i], ldist[p] = dist[i];
    if (i == n - 1) *n_out = (uint64_t) p + k;
}

}  // namespace oatk
