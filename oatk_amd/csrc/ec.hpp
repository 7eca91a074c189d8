// oatk_amd/csrc/ec.hpp -- per-read syncmer-chain error correction on the device.
//
// Replaces, on flat arrays that stay resident in HBM:
//   find_error_syncmers            syncerr.c:679-757   -> ec_mark_kernel, ec_arc_del_kernel
//   error blocks of a read         syncerr.c:339-612   -> ec_blocks (device function, shared by three kernels)
//   dfs_search + wf_ed_core        syncerr.c:144-286, levdist.c:75-310 -> ec_wave_kernel (ec_wave.hpp)
//   update_syncmer_db              syncerr.c:769-814   -> a stable radix sort of (syncmer, occurrence) pairs + ec_fwd_flag / ec_cov_sorted kernels (api_ec.inc)
//
// The graph is the reference's asmg_t in arc-array order (sorted (v,w), graph.c:70-83) as CSR over oriented vertices.
// Every vertex is one syncmer (utg id == syncmer id, syncerr.c:421-423) and its hoco consensus is the oriented k-mer of
// the syncmer's first occurrence (scg_syncmer_consensus in hoco mode, syncasm.c:910-940) -- so no 1002-byte string per
// vertex is materialised: bases are read in place from the resident hoco strings of the reads.
//
// Mapping: error blocks of one read are independent (they are delimited on the ORIGINAL chain), so the unit of work is
// one block: a depth-first search over the good-syncmer graph that extends a consensus string arc by arc and re-aligns
// it to the read segment with a resumable Landau-Vishkin wavefront.  One WAVE solves one block (ec_wave.hpp).
#pragma once
#include "common.hpp"

namespace oatk {

#define EC_FAILURE 0
#define EC_SUCCESS 1
#define EC_AMBISNQ 2
#define EC_AMBISEQ 3
#define EC_MAX_DFS_PATH 10000
#define EC_MIN_ERR_SEQ_LEN 10
#define EC_MIN_ERR_BASE 6
#define EC_NONE 0xFFFFFFFFFFFFFFFFULL

struct EcGraph {
    uint64_t n_vtx, n_arc;
    const uint64_t *idx_p;        // [2 n_vtx] first arc of an oriented vertex
    const uint32_t *idx_n;        // [2 n_vtx] arc count
    const uint64_t *arc_v, *arc_w;
    const uint32_t *arc_ls, *arc_cov;
    uint8_t *arc_del;
    uint8_t *scm_del;             // [n_vtx]
    const uint32_t *scm_cov;
    const uint64_t *scm_s;
    // where the bases of a vertex live: byte offset of the read's hoco string, and pos << 1 | rev on it
    const uint64_t *vtx_hs_off;
    const uint32_t *vtx_mpos;
    const uint8_t *other;         // light graph (ecgraph.hpp): [2 n_vtx] the oriented vertex has arcs that were never materialised; else null
};

struct EcReads {
    uint64_t n_reads, sid0;
    int K;
    const uint8_t *hoco_s;
    const uint64_t *off;          // packed-stream offsets; hoco string of read r at off[r] / 4
    const uint32_t *hoco_l;
    const uint64_t *scm_off;      // [n_reads + 1] slots of the per-read chains
    const uint64_t *k_mer;        // id << 1 (| corrected)
    const uint32_t *m_pos;
};

// ---- find_error_syncmers, first loop (syncerr.c:690-718) ----
__global__ void ec_mark_kernel(EcGraph g, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_vtx) return;
    if (g.scm_del[i] || g.scm_cov[i] >= max_err_c) return;
    if (g.scm_cov[i] < err_mer_c) { g.scm_del[i] = 1; return; }
    const uint32_t nv = g.scm_cov[i];
    int b[2] = {-1, -1};
    for (int k = 0; k < 2; ++k) {
        const uint64_t v = i << 1 | (uint64_t) k, p = g.idx_p[v];
        const uint32_t na = g.idx_n[v];
        uint32_t live = 0;
        for (uint32_t j = 0; j < na; ++j) live += !g.arc_del[p + j];
        if (!live && !(g.other && g.other[v])) continue;
        b[k] = 0;
        for (uint32_t j = 0; j < na; ++j) {
            if (g.arc_del[p + j]) continue;
            const uint32_t nw = g.scm_cov[g.arc_w[p + j] >> 1], mn = nv < nw? nv : nw, ac = g.arc_cov[p + j];
            if (ac >= err_arc_c && (double) ac >= (double) mn * max_arc_f) { b[k] = 1; break; }
        }
    }
    if (!b[0] || !b[1]) g.scm_del[i] = 1;
}

// asmg_vtx_del for every marked syncmer (syncerr.c:748-752, graph.h:101-122): in a symmetric graph that is every arc
// with a deleted endpoint.  Runs after ec_mark_kernel has finished (the marking reads the ORIGINAL arc flags).
__global__ void ec_arc_del_kernel(EcGraph g)
{
    uint64_t a = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= g.n_arc) return;
    if (g.scm_del[g.arc_v[a] >> 1] || g.scm_del[g.arc_w[a] >> 1]) g.arc_del[a] = 1;
}

// where each vertex's k-mer can be read: its first occurrence (syncasm.c:910-926; nothing is corrected yet)
// (sharded reads: `l2g` maps the shard's syncmer ids to the global ids the graph is built on; vertices this shard never saw
// keep EC_NO_SRC until their k-mer is imported)
#define EC_NO_SRC 0xFFFFFFFFFFFFFFFFULL
__global__ void ec_vtx_src_kernel(uint64_t n_local, const uint64_t *scm_loc, const uint32_t *l2g, uint64_t *vtx_hs_off, uint32_t *vtx_mpos)
{
    // scm_loc (count.hpp: the locator of a syncmer's first occurrence) = (32-bit word index of the read's hoco string) << 32 | pos << 1 | rev
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    const uint64_t loc = scm_loc[i], v = l2g? l2g[i] : i;
    vtx_hs_off[v] = (loc >> 32) << 2;
    vtx_mpos[v] = (uint32_t) loc;
}
__global__ void ec_remap_kernel(uint64_t n, const uint64_t *k_mer, const uint32_t *l2g, uint64_t *out)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint64_t) l2g[k_mer[i] >> 1] << 1 | (k_mer[i] & 1ULL);
}

__device__ __forceinline__ uint32_t hoco_base(const uint8_t *hs, uint32_t p)
{
    return (hs[p >> 2] >> (((p & 3u) ^ 3u) << 1)) & 3u;
}

// ---- one error block of a read ----
struct EcBlock {
    uint64_t beg_utg, end_utg;    // source / sink oriented vertices (EC_NONE = open end)
    uint32_t beg_pos;             // first base of the read segment (hoco)
    int32_t l;                    // its length
    int32_t beg, end;             // chain indices: left anchor (after adjustment) and right anchor
    int32_t r;                    // 1 = leading block, solved on the reverse complement
};

// Walks the blocks of one read exactly like the loop at syncerr.c:394-598 and calls
//   on_block(k, blk)                 for the k-th block that has an anchor, and
//   on_copy(first, last_exclusive)   for every run of original chain entries that is kept as is.
// Returns the number of blocks, or -1 when the read has no good syncmer at all (it is then left untouched).
template <class FB, class FC>
__device__ int ec_blocks(const uint8_t *scm_del, const uint64_t *km, const uint32_t *mp, int32_t n, uint32_t hoco_l, int K, FB on_block, FC on_copy)
{
    int32_t beg = -1, end, nb = 0;
    bool updated = true;
    for (;;) {
        uint32_t beg_pos = beg < 1? 0u : (mp[beg - 1] >> 1) + (uint32_t) K;
        beg_pos += EC_MIN_ERR_SEQ_LEN;
        for (end = beg + 1; end < n; ++end)
            if (!scm_del[km[end] >> 1] && !(km[end] & 1ULL) && (mp[end] >> 1) >= beg_pos) break;
        if (beg >= 0 || end < n) {
            EcBlock b;
            if (beg < 0) {
                beg = end;
                b.beg_utg = (km[beg] & ~1ULL) | (uint64_t) !(mp[beg] & 1u);
                b.beg_pos = 0, b.end_utg = EC_NONE, b.l = (int32_t) (mp[beg] >> 1), b.r = 1;
            } else {
                --beg;
                b.beg_utg = (km[beg] & ~1ULL) | (mp[beg] & 1u);
                b.beg_pos = (mp[beg] >> 1) + (uint32_t) K;
                if (end >= n) b.end_utg = EC_NONE, b.l = (int32_t) hoco_l - (int32_t) b.beg_pos;
                else b.end_utg = (km[end] & ~1ULL) | (mp[end] & 1u), b.l = (int32_t) (mp[end] >> 1) - (int32_t) b.beg_pos;
                b.r = 0;
            }
            b.beg = beg, b.end = end;
            on_block(nb, b);
            ++nb;
        } else {
            updated = false;
        }
        for (beg = end + 1; beg < n; ++beg)
            if (scm_del[km[beg] >> 1] || (km[end] & 1ULL)) break;     // [end], as written in the reference (syncerr.c:579)
        if (beg > n) break;
        on_copy(end, beg);
    }
    return updated? nb : -1;
}

struct EcBlockOut {
    uint32_t status;              // EC_*; EC_FAILURE also for blocks shorter than EC_MIN_ERR_SEQ_LEN
    uint32_t np;                  // entries of the optimum path
    uint64_t path_off;            // into the path pool
    uint32_t flags;               // 1 = did not fit the scratch slab (must be re-run with a large one)
    uint32_t short_block;         // 1 = l < EC_MIN_ERR_SEQ_LEN (stats[10])
    uint32_t tried, n_path;       // the search's effort: arcs followed (DFS steps), dead ends counted (syncerr.c:147) -- OATK_BUF_EC_BLOCK_OUT
    uint32_t wf_steps, wf_diag;   // ... wavefront steps taken, and the diagonals they covered in all (>> 6: units of 64)
    uint32_t ticks, tier;         // ... the time the wave that finished the block spent on it (s_memrealtime, 100 MHz), and the tier it ran in
};

// The arcs the search may follow: the graph's arc array with the deleted arcs squeezed out (same order), each carrying
// what the search needs to know about its target, so that one 32-byte load per arc is the only graph access.  At high
// coverage a good syncmer has thousands of deleted arcs to one-off error syncmers; the search never sees them.
struct __attribute__((aligned(32))) EcLiveArc {
    uint32_t w, ls;               // target oriented vertex, overlap
    uint32_t hs16, mpos;          // the target's k-mer: hoco byte offset / 16 of its read, pos << 1 | rev on it
    uint32_t lp, ln;              // the target's own live arcs
    uint32_t pad[2];
};
struct EcLive {
    const uint32_t *idx_p, *idx_n;  // [2 n_vtx] first live arc / live arc count of an oriented vertex
    const EcLiveArc *arc;
};

struct __attribute__((aligned(16))) EcWork {   // one block, self-contained for the solver
    uint64_t beg_utg, end_utg;
    uint32_t read, beg_pos;
    int32_t l, r;
    uint32_t hs16;                // hoco byte offset / 16 of the read
    uint32_t lp, ln;              // live arcs of beg_utg
    uint32_t pad;                 // (list kernel -> ec_new_n_kernel) chain entries the read keeps if this block is NOT corrected (syncerr.c:533-542)
};

__global__ void ec_live_flag_kernel(uint64_t n_arc, const uint8_t *arc_del, uint32_t *live)
{
    uint64_t a = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (a < n_arc) live[a] = !arc_del[a];
}
// ... and whether the live graph BRANCHES anywhere: *branches becomes nonzero when some oriented vertex has more than one live arc out (api_ec.inc: a graph that
// does not is searched by walking, and the solver without budgets and second stage is the faster one for it)
__global__ void ec_live_idx_kernel(uint64_t n_ovtx, const uint64_t *idx_p, const uint32_t *idx_n, const uint64_t *live_off, uint32_t *lidx_p, uint32_t *lidx_n, uint32_t *branches)
{
    uint64_t v = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_ovtx) return;
    const uint32_t n = idx_n[v];
    const uint64_t p = n? live_off[idx_p[v]] : 0;
    const uint32_t ln = n? (uint32_t) (live_off[idx_p[v] + n] - p) : 0u;
    lidx_p[v] = (uint32_t) p, lidx_n[v] = ln;
    if (ln > 1u && *(volatile uint32_t *) branches == 0u) atomicOr(branches, 1u);       // (read first: a graph may have millions of them, and they would all queue on one address)
}
__global__ void ec_live_arc_kernel(uint64_t n_arc, const uint8_t *arc_del, const uint64_t *live_off, const uint64_t *arc_w, const uint32_t *arc_ls,
                                   const uint64_t *vtx_hs_off, const uint32_t *vtx_mpos, const uint32_t *lidx_p, const uint32_t *lidx_n, EcLiveArc *larc,
                                   uint32_t *no_src)
{
    uint64_t a = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_arc || arc_del[a]) return;
    const uint64_t w = arc_w[a];
    if (vtx_hs_off[w >> 1] == EC_NO_SRC) *no_src = 1u;          // sharded reads: the k-mer of a live vertex was never imported
    EcLiveArc x;
    x.w = (uint32_t) w, x.ls = arc_ls[a], x.hs16 = (uint32_t) (vtx_hs_off[w >> 1] >> 4), x.mpos = vtx_mpos[w >> 1];
    x.lp = lidx_p[w], x.ln = lidx_n[w], x.pad[0] = x.pad[1] = 0;
    larc[live_off[a]] = x;
}

// ---- assemble the corrected chains (syncerr.c:513-542, :585-612): pass 0 counts, pass 1 writes ----
struct EcAssembleArgs {
    EcReads rd;
    const uint8_t *scm_del;
    const uint64_t *scm_s;
    const uint64_t *blk_off;
    const EcBlockOut *out;
    const uint64_t *path_pool;
    uint32_t *new_n;              // [n_reads]
    const uint64_t *new_off;      // [n_reads + 1] (pass 1)
    uint64_t *new_k_mer, *new_s_mer;
    uint32_t *new_m_pos;
    uint32_t *key_id;             // pass 1: the (syncmer, occurrence) pairs update_syncmer_db sorts, written along with the chains
    uint64_t *val_occ;
    uint64_t sid0;
    const uint64_t *old_s_mer;
    unsigned long long *stats;    // [11]
    int pass;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// The same walk with ONE WAVE PER READ.  A HiFi read carries a few dozen syncmers, so its whole chain sits in the lanes of a wave: lane j
// holds entry j, the two searches of the loop at syncerr.c:394-598 ("first good syncmer at or beyond a position", "first deleted syncmer
// after it") are one ballot and a count-trailing-zeros each, and everything the callbacks write goes out with the lanes side by side
// instead of one lane striding through its own read.  Four kernels use it -- block count, block list, and the two passes of the chain
// assembly -- and the lane-per-read versions above remain for reads with more than 64 syncmers (lane 0 runs them).
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ecr_u32(uint32_t v, int lane) { return (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) __shfl((int32_t) v, lane)); }
__device__ __forceinline__ uint64_t ecr_u64(uint64_t v, int lane) { return (uint64_t) ecr_u32((uint32_t) (v >> 32), lane) << 32 | ecr_u32((uint32_t) v, lane); }
__device__ __forceinline__ uint64_t ecr_above(int32_t j) { return j < 0? ~0ULL : (j >= 63? 0ULL : ~0ULL << (j + 1)); }      // lanes > j

// km, mp, del: this lane's chain entry (lane < n <= 64); every argument of the callbacks is wave-uniform
template <class FB, class FC>
__device__ int ec_blocks_wave(int lane, int32_t n, uint64_t km, uint32_t mp, bool del, uint32_t hoco_l, int K, FB on_block, FC on_copy)
{
    const bool in = lane < n;
    const uint64_t goodm = __ballot(in && !del && !(km & 1ULL)), delm = __ballot(in && del);
    const uint32_t pos = mp >> 1;
    int32_t beg = -1, end, nb = 0;
    bool updated = true;
    for (;;) {
        uint32_t beg_pos = beg < 1? 0u : ecr_u32(pos, beg - 1) + (uint32_t) K;
        beg_pos += EC_MIN_ERR_SEQ_LEN;
        const uint64_t m = __ballot(pos >= beg_pos) & goodm & ecr_above(beg);
        end = m? (int32_t) __builtin_ctzll(m) : (beg + 1 < n? n : beg + 1);
        if (beg >= 0 || end < n) {
            EcBlock b;
            if (beg < 0) {
                beg = end;
                const uint64_t kb = ecr_u64(km, beg);
                const uint32_t mb = ecr_u32(mp, beg);
                b.beg_utg = (kb & ~1ULL) | (uint64_t) !(mb & 1u);
                b.beg_pos = 0, b.end_utg = EC_NONE, b.l = (int32_t) (mb >> 1), b.r = 1;
            } else {
                --beg;
                const uint64_t kb = ecr_u64(km, beg);
                const uint32_t mb = ecr_u32(mp, beg);
                b.beg_utg = (kb & ~1ULL) | (mb & 1u);
                b.beg_pos = (mb >> 1) + (uint32_t) K;
                if (end >= n) b.end_utg = EC_NONE, b.l = (int32_t) hoco_l - (int32_t) b.beg_pos;
                else {
                    const uint64_t ke = ecr_u64(km, end);
                    const uint32_t me = ecr_u32(mp, end);
                    b.end_utg = (ke & ~1ULL) | (me & 1u), b.l = (int32_t) (me >> 1) - (int32_t) b.beg_pos;
                }
                b.r = 0;
            }
            b.beg = beg, b.end = end;
            on_block(nb, b);
            ++nb;
        } else {
            updated = false;
        }
        if (end + 1 < n) {
            if (ecr_u64(km, end) & 1ULL) beg = end + 1;                   // [end], as written in the reference (syncerr.c:579)
            else { const uint64_t m2 = delm & ecr_above(end); beg = m2? (int32_t) __builtin_ctzll(m2) : n; }
        } else {
            beg = end + 1;
        }
        if (beg > n) break;
        on_copy(end, beg);
    }
    return updated? nb : -1;
}

// A wave takes ECR_RPW consecutive reads: everything the walks need of them is requested first (the loads of all of them are in flight
// together), then the reads are walked one after the other.  Measured at config 3 (2 M reads): with two reads per wave the count + list
// kernels take what they take with one (2.9 ms) and the assembly passes are slower (7.0 against 6.3 ms) -- the walks are not bound by the
// rate at which waves launch but by what each does once its data is there -- so a wave takes ONE read.
#ifndef ECR_RPW
#define ECR_RPW 1
#endif
#define ECR_READS_PER_BLOCK (4 * ECR_RPW)

struct EcrRead {                  // one read of the wave's, as the lanes hold it
    uint64_t r, o;
    int32_t n;                    // -1: no such read
    uint64_t km;
    uint32_t mp, hoco_l;
    bool del;
};
__device__ __forceinline__ void ecr_load(const EcReads &rd, const uint8_t *scm_del, uint64_t r0, int lane, EcrRead (&q)[ECR_RPW])
{
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        q[k].r = r0 + k;
        const bool live = q[k].r < rd.n_reads;
        q[k].o = live? rd.scm_off[q[k].r] : 0;
        q[k].n = live? (int32_t) (rd.scm_off[q[k].r + 1] - q[k].o) : -1;
        q[k].hoco_l = live? rd.hoco_l[q[k].r] : 0;
    }
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        const bool in = q[k].n >= 0 && q[k].n <= 64 && lane < q[k].n;
        q[k].km = in? rd.k_mer[q[k].o + lane] : 0;
        q[k].mp = in? rd.m_pos[q[k].o + lane] : 0;
    }
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) q[k].del = q[k].n >= 0 && q[k].n <= 64 && lane < q[k].n && scm_del[q[k].km >> 1];
}

__global__ __launch_bounds__(256) void ec_count_blocks_wave_kernel(EcReads rd, const uint8_t *scm_del, uint32_t *n_blocks)
{
    const uint64_t r0 = ((uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * ECR_RPW;
    const int lane = threadIdx.x & 63;
    if (r0 >= rd.n_reads) return;
    EcrRead q[ECR_RPW];
    ecr_load(rd, scm_del, r0, lane, q);
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        const EcrRead &x = q[k];
        if (x.n < 0) continue;
        int nb = 0;
        if (x.n > 64) {
            if (lane == 0) {
                ec_blocks(scm_del, rd.k_mer + x.o, rd.m_pos + x.o, x.n, x.hoco_l, rd.K, [&](int, const EcBlock &) { ++nb; }, [](int32_t, int32_t) {});
                n_blocks[x.r] = (uint32_t) nb;
            }
            continue;
        }
        ec_blocks_wave(lane, x.n, x.km, x.mp, x.del, x.hoco_l, rd.K, [&](int, const EcBlock &) { ++nb; }, [](int32_t, int32_t) {});
        if (lane == 0) n_blocks[x.r] = (uint32_t) nb;
    }
}

__global__ __launch_bounds__(256) void ec_list_blocks_wave_kernel(EcReads rd, EcLive lv, const uint8_t *scm_del, const uint64_t *blk_off, EcWork *work, uint32_t *copy_n)
{
    const uint64_t r0 = ((uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * ECR_RPW;
    const int lane = threadIdx.x & 63;
    if (r0 >= rd.n_reads) return;
    EcrRead q[ECR_RPW];
    ecr_load(rd, scm_del, r0, lane, q);
    EcWork *wq[ECR_RPW];
    uint32_t hq[ECR_RPW];
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) wq[k] = q[k].n >= 0? work + blk_off[q[k].r] : nullptr, hq[k] = q[k].n >= 0? (uint32_t) (rd.off[q[k].r] >> 6) : 0u;
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        const EcrRead &x = q[k];
        if (x.n < 0) continue;
        EcWork *w = wq[k];
        const uint32_t hs16 = hq[k];
        const int32_t n = x.n;
        auto put = [&](int i, const EcBlock &b) {
            EcWork y;
            y.beg_utg = b.beg_utg, y.end_utg = b.end_utg, y.read = (uint32_t) x.r, y.beg_pos = b.beg_pos, y.l = b.l, y.r = b.r;
            y.hs16 = hs16, y.lp = lv.idx_p[b.beg_utg], y.ln = lv.idx_n[b.beg_utg];
            // what the assembly copies when the block is not corrected (ec_assemble_wave_kernel: copy(0, beg) / copy(beg + 1, min(end, n)))
            const int32_t last = b.end < n? b.end : n;
            y.pad = b.r? (uint32_t) b.beg : (b.beg + 1 < n && last > b.beg + 1? (uint32_t) (last - b.beg - 1) : 0u);
            w[i] = y;
        };
        // ... and what it copies between the blocks whatever becomes of them; a read without a good syncmer keeps its chain (syncerr.c:562-572)
        uint32_t cp = 0;
        auto on_copy = [&](int32_t first, int32_t last) { if (last > first) cp += (uint32_t) (last - first); };
        if (x.n > 64) {
            if (lane == 0) {
                const int nbs = ec_blocks(scm_del, rd.k_mer + x.o, rd.m_pos + x.o, x.n, x.hoco_l, rd.K, put, on_copy);
                copy_n[x.r] = nbs < 0? (uint32_t) n : cp;
            }
            continue;
        }
        // lane i keeps block i and writes it after the walk, so the gathers of a read's blocks are in flight together
        EcBlock mine;
        mine.beg_utg = 0, mine.end_utg = 0, mine.beg_pos = 0, mine.l = 0, mine.beg = 0, mine.end = 0, mine.r = 0;
        const int nb = ec_blocks_wave(lane, x.n, x.km, x.mp, x.del, x.hoco_l, rd.K,
                                      [&](int i, const EcBlock &b) { if (i < 64) { if (lane == i) mine = b; } else if (lane == 0) put(i, b); }, on_copy);
        if (lane < nb) put(lane, mine);
        if (lane == 0) copy_n[x.r] = nb < 0? (uint32_t) n : cp;
    }
}

// the corrected chains' lengths from the blocks' outcomes (r04; until then a third walk over the chains, ec_assemble_wave_kernel<0>): what the read keeps between
// its blocks, plus per block the optimum path's interior (syncerr.c:513-532) or the originals it keeps
__global__ __launch_bounds__(256) void ec_new_n_kernel(uint64_t n_reads, const uint32_t *copy_n, const uint64_t *blk_off, const EcWork *work, const EcBlockOut *out, uint32_t *new_n)
{
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint32_t s = copy_n[r];
    for (uint64_t i = blk_off[r]; i < blk_off[r + 1]; ++i) {
        if (out[i].status == EC_SUCCESS) {
            const int32_t np = (int32_t) out[i].np;
            s += work[i].r? (uint32_t) (np >= 1? np - 1 : 0) : (uint32_t) ((np >= 2? np - 2 : 0) + (work[i].end_utg == EC_NONE && np > 1? 1 : 0));
        } else s += work[i].pad;
    }
    new_n[r] = s;
}

// stats[11] of read_error_correction (syncerr.c:502-504, :513-542) from the solved blocks; a small fixed grid strides over them and every
// workgroup adds its eleven sums once (eleven addresses take atomics one at a time: 90 k of them cost 0.4 ms, 3 k nothing)
__global__ __launch_bounds__(256) void ec_block_stats_kernel(const EcWork *work, const EcBlockOut *out, uint64_t n_work, unsigned long long *stats)
{
    __shared__ uint32_t part[4][11];
    uint32_t loc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_work; i += (uint64_t) gridDim.x * blockDim.x) {
        const EcBlockOut x = out[i];
        if (x.short_block) ++loc[10];
        else if (work[i].end_utg == EC_NONE) ++loc[0], ++loc[1 + x.status];
        else ++loc[5], ++loc[6 + x.status];
    }
    for (int i = 0; i < 11; ++i) {
        uint32_t v = loc[i];
        for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        const uint32_t v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v) atomicAdd(&stats[threadIdx.x], (unsigned long long) v);
    }
}

// one read, lane-serial (the body of ec_assemble_kernel without the statistics): reads with more than 64 syncmers
__device__ inline void ec_assemble_read_serial(const EcAssembleArgs &a, uint64_t r)
{
    const uint64_t o = a.rd.scm_off[r];
    const int32_t n = (int32_t) (a.rd.scm_off[r + 1] - o);
    const uint64_t *km = a.rd.k_mer + o;
    const uint32_t *mp = a.rd.m_pos + o;
    const EcBlockOut *bo = a.out + a.blk_off[r];
    uint64_t wpos = a.pass? a.new_off[r] : 0;
    const uint64_t w0 = wpos;
    uint32_t cnt = 0;
    auto put = [&](uint64_t k, uint32_t m) {
        if (a.pass) {
            a.new_k_mer[wpos] = k, a.new_m_pos[wpos] = m, a.new_s_mer[wpos] = a.scm_s[k >> 1];
            a.key_id[wpos] = (uint32_t) (k >> 1), a.val_occ[wpos] = (a.sid0 + r) << 32 | (wpos - w0) << 1 | (m & 1u);
            ++wpos;
        }
        ++cnt;
    };
    int nb = ec_blocks(a.scm_del, km, mp, n, a.rd.hoco_l[r], a.rd.K,
        [&](int k, const EcBlock &b) {
            const EcBlockOut &x = bo[k];
            if (x.status == EC_SUCCESS) {
                const uint64_t *path = a.path_pool + x.path_off;
                const int32_t np = (int32_t) x.np;
                if (b.r) {
                    for (int32_t j = np - 1; j > 0; --j) put((path[j] & ~1ULL) | 1ULL, 0xFFFFFFFFu ^ (uint32_t) (path[j] & 1ULL));
                } else {
                    int32_t j;
                    for (j = 1; j < np - 1; ++j) put((path[j] & ~1ULL) | 1ULL, 0xFFFFFFFEu | (uint32_t) (path[j] & 1ULL));
                    if (b.end_utg == EC_NONE && np > 1) put((path[j] & ~1ULL) | 1ULL, 0xFFFFFFFEu | (uint32_t) (path[j] & 1ULL));
                }
            } else if (b.r) {
                for (int32_t j = 0; j < b.beg; ++j) put(km[j], mp[j]);
            } else if (b.beg + 1 < n) {
                for (int32_t j = b.beg + 1; j < b.end; ++j) put(km[j], mp[j]);
            }
        },
        [&](int32_t first, int32_t last) { for (int32_t j = first; j < last; ++j) put(km[j], mp[j]); });
    if (nb < 0) {
        if (a.pass) {
            uint64_t q = a.new_off[r];
            for (int32_t j = 0; j < n; ++j) {
                a.new_k_mer[q + j] = km[j], a.new_m_pos[q + j] = mp[j], a.new_s_mer[q + j] = a.old_s_mer[o + j];
                a.key_id[q + j] = (uint32_t) (km[j] >> 1), a.val_occ[q + j] = (a.sid0 + r) << 32 | (uint64_t) j << 1 | (mp[j] & 1u);
            }
        }
        cnt = (uint32_t) n;
    }
    if (!a.pass) a.new_n[r] = cnt;
}

// the corrected chains (syncerr.c:513-542, :585-612), pass 0 counts and pass 1 writes, one wave per read
template <int PASS>
__global__ __launch_bounds__(256) void ec_assemble_wave_kernel(EcAssembleArgs a)
{
    const uint64_t r0 = ((uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * ECR_RPW;
    const int lane = threadIdx.x & 63;
    if (r0 >= a.rd.n_reads) return;
    EcrRead q[ECR_RPW];
    ecr_load(a.rd, a.scm_del, r0, lane, q);
    // lane i holds the outcome of block i of each read: one load for all of a read's blocks instead of one after the other inside the walk
    const EcBlockOut *boq[ECR_RPW];
    uint32_t st_q[ECR_RPW], np_q[ECR_RPW];
    uint64_t path_q[ECR_RPW], w0_q[ECR_RPW], s_q[ECR_RPW];
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        const bool live = q[k].n >= 0;
        const uint64_t b0 = live? a.blk_off[q[k].r] : 0;
        const int32_t nbk = live? (int32_t) (a.blk_off[q[k].r + 1] - b0) : 0;
        boq[k] = a.out + b0;
        st_q[k] = EC_FAILURE, np_q[k] = 0, path_q[k] = 0;
        if (lane < nbk) st_q[k] = boq[k][lane].status, np_q[k] = boq[k][lane].np, path_q[k] = boq[k][lane].path_off;
        w0_q[k] = PASS && live? a.new_off[q[k].r] : 0;
        s_q[k] = PASS && live && q[k].n <= 64 && lane < q[k].n? a.old_s_mer[q[k].o + lane] : 0;    // a syncmer's s-mer is the same at every occurrence (count.hpp: check_smer_kernel)
    }
#pragma unroll
    for (int k = 0; k < ECR_RPW; ++k) {
        const EcrRead &x = q[k];
        if (x.n < 0) continue;
        if (x.n > 64) {
            if (lane == 0) ec_assemble_read_serial(a, x.r);
            continue;
        }
        const int32_t n = x.n;
        const uint64_t km = x.km, my_s = s_q[k], w0 = w0_q[k], my_path = path_q[k];
        const uint32_t mp = x.mp, my_status = st_q[k], my_np = np_q[k];
        const EcBlockOut *bo = boq[k];
        uint64_t wpos = w0;
        const uint64_t sid = (a.sid0 + x.r) << 32;
        auto write = [&](uint64_t at, uint64_t kk, uint32_t m, uint64_t sv) {
            a.new_k_mer[at] = kk, a.new_m_pos[at] = m, a.new_s_mer[at] = sv;
            a.key_id[at] = (uint32_t) (kk >> 1), a.val_occ[at] = sid | (at - w0) << 1 | (m & 1u);                 // syncerr.c:796-805
        };
        auto copy = [&](int32_t first, int32_t last) {                     // original entries [first, last) stay
            if (last <= first) return;
            if (PASS && lane >= first && lane < last) write(wpos + (uint32_t) (lane - first), km, mp, my_s);
            wpos += (uint32_t) (last - first);
        };
        const int nb = ec_blocks_wave(lane, n, km, mp, x.del, x.hoco_l, a.rd.K,
            [&](int i, const EcBlock &b) {
                const uint32_t status = i < 64? ecr_u32(my_status, i) : bo[i].status;
                if (status == EC_SUCCESS) {
                    const int32_t np = (int32_t) (i < 64? ecr_u32(my_np, i) : bo[i].np);
                    const uint64_t *path = a.path_pool + (i < 64? ecr_u64(my_path, i) : bo[i].path_off);
                    int32_t c;
                    if (b.r) c = np >= 1? np - 1 : 0;
                    else c = (np >= 2? np - 2 : 0) + (b.end_utg == EC_NONE && np > 1? 1 : 0);
                    if (PASS) {
                        for (int32_t t = lane; t < c; t += 64) {
                            const uint64_t p = b.r? path[np - 1 - t] : path[1 + t];
                            const uint64_t kk = (p & ~1ULL) | 1ULL;
                            write(wpos + (uint32_t) t, kk, b.r? 0xFFFFFFFFu ^ (uint32_t) (p & 1ULL) : 0xFFFFFFFEu | (uint32_t) (p & 1ULL), a.scm_s[kk >> 1]);
                        }
                    }
                    wpos += (uint32_t) c;
                } else if (b.r) {
                    copy(0, b.beg);
                } else if (b.beg + 1 < n) {
                    copy(b.beg + 1, b.end < n? b.end : n);
                }
            },
            copy);
        if (nb < 0) {                                    // no good syncmer: the read keeps its arrays (syncerr.c:562-572)
            wpos = w0;
            if (PASS && lane < n) write(w0 + (uint32_t) lane, km, mp, my_s);
            wpos += (uint32_t) n;
        }
        if (!PASS && lane == 0) a.new_n[x.r] = (uint32_t) (wpos - w0);
    }
}

// ---- update_syncmer_db (syncerr.c:769-814): coverage, forward-strand presence; occurrence lists come from a stable sort ----
__global__ void ec_fwd_flag_kernel(uint64_t tot, const uint64_t *occ, uint32_t *flag)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) flag[i] = !(occ[i] & 1ULL);
}
__global__ void ec_cov_sorted_kernel(uint64_t tot, const uint32_t *key_sorted, const uint32_t *fwd_incl, uint32_t *cov, uint32_t *fwd)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const uint32_t k = key_sorted[i];
    if (i + 1 < tot && key_sorted[i + 1] == k) return;
    uint64_t lo = 0, hi = i;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (key_sorted[mid] < k) lo = mid + 1; else hi = mid;
    }
    cov[k] = (uint32_t) (i - lo + 1);
    fwd[k] = fwd_incl[i] - (lo? fwd_incl[lo - 1] : 0u);
}

__global__ void ec_del_kernel(uint64_t n, const uint32_t *fwd, uint8_t *del)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) del[i] = !fwd[i];
}

}  // namespace oatk
