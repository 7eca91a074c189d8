// oatk_amd/csrc/ec.hpp -- per-read syncmer-chain error correction on the device.
//
// Replaces, on flat arrays that stay resident in HBM:
//   find_error_syncmers            syncerr.c:679-757   -> ec_mark_kernel, ec_arc_del_kernel
//   error blocks of a read         syncerr.c:339-612   -> ec_blocks (device function, shared by three kernels)
//   dfs_search + wf_ed_core        syncerr.c:144-286, levdist.c:75-310 -> ec_solve_kernel
//   update_syncmer_db              syncerr.c:769-814   -> ec_cov_kernel (+ a stable radix sort in api.hip)
//
// The graph is the reference's asmg_t in arc-array order (sorted (v,w), graph.c:70-83) as CSR over oriented vertices.
// Every vertex is one syncmer (utg id == syncmer id, syncerr.c:421-423) and its hoco consensus is the oriented k-mer of
// the syncmer's first occurrence (scg_syncmer_consensus in hoco mode, syncasm.c:910-940) -- so no 1002-byte string per
// vertex is materialised: bases are read in place from the resident hoco strings of the reads.
//
// Mapping: error blocks of one read are independent (they are delimited on the ORIGINAL chain), so the unit of work is
// one block = one lane: a depth-first search over the good-syncmer graph that extends a consensus string arc by arc
// and re-aligns it to the read segment with a resumable Landau-Vishkin wavefront.  Per-lane state (strings, DFS frames
// with the saved wavefront, paths) lives in a private scratch slab in HBM.  Blocks that outgrow the slab are flagged
// and re-run with large slabs.  This is irregular, latency-bound work (~4 % of the reference's CPU time); correctness
// (bit-identical chains) is the bar here, not a roofline.
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
        if (!live) continue;
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
__global__ void ec_vtx_src_kernel(uint64_t n_vtx, const uint64_t *occ_off, const uint64_t *occ, uint64_t sid0, const uint64_t *off,
                                  const uint64_t *scm_off, const uint32_t *m_pos, uint64_t *vtx_hs_off, uint32_t *vtx_mpos)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vtx) return;
    const uint64_t o = occ[occ_off[i]], rd = (o >> 32) - sid0, idx = (uint32_t) o >> 1;
    vtx_hs_off[i] = off[rd] >> 2;
    vtx_mpos[i] = m_pos[scm_off[rd] + idx];
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
};

// counts blocks per read (block slots are then laid out by a prefix sum)
__global__ void ec_count_blocks_kernel(EcReads rd, const uint8_t *scm_del, uint32_t *n_blocks)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rd.n_reads) return;
    const uint64_t o = rd.scm_off[r];
    const int32_t n = (int32_t) (rd.scm_off[r + 1] - o);
    int nb = 0;
    ec_blocks(scm_del, rd.k_mer + o, rd.m_pos + o, n, rd.hoco_l[r], rd.K, [&](int, const EcBlock &) { ++nb; }, [](int32_t, int32_t) {});
    n_blocks[r] = (uint32_t) nb;
}

struct EcWork {
    uint32_t read;
    EcBlock b;
};

__global__ void ec_list_blocks_kernel(EcReads rd, const uint8_t *scm_del, const uint64_t *blk_off, EcWork *work)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rd.n_reads) return;
    const uint64_t o = rd.scm_off[r];
    const int32_t n = (int32_t) (rd.scm_off[r + 1] - o);
    EcWork *w = work + blk_off[r];
    ec_blocks(scm_del, rd.k_mer + o, rd.m_pos + o, n, rd.hoco_l[r], rd.K,
              [&](int k, const EcBlock &b) { w[k].read = (uint32_t) r; w[k].b = b; }, [](int32_t, int32_t) {});
}

// ---- the solver ----
struct EcScratch {                // carved out of one lane's slab
    uint8_t *ts, *cs, *os;        // target, current consensus, optimum consensus (one base code per byte)
    int32_t cap_t, cap_c;
    uint64_t *c_path;
    int32_t cap_path;
    int32_t *wd, *wk, *nd, *nk;   // working wavefront + next
    int32_t cap_w;
    uint8_t *frames;              // LIFO arena of DFS frames
    int32_t cap_f;
};

struct EcFrame {                  // state at the entry of one DFS level (syncerr.c:158-171)
    uint64_t arc_i, arc_end;
    int32_t l0, score, t_end, q_end, n, prev_off;      // prev_off: arena offset of the parent's frame (-1 for the root)
};

// one wavefront step (levdist.c:156-224, extension mode, no traceback); returns 1 when an end was reached
__device__ int ec_wf_step(const uint8_t *ts, int32_t tl, const uint8_t *qs, int32_t ql, int32_t bw, int32_t *d, int32_t *k, int32_t *nd, int32_t *nk,
                          int32_t &n, int32_t &t_end, int32_t &q_end)
{
    t_end = q_end = -1;
    for (int32_t j = 0; j < n; ++j) {
        int32_t kk = k[j];
        const int32_t dd = d[j];
        if (kk >= tl || kk + dd >= ql) continue;
        const int32_t lim = (ql - dd < tl? ql - dd : tl) - 1;
        while (kk < lim && ts[kk + 1] == qs[kk + dd + 1]) ++kk;
        if (kk + dd == ql - 1 || kk == tl - 1) { t_end = kk, q_end = kk + dd; return 1; }
        k[j] = kk;
    }
    nd[0] = d[0] - 1, nk[0] = k[0] + 1;
    nd[1] = d[0], nk[1] = ((n == 1 || k[0] > k[1])? k[0] : k[1]) + 1;
    for (int32_t j = 1; j < n - 1; ++j) {
        int32_t kk = k[j - 1];
        if (k[j] + 1 > kk) kk = k[j] + 1;
        if (k[j + 1] + 1 > kk) kk = k[j + 1] + 1;
        nd[j + 1] = d[j], nk[j + 1] = kk;
    }
    if (n >= 2) nd[n] = d[n - 1], nk[n] = k[n - 2] > k[n - 1] + 1? k[n - 2] : k[n - 1] + 1;
    nd[n + 1] = d[n - 1] + 1, nk[n + 1] = k[n - 1];
    int32_t st = 0, en = n + 2;
    if (bw < 0 || n < 2 * bw + 1) {
        if (nd[0] < -tl) ++st;
        if (nd[n + 1] > ql) --en;
    } else {
        const int32_t lo = -bw > -tl? -bw : -tl, hi = bw > ql? bw : ql;     // the LARGER of bw and ql, as in levdist.c:108
        while (nd[st] < lo) ++st;
        while (nd[en - 1] > hi) --en;
    }
    n = en - st;
    for (int32_t j = 0; j < n; ++j) d[j] = nd[st + j], k[j] = nk[st + j];
    return 0;
}

// Solve one block.  Returns false when the scratch slab is too small (nothing is written then).
__device__ bool ec_solve_block(const EcGraph &g, const EcReads &rd, const EcWork &wk, const EcScratch &s, double max_edist,
                               uint32_t &status_out, uint32_t &np_out, uint64_t *path_out, int32_t path_cap)
{
    const EcBlock &b = wk.b;
    const int K = rd.K;
    const int32_t tl = b.l;
    int32_t bw = (int32_t) ceil((double) tl * max_edist);
    if (bw < EC_MIN_ERR_BASE) bw = EC_MIN_ERR_BASE;
    if (tl > s.cap_t || 2 * bw + 8 > s.cap_w) return false;
    // target: the read segment, reverse-complemented for a leading block (get_kmer_dna_seq, syncmer.c:1237)
    const uint8_t *hs = rd.hoco_s + (rd.off[wk.read] >> 2);
    for (int32_t i = 0; i < tl; ++i)
        s.ts[i] = b.r? (uint8_t) (3u ^ hoco_base(hs, b.beg_pos + (uint32_t) (tl - 1 - i))) : (uint8_t) hoco_base(hs, b.beg_pos + (uint32_t) i);

    int32_t status = EC_FAILURE, n_path = 0, edist = INT32_MAX, s_edist = INT32_MAX;
    int32_t c_len = 0, o_len = 0, np = 0;            // consensus length, optimum consensus length, optimum path entries
    int32_t depth = 0;                               // c_path holds depth + 1 entries while iterating a level
    int32_t fsz = 0;                                 // bytes used in the frame arena
    // working alignment state
    int32_t score = 0, t_end = 0, q_end = 0, n = 1;
    s.wd[0] = 0, s.wk[0] = -1;
    s.c_path[0] = b.beg_utg;

    int32_t top = -1, nfr = 0;                       // arena offset of the innermost frame, number of frames
    auto push_frame = [&](uint64_t src) -> bool {
        const int32_t need = ((int32_t) sizeof(EcFrame) + 8 * n + 7) & ~7;
        if (fsz + need > s.cap_f) return false;
        EcFrame *f = (EcFrame *) (s.frames + fsz);
        f->arc_i = g.idx_p[src], f->arc_end = f->arc_i + g.idx_n[src];
        f->l0 = c_len, f->score = score, f->t_end = t_end, f->q_end = q_end, f->n = n, f->prev_off = top;
        int32_t *sv = (int32_t *) (f + 1);
        for (int32_t j = 0; j < n; ++j) sv[2 * j] = s.wd[j], sv[2 * j + 1] = s.wk[j];
        top = fsz;
        fsz += need;
        ++nfr;
        return true;
    };
    if (!push_frame(b.beg_utg)) return false;

    while (nfr > 0) {
        EcFrame *f = (EcFrame *) (s.frames + top);
        depth = nfr - 1;
        if (f->arc_i == f->arc_end) {                 // level exhausted: return to the parent
            fsz = top;
            top = f->prev_off;
            --nfr;
            continue;
        }
        const uint64_t a = f->arc_i++;
        if (g.arc_del[a]) continue;
        // restore the state this level was entered with (syncerr.c:277-284)
        c_len = f->l0, score = f->score, t_end = f->t_end, q_end = f->q_end, n = f->n;
        {
            const int32_t *sv = (const int32_t *) (f + 1);
            for (int32_t j = 0; j < n; ++j) s.wd[j] = sv[2 * j], s.wk[j] = sv[2 * j + 1];
        }
        const int32_t t_end0 = f->t_end;
        const uint64_t w = g.arc_w[a];
        const int32_t ls = (int32_t) g.arc_ls[a], ext = K - ls;
        if (depth + 2 > s.cap_path || c_len + ext > s.cap_c) return false;
        s.c_path[depth + 1] = w;
        int32_t cn = depth + 2;                       // entries in c_path
        {   // append the part of w's k-mer that lies beyond the overlap (syncerr.c:186-190)
            const uint8_t *vs = rd.hoco_s + g.vtx_hs_off[w >> 1];
            const uint32_t mp = g.vtx_mpos[w >> 1], pos = mp >> 1, vrev = mp & 1u;
            // forward string of the vertex: F[j] = vrev ? comp(base[pos + K-1-j]) : base[pos + j]
            for (int32_t t = 0; t < ext; ++t) {
                // w forward: F[ls + t];  w reverse: comp(F[K - ls - 1 - t])
                const int32_t j = (w & 1ULL)? K - ls - 1 - t : ls + t;
                uint32_t c = vrev? 3u ^ hoco_base(vs, pos + (uint32_t) (K - 1 - j)) : hoco_base(vs, pos + (uint32_t) j);
                if (w & 1ULL) c ^= 3u;
                s.cs[c_len + t] = (uint8_t) c;
            }
            c_len += ext;
        }
        // wf_ed_core (levdist.c:265-310)
        for (;;) {
            if (ec_wf_step(s.ts, tl, s.cs, c_len, bw, s.wd, s.wk, s.nd, s.nk, n, t_end, q_end)) break;
            ++score;
            if (score > bw) break;
        }
        t_end += 1, q_end += 1;
        const int32_t ql = c_len;
        const int32_t sc = score + tl - t_end;        // syncerr.c:209
        if (sc <= bw && (b.end_utg == EC_NONE || b.end_utg == w)) {
            status = EC_SUCCESS;
            if (sc <= edist) {
                if (t_end > t_end0) s_edist = edist;
                edist = sc;
                if (b.end_utg == EC_NONE && q_end < ql) --cn;
                if (edist == s_edist) {
                    bool diff = q_end != o_len;
                    for (int32_t i = 0; !diff && i < q_end; ++i) diff = s.cs[i] != s.os[i];
                    if (diff) status = EC_AMBISEQ;
                    if (status == EC_SUCCESS) {
                        bool pd = cn != np;
                        for (int32_t i = 0; !pd && i < cn; ++i) pd = s.c_path[i] != path_out[i];
                        if (pd) status = EC_AMBISNQ;
                    }
                }
                if (cn > path_cap) return false;
                for (int32_t i = 0; i < q_end; ++i) s.os[i] = s.cs[i];
                o_len = q_end;
                for (int32_t i = 0; i < cn; ++i) path_out[i] = s.c_path[i];
                np = cn;
            } else if (sc < s_edist) {
                s_edist = sc;
            }
        }
        if (score <= bw && ql - K <= tl + bw && ((b.end_utg != EC_NONE && b.end_utg != w) || t_end < tl)) {
            if (n_path < EC_MAX_DFS_PATH) {           // the callee would return at once otherwise (syncerr.c:146-148)
                if (!push_frame(w)) return false;
            }
        } else {
            ++n_path;
        }
    }
    status_out = (uint32_t) status, np_out = (uint32_t) np;
    return true;
}

struct EcSolveArgs {
    EcGraph g;
    EcReads rd;
    const EcWork *work;
    uint64_t n_work;
    const uint32_t *todo;         // optional list of work indices (second, large-slab pass); NULL = all
    uint64_t n_todo;
    double max_edist;
    uint8_t *slabs;               // one slab per launched lane
    uint64_t slab_bytes;
    int32_t cap_t, cap_path, cap_w, cap_f;
    EcBlockOut *out;              // [n_work]
    uint64_t *path_pool;          // optimum paths; bump-allocated
    uint64_t pool_cap;
    unsigned long long *pool_cursor;
};

__global__ __launch_bounds__(64) void ec_solve_kernel(EcSolveArgs a)
{
    const uint64_t lane_id = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, n_lanes = (uint64_t) gridDim.x * blockDim.x;
    uint8_t *slab = a.slabs + lane_id * a.slab_bytes;
    EcScratch s;
    const int32_t cap_c = a.cap_t + a.cap_t / 8 + 2 * a.rd.K + 64;
    uint8_t *p = slab;
    s.cap_t = a.cap_t, s.cap_c = cap_c, s.cap_path = a.cap_path, s.cap_w = a.cap_w, s.cap_f = a.cap_f;
    s.ts = p, p += (a.cap_t + 7) & ~7;
    s.cs = p, p += (cap_c + 7) & ~7;
    s.os = p, p += (cap_c + 7) & ~7;
    s.c_path = (uint64_t *) p, p += 8 * (size_t) a.cap_path;
    uint64_t *path_tmp = (uint64_t *) p; p += 8 * (size_t) a.cap_path;
    s.wd = (int32_t *) p, p += 4 * (size_t) a.cap_w;
    s.wk = (int32_t *) p, p += 4 * (size_t) a.cap_w;
    s.nd = (int32_t *) p, p += 4 * (size_t) a.cap_w;
    s.nk = (int32_t *) p, p += 4 * (size_t) a.cap_w;
    s.frames = p;
    const uint64_t total = a.todo? a.n_todo : a.n_work;
    for (uint64_t t = lane_id; t < total; t += n_lanes) {
        const uint64_t wi = a.todo? a.todo[t] : t;
        const EcWork &wk = a.work[wi];
        EcBlockOut o;
        o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0;
        if (wk.b.l < EC_MIN_ERR_SEQ_LEN) {
            o.short_block = 1;                         // syncerr.c:502-504
        } else {
            uint32_t st = 0, np = 0;
            if (!ec_solve_block(a.g, a.rd, wk, s, a.max_edist, st, np, path_tmp, a.cap_path)) {
                o.flags = 1;
            } else {
                o.status = st, o.np = np;
                if (st == EC_SUCCESS && np) {
                    const unsigned long long off = atomicAdd(a.pool_cursor, (unsigned long long) np);
                    o.path_off = off;
                    if (off + np <= a.pool_cap) for (uint32_t i = 0; i < np; ++i) a.path_pool[off + i] = path_tmp[i];
                }
            }
        }
        a.out[wi] = o;
    }
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
    const uint64_t *old_s_mer;
    unsigned long long *stats;    // [11]
    int pass;
};

__global__ void ec_assemble_kernel(EcAssembleArgs a)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.rd.n_reads) return;
    const uint64_t o = a.rd.scm_off[r];
    const int32_t n = (int32_t) (a.rd.scm_off[r + 1] - o);
    const uint64_t *km = a.rd.k_mer + o;
    const uint32_t *mp = a.rd.m_pos + o;
    const EcBlockOut *bo = a.out + a.blk_off[r];
    uint64_t wpos = a.pass? a.new_off[r] : 0;
    uint32_t cnt = 0;
    auto put = [&](uint64_t k, uint32_t m) {
        if (a.pass) { a.new_k_mer[wpos] = k, a.new_m_pos[wpos] = m, a.new_s_mer[wpos] = a.scm_s[k >> 1]; ++wpos; }
        ++cnt;
    };
    int nb = ec_blocks(a.scm_del, km, mp, n, a.rd.hoco_l[r], a.rd.K,
        [&](int k, const EcBlock &b) {
            const EcBlockOut &x = bo[k];
            if (a.pass == 0) {
                if (x.short_block) atomicAdd(&a.stats[10], 1ULL);
                else if (b.end_utg == EC_NONE) { atomicAdd(&a.stats[0], 1ULL); atomicAdd(&a.stats[1 + x.status], 1ULL); }
                else { atomicAdd(&a.stats[5], 1ULL); atomicAdd(&a.stats[6 + x.status], 1ULL); }
            }
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
    if (nb < 0) {                                    // no good syncmer: the read keeps its arrays (syncerr.c:562-572)
        if (a.pass) {
            uint64_t q = a.new_off[r];
            for (int32_t j = 0; j < n; ++j) a.new_k_mer[q + j] = km[j], a.new_m_pos[q + j] = mp[j], a.new_s_mer[q + j] = a.old_s_mer[o + j];
        }
        cnt = (uint32_t) n;
    }
    if (!a.pass) a.new_n[r] = cnt;
}

// ---- update_syncmer_db (syncerr.c:769-814): coverage, forward-strand presence; occurrence lists come from a stable sort ----
__global__ void ec_cov_kernel(uint64_t tot, const uint64_t *new_k_mer, const uint32_t *new_m_pos, uint32_t *cov, uint32_t *fwd)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const uint64_t k = new_k_mer[i] >> 1;
    atomicAdd(&cov[k], 1u);
    if (!(new_m_pos[i] & 1u)) atomicAdd(&fwd[k], 1u);
}

__global__ void ec_occ_keys_kernel(uint64_t n_reads, uint64_t sid0, const uint64_t *new_off, const uint64_t *new_k_mer, const uint32_t *new_m_pos,
                                   uint32_t *key_id, uint64_t *val_occ)
{
    uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    for (uint64_t i = new_off[r], j = 0; i < new_off[r + 1]; ++i, ++j) {
        key_id[i] = (uint32_t) (new_k_mer[i] >> 1);
        val_occ[i] = (sid0 + r) << 32 | j << 1 | (new_m_pos[i] & 1u);
    }
}

__global__ void ec_del_kernel(uint64_t n, const uint32_t *fwd, uint8_t *del)
{
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) del[i] = !fwd[i];
}

}  // namespace oatk
