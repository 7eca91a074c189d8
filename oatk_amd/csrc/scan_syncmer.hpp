// oatk_amd/csrc/scan_syncmer.hpp -- kernel B of the read scan: closed-syncmer selection + k-mer hash.
//
// Replaces the second half of the reference's per-read loop (syncmer.c:306-394) and `kmer_hash64`
// (syncmer.c:175-226).  Input is the 2-bit hoco string written by kernel A (scan_hpc.hpp); output is
// one record per closed syncmer: (sid<<32 | ordinal<<1 | rev, s-mer code, pos<<1 | rev); the MurmurHash64A of
// the oriented k-mer is added by kmer_hash_kernel.  Records are appended unordered; (sid, ordinal) makes them self-describing.
//
// The reference keeps a Q-slot ring with a tracked minimum and re-scans it when the minimum expires.
// Here every k-mer is decided independently from the array M[] of s-mer hashes (indexed by the END
// position of the s-mer), see DESIGN.md "Stateless closed-syncmer rule".  For the k-mer whose last base
// is E, with w = K - S:
//     y = M[E]  (last s-mer)   f = M[E-w] (first s-mer)   x = M[E-w-1]   b = min M[E-w .. E-1]
//     Close <=> valid && y != MAX && (y < b || (y == b && (x >= b || f == b)))
//     Open  <=> valid && next_base_ok && f != MAX && f <= b && f <= y
//     both   => neither (syncmer.c:337,393); Close is listed before Open.
//
// MI355X mapping: one 256-thread workgroup per read walks tiles of T = 256*C hoco positions.  M lives in
// an LDS ring (no halo recompute across tiles); b is a van-Herk style window minimum assembled from a
// lane's own chunk suffix, a sparse-table range minimum over chunk minima, and the far chunk's prefix.
// Selected syncmers (about one per 486 positions at K=1001) are compacted with a workgroup scan; their 251-byte
// k-mers are hashed afterwards by kmer_hash_kernel (kmer_hash.hpp), one lane per syncmer.
#pragma once
#include "common.hpp"

#define OATK_MI(i) ((((uint32_t) (i)) & (R - 1)) + ((((uint32_t) (i)) & (R - 1)) >> 5))

namespace oatk {

constexpr int SYN_NT = 256;
constexpr uint32_t OATK_REC_SHARDS = 1024;
constexpr int SYN_EM_CAP = 8;     // syncmers hashed per cooperative round

struct SynArgs {
    const uint8_t *hoco_s;    // read r at off[r] / 4
    uint32_t *nbits;          // ambiguous-base bitmap, read r at off[r] / 32 words; consumed and zeroed here
    const uint64_t *off;
    const uint32_t *hoco_l;
    const uint32_t *n_nn;     // ambiguous bases per read (0 selects the fast path)
    uint64_t sid0;
    int K, S;
    int want_n;               // 1: process only reads with ambiguous bases; 0: only reads without
    int list_cap;             // fast kernel: syncmers collected per read before records are written (<= SYF_LIST)
    uint32_t *n_scm;          // per read
    uint64_t *rec_hash, *rec_lo, *rec_smer;
    uint32_t *rec_mpos;
    // Records are appended to one of OATK_REC_SHARDS regions (shard = blockIdx & (SHARDS-1)), each with its own counter:
    // a single global counter took ~260 k atomics per 50 k reads and, at ~88 atomics/us on one address, WAS the kernel time.
    uint32_t region_cap;      // slots per shard region
    uint32_t *shard_cnt;      // [OATK_REC_SHARDS] records appended per shard (may exceed region_cap)
};

template <int C, int R, bool HAS_N>
__global__ __launch_bounds__(SYN_NT) void syncmer_kernel(SynArgs a)
{
    constexpr int T = SYN_NT * C;          // END positions per tile
    constexpr int NCH = R / C;             // chunk ring
    constexpr int PBW = R / 16;            // packed-base ring, 16 bases per word
    constexpr int NWAVE = SYN_NT / OATK_WAVE;

    // one pad slot per 32: lanes walk this ring with a stride of C slots, which would otherwise land on 4 banks
    __shared__ uint64_t m_ring[R + R / 32];
    __shared__ uint64_t cm_ring[NCH];
    __shared__ uint64_t st_a[NCH], st_b[NCH];
    __shared__ uint32_t pb[PBW];
    __shared__ uint16_t lr_ring[HAS_N? R : 1];
    __shared__ uint32_t nb_ring[HAS_N? R / 32 : 1];
    __shared__ uint32_t w_cnt[NWAVE];
    __shared__ int32_t w_max[NWAVE];
    __shared__ uint32_t s_gbase;

    const uint32_t r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t nn = a.n_nn[r];
    if ((nn != 0) != (a.want_n != 0)) return;

    const int K = a.K, S = a.S, w = K - S;
    const uint32_t hl = a.hoco_l[r];
    if (hl < (uint32_t) K) {               // no k-mer fits; still hand the bitmap back clean
        if (HAS_N) {
            uint32_t *gnb = a.nbits + (a.off[r] >> 5);
            for (uint32_t i = tid; i < (hl + 31u) / 32u; i += SYN_NT) gnb[i] = 0;
        }
        if (tid == 0) a.n_scm[r] = 0;
        return;
    }
    const uint64_t sid = a.sid0 + r;
    const uint64_t mask = (1ULL << (2 * S)) - 1;
    const uint32_t *ghs = (const uint32_t *) (a.hoco_s + (a.off[r] >> 2));
    uint32_t *gnb = a.nbits + (a.off[r] >> 5);
    const int D = (w - 1) / C, rem = (w - 1) % C;      // far-chunk geometry of the window [E-w, E-1]
    int st_level = 0;                                   // sparse-table level for ranges of D-1 chunks
    while ((2 << st_level) <= D - 1) ++st_level;

    for (uint32_t i = tid; i < R + R / 32; i += SYN_NT) m_ring[i] = UINT64_MAX;
    for (uint32_t i = tid; i < NCH; i += SYN_NT) cm_ring[i] = UINT64_MAX;
    for (uint32_t i = tid; i < PBW; i += SYN_NT) pb[i] = 0;
    __syncthreads();

    // 32 bases MSB-first starting at base index t (may be negative / beyond the read: masked by callers)
    auto get64 = [&](int32_t t) -> uint64_t {
        int32_t wi = t >> 4;
        uint32_t sh = ((uint32_t) t & 15u) * 2u;
        uint64_t hi = (uint64_t) pb[wi & (PBW - 1)] << 32 | pb[(wi + 1) & (PBW - 1)];
        uint32_t w2 = pb[(wi + 2) & (PBW - 1)];
        return sh? (hi << sh) | ((uint64_t) w2 >> (32u - sh)) : hi;
    };
    // canonical s-mer code (canon << 1 | strand) of the s-mer ending at base e
    auto smer_code = [&](int32_t e) -> uint64_t {
        uint64_t X = get64(e - S + 1) & (~0ULL << (64 - 2 * S));
        uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
        return fw < rv? fw << 1 : rv << 1 | 1ULL;
    };

    uint32_t ord0 = 0;            // syncmers emitted so far on this read
    int32_t last_n = -1;          // most recent ambiguous position (HAS_N)
    int32_t eval_lo = 0;          // first window-start chunk not yet decided

    for (uint32_t I0 = 0; I0 < hl; I0 += T) {
        const uint32_t I1 = I0 + T;
        // ---- P0: new packed bases (and ambiguity bits) into LDS ----
        if (tid < T / 16) {
            uint32_t wi = I0 / 16 + tid;
            uint32_t v = wi * 16 < hl? ghs[wi] : 0u;
            pb[wi & (PBW - 1)] = __builtin_bswap32(v);     // hoco_s bytes are MSB-first; make the word MSB-first too
        }
        if (HAS_N && tid < T / 32) {
            uint32_t wi = I0 / 32 + tid;
            uint32_t v = 0;
            if (wi * 32 < hl) { v = gnb[wi]; if (v) gnb[wi] = 0; }
            nb_ring[wi & (R / 32 - 1)] = v;
        }
        __syncthreads();

        // ---- P1: s-mer hashes for the C new END positions of this lane ----
        {
            const int32_t i0 = (int32_t) (I0 + tid * C);
            uint64_t vb = get64(i0);
            uint64_t X = get64(i0 - S) & (~0ULL << (64 - 2 * S));
            uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
            uint32_t nbm = 0;
            int32_t ln = -1;
            if (HAS_N) {
                nbm = (nb_ring[((uint32_t) i0 >> 5) & (R / 32 - 1)] >> ((uint32_t) i0 & 31u)) & ((1u << C) - 1u);
                int32_t mine = nbm? i0 + 31 - __builtin_clz(nbm) : -1;
                int32_t inc = wave_incl_max(mine, lane);
                if (lane == 63) w_max[wid] = inc;
                __syncthreads();
                ln = __shfl_up(inc, 1);
                if (lane == 0) ln = -1;
                if (last_n > ln) ln = last_n;
                for (uint32_t ww = 0; ww < wid; ++ww) if (w_max[ww] > ln) ln = w_max[ww];
            }
            uint64_t cmin = UINT64_MAX;
#pragma unroll
            for (int b = 0; b < C; ++b) {
                const int32_t i = i0 + b;
                const uint64_t c = (vb >> (62 - 2 * b)) & 3ULL;
                fw = (fw << 2 | c) & mask;
                rv = rv >> 2 | (3ULL ^ c) << (2 * S - 2);
                bool ok;
                if (HAS_N) {
                    if ((nbm >> b) & 1u) ln = i;
                    uint32_t l = (uint32_t) (i - ln);
                    lr_ring[i & (R - 1)] = (uint16_t) (l > 65535u? 65535u : l);
                    ok = l >= (uint32_t) S && (uint32_t) i < hl;
                } else {
                    ok = i + 1 >= S && (uint32_t) i < hl;
                }
                uint64_t mv = UINT64_MAX;
                if (ok && fw != rv) mv = hash64(fw < rv? fw : rv, mask);
                m_ring[OATK_MI(i)] = mv;
                cmin = mv < cmin? mv : cmin;
            }
            cm_ring[((uint32_t) i0 / C) & (NCH - 1)] = cmin;
        }
        __syncthreads();
        if (HAS_N) {
            for (uint32_t ww = 0; ww < NWAVE; ++ww) if (w_max[ww] > last_n) last_n = w_max[ww];
        }

        // ---- which window-start chunks can be decided now ----
        int32_t eval_hi;
        const bool last_tile = I1 >= hl;
        if (!last_tile) {
            int32_t lim = (int32_t) I1 - w - 1;
            eval_hi = lim < 0? 0 : lim / C;
        } else {
            eval_hi = ((int32_t) hl - w + C - 1) / C;      // a <= hl - w - 1, checked per position
        }

        // ---- P2: sparse table over chunk minima (only the level the window needs) ----
        const uint64_t *st = cm_ring;
        if (D >= 2 && eval_hi > eval_lo) {
            const int32_t c_lo = eval_lo + 1, c_hi = eval_hi + D;   // entries [c_lo, c_hi)
            const uint64_t *src = cm_ring;
            for (int lv = 1; lv <= st_level; ++lv) {
                uint64_t *dst = (lv & 1)? st_a : st_b;
                const int half = 1 << (lv - 1);
                for (int32_t c = c_lo + (int32_t) tid; c < c_hi; c += SYN_NT) {
                    uint64_t u = src[c & (NCH - 1)], v = src[(c + half) & (NCH - 1)];
                    dst[c & (NCH - 1)] = u < v? u : v;
                }
                __syncthreads();
                src = dst;
            }
            st = src;
        }

        // ---- P3 + P4: decide windows, compact, hash ----
        for (int32_t cbase = eval_lo; cbase < eval_hi; cbase += SYN_NT) {
            const int32_t ca = cbase + (int32_t) tid;
            uint32_t close_m = 0, open_m = 0;
            if (ca < eval_hi) {
                const int32_t a0 = ca * C;
                uint64_t v[C], g[C + 1], sv[C];
#pragma unroll
                for (int o = 0; o < C; ++o) v[o] = m_ring[OATK_MI(a0 + o)];
                uint64_t x = a0 > 0? m_ring[OATK_MI(a0 - 1)] : UINT64_MAX;
#pragma unroll
                for (int t = 0; t <= C; ++t) g[t] = m_ring[OATK_MI(a0 + w - 1 + t)];
                uint64_t bmin[C];
                if (D >= 1) {
                    uint64_t run = UINT64_MAX;
#pragma unroll
                    for (int o = C - 1; o >= 0; --o) { run = v[o] < run? v[o] : run; sv[o] = run; }
                    uint64_t E0 = UINT64_MAX;                    // far chunk, part before the window end
                    for (int t = 0; t < rem; ++t) {
                        uint64_t u = m_ring[OATK_MI(a0 + w - 1 - rem + t)];
                        E0 = u < E0? u : E0;
                    }
                    uint64_t rmq1 = UINT64_MAX;                  // chunks ca+1 .. ca+D-1
                    if (D >= 2) {
                        uint64_t u = st[(ca + 1) & (NCH - 1)], z = st[(ca + D - (1 << st_level)) & (NCH - 1)];
                        rmq1 = u < z? u : z;
                    }
                    uint64_t far_full = cm_ring[(ca + D) & (NCH - 1)];
                    uint64_t rmq2 = rmq1 < far_full? rmq1 : far_full;
                    uint64_t pre = E0;
#pragma unroll
                    for (int o = 0; o < C; ++o) {
                        if (o == C - rem) pre = UINT64_MAX;       // crossed into the next chunk: prefix restarts
                        pre = g[o] < pre? g[o] : pre;
                        uint64_t full = o < C - rem? rmq1 : rmq2;
                        uint64_t bb = sv[o] < full? sv[o] : full;
                        bmin[o] = bb < pre? bb : pre;
                    }
                } else {                                         // window shorter than a chunk (small K only)
#pragma unroll
                    for (int o = 0; o < C; ++o) {
                        uint64_t bb = UINT64_MAX;
                        for (int t = 0; t < w; ++t) {
                            uint64_t u = m_ring[OATK_MI(a0 + o + t)];
                            bb = u < bb? u : bb;
                        }
                        bmin[o] = bb;
                    }
                }
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    const int32_t E = a0 + o + w;
                    const uint64_t f = v[o], y = g[o + 1], b = bmin[o];
                    bool valid, nextok = true;
                    if (HAS_N) {
                        valid = (uint32_t) E < hl && lr_ring[E & (R - 1)] >= (uint32_t) K;
                        nextok = (uint32_t) (E + 1) == hl || lr_ring[(E + 1) & (R - 1)] >= 1u;
                    } else {
                        valid = (uint32_t) E < hl && E + 1 >= K;
                    }
                    bool cl = valid && y != UINT64_MAX && (y < b || (y == b && (x >= b || f == b)));
                    bool op = valid && nextok && f != UINT64_MAX && f <= b && f <= y;
                    if (cl && op) cl = op = false;
                    close_m |= (uint32_t) cl << o;
                    open_m |= (uint32_t) op << o;
                    x = f;
                }
            }
            // compaction: ordinals follow position order, Close before Open
            const uint32_t cnt = __builtin_popcount(close_m) + __builtin_popcount(open_m);
            const uint32_t inc = wave_incl_sum(cnt, lane);
            if (lane == 63) w_cnt[wid] = inc;
            __syncthreads();
            uint32_t ex = inc - cnt, total = 0;
            for (uint32_t ww = 0; ww < NWAVE; ++ww) {
                if (ww < wid) ex += w_cnt[ww];
                total += w_cnt[ww];
            }
            if (total) {
                if (tid == 0) s_gbase = atomicAdd(&a.shard_cnt[blockIdx.x & (OATK_REC_SHARDS - 1)], total);
                __syncthreads();
                // records carry (sid | ordinal | strand, s-mer code, position); the k-mer hash is filled in afterwards by
                // kmer_hash_kernel, one lane per record
                if (cnt) {
                    uint32_t q = ex;
                    for (int o = 0; o < C; ++o) {
                        const uint32_t sel = ((close_m >> o) & 1u) | (((open_m >> o) & 1u) << 1);
                        if (!sel) continue;
                        const int32_t E = ca * C + o + w, j = E - K + 1;
                        const bool is_open = sel == 2u;
                        uint64_t code = smer_code(is_open? E - w : E);          // first s-mer ends at E-w, last at E
                        const uint32_t rev = (uint32_t) (code & 1ULL);
                        if (!is_open) code ^= 1ULL;                              // Close stores S ^ 1 (syncmer.c:345)
                        const uint32_t loc = s_gbase + q, ordn = ord0 + q;
                        if (loc < a.region_cap) {
                            const size_t slot = (size_t) (blockIdx.x & (OATK_REC_SHARDS - 1)) * a.region_cap + loc;
                            a.rec_lo[slot] = sid << 32 | (uint64_t) ordn << 1 | rev;
                            a.rec_smer[slot] = code;
                            a.rec_mpos[slot] = (uint32_t) j << 1 | rev;
                        }
                        ++q;
                    }
                }
                ord0 += total;
            }
            __syncthreads();
        }
        eval_lo = eval_hi > eval_lo? eval_hi : eval_lo;
    }
    if (tid == 0) a.n_scm[r] = ord0;
}

}  // namespace oatk
