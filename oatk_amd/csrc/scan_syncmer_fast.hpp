// oatk_amd/csrc/scan_syncmer_fast.hpp -- kernel B, fast path (reads without ambiguous bases, 520 <= K-S <= 1031).
//
// Same contract and rule as scan_syncmer.hpp (the general kernel, kept for reads with N and unusual K/S); the
// difference is HOW MUCH work and HOW MANY stalls are spent per position.  PMC on MI355X showed the general kernel
// issuing ~1500 VALU wave-instructions per 2048-position tile per wave (486 of them in the unavoidable s-mer
// hashing).  Here:
//
//  * a window of w = K - S = 8 (D + 1) + r positions is D WHOLE chunks of 8 plus ragged ends.  The whole chunks are a range minimum over
//    per-chunk minima (one wave-level prefix-min and suffix-min over 32-bit keys per tile, DPP row shifts + v_readlane, no LDS round trips;
//    64 <= D <= 127 means at most one whole block of 64 chunks inside a range); the ragged ends are the lane's own positions (registers) and
//    the two chunks the lane's eight windows start in -- sixteen ring entries, four 16-byte LDS reads at compile-time offsets.  Only the top 32
//    bits of the hashes take part.
//  * every lane decides its own eight positions in straight-line code: per position a three-instruction necessary condition
//      Close at k-mer end E needs  M[E]   <= the D chunk minima before E's chunk
//      Open  at k-mer end E needs  M[E-w] <= the D chunk minima after (E-w)'s chunk, and <= M[E]
//    whose result stays in a scalar register pair, and the exact rule (~18 instructions: minima over registers with constant indices)
//    only for the positions where some lane of the wave passes it -- about one per wave and tile on random sequence (closed syncmers have
//    density 2/(w+1) = 1/486), every position inside a tandem repeat, at the same price per position.  (Until r03h the wave decided its
//    candidates together, one after the other: 11.5 % of the kernel's time for two positions in a thousand.)
//  * the ring holds TOP WORDS only (16 KB + pad words; whole hashes were 33 KB and four workgroups per CU).  Top words that tie send a
//    position to the rule on full 64-bit hashes, lane by lane: a value whose top word equals the one it is compared with is hashed again from the
//    read's packed bases, and the whole chunks of the window come from 64-bit chunk minima the hashing phase leaves in a 4 KB ring.
//  * syncmers are appended to a per-read list in LDS at positions that follow from the waves' counts alone (no atomics, position
//    order, index = ordinal) and become records when the read is done: one record-slot atomic per read.  A tile has two workgroup
//    barriers, placed so that the s-mer hashing of the next tile (registers only) overlaps with the slower waves' decisions on this one.
//  * two forms: two waves per workgroup on a 2048-slot ring (tiles of 1024 positions; K - S <= 1023) and four waves on 4096 slots.  A read's last tile
//    costs a tile's time however little of it lies inside the read, and two waves meet at a barrier sooner than four (r03p).
//  * what bounds it: 13.6 KB of LDS and 80 registers let eleven two-wave workgroups onto a CU; 72 VALU wave-instructions per position, 39 of them
//    the rolling canonical s-mer and hash64, at 0.95 of the issue ceiling of that opcode mix.  Ring addresses alternate between two values per
//    lane (a tile is half the ring) and are toggled, not recomputed; end-of-read special cases are decided per wave, not per lane.
//    DESIGN.md 5 has the measurements.
//  * selected syncmers leave as (sid|ordinal|rev, s-mer code, pos) records; their 251-byte k-mers are hashed
//    afterwards by kmer_hash_kernel (one lane per syncmer) instead of by a lone lane inside this kernel.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "scan_syncmer.hpp"
#include "scan_hpc.hpp"

namespace oatk {

constexpr int SYF_C = 8;                 // positions per lane per tile (one chunk)
constexpr int SYF_T = SYN_NT * SYF_C;    // 2048 positions per tile
constexpr int SYF_BLK = 64;              // chunks per wave = block of the prefix/suffix minima
constexpr int SYF_LIST = 128;            // syncmers a read collects in LDS before they become records (a 15 kb read has ~25)

// ring size (positions) the fast kernel needs for this K, or 0 if it does not apply
static inline int syncmer_fast_ring(int K, int S)
{
    const int w = K - S;
    if (w / SYF_C - 1 < SYF_BLK || w / SYF_C - 1 >= 2 * SYF_BLK) return 0;     // 64 <= D <= 127: at most one whole block in a filter range
    if (K + SYF_T + SYF_C + 64 <= 4096) return 4096;
    return 0;
}

// ---- DPP helpers (gfx9 DPP: row = 16 lanes) ----
#define OATK_DPP_ROW_SHL(n) (0x100 + (n))
#define OATK_DPP_ROW_SHR(n) (0x110 + (n))
#define OATK_DPP_ROW_BCAST15 0x142
#define OATK_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) src, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix-min over the 64 lanes of a wave: four shifts inside the rows of sixteen, then lane 15 of rows 0 and 2 into rows 1 and 3 and lane 31 into
// rows 2 and 3 (row_bcast, row masks 0xa and 0xc) -- six v_min_u32_dpp.  (Until r04 the row totals went through v_readlane and came back through a v_mov and a
// v_cndmask each: fourteen instructions per direction.)
__device__ __forceinline__ uint32_t wave_prefix_min_u32(uint32_t v)
{
    uint32_t p = v, t;                // (old = the identity of min: the compiler folds shift and minimum into one v_min_u32_dpp)
    t = dpp_u32<OATK_DPP_ROW_SHR(1)>(0xFFFFFFFFu, p); p = t < p? t : p;
    t = dpp_u32<OATK_DPP_ROW_SHR(2)>(0xFFFFFFFFu, p); p = t < p? t : p;
    t = dpp_u32<OATK_DPP_ROW_SHR(4)>(0xFFFFFFFFu, p); p = t < p? t : p;
    t = dpp_u32<OATK_DPP_ROW_SHR(8)>(0xFFFFFFFFu, p); p = t < p? t : p;
    t = dpp_u32<OATK_DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, p); p = t < p? t : p;
    t = dpp_u32<OATK_DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, p); p = t < p? t : p;
    return p;
}
// ... and the inclusive suffix-min, MIRRORED: lane l returns the suffix minimum of lane 63 - l (there is no row_bcast towards lower lanes: the values are turned
// round by one ds_bpermute_b32 and scanned the same way; the caller stores the result at the mirrored address)
__device__ __forceinline__ uint32_t wave_suffix_min_mirrored_u32(uint32_t v, uint32_t lane)
{
    return wave_prefix_min_u32((uint32_t) __builtin_amdgcn_ds_bpermute((int) ((63u - lane) << 2), (int) v));
}

#define OATK_SYF_PADW(R) ((R) / 8)        // four pad words per 32 positions of the top-word ring
#ifndef OATK_SYF_EXP
#define OATK_SYF_EXP 0                    // timing experiments (development aid; results are wrong with any of them)
#endif
#ifndef OATK_SYF_WAVES
#define OATK_SYF_WAVES 6                  // waves per SIMD the register allocation aims at (six workgroups of four waves per CU)
#endif
// SH: (-(K - S)) & 7 when known at compile time (the offsets of the Open filter's eight reads become immediates), -1 otherwise
template <int R, bool S31, int NT, int SH>
__global__ __launch_bounds__(NT, OATK_SYF_WAVES) void syncmer_fast_kernel(SynArgs a)
{
    constexpr int C = SYF_C, T = NT * SYF_C, NCH = R / C, NWAVE = NT / OATK_WAVE;
    static_assert((R & (R - 1)) == 0, "power-of-two ring: index arithmetic is one AND (a 3200-slot ring raised occupancy from 3 to 4\n"
                  "workgroups per CU but its modulo arithmetic cost more issue slots than the occupancy returned)");

    __shared__ uint32_t m_top[R + OATK_SYF_PADW(R)];   // TOP WORDS of the s-mer hashes by END position: all that the filter and the decision of a candidate
                                                // look at; a position whose top word ties with its window's is re-hashed from the read's bases (r03h)
    __shared__ uint64_t c_min[NCH];             // 64-bit minimum of every chunk: the whole chunks of a window when top words tie.  Kept up to date by the
                                                // hashing phase only while the read is inside a stretch that ties ("repeat mode", below); otherwise the
                                                // waves that meet a tie fill in what their windows need themselves
    __shared__ uint32_t s_tie[2];               // some wave met a tie in the tile of this parity
    __shared__ uint32_t pre32[NCH], suf32[NCH]; // per-wave-block inclusive prefix / suffix minima of the chunk minima's top 32 bits
    __shared__ uint32_t w_cnt[2][NWAVE];         // syncmers per wave of a tile, double-buffered (two barriers per tile)
    __shared__ uint32_t sl_e[SYF_LIST];          // syncmers of the read so far, in position order: k-mer end | kind << 30 (1 Close, 2 Open);
                                                 // written out when the read is done (or the list is full)
    // (LDS and registers limit residency.  r03h: with whole 64-bit hashes in the ring a workgroup took 38.4 KB -- four per CU, and the kernel's time
    //  still fell by 13 % from three to four (profiles/r03h_b_residency.txt).  Top words alone are 16 KB; the low words were only ever read when top
    //  words tied, and those few positions are re-hashed from the packed bases instead.  24.6 KB: six workgroups per CU.)
    __shared__ uint32_t s_gb;                    // record slots of the list being written out

    const uint32_t r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (a.n_nn[r] != 0) return;                 // reads with ambiguous bases take the general kernel

    const int K = a.K, S = S31? 31 : a.S, w = K - S;
    const uint32_t hl = a.hoco_l[r];
    if (hl < (uint32_t) K) { if (tid == 0) a.n_scm[r] = 0; return; }
    const uint64_t sid = a.sid0 + r;
    const uint64_t mask = (1ULL << (2 * S)) - 1;
    const uint32_t *ghs = (const uint32_t *) (a.hoco_s + (a.off[r] >> 2));
    const int D = w / C - 1;                    // chunks that lie inside the window of EVERY position of a chunk

    auto rpos = [](int32_t i) -> uint32_t { return (uint32_t) i & (uint32_t) (R - 1); };
    // four pad words per 32 positions: a chunk stays 16-byte aligned, and lanes that read the same offset of consecutive chunks spread over the banks
    // (r03h, 400 k reads: 11.1 ms without the pad words, 10.3 with them, both at four workgroups per CU)
    auto mi = [&](int32_t i) -> uint32_t { const uint32_t p = rpos(i); return p + ((p >> 5) << 2); };
    auto rch = [](int32_t c) -> uint32_t { return (uint32_t) c & (uint32_t) (NCH - 1); };
    for (uint32_t i = tid; i < R + OATK_SYF_PADW(R); i += NT) m_top[i] = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < NCH; i += NT) pre32[i] = suf32[i] = 0xFFFFFFFFu, c_min[i] = UINT64_MAX;
    if (tid < 2) s_tie[tid] = 0;
    __syncthreads();

    // The bases a lane needs for a tile -- the 32 before its chunk (the s-mer that ends just in front of it) and the chunk's own 8 -- are
    // 80 bits that start on a byte boundary: three consecutive 32-bit words of the read's packed string, at a bit offset of 0 (even lanes)
    // or 16 (odd lanes) that never changes.  They come straight from HBM into registers, one tile AHEAD (the load is in flight during the
    // hashing of the tile before), and one v_perm_b32 per output word undoes the byte order (the string is MSB-first, a dword load is
    // little-endian) and the lane's offset at once.  (An LDS ring of bases fed by half the threads, two unaligned 64-bit extractions per lane
    // and tile with their modulo arithmetic and variable shifts: 50 VALU per lane and tile more, and 2 KB of LDS.)
    struct __attribute__((packed, aligned(4))) Words3 { uint32_t a, b, c; };
    const uint32_t bsel = (tid & 1u)? 0x06070001u : 0x04050607u;    // {S0 = word k, S1 = word k + 1}: bytes of the MSB-first string from bit 16 / bit 0 on
    const uint32_t last_w = (hl - 1u) >> 4;     // (two words of slack lie behind every read's string)
    uint32_t rw0, rw1, rw2;                     // this tile's raw words: bases (i0 - 32) & ~15 ...
    {
        const int32_t wi0 = ((int32_t) (tid * C) - 32) >> 4;         // negative for the first four lanes: those bases lie before the read
        rw0 = ghs[wi0 < 0? 0 : wi0], rw1 = ghs[wi0 + 1 < 0? 0 : wi0 + 1], rw2 = ghs[wi0 + 2];       // (and no k-mer can end there: any value does)
    }

    uint32_t ord0 = 0, par = 0;                 // syncmers already turned into records; parity of the count buffer
    uint32_t sl_n = 0;                          // syncmers in the list (the same value in every thread)
    const uint32_t lcap = hl < (1u << 30)? (uint32_t) a.list_cap : 0u;       // (list entries keep the kind above bit 29)
    const int HW = (w & (C - 1)) + C;           // window positions not covered by D whole chunks: w - C * D

    // s-mer code of a record from the read's packed bases in HBM (the LDS base ring only holds the last tiles)
    auto get64g = [&](int32_t t) -> uint64_t {
        const uint32_t wi = (uint32_t) t >> 4, sh = ((uint32_t) t & 15u) * 2u;
        const uint64_t hi = (uint64_t) __builtin_bswap32(ghs[wi]) << 32 | __builtin_bswap32(ghs[wi + 1]);
        const uint32_t w2 = __builtin_bswap32(ghs[wi + 2]);            // the slab has slack behind the last read
        return sh? (hi << sh) | ((uint64_t) w2 >> (32u - sh)) : hi;
    };
    // the whole hash of the s-mer that ends at position q, as the hashing phase computed it (MAX where no s-mer ends): for the handful of
    // positions whose top word ties with the value it is compared with
    auto hash_at = [&](int32_t q) -> uint64_t {
        if (q + 1 < S || (uint32_t) q >= hl) return UINT64_MAX;
        const uint64_t X = get64g(q - S + 1) & (~0ULL << (64 - 2 * S));
        const uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
        return S31? hash64_s31(fw < rv? fw : rv) : (fw != rv? hash64(fw < rv? fw : rv, mask) : UINT64_MAX);
    };
    auto write_record = [&](int32_t E, uint32_t kind, uint32_t loc, uint32_t ordn) __attribute__((always_inline)) {
        const int32_t e = kind == 2u? E - w : E, j = E - K + 1;        // Open: first s-mer; Close: last s-mer
        const uint64_t X = get64g(e - S + 1) & (~0ULL << (64 - 2 * S));
        const uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
        uint64_t code = fw < rv? fw << 1 : rv << 1 | 1ULL;
        const uint32_t rev = (uint32_t) (code & 1ULL);
        if (kind == 1u) code ^= 1ULL;                                    // Close stores S ^ 1 (syncmer.c:345)
        if (loc < a.region_cap) {
            const size_t slot = (size_t) (blockIdx.x & (OATK_REC_SHARDS - 1)) * a.region_cap + loc;
            a.rec_lo[slot] = sid << 32 | (uint64_t) ordn << 1 | rev;
            a.rec_smer[slot] = code;
            a.rec_mpos[slot] = (uint32_t) j << 1 | rev;
        }
    };
    // the list -> records, all threads; the entries must be visible (a barrier since the last append).  One slot atomic per call.
    auto emit_list = [&]() __attribute__((always_inline)) {
        if (sl_n) {
            if (tid == 0) s_gb = atomicAdd(&a.shard_cnt[blockIdx.x & (OATK_REC_SHARDS - 1)], sl_n);
            __syncthreads();
            const uint32_t gb = s_gb;
            for (uint32_t i = tid; i < sl_n; i += NT) write_record((int32_t) (sl_e[i] & 0x3FFFFFFFu), sl_e[i] >> 30, gb + i, ord0 + i);
            __syncthreads();                    // the list may be refilled, s_gb rewritten
        }
        ord0 += sl_n, sl_n = 0;
    };

    // Syncmers of the previous tile: once the other waves' counts are known (across the next tile's barrier) every wave appends
    // its own to the list at a position that follows from the counts alone -- no atomics, and the list comes out in position
    // order, so an entry's index is its ordinal.  Records are written when the read is done: one slot atomic per read instead of
    // one per wave and tile, and the k-mer codes are computed by 256 lanes at once instead of by lone lanes between two barriers.
    uint32_t pend_kinds = 0, pend_rank = 0, pend_wtot = 0, pend_par = 0, pend_any = 0;
    int32_t pend_i0 = 0;
    auto flush = [&]() __attribute__((always_inline)) {      // call after a barrier that follows the tile
        if (!pend_any) return;
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (int ww = 0; ww < NWAVE; ++ww) { const uint32_t c = w_cnt[pend_par][ww]; tot += c; before += ww < (int) wid? c : 0u; }
        pend_any = 0;
        if (tot == 0) return;
        if (sl_n + tot > lcap) emit_list();               // uniform: sl_n and tot are the same in every thread
        const bool direct = tot > lcap;                    // a tile with more syncmers than the list holds: straight to records
        if (direct) {
            if (tid == 0) s_gb = atomicAdd(&a.shard_cnt[blockIdx.x & (OATK_REC_SHARDS - 1)], tot);
            __syncthreads();
            uint32_t rank = before + pend_rank, kk = pend_kinds;
            while (kk) {
                const int o = __builtin_ctz(kk) >> 1;
                const uint32_t kind = (pend_kinds >> (2 * o)) & 3u;
                kk &= ~(3u << (2 * o));
                write_record(pend_i0 + o, kind, s_gb + rank, ord0 + rank);
                ++rank;
            }
            __syncthreads();
        } else if (pend_wtot) {
            uint32_t idx = sl_n + before + pend_rank, kk = pend_kinds;
            while (kk) {
                const int o = __builtin_ctz(kk) >> 1;
                sl_e[idx] = (uint32_t) (pend_i0 + o) | ((pend_kinds >> (2 * o)) & 3u) << 30;
                kk &= ~(3u << (2 * o));
                ++idx;
            }
        }
        // (both counters move in straight-line code: two "+= tot" in sibling branches are merged by the compiler into one store
        //  through a selected address, which pins the counters in scratch memory -- a vector-memory round trip per tile)
        ord0 += direct? tot : 0u;
        sl_n += direct? 0u : tot;
    };

    // Ring addresses.  A tile advances every lane by T positions = HALF the ring (and half the chunk ring), so each address a lane
    // uses alternates between two values: a chunk-ring byte offset flips one bit, a top-word index is mirrored in the sum of its two
    // values (the pad words make it more than a bit flip).  Twelve registers updated by one instruction each per tile instead of
    // ~40 instructions of shifts, masks and adds; which ranges hold a whole block in their middle does not change at all.
    static_assert(2 * T == R, "the address toggling below needs tile = half the ring");
    struct RangeAddr { uint32_t lo, hi, mid; bool whole; };
    auto range_addr = [&](int32_t lo, int32_t hi) -> RangeAddr {
        return RangeAddr{rch(lo) * 4u, rch(hi) * 4u, rch(((lo >> 6) + 1) * 64 + 63) * 4u, (hi >> 6) - (lo >> 6) == 2};
    };
    const int32_t ch_0 = (int32_t) tid, ca0_0 = ((int32_t) (tid * C) - w) >> 3;       // tile 0: the lane's chunk, the chunk of its first window start
    RangeAddr rB = range_addr(ch_0 - D, ch_0 - 1), rF0 = range_addr(ca0_0 + 1, ca0_0 + D), rF1 = range_addr(ca0_0 + 2, ca0_0 + 1 + D);
    uint32_t o_cs = rch(ch_0) * 4u;
    uint32_t m_own = mi((int32_t) (tid * C)), m_fa = mi(ca0_0 * C), m_fb = mi((ca0_0 + 1) * C);      // (indices of 8-aligned positions)
    const uint32_t m_own_sum = m_own + mi((int32_t) (tid * C) + T), m_fa_sum = m_fa + mi(ca0_0 * C + T), m_fb_sum = m_fb + mi((ca0_0 + 1) * C + T);
    auto next_tile_addr = [&]() __attribute__((always_inline)) {
        constexpr uint32_t FLIP = (uint32_t) (NCH / 2) * 4u;
        o_cs ^= FLIP;
        rB.lo ^= FLIP, rB.hi ^= FLIP, rB.mid ^= FLIP, rF0.lo ^= FLIP, rF0.hi ^= FLIP, rF0.mid ^= FLIP, rF1.lo ^= FLIP, rF1.hi ^= FLIP, rF1.mid ^= FLIP;
        m_own = m_own_sum - m_own, m_fa = m_fa_sum - m_fa, m_fb = m_fb_sum - m_fb;
    };
    auto ld32 = [](const uint32_t *base, uint32_t byte_off) -> uint32_t { return *(const uint32_t *) ((const char *) base + byte_off); };

    // Repeat mode.  The 64-bit chunk minima are only ever read when top words tie -- 2^-28 of the candidates on ordinary sequence, every position
    // inside a tandem repeat.  Keeping them (a 64-bit compare and two selects per position: the slowest pair of the loop, 5 % of the kernel) is
    // therefore switched on by the ties themselves: a wave that meets a tie says so in LDS; one barrier later every thread knows and the hashing
    // of the tile after keeps the minima -- until two tiles have passed without a tie.  What a tying wave needs from tiles hashed without them
    // (the first two tiles of a repeat, and one window's reach behind) it computes itself from the packed bases, wave-wide (fill_c_min below).
    bool rep_mode = false, rep_next = false;
    int32_t rep_start = 0x7FFFFFFF;             // first chunk hashed in repeat mode (valid while rep_mode)
    uint32_t quiet = 0, tpar = 0;
    const int32_t wave_first = __builtin_amdgcn_readfirstlane((int) (wid * OATK_WAVE * C));
    for (uint32_t I0 = 0; I0 < hl; I0 += T, next_tile_addr(), tpar ^= 1u) {
        if (rep_next != rep_mode) { rep_mode = rep_next; rep_start = rep_mode? (int32_t) (I0 / C) : 0x7FFFFFFF; }
        // ---- P1: s-mer hashes of this lane's chunk, chunk minimum, wave prefix/suffix minima ----
        const int32_t i0 = (int32_t) (I0 + tid * C);
        const int32_t wb = (int32_t) I0 + wave_first;   // the wave's first position, in a scalar register: what is decided for the wave is decided there
        const int32_t ch = i0 / C;                      // chunk index; ch % 64 == lane
        uint32_t y[C];                                  // top words of the chunk's hashes
        uint32_t cmin_keep;                             // top word of the chunk's minimum
        {
            const uint32_t a_hi = __builtin_amdgcn_perm(rw0, rw1, bsel), a_lo = __builtin_amdgcn_perm(rw1, rw2, bsel);   // bases i0 - 32 .. i0 - 1
            const uint32_t vbh = __builtin_amdgcn_perm(rw2, 0u, bsel);  // the chunk's 8 bases sit in the top half
            const uint64_t vb = (uint64_t) vbh << 32;
            {   // the next tile's words, while this one is hashed (same lanes, 2048 positions on: 128 words).  Unconditional -- a load whose
                // result is selected under a condition is waited for on the spot --, with the word index held inside the read: lanes (and a
                // whole last tile) behind the read's end fetch its last words, which nobody looks at
                const uint32_t wn = ((I0 + T + tid * C) - 32u) >> 4;
                const Words3 nx = *(const Words3 *) (ghs + (wn < last_w? wn : last_w));
                rw0 = nx.a, rw1 = nx.b, rw2 = nx.c;
            }
            uint64_t cm = UINT64_MAX;                   // the chunk minimum in full (repeat mode only)
            uint32_t cmin = 0xFFFFFFFFu;                // its top word: what the filter looks at
            // (the test is made for the WAVE: a wave with one lane at either end of the read would otherwise run both branches, eight
            //  hashes each -- two waves per read, 5 % of the kernel)
            if (wb + 1 >= S && (uint32_t) (wb + OATK_WAVE * C) <= hl) {
                if (S31) {
                    // The s-mer that ends at position i0 + b is a FIXED bit field of the 96-bit window [a_hi : a_lo : vbh] (bases i0 - 32 .. i0 + 15): bits
                    // [91 - 2b : 30 - 2b]; its reverse complement is the field [65 + 2b : 4 + 2b] of the window's reverse complement [r2 : r1 : r0].  Two
                    // v_alignbit_b32 per strand and position, against the six + four instructions of rolling both 62-bit values along (r04: both forms
                    // cost the same per instruction class, profiles/r04a_valu_rates.txt, so the count decides).
                    auto rc16 = [](uint32_t x) -> uint32_t {
                        const uint32_t r = __builtin_bitreverse32(x);
                        return ~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1));
                    };
                    const uint32_t r0 = rc16(a_hi), r1 = rc16(a_lo), r2 = rc16(vbh);
                    auto hash_chunk = [&](auto keep) __attribute__((always_inline)) {
#pragma unroll
                        for (int b = 0; b < C; ++b) {
                            // both strands LEFT-aligned in 64 bits, two bits of the neighbouring base below them: they never decide the comparison (an odd-length
                            // s-mer is never its own reverse complement, so the 62 bits above differ), and no mask is needed to form the operands
                            uint64_t fw = (uint64_t) __builtin_amdgcn_alignbit(a_hi, a_lo, 28 - 2 * b) << 32 | __builtin_amdgcn_alignbit(a_lo, vbh, 28 - 2 * b);
                            uint64_t rv = (uint64_t) __builtin_amdgcn_alignbit(r2, r1, 2 + 2 * b) << 32 | __builtin_amdgcn_alignbit(r1, r0, 2 + 2 * b);
                            asm("" : "+v"(fw), "+v"(rv));        // (the halves are read back out of the register pairs: without this the compiler keeps each half twice, a v_mov per word)
                            const uint64_t cn = fw < rv? fw : rv;
                            const uint64_t mv4 = hash64_s31_left(cn);   // four times the hash: the order is the hash's, the top word a shift away
                            y[b] = (uint32_t) (mv4 >> 34);
                            cmin = y[b] < cmin? y[b] : cmin;
                            if (decltype(keep)::value) cm = mv4 < cm? mv4 : cm;
                        }
                        if (decltype(keep)::value) cm >>= 2;
                    };
                    // (two copies of the loop, chosen by a scalar branch: as one loop with a per-position select the compiler keeps the 64-bit chain for every position)
                    if (__builtin_amdgcn_readfirstlane((int) rep_mode)) hash_chunk(std::true_type()); else hash_chunk(std::false_type());
                } else {
                    const uint64_t X = ((uint64_t) a_hi << 32 | a_lo) << (2 * (32 - S));       // the S bases that end at i0 - 1, at the top
                    uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
#pragma unroll
                    for (int b = 0; b < C; ++b) {
                        const uint64_t c = (vbh >> (30 - 2 * b)) & 3u;
                        fw = (fw << 2 | c) & mask;
                        rv = rv >> 2 | (3ULL ^ c) << (2 * S - 2);
                        const uint64_t mv = fw != rv? hash64(fw < rv? fw : rv, mask) : UINT64_MAX;
                        y[b] = (uint32_t) (mv >> 32);
                        cmin = y[b] < cmin? y[b] : cmin;
                        cm = mv < cm? mv : cm;      // (S != 31: the chain is always kept; this form of the kernel is not the one the headline workload runs)
                    }
                }
            } else {                                    // first / last chunk of the read
                const uint64_t X = ((uint64_t) a_hi << 32 | a_lo) << (2 * (32 - S));       // the S bases that end at i0 - 1, at the top
                uint64_t fw = X >> (64 - 2 * S), rv = revcomp32(X) & mask;
#pragma unroll
                for (int b = 0; b < C; ++b) {
                    const int32_t i = i0 + b;
                    const uint64_t c = (vb >> (62 - 2 * b)) & 3ULL;
                    fw = (fw << 2 | c) & mask;
                    rv = rv >> 2 | (3ULL ^ c) << (2 * S - 2);
                    uint64_t mv = UINT64_MAX;
                    if (i + 1 >= S && (uint32_t) i < hl && fw != rv) mv = S31? hash64_s31(fw < rv? fw : rv) : hash64(fw < rv? fw : rv, mask);
                    y[b] = (uint32_t) (mv >> 32);
                    cmin = y[b] < cmin? y[b] : cmin;
                    if (rep_mode) cm = mv < cm? mv : cm;
                }
            }
            cmin_keep = cmin;
            uint32_t pre, suf;
#ifdef OATK_SCAN_SHFL
            pre = suf = cmin;
            for (int d = 1; d < OATK_WAVE; d <<= 1) {
                uint32_t up = __shfl_up(pre, d), dn = __shfl_down(suf, d);
                if ((int) lane >= d) pre = up < pre? up : pre;
                if ((int) lane + d < OATK_WAVE) suf = dn < suf? dn : suf;
            }
#else
            pre = wave_prefix_min_u32(cmin);
            suf = wave_suffix_min_mirrored_u32(cmin, lane);
#endif
            // everything above lives in registers: a wave that is done with the previous tile hashes ahead while the others
            // still read that tile's windows from the ring.  Ring slots are only overwritten behind this barrier.
            __syncthreads();
            static_assert(C == 8, "two 16-byte stores per lane");
            *(uint4 *) (m_top + m_own) = make_uint4(y[0], y[1], y[2], y[3]);
            *(uint4 *) (m_top + m_own + 4) = make_uint4(y[4], y[5], y[6], y[7]);
            if (rep_mode) *(uint64_t *) ((char *) c_min + 2u * o_cs) = cm;
            {   // what the tile before saw (its flags are behind a barrier now) decides how the tile AFTER this one is hashed
                const uint32_t seen = s_tie[tpar ^ 1u];
                quiet = seen? 0u : quiet + 1u;
                rep_next = seen != 0u || (rep_mode && quiet < 2u);
            }
            *(uint32_t *) ((char *) pre32 + o_cs) = pre;
#ifdef OATK_SCAN_SHFL
            *(uint32_t *) ((char *) suf32 + o_cs) = suf;
#else
            *(uint32_t *) ((char *) suf32 + (o_cs ^ 252u)) = suf;        // lane l holds the suffix minimum of the chunk of lane 63 - l
#endif
        }
        __syncthreads();
        if (tid == 0) s_tie[tpar ^ 1u] = 0;             // (read above by everybody, written next by the tile after this one, two barriers on)
        flush();                                        // the previous tile's records

        // ---- P3: filter on 32-bit keys (straight-line code: 64 <= D <= 127 means at most ONE whole block inside a range), and the decision.
        //
        // The window of the k-mer that ends at E = i0 + o starts at lo = E - w, and w = 8 (D + 1) + r.  Close compares M[E] with
        //      [lo, E - 1]  =  the rest of lo's chunk from lo on  (+ one whole chunk when o < r)  +  D whole chunks  +  the lane's own [i0, E)
        // and Open compares M[lo] with
        //      (lo, E - 1]  =  the rest of lo's chunk behind lo  +  D whole chunks  (+ the chunk before the lane's own when o < r)  +  [i0, E).
        // The D whole chunks are a range minimum over the prefix / suffix arrays, the lane's own positions are in registers, and the two chunks
        // the eight windows start in are sixteen ring entries -- four 16-byte reads at offsets that are compile-time constants -- that serve all
        // eight positions, for Close and for Open ("the rest of a chunk from lo on" is a minimum over registers with constant indices).  So every
        // lane decides its own positions in straight-line code on top words; what is left for the tie path is a top word EQUAL to the minimum it
        // is compared with.
        // (Until r03h a filter picked ~1 candidate per wave and tile and the wave then decided its candidates one after the other, sixteen lanes
        //  fetching the ragged ends of one window: 11.5 % of the kernel's time for two positions in a thousand, profiles/r03i_b_phase_experiments.txt.)
        static_assert(SH >= 0 && SH < C, "the window's alignment against the chunks is a template argument");
        constexpr int RR = (C - SH) & (C - 1);          // r = w mod 8
        uint32_t backF_keep = 0, fwd0_keep = 0, fwd1_keep = 0;
        uint32_t kinds = 0;                             // 2 bits per position of the chunk: 0 none, 1 Close, 2 Open
        uint32_t tiemask = 0;
        {
            // minimum of the chunk minima over chunks [lo, hi], hi - lo = D - 1: suffix of lo's block, prefix of hi's block and,
            // when the two are not adjacent, the one whole block between them.  Chunks before the read map to ring slots that
            // still hold the initial MAX, which is exactly "no constraint".
            auto range_min = [&](const RangeAddr &ra) -> uint32_t {
                const uint32_t a0 = ld32(suf32, ra.lo), a1 = ld32(pre32, ra.hi);
                const uint32_t mid = ld32(pre32, ra.mid);
                const uint32_t a2 = ra.whole? mid : 0xFFFFFFFFu;
                const uint32_t v = a0 < a1? a0 : a1;
                return a2 < v? a2 : v;
            };
            const uint32_t backF = range_min(rB);            // Close bound: chunks [ch - D, ch - 1]
            const uint32_t fwd0 = range_min(rF0);            // Open bound, first s-mers ending in chunk ca0: chunks [ca0 + 1, ca0 + D]
            const uint32_t fwd1 = range_min(rF1);            // ... and in chunk ca0 + 1: chunks [ca0 + 2, ca0 + 1 + D]
            backF_keep = backF, fwd0_keep = fwd0, fwd1_keep = fwd1;
            // W[j] = top word of position lo(o = 0) + j: chunk ca0 from offset SH on (j < BD), then chunk ca0 + 1
            constexpr int BD = RR? RR : C;              // index of the first position of the second chunk
            uint32_t W[2 * C];
            {
                const uint4 a0 = *(const uint4 *) (m_top + m_fa), a1 = *(const uint4 *) (m_top + m_fa + 4);
                const uint4 b0 = *(const uint4 *) (m_top + m_fb), b1 = *(const uint4 *) (m_top + m_fb + 4);
                const uint32_t all[2 * C] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 2 * C - SH; ++j) W[j] = all[SH + j];
            }
            // the minimum of the chunk before the lane's own (Open with o < r): the neighbour lane's, or for lane 0 the last chunk of the wave before
            uint32_t cprev = 0xFFFFFFFFu;
            if (RR > 0) {
                const uint32_t nb = dpp_u32<0x138>(0xFFFFFFFFu, cmin_keep);                   // wave_shr:1
                const uint32_t lb = ld32(suf32, (o_cs - 4u) & (uint32_t) (NCH * 4 - 1));         // (the suffix minimum of a block's last chunk is that chunk's own)
                cprev = lane == 0? lb : nb;
            }
            // per position: a cheap necessary condition -- a minimum and two compares whose results are LANE MASKS in scalar registers, or'ed and tested there --
            // and the rule itself only where some lane of the wave passes it: about one position per wave and tile on random sequence.
            // (r04: written on the masks themselves.  As `bool c = ...; if (__ballot(c))` the compiler rebuilt c as a vector 0/1 and compared that again, and
            //  carried the end-of-read test of the two edge waves of a read through every wave under a saved exec mask: eleven VALU per position, not three.)
            const bool edge = !(wb + 1 >= K && (uint32_t) (wb + OATK_WAVE * C) <= hl);         // k-mers that do not fit the read: E + 1 < K, E >= hoco_l
            auto decide = [&](auto at_edge) __attribute__((always_inline)) {
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    const uint32_t yhi = y[o], fhi = W[o];
                    const uint32_t fb = o + SH < C? fwd0 : fwd1;
                    const uint32_t fy = fb < yhi? fb : yhi;
                    const bool c1 = yhi <= backF, c2 = fhi <= fy;
                    uint64_t cand = __ballot(c1) | __ballot(c2);
                    bool fits = true;
                    if (decltype(at_edge)::value) {
                        fits = (uint32_t) (i0 + o) < hl && i0 + o + 1 >= K;
                        cand &= __ballot(fits);
                    }
#if OATK_SYF_EXP == 1
                    if (!a.want_n) cand = 0;                // (timing experiment: nothing survives the filter)
#endif
                    if (cand) {
                        const bool c = (c1 | c2) && fits;
                        uint32_t pmin = 0xFFFFFFFFu;        // the lane's own positions before E
#pragma unroll
                        for (int j = 0; j < o; ++j) pmin = y[j] < pmin? y[j] : pmin;
                        const int cend = o < BD? BD - 1 : BD + C - 1;       // last position of lo's chunk
                        uint32_t rest = 0xFFFFFFFFu;                        // lo's chunk behind lo
#pragma unroll
                        for (int j = o + 1; j <= cend; ++j) rest = W[j] < rest? W[j] : rest;
                        // Close: M[E] against the window's minimum
                        uint32_t head = fhi < rest? fhi : rest;             // lo's chunk from lo on ...
                        if (o < RR) {                                       // ... and the whole chunk behind it
#pragma unroll
                            for (int j = RR; j < RR + C; ++j) head = W[j] < head? W[j] : head;
                        }
                        const uint32_t hb = head < backF? head : backF, bh = hb < pmin? hb : pmin;
                        const bool cl = yhi < bh, tie_c = yhi == bh;
                        // Open: M[lo] against everything else in the window and M[E]
                        const uint32_t tailp = o < RR? (cprev < pmin? cprev : pmin) : pmin;
                        const uint32_t rf = rest < fb? rest : fb, rh = rf < tailp? rf : tailp;
                        const bool le = fhi <= rh && fhi <= yhi, op = fhi < rh && fhi < yhi;
                        const bool tie = tie_c || (le && !op);
                        const uint32_t k = tie? 0u : (cl && op? 0u : (cl? 1u : (op? 2u : 0u)));
                        if (c) kinds |= k << (2 * o), tiemask |= (tie? 1u : 0u) << o;
                    }
                }
            };
            // (two copies, chosen by a scalar branch: the ordinary one knows nothing of read ends)
            if (edge) decide(std::true_type()); else decide(std::false_type());
        }
        // Top words tied: ~2^-28 per candidate on random sequence, but the rule rather than the exception inside tandem repeats
        // (telomeres, microsatellites), where the window minimum comes back every period and every lane of the wave holds ties.
        // Those are decided lane by lane on whole 64-bit hashes.  The ring holds top words only, so: a value whose top word differs from
        // the one it is compared with is settled by the top words (its low word is taken as 0 -- no comparison below can tell); a value
        // whose top word is EQUAL is hashed again from the read's packed bases (`hash_at`; in a repeat of period p that is one ragged
        // position in p); and the whole chunks of a window come from the 64-bit chunk minima the hashing phase leaves in `c_min`.
        if (__ballot(tiemask != 0)) {
            constexpr int sh = SH;
            {   // the 64-bit minima of the chunks this wave's windows cover, where the hashing phase did not keep them (see "repeat mode" above): the wave
                // computes them from the packed bases, a chunk per lane and turn, and leaves them in the ring (another wave with ties may write the same
                // values to the same slots).  The windows of the wave's 64 chunks reach from chunk ca0(lane 0) + 1 to ca0(lane 63) + 1 + D: fewer than the ring holds.
                if (lane == 0) s_tie[tpar] = 1u;
                const int32_t ca0_l = (i0 - w) >> 3;
                const int32_t c_lo = __builtin_amdgcn_readlane(ca0_l, 0) + 1, c_hi_all = __builtin_amdgcn_readlane(ca0_l, 63) + 1 + D;
                const int32_t c_hi = rep_mode && rep_start - 1 < c_hi_all? rep_start - 1 : c_hi_all;
                for (int32_t cc = c_lo + (int32_t) lane; cc <= c_hi; cc += 64) {
                    uint64_t u = UINT64_MAX;
                    for (int b = 0; b < C; ++b) { const uint64_t h = hash_at(cc * C + b); u = h < u? h : u; }
                    c_min[rch(cc)] = u;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            auto full = [&](int32_t q, uint32_t ref) -> uint64_t {
                const uint32_t t = m_top[mi(q)];
                return t == ref? hash_at(q) : (uint64_t) t << 32;
            };
            // The whole chunks inside the windows of a lane's eight positions are only TWO ranges, [ca0 + 1, ca0 + D] and
            // [ca0 + 2, ca0 + 1 + D] (ca0: the chunk the first window starts in), and the Close range [ch - D, ch - 1] is one of them:
            // one walk over their common part serves every tie of the lane (in a repeat of period 2 that is four ties, eight ranges).
            uint64_t r0 = UINT64_MAX, r1 = UINT64_MAX;
            if (tiemask) {
                const int32_t ca0 = (i0 - w) >> 3;
                uint64_t mid = UINT64_MAX;
                for (int32_t cc = ca0 + 2; cc <= ca0 + D; ++cc) { const uint64_t u = c_min[rch(cc)]; mid = u < mid? u : mid; }
                const uint64_t e0 = c_min[rch(ca0 + 1)], e1 = c_min[rch(ca0 + 1 + D)];
                r0 = e0 < mid? e0 : mid, r1 = e1 < mid? e1 : mid;
            }
            const uint64_t r_close = (w & (C - 1))? r1 : r0;                // ch - D = ca0 + 2 unless w is a multiple of 8
            uint32_t tm = tiemask;
            while (tm) {                                // the rule in full for one position (scan_syncmer.hpp states it; window = [E - w, E - 1])
                const int o = __builtin_ctz(tm);
                tm &= tm - 1;
                const int32_t E = i0 + o, lo = E - w;
                const uint32_t yhi = m_top[mi(E)], fhi = m_top[mi(lo)];
                const uint64_t yy = hash_at(E), f = hash_at(lo);
                const uint32_t fb = o + sh < C? fwd0_keep : fwd1_keep;
                bool cl = false, op = false;
                if (yhi <= backF_keep) {                // Close: window = head [lo, lo + HW - o) + D chunks before `ch` + tail [i0, E)
                    uint64_t b = UINT64_MAX;
                    for (int t = 0; t < HW - o; ++t) { const uint64_t u = t? full(lo + t, yhi) : f; b = u < b? u : b; }
                    for (int t = 0; t < o; ++t) { const uint64_t u = full(i0 + t, yhi); b = u < b? u : b; }
                    if (yhi == backF_keep) b = r_close < b? r_close : b;
                    if (yy != UINT64_MAX && yy <= b) {
                        const uint64_t x = full(lo - 1, yhi);
                        cl = yy < b || x >= b || f == b;
                    }
                }
                if (fhi <= fb && fhi <= yhi) {          // Open: f must not exceed anything else in the window
                    const int32_t ca = lo >> 3, tail0 = (ca + 1 + D) * C;
                    uint64_t b = UINT64_MAX;
                    for (int32_t q = lo + 1; q < (ca + 1) * C; ++q) { const uint64_t u = full(q, fhi); b = u < b? u : b; }
                    for (int32_t q = tail0; q < E; ++q) { const uint64_t u = full(q, fhi); b = u < b? u : b; }
                    if (fhi == fb) { const uint64_t u = ca == ((i0 - w) >> 3)? r0 : r1; b = u < b? u : b; }
                    op = f != UINT64_MAX && f <= b && f <= yy;
                }
                kinds |= (cl && op? 0u : (cl? 1u : (op? 2u : 0u))) << (2 * o);
            }
        }
        // ---- syncmers -> records, every wave its own.  Two things are only known a little later and neither is waited for:
        //      the counts of the other waves (ordinals), exchanged through LDS across the NEXT tile's barrier, and the record
        //      slots, from a returning atomic (an HBM round trip) that the next tile's hashing covers.  So a tile's records
        //      are written one tile late (the packed-base ring still holds its bases then). ----
#if OATK_SYF_EXP == 5
        kinds &= (uint32_t) -a.want_n;                  // (timing experiment: candidates are decided, none becomes a syncmer)
#endif
        const uint32_t ns = (uint32_t) __builtin_popcount((kinds | kinds >> 1) & 0x5555u);
        const uint32_t incl = wave_incl_sum_dpp(ns, lane);
        const uint32_t wtot = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
        if (lane == 0) w_cnt[par][wid] = wtot;
        pend_kinds = kinds, pend_rank = incl - ns, pend_i0 = i0, pend_wtot = wtot, pend_par = par, pend_any = 1u;
        par ^= 1u;
    }
    __syncthreads();
    flush();
    __syncthreads();
    emit_list();
    if (tid == 0) a.n_scm[r] = ord0;            // tid 0 is in wave 0
}

}  // namespace oatk
