// oatk_amd/csrc/scan_hpc.hpp -- kernel A of the read scan: homopolymer compression + 2-bit pack.
//
// Replaces the first half of the reference's per-read loop (syncmer.c:284-323): every maximal run of
// one ACGT base becomes one "hoco" position with a 2-bit base in hoco_s (MSB-first, 4 per byte,
// syncmer.c:290) and min(run,256)-1 in ho_rl (:303-304); runs > 255 also go to the ho_l_rl overflow
// list (:301-302); every non-ACGT byte is its own position stored as base A with run 1, and its raw
// coordinate goes to n_nucl (:316-321).
//
// MI355X mapping: the HBM-streaming kernel of the scan (1 B read + 1.25*rho B written per raw base).
// One WAVE per read walks 1 KiB tiles (four waves to a workgroup, which share the tables and nothing else); each lane owns one
// aligned 16-byte vector (reads start on 64-byte boundaries of the packed stream) and the next tile's vector is requested before the
// current one is processed.  Bytes are classified through a 256-entry LDS table (the reference's own
// table semantics, syncmer.c:47-64); a wave scan of (run-start count, last run-start position) -- DPP row shifts -- turns raw
// coordinates into hoco coordinates; finished runs are staged in the wave's LDS ring (one byte store per run, the 2-bit codes of a lane
// OR-ed in as at most two words) and leave as full 16-byte stores.  Ambiguous bases and runs > 255 are
// rare and handled off the fast path.
//
// The kernel is bound by VALU issue, not by HBM (PMC, r06: 259 VALU wave-instructions per wave and KiB before the waves went their own ways, times
// 4 cycles, over 1024 SIMDs at the 2.0 GHz it runs at IS its run time: profiles/r07z_pmc_scan.csv, r07z_pmc_clock_config3.csv), so what counts is
// instructions per 16-byte lane:
//  * a lane whose sixteen bytes (and the byte before them) are all ACGTU in either case -- every lane of a HiFi read -- never
//    touches the table: the low three bits of such a byte are distinct (A 1, C 3, T 4, U 5, G 7), so one v_perm_b32 turns
//    four bytes into four codes and a second one into the four bytes they SHOULD be; any difference sends the lane to the
//    table path.  Codes are packed to sixteen 2-bit fields and run starts fall out of field XOR previous field.
//  * the codes of the runs a lane finishes are squeezed together through a 4 KiB table (4 positions at a time: 4-bit
//    keep mask x 4 codes -> the kept codes, first one on top) instead of a loop that drops one field per turn -- such a loop
//    runs as long as the unluckiest lane of the wave (9 turns for a mean of 4).
//  * runs longer than one base that begin and end inside the lane are at most 15 long: their loop carries no clamp and no
//    overflow test; the one run that enters the lane from the left is handled before it.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace oatk {

// r03p: 128 threads.  A read's last tile takes a tile's time however little of it lies inside the read (a 15 kb read is 3.7 tiles of 4 KiB: an eighth of the
// kernel), two waves meet at the tile's two barriers sooner than four, and the rings shrink with the tile (11.3 KB with the tables: 14 workgroups = 28 waves per CU,
// where the kernel is saturated): 3.50 -> 3.15 ms at 400 k reads; one wave per workgroup: 3.7 ms (18 waves per CU: the tables are 6.3 KB per workgroup).
constexpr int HPC_BPT = 16;

// squeeze table: index = keep << 8 | four 2-bit codes (position i in bits 2i+1:2i); value = the kept codes in position
// order, the first in bits 7:6
struct HpcSqueezeTab {
    uint8_t v[4096];
    constexpr HpcSqueezeTab() : v()
    {
        for (int m = 0; m < 16; ++m)
            for (int c = 0; c < 256; ++c) {
                int out = 0, k = 0;
                for (int i = 0; i < 4; ++i)
                    if (m >> i & 1) out |= ((c >> (2 * i)) & 3) << (6 - 2 * k), ++k;
                v[m << 8 | c] = (uint8_t) out;
            }
    }
};
__device__ const HpcSqueezeTab hpc_squeeze_tab = HpcSqueezeTab();

// gap table: index = eight run-start bits; byte k of the value = positions between the k-th start and the one before it (the run the k-th
// start finishes is that much longer than one base); byte 0 is 0 -- the run the first start finishes began somewhere else
struct HpcGapTab {
    uint64_t v[256];
    constexpr HpcGapTab() : v()
    {
        for (int m = 0; m < 256; ++m) {
            uint64_t out = 0;
            int k = 0, prev = -1;
            for (int i = 0; i < 8; ++i)
                if (m >> i & 1) {
                    if (prev >= 0) out |= (uint64_t) (i - prev - 1) << (8 * k);
                    prev = i, ++k;
                }
            v[m] = out;
        }
    }
};
__device__ const HpcGapTab hpc_gap_tab = HpcGapTab();

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
    uint32_t n_reads;         // the waves stride over the reads (the tables and the zeroed rings are set up once per workgroup, not once per read)
};

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t hpc_dpp(uint32_t old, uint32_t src)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) src, CTRL, ROW_MASK, 0xf, false);
}

// inclusive wave scans with DPP: four shifts inside the rows of sixteen lanes, then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15, row mask 0xa)
// and lane 31 into rows 2 and 3 (row_bcast:31, row mask 0xc) -- six instructions (r04; the row totals went through v_readlane and a chain of selects before)
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v, uint32_t lane)
{
    (void) lane;
    v += hpc_dpp<0x111>(0u, v);
    v += hpc_dpp<0x112>(0u, v);
    v += hpc_dpp<0x114>(0u, v);
    v += hpc_dpp<0x118>(0u, v);
    v += hpc_dpp<0x142, 0xa>(0u, v);
    v += hpc_dpp<0x143, 0xc>(0u, v);
    return v;
}
__device__ __forceinline__ int32_t wave_incl_max_dpp(int32_t v, uint32_t lane)
{
    (void) lane;
    int32_t t;
    t = (int32_t) hpc_dpp<0x111>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    t = (int32_t) hpc_dpp<0x112>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    t = (int32_t) hpc_dpp<0x114>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    t = (int32_t) hpc_dpp<0x118>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    t = (int32_t) hpc_dpp<0x142, 0xa>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    t = (int32_t) hpc_dpp<0x143, 0xc>((uint32_t) v, (uint32_t) v); v = t > v? t : v;
    return v;
}

// Every wave of the workgroup walks reads of its own, in tiles of 64 x 16 bytes, on rings of its own: the three tables are all the waves share, and there is no
// barrier and no exchange between waves inside a read (r06; until then the two waves of a workgroup walked one read together in tiles of 2 KiB, with an exchange of
// their counts through LDS and two barriers per tile: 3.14 -> 2.94 ms at 400 k reads, profiles/r08a_hpc_wave_per_read.txt).
template <int NW>
__global__ __launch_bounds__(NW * OATK_WAVE) void hpc_pack_kernel(HpcArgs a)
{
    constexpr int NT = NW * OATK_WAVE;
    constexpr int TILE = OATK_WAVE * HPC_BPT;       // bytes a wave takes at a time
    constexpr int RING = 2 * TILE;                  // staged hoco positions; > one tile + one unflushed 64-group
    __shared__ uint4 ring_rl4_all[NW][RING / 16];   // run lengths, one byte per staged hoco position
    __shared__ uint4 ring_hs4_all[NW][RING / 64];   // 2-bit codes, 16 per word, MSB-first words (byte-swapped on the way out)
    __shared__ uint8_t lut[256];
    __shared__ uint4 sq4[256];                      // the squeeze table, 4 KiB
    __shared__ uint64_t gap8[256];                  // the gap table, 2 KiB
    __shared__ uint32_t s_rare[NW][2];              // ambiguous bases, runs > 255 of the wave's read

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wid = (uint32_t) __builtin_amdgcn_readfirstlane((int) (tid >> 6));        // (a scalar: the read, its length and its addresses are the wave's)
    uint4 *ring_rl4 = ring_rl4_all[wid], *ring_hs4 = ring_hs4_all[wid];
    uint8_t *ring_rl = (uint8_t *) ring_rl4;
    uint32_t *ring_hs = (uint32_t *) ring_hs4;

    for (uint32_t i = tid; i < 256; i += NT) {
        lut[i] = (uint8_t) nt4_code(i);
        sq4[i] = ((const uint4 *) hpc_squeeze_tab.v)[i];
        gap8[i] = hpc_gap_tab.v[i];
    }
    const uint8_t *sq = (const uint8_t *) sq4;
    for (uint32_t i = tid; i < NW * RING / 64; i += NT) (&ring_hs4_all[0][0])[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < NW * RING / 16; i += NT) (&ring_rl4_all[0][0])[i] = make_uint4(0, 0, 0, 0);      // runs of one base (most) never write their 0
    // (a read leaves the rings as it found them: what is flushed is zeroed, and everything staged is flushed at the read's end)
    __syncthreads();
    // between the lanes of the wave: what one wrote to LDS the others may read behind this (the wave's LDS operations are carried out in the order they were issued;
    // this keeps the compiler from changing that order)
    auto sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    for (uint32_t r = blockIdx.x * NW + wid; r < a.n_reads; r += gridDim.x * NW) {
    const uint64_t o = a.off[r];
    const uint32_t L = a.len[r];
    const uint64_t sid = a.sid0 + r;
    const uint8_t *in = a.seq + o;
    uint8_t *out_rl = a.ho_rl + o;
    uint8_t *out_hs = a.hoco_s + (o >> 2);
    uint32_t *out_nb = a.nbits + (o >> 5);
    if (lane == 0) s_rare[wid][0] = 0, s_rare[wid][1] = 0;

    // move finished 64-position groups [g0, g1) from the LDS ring to HBM as 16-byte stores: four vectors of run lengths per group
    // (consecutive in the ring and in HBM alike), then one vector of codes per group; what leaves is zeroed for its next use
    auto flush = [&](uint32_t g0, uint32_t g1) {
        const uint32_t n_rl = (g1 - g0) * 4u, n_item = n_rl + (g1 - g0);
        for (uint32_t it = lane; it < n_item; it += OATK_WAVE) {
            if (it < n_rl) {
                const uint32_t idx = g0 * 4u + it;
                ((uint4 *) out_rl)[idx] = ring_rl4[idx & (RING / 16 - 1)];
                ring_rl4[idx & (RING / 16 - 1)] = make_uint4(0, 0, 0, 0);
            } else {
                const uint32_t g = g0 + (it - n_rl);
                uint4 v = ring_hs4[g & (RING / 64 - 1)];
                ring_hs4[g & (RING / 64 - 1)] = make_uint4(0, 0, 0, 0);
                v.x = __builtin_bswap32(v.x), v.y = __builtin_bswap32(v.y), v.z = __builtin_bswap32(v.z), v.w = __builtin_bswap32(v.w);
                ((uint4 *) out_hs)[g] = v;
            }
        }
    };
    // rare events of one finished run: hoco index h, raw start p0, length rl, class c
    auto rare = [&](uint32_t h, uint32_t p0, uint32_t rl, uint32_t c) {
        if (rl > 255u) {
            uint32_t idx = atomicAdd(&a.counters[1], 1u);
            if (idx < a.lrl_cap) a.lrl_key[idx] = sid << 32 | h, a.lrl_val[idx] = rl - 1u;
            atomicAdd(&s_rare[wid][1], 1u);
        }
        if (c == 4u) {
            atomicOr(&out_nb[h >> 5], 1u << (h & 31u));
            uint32_t idx = atomicAdd(&a.counters[0], 1u);
            if (idx < a.nn_cap) a.nn_key[idx] = sid << 32 | p0;
            atomicAdd(&s_rare[wid][0], 1u);
        }
    };

    uint32_t nstart = 0;      // run starts seen in earlier tiles
    int32_t last_start = -1;  // raw position of the most recent one
    uint32_t flushed = 0;     // 64-groups already in HBM

    uint4 vnext = make_uint4(0, 0, 0, 0);
    uint32_t upnext = 0;      // lane 0 of a wave: the byte before the wave's first (the other lanes get theirs from their neighbour)
    if (lane * HPC_BPT < L) vnext = *(const uint4 *) (in + lane * HPC_BPT);

    // one tile.  FULL: it lies inside the read -- sixteen valid bytes in every lane; HP: it is not the read's first -- a byte before every lane; both known when the
    // code is compiled (r06: the tests for the read's two ends were a twelfth of the kernel's instructions; 2.94 -> 2.79 ms at 400 k reads)
    auto tile = [&](const uint32_t t0, auto full_tag, auto hp_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value, HP = decltype(hp_tag)::value;
        const uint32_t b0 = t0 + lane * HPC_BPT;
        const uint4 v = vnext;
        uint32_t upb = hpc_dpp<0x138>(0u, v.w >> 24);                             // wave_shr:1 -- the previous lane's last byte
        if (lane == 0) upb = upnext;
        if (b0 + TILE < L) vnext = *(const uint4 *) (in + b0 + TILE);      // next tile's bytes, in flight while this one is processed
        if (lane == 0 && b0 + TILE <= L) upnext = in[b0 + TILE - 1];
        const int nvalid = FULL? HPC_BPT : b0 < L? (int) (L - b0 < (uint32_t) HPC_BPT? L - b0 : (uint32_t) HPC_BPT) : 0;

        uint32_t smask = 0;       // run starts among the lane's sixteen positions
        uint32_t pf = 0;          // sixteen 2-bit fields: the code of the byte BEFORE position b in field b
        uint32_t cx = 0, cy = 0, up = 0;      // classes as nibbles, table path only (ambiguous bases need them further down)
        bool slow = !FULL && nvalid > 0 && nvalid < HPC_BPT;
        const bool wave_slow = !FULL && __ballot(slow) != 0;        // the wave with the read's last, partial vector takes the table path as a whole
        if (nvalid == HPC_BPT && wave_slow) slow = true;   // (it would run both paths otherwise)
        else if (nvalid == HPC_BPT) {
            // ---- all ACGTU?  index = byte & 7 into two 8-byte tables: the code, and the case-folded byte that has this code ----
            constexpr uint32_t TL = 0x01000000u, TH = 0x02000303u;               // idx 1 A 0, 3 C 1, 4 T 3, 5 U 3, 7 G 2
            constexpr uint32_t EL = 0x43FF41FFu, EH = 0x47FF5554u;               // idx 1 'A', 3 'C', 4 'T', 5 'U', 7 'G'; 0xFF never matches
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
            uint32_t diff = 0, f = 0;
            // (r04) the code of an ACGTU byte in either case is ((b >> 1) ^ (b >> 2)) & 3 -- A 0, C 1, G 2, T / U 3 --, and one product gathers the four codes of a word
            // into its top byte: code i sits at bit 8 i and goes to bit 24 + 2 i, the other partial products land on bits of their own below or fall off the top
            uint32_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t sel = wds[q] & 0x07070707u;
                diff |= (wds[q] & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(EH, EL, sel);
                const uint32_t c = ((wds[q] >> 1) ^ (wds[q] >> 2)) & 0x03030303u;
                pk[q] = c * 0x01041040u;
            }
            f = __builtin_amdgcn_perm(__builtin_amdgcn_perm(pk[3], pk[2], 0x00000703u), __builtin_amdgcn_perm(pk[1], pk[0], 0x00000703u), 0x05040100u);      // the four top bytes, word 0's lowest
            uint32_t cu = 0;
            if (HP || b0) {
                const uint32_t sel = upb & 7u;
                cu = __builtin_amdgcn_perm(TH, TL, sel) & 3u;
                diff |= ((upb & 0xDFu) ^ __builtin_amdgcn_perm(EH, EL, sel)) & 0xFFu;
            }
            pf = f << 2 | cu;
            uint32_t s = f ^ pf;                                                  // field b != 0: position b starts a run
            s = (s | s >> 1) & 0x55555555u;
            s = (s | s >> 1) & 0x33333333u;
            s = (s | s >> 2) & 0x0F0F0F0Fu;
            s = (s | s >> 4) & 0x00FF00FFu;
            smask = (s | s >> 8) & 0xFFFFu;
            if (!HP && b0 == 0) smask |= 1u;
            slow = diff != 0;
        }
        if (slow) {
            // ---- classes as 16 nibbles through the table (cx: bytes 0-7, cy: bytes 8-15); bytes past the read end get class 7 ----
            cx = cy = 0;
            up = 7u;
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                cx |= (uint32_t) lut[(wds[b >> 2] >> (8 * (b & 3))) & 0xffu] << (4 * b);
                cy |= (uint32_t) lut[(wds[2 + (b >> 2)] >> (8 * (b & 3))) & 0xffu] << (4 * b);
            }
            if (nvalid < HPC_BPT) {     // tail lane: blank the nibbles beyond the read
                const uint64_t keep = (1ULL << (4 * nvalid)) - 1ULL;
                uint64_t cc = ((uint64_t) cy << 32 | cx);
                cc = (cc & keep) | (0x7777777777777777ULL & ~keep);
                cx = (uint32_t) cc, cy = (uint32_t) (cc >> 32);
            }
            if (HP || b0 > 0) up = lut[upb & 0xffu];
            // run starts: class differs from the previous byte's, or the byte is ambiguous (class 4); nibble-parallel:
            // d = c ^ prev; start <=> d != 0 or c == 4; positions past the end (class 7) never start
            const uint32_t px = cx << 4 | up, py = cy << 4 | cx >> 28;
            const uint32_t dx = cx ^ px, dy = cy ^ py;
            const uint32_t LOW = 0x11111111u;
            uint32_t sx = (dx | dx >> 1 | dx >> 2) & LOW, sy = (dy | dy >> 1 | dy >> 2) & LOW;
            sx |= (cx >> 2) & ~(cx >> 1) & ~cx & LOW;           // class 4 = 0b100
            sy |= (cy >> 2) & ~(cy >> 1) & ~cy & LOW;
            sx &= ~((cx >> 2) & (cx >> 1) & cx) & LOW;          // class 7 = padding
            sy &= ~((cy >> 2) & (cy >> 1) & cy) & LOW;
            auto squeeze = [](uint32_t s) -> uint32_t {        // bits 0, 4, ..., 28 -> bits 0..7
                s = (s | s >> 3) & 0x03030303u;
                s = (s | s >> 6) & 0x000f000fu;
                s = (s | s >> 12) & 0x000000ffu;
                return s;
            };
            smask = squeeze(sx) | squeeze(sy) << 8;
            // the 2-bit code of the byte before every position (an ambiguous base is stored as A, syncmer.c:316-321)
            uint64_t x = (((uint64_t) cy << 32 | cx) << 4 | up) & 0x3333333333333333ULL;
            x = (x | x >> 2) & 0x0f0f0f0f0f0f0f0fULL;
            x = (x | x >> 4) & 0x00ff00ff00ff00ffULL;
            x = (x | x >> 8) & 0x0000ffff0000ffffULL;
            pf = (uint32_t) (x | x >> 16);
        }
        const uint32_t cnt = __builtin_popcount(smask);
        const int32_t lpos = smask? (int32_t) (b0 + 31 - __builtin_clz(smask)) : -1;
        // ---- wave scan: run starts before this lane, and the position of the latest one ----
        const uint32_t icnt = wave_incl_sum_dpp(cnt, lane);
        // latest start at or before this lane: positions grow with the lane, so when every lane holds a start (always, outside
        // homopolymers of 16+ bases and the tail of the read) it is the lane's own
        const int32_t imax = __ballot(lpos < 0)? wave_incl_max_dpp(lpos, lane) : lpos;
        // the tile's totals are lane 63's: scalar registers
        const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) icnt, 63);
        const int32_t wmax = __builtin_amdgcn_readlane(imax, 63), tmax = wmax > last_start? wmax : last_start;
        sync();                     // (the flush of the tile before zeroed ring slots: the wave's LDS operations stay in this order)
        int32_t ls = (int32_t) hpc_dpp<0x138>((uint32_t) -1, (uint32_t) imax);   // wave_shr:1 -> the previous lane's inclusive max
        if (lane == 0) ls = -1;
        const uint32_t n = nstart + icnt - cnt;
        if (last_start > ls) ls = last_start;
        // ---- a run is finished when the next one starts: this lane finishes one run per start it holds ----
        if (smask) {
            // hoco index of the run finished by the k-th start of this lane: (n + k) - 1; the very first start of a
            // read (position 0) finishes nothing.  Three out of four positions are starts and most runs are one base long, so
            // nothing here walks the sixteen positions:
            //   codes  : the fields of pf at finishing starts, squeezed together four positions at a time through the table
            //   lengths: the ring is all zero (min(rl, 256) - 1 of a one-base run); only starts whose previous byte is not a
            //            start finish a longer run (a loop over ~3 set bits)
            const uint64_t cls64 = (uint64_t) cy << 32 | cx;               // zero unless the lane took the table path
            uint32_t fin = smask;                                          // starts that finish a run
            if (!HP && b0 == 0) fin &= ~1u;
            uint32_t special = 0;
            if (slow) {     // an ambiguous byte (class 4) before a finishing start?
                const uint64_t prevc = cls64 << 4 | up;                    // class of the byte before position b, nibble b
                const uint64_t is4 = (prevc >> 2) & ~(prevc >> 1) & ~prevc & 0x1111111111111111ULL;
                uint64_t spread = fin;                                     // bit b -> bit 4 b
                spread = (spread | spread << 24) & 0x000000ff000000ffULL;
                spread = (spread | spread << 12) & 0x000f000f000f000fULL;
                spread = (spread | spread << 6) & 0x0303030303030303ULL;
                spread = (spread | spread << 3) & 0x1111111111111111ULL;
                special = (is4 & spread) != 0;
            }
            if (fin) {
                // kept codes of positions 4j .. 4j+3 on top of a byte; the bytes butt together, the first finished run on top
                const uint32_t t0b = sq[(fin & 0xFu) << 8 | (pf & 0xFFu)];
                const uint32_t t1b = sq[(fin & 0xF0u) << 4 | ((pf >> 8) & 0xFFu)];
                const uint32_t t2b = sq[(fin & 0xF00u) | ((pf >> 16) & 0xFFu)];
                const uint32_t t3b = sq[(fin & 0xF000u) >> 4 | pf >> 24];
                const uint32_t c1 = (uint32_t) __builtin_popcount(fin & 0xFu), c2 = (uint32_t) __builtin_popcount(fin & 0xFFu),
                               c3 = (uint32_t) __builtin_popcount(fin & 0xFFFu);
                const uint32_t rv = t0b << 24 | t1b << (24u - 2u * c1) | t2b << (24u - 2u * c2) | t3b << (24u - 2u * c3);
                const uint32_t hfirst = n - 1u + (!HP && b0 == 0? 1u : 0u);
                const uint32_t off = (hfirst & 15u) * 2u;
                const uint64_t sh = ((uint64_t) rv << 32) >> off;
                const uint32_t w0 = (hfirst & (RING - 1)) >> 4;
                const uint32_t hiw = (uint32_t) (sh >> 32), low = (uint32_t) sh;
                if (hiw) atomicOr(&ring_hs[w0], hiw);
                if (low) atomicOr(&ring_hs[(w0 + 1) & (RING / 16 - 1)], low);
                // the run that enters the lane from the left ends at the lane's first start: the only one that can be long
                const uint32_t bf = (uint32_t) __builtin_ctz(smask);
                uint32_t first = 0;                                        // its length - 1, clamped (0 for the start at position 0 of the read)
                if ((fin >> bf) & 1u) {
                    const uint32_t rl = (uint32_t) ((int32_t) (b0 + bf) - ls);
                    first = (rl > 256u? 256u : rl) - 1u;
                    special |= rl > 255u;
                }
                // runs longer than one base inside the lane (2 .. 15) -- no loop (one that visits them start by start runs as long as the
                // unluckiest lane of the wave: seven turns for a mean of three).  The k-th start of the lane finishes hoco position n - 1 + k, so
                // the lengths (minus one) of everything the lane finishes are consecutive BYTES of the ring: eight starts at a time through the
                // gap table, the first start of the upper half fixed up with the zeros that end the lower half, both halves shifted to their
                // place and OR-ed into the (zeroed) ring as whole words.  Byte 0 is the run that came in from the left.
                {
                    const uint32_t slo = smask & 0xFFu, shi = smask >> 8;
                    uint64_t glo = gap8[slo], ghi = gap8[shi];
                    const uint32_t between = (uint32_t) (__ffs((int) shi) + __clz((int) slo)) - 25u;     // ctz(shi) + 7 - top(slo)
                    glo |= slo? first : 0u;
                    ghi |= slo? (shi? between : 0u) : first;
                    const uint32_t h0 = n - 1u;
                    const uint32_t sl = (h0 & 3u) * 8u, shh = sl + (uint32_t) __builtin_popcount(slo) * 8u;       // bit shifts of the two halves: 0 .. 24, 0 .. 88
                    const uint32_t t = shh & 31u, q4 = (shh >> 5) * 4u;
                    const uint64_t L0 = (uint64_t) (uint32_t) glo << sl, L1 = (glo >> 32) << sl;
                    const uint64_t H0 = (uint64_t) (uint32_t) ghi << t, H1 = (ghi >> 32) << t;
                    const uint32_t l0 = (uint32_t) L0, l1 = (uint32_t) (L0 >> 32) | (uint32_t) L1, l2 = (uint32_t) (L1 >> 32);
                    const uint32_t g0 = (uint32_t) H0, g1 = (uint32_t) (H0 >> 32) | (uint32_t) H1, g2 = (uint32_t) (H1 >> 32);
                    const uint32_t ab = h0 & (uint32_t) (RING - 1) & ~3u;                              // byte address of the first word
                    if (ab <= (uint32_t) RING - 24u) {                                                 // (five words at most)
                        uint32_t *pl = (uint32_t *) ((char *) ring_rl4 + ab), *ph = (uint32_t *) ((char *) ring_rl4 + ab + q4);
                        atomicOr(pl, l0), atomicOr(pl + 1, l1), atomicOr(pl + 2, l2);
                        atomicOr(ph, g0), atomicOr(ph + 1, g1), atomicOr(ph + 2, g2);
                    } else {                                                                                // ... that wrap around the ring
                        auto at = [&](uint32_t b) -> uint32_t * { return (uint32_t *) ((char *) ring_rl4 + (b & (uint32_t) (RING - 1))); };
                        atomicOr(at(ab), l0), atomicOr(at(ab + 4u), l1), atomicOr(at(ab + 8u), l2);
                        atomicOr(at(ab + q4), g0), atomicOr(at(ab + q4 + 4u), g1), atomicOr(at(ab + q4 + 8u), g2);
                    }
                }
            }
            if (special) {                                             // ambiguous bases / very long runs: walk again, slowly
                uint32_t k2 = 0;
                int32_t p2 = ls;
                for (int b = 0; b < HPC_BPT; ++b) {
                    if ((smask >> b) & 1u) {
                        const int32_t i = (int32_t) (b0 + b);
                        if (i > 0) {
                            const uint32_t pc = b? (uint32_t) (cls64 >> (4 * (b - 1))) & 7u : up;
                            rare(n + k2 - 1u, (uint32_t) p2, (uint32_t) (i - p2), pc);
                        }
                        p2 = i;
                        ++k2;
                    }
                }
            }
        }
        sync();
        nstart += tot;
        last_start = tmax;
        const uint32_t done = nstart? (nstart - 1u) >> 6 : 0u;   // complete 64-groups among finished runs
        flush(flushed, done);
        flushed = done;
        // (what this zeroes and what the next tile fills are different 64-groups)
    };
    for (uint32_t t0 = 0; t0 < L; t0 += TILE) {
        if (t0 && t0 + TILE <= L) tile(t0, std::true_type(), std::true_type()); else tile(t0, std::false_type(), std::false_type());
        // (a third form for a read's first tile -- FULL, not HP -- takes 79 registers for the three of them: six waves per SIMD instead of eight)
    }
    // (tried, r06: two vectors per lane and tile, classified and finished one after the other by the same code, one scan and one prefetch for both --
    //  79 registers and 27 KB of LDS for a tenth fewer instructions: 2.77 against 2.79 ms at 400 k reads, within the noise; profiles/r08e_*)
    // the last run ends with the read
    if (lane == 0 && nstart) {
        const uint32_t h = nstart - 1u, rl = L - (uint32_t) last_start, c = lut[in[L - 1]];
        ring_rl[h & (RING - 1)] = (uint8_t) ((rl > 256u? 256u : rl) - 1u);
        if (c & 3u & (c < 4u? 3u : 0u)) atomicOr(&ring_hs[(h & (RING - 1)) >> 4], (c & 3u) << (30u - 2u * (h & 15u)));
        rare(h, (uint32_t) last_start, rl, c);
    }
    sync();
    flush(flushed, (nstart + 63u) >> 6);
    if (lane == 0) {
        a.hoco_l[r] = nstart;
        a.n_nn[r] = s_rare[wid][0];
        a.n_lrl[r] = s_rare[wid][1];
    }
    sync();                   // the rings are clean and the counters read before the next read touches them
    }
}

}  // namespace oatk
