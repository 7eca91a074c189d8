// oatk_amd/csrc/ec_fused.hpp -- the error-block solver for the blocks that are NOT small: a workgroup per block, SEVERAL wavefront steps per barrier (round 5).
//
// The search is that of ec_heavy.hpp / ec_wave.hpp (dfs_search + wf_ed_core, syncerr.c:144-286, levdist.c:75-310), statement for statement, and so is the block's carve-up
// (strings in LDS, frames in an LDS arena that spills into the HBM slab).  What differs is the wavefront step.  ec_heavy.hpp puts diagonal d into slot d + bw + 1, slot s
// into lane s % 256 of four waves, and pays for every step a barrier and an LDS exchange of the entries at the waves' seams and of "has any lane reached an end": ~1 us a
// step on the config-1 surrogate's tandem arrays, where a block runs into MAX_DFS_PATH = 10 000 dead ends, a hundred steps each on hundreds of diagonals (tools/stepbench.py,
// profiles/r05e_stepbench.txt: the four waves save a third of a single wave's time, no more) -- and the tree of the search does not parallelise (tools/experiments/ec_tree.hpp).  Here
//   * a wave's 64 lanes hold 64 CONSECUTIVE slots of which it OWNS the middle 64 - 2 ECF_S: ECF_S slots at either end are copies of its neighbours' (a halo).  The next
//     wavefront takes max(k[d-1], k[d]+1, k[d+1]+1): after one step a wave's outermost lane at either end is stale, after ECF_S steps its halo is -- and its own slots are not.
//     So a wave takes ECF_S steps on its own (extension against the strings in LDS, neighbours through DPP shifts), and only then do the waves meet: one barrier and one
//     exchange of 2 ECF_S entries per wave and ECF_S steps;
//   * a step ENDS the alignment when a diagonal reaches an end of either string (levdist.c:166-180), at the LOWEST such diagonal, with only the diagonals below it stored.
//     Inside a run of steps a wave only notes the first step at which one of its own slots reached an end; when the waves meet and some wave has noted one, every wave goes
//     back to the state it began the run with, repeats the steps before that one (known to reach no end), and takes the last step the careful way (one more barrier: the
//     lowest slot of all).  That happens once per arc tried; the steps are a hundred;
//   * the `score > bw` stop (levdist.c:303) is known in advance: a run is cut to the steps that are left.
// Results are those of ec_heavy.hpp (and of the reference) bit for bit.
#pragma once
#include "ec_heavy.hpp"
#include "ec_tables.hpp"

namespace oatk {

#define ECF_S 4                   // steps per barrier = slots of halo at either end of a wave
#define ECF_OWN (64 - 2 * ECF_S)  // slots a wave owns
#define ECF_NW 16                 // waves per block: 16 x 56 = 896 slots, 2 bw + 3 <= 896
#define ECF_TIER 32u
#define ECF_NT 64                 // (CERT) pairs of tables a workgroup keeps per block: one per arc with a long string (14 - 17 in the heavy blocks of 40 k surrogate reads; at 200 k reads 24 were too few for half the long arcs that die: profiles/r06g)
// (CERT) a workgroup's table region in HBM, 32-bit words: [0] pairs in use, [1 .. ECF_NT] their arcs, then (from word 128 on) ECF_NT x 2 x (cap_t + 1) numbers
__host__ __device__ inline uint64_t ecf_tab_words(int32_t cap_t) { return 128 + (uint64_t) ECF_NT * 2 * (uint64_t) (cap_t + 1); }

// LDS carve-up (32-bit words): [endt: 2 x NW][bnd: 2 x NW x 2 S][red: 2 x NW x 2][any: 2 x NW][bc: 8] ts cs frames
__host__ __device__ inline uint32_t ecf_misc_words(int NW) { return (uint32_t) (2 * NW + 2 * NW * 2 * ECF_S + 2 * NW * 2 + 2 * NW + 8 + 1) & ~1u; }
__host__ __device__ inline uint32_t ecf_lds_words(int32_t cap_t, int32_t cap_c, int32_t cap_fl, int NW)
{
    return ((ecf_misc_words(NW) + ecw_words(cap_t) + ecw_words(cap_c) + 1u) & ~1u) + (uint32_t) cap_fl / 4u;
}

#ifdef ECF_PROF2
// development builds (-DECF_PROF2): cycles of thread 0 by phase of the per-arc loop, summed over blocks, one row per class of workgroups (tools/r06_prof_arcs.sh)
//   0 arcs  1 from a frame  2 without alignment  3 known dead (CERT)  4 tables built  5 aligned  6 arcs of >= 32 bases  7 ... with a lagging wavefront
//   8 top (barrier + restore + arc)  9 append + barrier  10 CERT test  11 alignment  12 outcome + frame push  13 table building  14 blocks  15 wavefront steps
__device__ unsigned long long ecf_prof2[5][28];      // 16 .. 19 aligned arcs by (string >= 32 bases) x 2 + (dies by score), 20 .. 23 their wavefront steps
#define ECF_P2_T(i) do { const uint64_t now_ = __builtin_readcyclecounter(); p2[i] += now_ - p2_last; p2_last = now_; } while (0)
#define ECF_P2_C(i, v) (p2[i] += (uint64_t) (v))
#else
#define ECF_P2_T(i) do { } while (0)
#define ECF_P2_C(i, v) do { } while (0)
#endif
struct EcfShared {
    int32_t *endt, *bnd, *red, *any, *bc;
    uint32_t *ts, *cs;
    uint8_t *fl;                  // LDS frame arena
    uint8_t *fh;                  // HBM frame arena (behind it)
    uint32_t *os;
    uint64_t *c_path, *o_path;
    int32_t cap_t, cap_c, cap_path, cap_fl, cap_fh;
    int32_t *tab;                 // (CERT) this workgroup's table region, ecf_tab_words(cap_t) words
};

// what one lane has to do in one step (levdist.c:156-205, extension mode, no traceback): its diagonal run down (kn, and whether that reached an end of a string), and the
// entry of the next wavefront in its slot from the wave's own lanes (knext: stale in the outermost lane at either end, one lane further in after every step)
struct EcfLane { int32_t kn, knext; bool act, reached; };
template <int NW>
__device__ __forceinline__ EcfLane ecf_lane_step(const uint32_t *ts, const uint32_t *qs, int32_t tl, int32_t ql, int32_t bw, int32_t OFF, int32_t s, int32_t d, int32_t lim, int32_t k,
                                                 int32_t &s_lo, int32_t &n, int stale)
{
    const int lane = (int) threadIdx.x & 63;
    EcfLane o;
    // (`stale` lanes at either end of the wave hold what is left of a maximum whose other terms lay outside the wave: too small, possibly so small that k + d points before
    //  the query -- they take no part; what they would feed is stale one step later anyway)
    const bool in_range = s >= s_lo && s < s_lo + n && lane >= stale && lane < 64 - stale;
    int32_t kk = in_range? k : 0;
    const bool act = in_range && kk < tl && kk + d < ql;
    {
        const int32_t p = act? kk + 1 : 0;             // (lanes without work read, harmlessly, the head of the strings)
        const uint32_t x = ecw_win16(ts, p) ^ ecw_win16(qs, act? p + d : 0);
        int32_t m = x? __builtin_ctz(x) >> 1 : 16;
        const int32_t rem = lim - kk;
        m = m < rem? m : rem;
        m = act? m : 0;
        kk += m;
        bool more = act && m == 16 && kk < lim;
        if (__ballot(more)) {
            // diagonals that matched all sixteen (the path that follows the read; every p-th diagonal inside a tandem array): three more windows lane by lane ...
            for (int it = 0; it < 3; ++it) {
                const int32_t p2 = more? kk + 1 : 0;
                const uint32_t x2 = ecw_win16(ts, p2) ^ ecw_win16(qs, more? p2 + d : 0);
                int32_t m2 = x2? __builtin_ctz(x2) >> 1 : 16;
                const int32_t rem2 = lim - kk;
                m2 = m2 < rem2? m2 : rem2;
                m2 = more? m2 : 0;
                kk += m2;
                more = more && m2 == 16 && kk < lim;
                if (!__ballot(more)) break;
            }
            // ... and what still goes on is run down by the whole wave, 1024 bases a turn
            uint64_t mb = __ballot(more);
            while (mb) {
                const int l = __builtin_ctzll(mb);
                mb &= mb - 1;
                int32_t bk = (int32_t) ecw_lane((uint32_t) kk, l);
                const int32_t bd = (int32_t) ecw_lane((uint32_t) d, l), blim = (int32_t) ecw_lane((uint32_t) lim, l);
                for (;;) {
                    const int32_t rr = blim - bk - (lane << 4);                 // bases left from this lane's window on
                    const int32_t oo = rr > 0? lane << 4 : 0;
                    const uint32_t xx = ecw_win16(ts, bk + 1 + oo) ^ ecw_win16(qs, bk + bd + 1 + oo);
                    int32_t mm = xx? __builtin_ctz(xx) >> 1 : 16;
                    mm = mm < rr? mm : rr;
                    mm = rr > 0? mm : 0;
                    const uint64_t nb = __ballot(mm != 16);
                    if (nb) {
                        const int fl = __builtin_ctzll(nb);
                        bk += (fl << 4) + (int32_t) ecw_lane((uint32_t) mm, fl);
                        break;
                    }
                    bk += 1024;
                }
                kk = lane == l? bk : kk;
            }
        }
    }
    o.act = act;
    o.kn = act? kk : k;
    o.reached = act && kk == lim;                      // k + d == ql - 1 || k == tl - 1 (levdist.c:171), with k <= lim = min(ql - d, tl) - 1
    // next wavefront: diagonals d0 - 1 .. d0 + n (levdist.c:183-205), trimmed (:207-210) or pruned (wf_prune_bw, :99-113)
    int32_t st = 0, en = n + 2;
    const int32_t ns = s_lo - 1, nd0 = ns - OFF;
    if (ECW_LIKELY(bw < 0 || n < 2 * bw + 1)) {
        if (nd0 < -tl) ++st;
        if (nd0 + n + 1 > ql) --en;
    } else {
        const int32_t lo = -bw > -tl? -bw : -tl, hi = bw > ql? bw : ql;          // the LARGER of bw and ql, as in levdist.c:108
        while (nd0 + st < lo) ++st;
        while (nd0 + en - 1 > hi) --en;
    }
    const int32_t n_lo = ns + st, n_n = en - st;
    const int32_t c = o.kn;
    const int32_t left = ech_dpp<0x138>(ECH_NEG, c);                               // wave_shr:1 -- lane i takes lane i - 1 (slot s - 1); lane 0 keeps `old`
    const int32_t right = ech_dpp<0x130>(ECH_NEG, c);                              // wave_shl:1 -- lane i takes lane i + 1 (slot s + 1); lane 63 keeps `old`
    int32_t v = left;
    v = c + 1 > v? c + 1 : v;
    v = right + 1 > v? right + 1 : v;
    o.knext = s >= n_lo && s < n_lo + n_n? v : ECH_NEG;
    s_lo = n_lo, n = n_n;
    return o;
}

// wf_ed_core's loop (levdist.c:296-304) from the state (k, s_lo, n, score) on a query of ql bases: until a step reaches an end (t_end, q_end >= 0, the state as that step leaves
// it) or the score passes bw (t_end = q_end = -1).  par = the parity of the double-buffered exchange words (flipped at every barrier).
template <int NW>
__device__ void ecf_align(const EcfShared &sh, int32_t tl, int32_t ql, int32_t bw, int32_t OFF, int32_t &k, int32_t &s_lo, int32_t &n, int32_t &score, int32_t &t_end, int32_t &q_end,
                          uint32_t &par, uint32_t &wf_steps, uint64_t &wf_diag)
{
    const int t = (int) threadIdx.x, lane = t & 63;
    const int wave = ecw_uni(t >> 6);
    const uint32_t *ts = sh.ts, *qs = sh.cs;
    const int32_t s = wave * ECF_OWN - ECF_S + lane, d = s - OFF;
    const bool owned = lane >= ECF_S && lane < 64 - ECF_S;
    const int32_t lim = (ql - d < tl? ql - d : tl) - 1;
    // the waves meet: this wave's entries next to its halo go to its neighbours, theirs come into its halo
    auto exchange = [&](int32_t note) -> int32_t {
        int32_t *endt = sh.endt + par * NW, *bnd = sh.bnd + par * (NW * 2 * ECF_S);
        if (lane == 0) endt[wave] = note;
        if (lane >= ECF_S && lane < 2 * ECF_S) bnd[wave * 2 * ECF_S + (lane - ECF_S)] = k;                         // my lowest slots: the halo at the upper end of the wave below
        if (lane >= 64 - 2 * ECF_S && lane < 64 - ECF_S) bnd[wave * 2 * ECF_S + ECF_S + (lane - (64 - 2 * ECF_S))] = k;      // my highest slots: the halo at the lower end of the wave above
        __syncthreads();
        int32_t first = ECH_INF;
        {   // (a lane per wave's note; almost always nobody has one)
            const int32_t e = lane < NW? endt[lane] : ECH_INF;
            uint64_t eb = __ballot(e != ECH_INF);
            while (ECW_RARE(eb != 0)) { const int l = __builtin_ctzll(eb); eb &= eb - 1; const int32_t v = (int32_t) ecw_lane((uint32_t) e, l); first = v < first? v : first; }
        }
        if (lane < ECF_S) k = wave > 0? bnd[(wave - 1) * 2 * ECF_S + ECF_S + lane] : ECH_NEG;
        if (lane >= 64 - ECF_S) k = wave < NW - 1? bnd[(wave + 1) * 2 * ECF_S + (lane - (64 - ECF_S))] : ECH_NEG;
        par ^= 1u;
        return first;
    };
    // one step the careful way, from a state whose halos are fresh (every lane's entry is right, so every lane's run down its diagonal is): the lowest slot of all that
    // reached an end decides (levdist.c:166-180); without one the waves' next entries cross in the same meeting.  One barrier.
    auto careful_step = [&]() -> bool {
        int32_t s_lo1 = s_lo, n1 = n;
        const EcfLane o = ecf_lane_step<NW>(ts, qs, tl, ql, bw, OFF, s, d, lim, k, s_lo1, n1, 0);
        int32_t first_slot = ECH_INF, first_k = 0;
        const uint64_t rb = __ballot(o.reached && owned);
        if (rb) { const int fl = __builtin_ctzll(rb); first_slot = wave * ECF_OWN - ECF_S + fl, first_k = (int32_t) ecw_lane((uint32_t) o.kn, fl); }
        int32_t *red = sh.red + par * (NW * 2), *bnd = sh.bnd + par * (NW * 2 * ECF_S);
        if (lane == 0) red[wave * 2] = first_slot, red[wave * 2 + 1] = first_k;
        if (lane >= ECF_S && lane < 2 * ECF_S) bnd[wave * 2 * ECF_S + (lane - ECF_S)] = o.knext;
        if (lane >= 64 - 2 * ECF_S && lane < 64 - ECF_S) bnd[wave * 2 * ECF_S + ECF_S + (lane - (64 - 2 * ECF_S))] = o.knext;
        __syncthreads();
        {
            const int32_t fs = lane < NW? red[lane * 2] : ECH_INF;
            uint64_t eb = __ballot(fs != ECH_INF);
            while (eb) {
                const int l = __builtin_ctzll(eb);
                eb &= eb - 1;
                const int32_t v = (int32_t) ecw_lane((uint32_t) fs, l);
                if (v < first_slot) first_slot = v, first_k = ecw_uni(red[l * 2 + 1]);
            }
        }
        par ^= 1u;
        if (first_slot != ECH_INF) {
            if (o.act && s < first_slot) k = o.kn;     // only the diagonals below it are stored; the wavefront stays the one the step began with
            t_end = first_k, q_end = first_k + (first_slot - OFF);
            return true;
        }
        k = o.knext;
        if (lane < ECF_S) k = wave > 0? bnd[(wave - 1) * 2 * ECF_S + ECF_S + lane] : ECH_NEG;
        if (lane >= 64 - ECF_S) k = wave < NW - 1? bnd[(wave + 1) * 2 * ECF_S + (lane - (64 - ECF_S))] : ECH_NEG;
        s_lo = s_lo1, n = n1;
        return false;
    };
    t_end = q_end = -1;
    // (most alignments of a search end within a step or two -- the path that follows the read: a run of four steps and the way back would triple their cost.  Runs begin
    //  with one step and double while no end is met; the alignments that matter here are a hundred steps long)
    int grow = 1;
    for (;;) {
        const int32_t left = bw - score + 1;           // steps before `score > bw` (bw < 0: no band, wf_ed in the tests)
        int L = bw < 0 || left > ECF_S? ECF_S : left;
        L = L < grow? L : grow;
        grow = grow < ECF_S? grow * 2 : ECF_S;
        if (L == 1) {
            ++wf_steps, wf_diag += (uint64_t) n;
            if (careful_step()) return;
            ++score;
            if (bw >= 0 && score > bw) return;
            continue;
        }
        const int32_t k0 = k, s_lo0 = s_lo, n0 = n;
        int32_t note = ECH_INF;
#ifdef ECF_PROF
        const uint64_t p0 = __builtin_readcyclecounter();
#endif
        for (int i = 0; i < L; ++i) {
            const EcfLane o = ecf_lane_step<NW>(ts, qs, tl, ql, bw, OFF, s, d, lim, k, s_lo, n, i);
            if (note == ECH_INF && __ballot(o.reached && owned)) note = i;
            k = o.knext;
        }
        // (the halo is stale now; a wave's own slots are not.  What a halo lane holds is some maximum over entries of the wavefront: a position of the strings, never out of bounds)
#ifdef ECF_PROF
        const uint64_t p1 = __builtin_readcyclecounter();
#endif
        const int32_t first = exchange(note);
#ifdef ECF_PROF
        { const uint64_t p2 = __builtin_readcyclecounter(); if (t == 0) { sh.bc[4] += (int32_t) (p1 - p0), sh.bc[5] += (int32_t) (p2 - p1), sh.bc[6] += 1; } }
#endif
        if (ECW_LIKELY(first == ECH_INF)) {
            wf_steps += (uint32_t) L, wf_diag += (uint64_t) L * (uint64_t) n0;
            score += L;
            if (bw >= 0 && score > bw) return;
            continue;
        }
        // a step of this run reached an end: back, the steps before it once more, and that one with the lowest slot of all
        k = k0, s_lo = s_lo0, n = n0;
        if (first == 0) { ++wf_steps, wf_diag += (uint64_t) n0; (void) careful_step(); return; }
        for (int i = 0; i < first; ++i) {
            const EcfLane o = ecf_lane_step<NW>(ts, qs, tl, ql, bw, OFF, s, d, lim, k, s_lo, n, i);
            k = o.knext;
        }
        score += first;
        wf_steps += (uint32_t) first + 1u, wf_diag += (uint64_t) (first + 1) * (uint64_t) n0;
        int32_t s_lo1 = s_lo, n1 = n;
        const EcfLane o = ecf_lane_step<NW>(ts, qs, tl, ql, bw, OFF, s, d, lim, k, s_lo1, n1, first);
        int32_t first_slot = ECH_INF, first_k = 0;
        {
            const uint64_t rb = __ballot(o.reached && owned);
            if (rb) { const int fl = __builtin_ctzll(rb); first_slot = wave * ECF_OWN - ECF_S + fl, first_k = (int32_t) ecw_lane((uint32_t) o.kn, fl); }
            int32_t *red = sh.red + par * (NW * 2);
            if (lane == 0) red[wave * 2] = first_slot, red[wave * 2 + 1] = first_k;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int32_t fs = ecw_uni(red[w * 2]), fk = ecw_uni(red[w * 2 + 1]);
                if (fs < first_slot) first_slot = fs, first_k = fk;
            }
        }
        // only the diagonals below the first one that reached an end are stored (levdist.c:166-180); the wavefront stays the one the step began with
        if (o.act && s < first_slot) k = o.kn;
        (void) exchange(ECH_INF);                      // (the halos, for the next call)
        t_end = first_k, q_end = first_k + (first_slot - OFF);
        return;
    }
}

// wf_ed_core on its own over ecf_align (test entry: oatk_hip_debug_wf_ed_wg with R = 16): one workgroup per job
template <int NW>
__global__ __launch_bounds__(64 * NW) void ecf_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                            const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3, int32_t cap_words)
{
    constexpr int T = 64 * NW;
    extern __shared__ uint32_t ecf_lds[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x, lane = t & 63, wave = t >> 6;
    EcfShared sh;
    sh.endt = (int32_t *) ecf_lds, sh.bnd = sh.endt + 2 * NW, sh.red = sh.bnd + 2 * NW * 2 * ECF_S, sh.any = sh.red + 2 * NW * 2, sh.bc = sh.any + 2 * NW;
    sh.ts = ecf_lds + ecf_misc_words(NW), sh.cs = sh.ts + cap_words;
    const int32_t tlen = tl[j], band = bw[j];
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];      // (packed with their pad words: ecw_words)
    for (uint64_t i = t; i < nt; i += T) sh.ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += T) sh.cs[i] = qw[qw_off[j] + i];
    const int32_t OFF = (band < 0? tlen : band) + 1;
    const int32_t s = wave * ECF_OWN - ECF_S + lane;
    int32_t k = s == OFF? -1 : ECH_NEG;                // the caller's initial state: diagonal 0, nothing matched, score 0 (syncerr.c:465-482)
    int32_t s_lo = OFF, n = 1, score = 0, t_end = -1, q_end = -1;
    uint32_t par = 0, wfs = 0;
    if (t < 8) sh.bc[t] = 0;
    uint64_t wfd = 0;
    __syncthreads();
    for (uint64_t st = step_off[j]; st < step_off[j + 1]; ++st) {
        const int32_t ql = step_ql[st];
        ecf_align<NW>(sh, tlen, ql, band, OFF, k, s_lo, n, score, t_end, q_end, par, wfs, wfd);
        if (t == 0) out3[3 * st] = score, out3[3 * st + 1] = t_end + 1, out3[3 * st + 2] = q_end + 1;
#ifdef ECF_PROF
        if (t == 0) out3[3 * st] = sh.bc[4], out3[3 * st + 1] = sh.bc[5], out3[3 * st + 2] = sh.bc[6];
#endif
    }
}

// Solve one block with the whole workgroup (ech_solve_block of ec_heavy.hpp with the wavefront one slot per lane and ecf_align for wf_ed_core).  Returns false when the
// block outgrows the carve-up (it is then re-run by the next class or the slab tier of ec_wave.hpp).
// CERT (experimental, OATK_DEBUG_EC_CERT=1; written after round 5's last GPU run and never executed): an arc that appends a long string is first asked whether it can be
// alive at all -- min over the band of (what the parent's wavefront knows of its last row + the string's tables) beyond bw: dead by score, no step taken (DESIGN.md 8.3,
// tests/trace/ec_trace.c ECT_ROWS: 92.8 % of such arcs' steps on the config-1 surrogate, no living arc).  Without CERT the code is what it was.
template <int NW, bool CERT = false>
__device__ bool ecf_solve_block(const EcLive &lv, const EcReads &rd, const EcWork &wk, const EcfShared &sh, double max_edist,
                                uint32_t &status_out, uint32_t &np_out, uint32_t &tried_out, uint32_t &n_path_out, uint32_t &wf_steps_out, uint32_t &wf_diag_out)
{
    constexpr int T = 64 * NW;
    const int t = (int) threadIdx.x, lane = t & 63;
    const int wave = ecw_uni(t >> 6);
    const int K = rd.K;
    const int32_t tl = wk.l;
    int32_t bw = (int32_t) ceil((double) tl * max_edist);
    if (bw < EC_MIN_ERR_BASE) bw = EC_MIN_ERR_BASE;
    const int32_t OFF = bw + 1;
    if (ECW_RARE(tl > sh.cap_t || 2 * bw + 3 > NW * ECF_OWN)) return false;
    EcwArcRegs pre;
    pre.a = make_uint4(0, 0, 0, 0), pre.b = make_uint2(0, 0);
    uint32_t pre_idx = 0xFFFFFFFFu;
    if (wk.ln) pre = ecw_arc_load(lv.arc, wk.lp), pre_idx = wk.lp;
    // target: the read segment, reverse-complemented for a leading block (get_kmer_dna_seq, syncmer.c:1237)
    const uint8_t *hs = rd.hoco_s + ((uint64_t) wk.hs16 << 4);
    {
        const bool R_ = wk.r != 0;
        for (int32_t wb = 0; (wb << 4) < tl; wb += 2 * T) {                    // two windows per lane with their loads in flight together
            uint32_t w0[2], w1[2], pp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int32_t wi = wb + t + T * u;
                const int64_t start = R_? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
                pp[u] = ecw_gather16_at(start, R_);
                w0[u] = w1[u] = 0;
                if ((wi << 4) < tl) { const uint32_t *q = (const uint32_t *) hs + (pp[u] >> 4); w0[u] = q[0], w1[u] = q[1]; }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int32_t wi = wb + t + T * u;
                const int64_t start = R_? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
                if ((wi << 4) < tl) sh.ts[wi] = ecw_gather16_fin(w0[u], w1[u], pp[u], start, R_);
            }
        }
    }
    int32_t status = EC_FAILURE, n_path = 0, edist = INT32_MAX, s_edist = INT32_MAX;
    int32_t c_len = 0, o_len = 0, np = 0;
    uint32_t tried = 0, wf_steps = 0;
    uint64_t wf_diag = 0;
    int32_t score = 0, t_end = 0, q_end = 0;
    const int32_t slot = wave * ECF_OWN - ECF_S + lane;             // this lane's slot: diagonal slot - OFF; the wave owns the lanes ECF_S .. 63 - ECF_S, the rest is halo
    const bool owned = lane >= ECF_S && lane < 64 - ECF_S;
    int32_t k = slot == OFF? -1 : ECH_NEG;
    int32_t s_lo = OFF, n = 1;
    uint32_t par = 0, apar = 0;
    if (t == 0) sh.c_path[0] = wk.beg_utg;
    int32_t fsz = 0, top = -1, nfr = 0;
    bool vpend = false;
    uint32_t v_arc = 0;
    int32_t v_depth = 0;
    int32_t aligned_len = 0;                          // (CERT) the consensus length the wavefront stands for: behind c_len where alignments were skipped
#ifdef ECF_PROF2
    uint64_t p2[28] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p2_last = __builtin_readcyclecounter();
#endif
    if (CERT) { if (t == 0 && sh.tab) sh.tab[0] = 0; }

    // workgroup-wide "any lane": rare paths only (ties between optimum paths)
    auto wg_any = [&](bool p) -> bool {
        const uint64_t b = __ballot(p);
        if (NW == 1) return b != 0;
        int32_t *any = sh.any + apar * NW;
        apar ^= 1u;
        if (lane == 0) any[wave] = b != 0;
        __syncthreads();
        bool r = false;
#pragma unroll
        for (int w = 0; w < NW; ++w) r |= ecw_uni(any[w]) != 0;
        return r;
    };
    // frames: offsets below cap_fl lie in the LDS arena, the rest in the slab; a frame never straddles
    auto push_frame = [&](uint32_t lp, uint32_t ln, int32_t depth) -> bool {
        const int32_t need = ((int32_t) sizeof(EchFrame) + 4 * n + 7) & ~7;
        int32_t at = fsz;
        if (at < sh.cap_fl && at + need > sh.cap_fl) at = sh.cap_fl;
        if (at + need > sh.cap_fl + sh.cap_fh) return false;
        const bool in_lds = at < sh.cap_fl;
        EchFrame hd;
        hd.arc_i = lp, hd.arc_end = lp + ln, hd.l0 = c_len, hd.score = score, hd.t_end = t_end, hd.q_end = q_end, hd.n = n, hd.s_lo = s_lo, hd.prev_off = top, hd.depth = depth;
        if (in_lds) {
            EchFrame *f = (EchFrame *) (sh.fl + at);
            if (t == 0) *f = hd;
            int32_t *sv = (int32_t *) (f + 1);
            if (owned && slot >= s_lo && slot < s_lo + n) sv[slot - s_lo] = k;
        } else {
            EchFrame *f = (EchFrame *) (sh.fh + (at - sh.cap_fl));
            if (t == 0) *f = hd;
            int32_t *sv = (int32_t *) (f + 1);
            if (owned && slot >= s_lo && slot < s_lo + n) sv[slot - s_lo] = k;
        }
        top = at;
        fsz = at + need;
        ++nfr;
        return true;
    };
    ech_barrier<NW>();
    if (ECW_LIKELY(wk.ln == 1)) vpend = true, v_arc = wk.lp, v_depth = 0;
    else if (!push_frame(wk.lp, wk.ln, 0)) return false;

    while (nfr > 0 || vpend) {
        ech_barrier<NW>();
        uint32_t a;
        int32_t depth;
        bool from_frame = false;
        if (ECW_LIKELY(vpend)) {                      // carry on where the search stands: nothing to restore
            vpend = false;
            a = v_arc, depth = v_depth;
        } else {
            const bool in_lds = top < sh.cap_fl;
            EchFrame hd;
            if (in_lds) hd = *(const EchFrame *) (sh.fl + top); else hd = *(const EchFrame *) (sh.fh + (top - sh.cap_fl));
            a = ecw_uniu(hd.arc_i);
            const uint32_t a_end = ecw_uniu(hd.arc_end);
            if (ECW_RARE(a == a_end)) {               // level exhausted: return to the nearest level with siblings left
                fsz = top;
                top = ecw_uni(hd.prev_off);
                --nfr;
                continue;
            }
            from_frame = true;
            // restore the state this level was entered with (syncerr.c:277-284)
            depth = ecw_uni(hd.depth);
            c_len = ecw_uni(hd.l0), score = ecw_uni(hd.score), t_end = ecw_uni(hd.t_end), q_end = ecw_uni(hd.q_end);
            aligned_len = c_len;                      // (a frame is pushed right after an alignment)
            n = ecw_uni(hd.n), s_lo = ecw_uni(hd.s_lo);
            if (in_lds) {
                const int32_t *sv = (const int32_t *) (sh.fl + top + sizeof(EchFrame));
                k = slot >= s_lo && slot < s_lo + n? sv[slot - s_lo] : ECH_NEG;
            } else {
                const int32_t *sv = (const int32_t *) (sh.fh + (top - sh.cap_fl) + sizeof(EchFrame));
                k = slot >= s_lo && slot < s_lo + n? sv[slot - s_lo] : ECH_NEG;
            }
        }
        ++tried;
        ECF_P2_C(0, 1); ECF_P2_C(1, from_frame);
        if (ECW_RARE(pre_idx != a)) pre = ecw_arc_load(lv.arc, a);
        const uint64_t w = ecw_uniu(pre.a.x);
        const int32_t ls = (int32_t) ecw_uniu(pre.a.y), ext = K - ls;
        const uint32_t w_hs16 = ecw_uniu(pre.a.z), w_mpos = ecw_uniu(pre.a.w), w_lp = ecw_uniu(pre.b.x), w_ln = ecw_uniu(pre.b.y);
        const int32_t t_end0 = t_end;
        ECF_P2_T(8);
        if (ECW_RARE(depth + 2 > sh.cap_path || c_len + ext > sh.cap_c)) return false;
        int32_t cn = depth + 2;                       // entries in c_path
        // the arc most likely to be tried next: the first one out of w (in flight during the gather and the alignment)
        pre_idx = 0xFFFFFFFFu;
        if (ECW_LIKELY(w_ln)) pre = ecw_arc_load(lv.arc, w_lp), pre_idx = w_lp;
        {   // append the part of w's k-mer that lies beyond the overlap (syncerr.c:186-190); see ec_wave.hpp
            const uint8_t *vs = rd.hoco_s + ((uint64_t) w_hs16 << 4);
            const uint32_t pos = w_mpos >> 1;
            const bool asc = (uint32_t) (w & 1ULL) == (w_mpos & 1u);
            const int32_t w0 = c_len >> 4, w1 = (c_len + ext - 1) >> 4;
            for (int32_t wb = w0; wb <= w1; wb += T) {
                const int32_t wi = wb + t;
                if (wi > w1) continue;
                const int32_t t0 = (wi << 4) - c_len;
                uint32_t x = asc? ecw_gather16(vs, (int64_t) pos + ls + t0, false) : ecw_gather16(vs, (int64_t) pos + K - 1 - ls - t0, true);
                if (t0 < 0) {
                    const uint32_t keep = (1u << ((uint32_t) (-t0) << 1)) - 1u;
                    x = (sh.cs[wi] & keep) | (x & ~keep);
                }
                sh.cs[wi] = x;
            }
            c_len += ext;
        }
        ech_barrier<NW>();
        // (every wave has read the frame by now: the cursor moves on, and the path takes its entry)
        if (t == 0) {
            sh.c_path[depth + 1] = w;
            if (from_frame) {
                if (top < sh.cap_fl) ((EchFrame *) (sh.fl + top))->arc_i = a + 1; else ((EchFrame *) (sh.fh + (top - sh.cap_fl)))->arc_i = a + 1;
            }
        }
        ECF_P2_T(9);
        // a vertex on an unbranched stretch that cannot be the end of the path needs no alignment of its own (ec_wave.hpp)
        if (edist == INT32_MAX && wk.end_utg != EC_NONE && wk.end_utg != w && w_ln == 1 && n_path < EC_MAX_DFS_PATH && c_len - K <= tl + bw && c_len >= bw + 3) {
            vpend = true, v_arc = w_lp, v_depth = depth + 1;
            ECF_P2_C(2, 1);
            continue;
        }
        bool known_dead = false;
        if constexpr (CERT) {
            const int32_t l0 = c_len - ext;
            if (ext >= 32) { ECF_P2_C(6, 1); if (aligned_len != l0) ECF_P2_C(7, 1); else if (ext > 1024) ECF_P2_C(24, 1); else if (!(l0 >= 1 && l0 - 1 + bw + 2 < tl - 1)) ECF_P2_C(25, 1); }
            // a long string, the wavefront standing for exactly the consensus before it, and no row before the new ones within reach of the target's last column
            if (sh.tab && ext >= 32 && ext <= 1024 && aligned_len == l0 && l0 >= 1 && l0 - 1 + bw + 2 < tl - 1) {
                int32_t *tabs = sh.tab + 128;
                const uint64_t pair = 2 * (uint64_t) (sh.cap_t + 1);
                int32_t idx = -1;
                {
                    const int32_t nt = ecw_uni(sh.tab[0]);
                    const uint32_t key = lane < nt? (uint32_t) sh.tab[1 + lane] : 0xFFFFFFFFu;      // (ECF_NT <= 64: a lane per pair)
                    const uint64_t hit = __ballot(lane < nt && key == a);
                    if (hit) idx = __builtin_ctzll(hit);
                    else if (nt < ECF_NT) {
                        // the string's two tables by all the waves (ec_tables.hpp: ecb_tables_wg), kept for the block
                        ecb_tables_wg<NW>(sh.ts, tl, sh.cs, l0, ext, bw, tabs + (uint64_t) nt * pair, tabs + (uint64_t) nt * pair + (uint64_t) (tl + 1));
                        __syncthreads();
                        if (t == 0) sh.tab[1 + nt] = (int32_t) a, sh.tab[0] = nt + 1;
                        __syncthreads();
                        idx = nt;
                        ECF_P2_C(4, 1); ECF_P2_T(13);
                    }
                }
                if (idx < 0) ECF_P2_C(26, 1); else ECF_P2_C(27, 1);
                if (idx >= 0) {
                    const int32_t *t0 = tabs + (uint64_t) idx * pair, *t1 = t0 + (tl + 1);
                    const bool far = c_len - 1 + bw < tl - 1;                                       // the new rows do not reach the target's last column either: the first table alone
                    // this lane's diagonal run down as the child's first step would (against the consensus BEFORE the string), then what that says of the parent's last row
                    const int32_t d = slot - OFF;
                    int32_t s_lo_c = s_lo, n_c = n;
                    const int32_t lim0 = (l0 - d < tl? l0 - d : tl) - 1;
                    const EcfLane o = ecf_lane_step<NW>(sh.ts, sh.cs, tl, l0, bw, OFF, slot, d, lim0, k, s_lo_c, n_c, 0);
                    const int32_t tp = l0 - 1 - d;                                                  // the cell of the parent's last row on this diagonal
                    int32_t v = ECH_INF;
                    if (owned && slot >= 0 && tp >= -1 && tp < tl) {
                        const bool inwf = slot >= s_lo && slot < s_lo + n;
                        const bool reached = inwf && o.kn >= tp;
                        // (round 6: a cell the wavefront HAS reached costs exactly the parent's score -- a call ends in the step that first reaches its consensus's last row, so no cell
                        //  of that row was within reach of a lower score; tests/trace/ec_trace.c: 0 of 10^5 reached cells differ, and the test knows 99 % of the long arcs that die
                        //  by score instead of 54 %.  Until round 6 such a cell was priced at |diagonal|.)
                        int32_t lb = d < 0? -d : d;
                        const int32_t fl = reached? score : score + 1;
                        if (fl > lb) lb = fl;
                        int32_t tv = t0[tp + 1];
                        if (!far && tp + 1 >= ecb_table1_lo(tl, ext, bw)) { const int32_t t2 = t1[tp + 1]; tv = t2 < tv? t2 : tv; }      // (below that the second table is not written: it is beyond bw there)
                        v = lb + tv;
                    }
#pragma unroll
                    for (int o2 = 32; o2 >= 1; o2 >>= 1) { const int32_t y = __shfl_xor(v, o2, 64); v = y < v? y : v; }
                    int32_t *any = sh.any + apar * NW;
                    apar ^= 1u;
                    if (lane == 0) any[wave] = v;
                    __syncthreads();
                    int32_t best = ECH_INF;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) { const int32_t y = ecw_uni(any[w2]); best = y < best? y : best; }
                    known_dead = best > bw;
                }
            }
        }
        // wf_ed_core (levdist.c:265-310)
        ECF_P2_T(10);
        if (CERT && known_dead) { score = bw + 1, t_end = -1, q_end = -1; ECF_P2_C(3, 1); }          // (what the steps would have left: levdist.c:303, syncerr.c:195 "zero if not aligned")
        else {
#ifdef ECF_PROF2
            const uint32_t st0_ = wf_steps;
#endif
            ecf_align<NW>(sh, tl, c_len, bw, OFF, k, s_lo, n, score, t_end, q_end, par, wf_steps, wf_diag); ECF_P2_C(5, 1);
#ifdef ECF_PROF2
            { const int cat_ = (ext >= 32? 2 : 0) + (score > bw? 1 : 0); p2[16 + cat_] += 1, p2[20 + cat_] += wf_steps - st0_; }
#endif
        }
        ECF_P2_T(11);
        aligned_len = c_len;
        t_end += 1, q_end += 1;
        const int32_t ql = c_len;
        const int32_t sc = score + tl - t_end;        // syncerr.c:209
        bool new_opt = false;
        if (sc <= bw && (wk.end_utg == EC_NONE || wk.end_utg == w)) {
            status = EC_SUCCESS;
            if (sc <= edist) {
                if (t_end > t_end0) s_edist = edist;
                edist = sc;
                if (wk.end_utg == EC_NONE && q_end < ql) --cn;
                ech_barrier<NW>();                    // (c_path[depth + 1] is in place for every wave)
                if (ECW_RARE(edist == s_edist)) {
                    bool diff = q_end != o_len;
                    if (!diff) {
                        bool d = false;
                        const int32_t nw = (q_end + 15) >> 4;
                        for (int32_t wi = t; wi < nw; wi += T) {
                            uint32_t x = sh.cs[wi] ^ sh.os[wi];
                            if (wi == nw - 1 && (q_end & 15)) x &= (1u << ((q_end & 15) << 1)) - 1u;
                            d |= x != 0;
                        }
                        diff = wg_any(d);
                    }
                    if (diff) status = EC_AMBISEQ;
                    if (status == EC_SUCCESS) {
                        bool pd = cn != np;
                        if (!pd) {
                            bool d = false;
                            for (int32_t i = t; i < cn; i += T) d |= sh.c_path[i] != sh.o_path[i];
                            pd = wg_any(d);
                        }
                        if (pd) status = EC_AMBISNQ;
                    }
                    ech_barrier<NW>();                // (the comparisons are done before the optimum is overwritten)
                }
                new_opt = true;
                o_len = q_end;
                for (int32_t i = t; i < cn; i += T) sh.o_path[i] = sh.c_path[i];
                np = cn;
            } else if (sc < s_edist) {
                s_edist = sc;
            }
        }
        if (score <= bw && ql - K <= tl + bw && ((wk.end_utg != EC_NONE && wk.end_utg != w) || t_end < tl)) {
            if (n_path < EC_MAX_DFS_PATH) {           // the callee would return at once otherwise (syncerr.c:146-148)
                if (ECW_LIKELY(w_ln == 1)) vpend = true, v_arc = w_lp, v_depth = depth + 1;
                else if (w_ln > 1 && !push_frame(w_lp, w_ln, depth + 1)) return false;      // (no arcs: the callee's loop does not run)
            }
        } else {
            ++n_path;
        }
        // the optimum consensus is only ever compared with a LATER path's (a tie): when the search ends here nobody reads it
        if (new_opt && (nfr > 0 || vpend)) {
            for (int32_t wi = t; wi < ((o_len + 15) >> 4); wi += T) sh.os[wi] = sh.cs[wi];
        }
        ECF_P2_T(12);
    }
    ech_barrier<NW>();
#ifdef ECF_PROF2
    if (t == 0) { p2[14] = 1, p2[15] = wf_steps; const int row = NW == 2? 0 : (NW == 4? 1 : (NW == 8? 2 : (NW == 16? 3 : 4))); for (int i = 0; i < 28; ++i) atomicAdd(&ecf_prof2[row][i], (unsigned long long) p2[i]); }
#endif
    status_out = (uint32_t) status, np_out = (uint32_t) np, tried_out = tried, n_path_out = (uint32_t) n_path, wf_steps_out = wf_steps, wf_diag_out = (uint32_t) (wf_diag >> 6);
    return true;
}

// One workgroup per block, blocks taken one at a time from the list.  EcwArgs as for ec_heavy_kernel.
template <int NW, bool CERT = false>
__global__ __launch_bounds__(64 * NW) void ec_fused_kernel(EcwArgs a)
{
    extern __shared__ uint32_t ecf_lds[];
    const int t = (int) threadIdx.x;
    EcfShared sh;
    sh.cap_t = a.cap_t, sh.cap_c = a.cap_c, sh.cap_path = a.cap_path, sh.cap_fl = a.cap_f, sh.cap_fh = (int32_t) a.os_words;
    sh.endt = (int32_t *) ecf_lds, sh.bnd = sh.endt + 2 * NW, sh.red = sh.bnd + 2 * NW * 2 * ECF_S, sh.any = sh.red + 2 * NW * 2, sh.bc = sh.any + 2 * NW;
    sh.ts = ecf_lds + ecf_misc_words(NW), sh.cs = sh.ts + ecw_words(a.cap_t);
    uint32_t *p = sh.cs + ecw_words(a.cap_c);
    p += (p - ecf_lds) & 1;
    sh.fl = (uint8_t *) p;
    uint8_t *slab = a.slabs + (uint64_t) blockIdx.x * a.slab_bytes;
    sh.c_path = (uint64_t *) slab, sh.o_path = sh.c_path + a.cap_path;
    sh.os = (uint32_t *) (sh.o_path + a.cap_path);
    sh.fh = (uint8_t *) (sh.os + ecw_words(a.cap_c));
    sh.tab = CERT && a.os_slabs? (int32_t *) a.os_slabs + (uint64_t) blockIdx.x * ecf_tab_words(a.cap_t) : nullptr;      // (the fused launches do not use os_slabs otherwise)
    const uint64_t total = a.todo? a.n_todo : a.n_work;
    uint64_t pool_at = 0, pool_end = 0;
    for (;;) {
        if (t == 0) {
            const unsigned long long t0 = atomicAdd(a.next, 1ULL);
            sh.bc[0] = (int32_t) (uint32_t) t0, sh.bc[1] = (int32_t) (uint32_t) (t0 >> 32);
        }
        __syncthreads();
        const uint64_t t0 = (uint64_t) ecw_uniu((uint32_t) sh.bc[1]) << 32 | ecw_uniu((uint32_t) sh.bc[0]);
        __syncthreads();
        if (t0 >= total) break;
        const uint64_t wi = a.todo? a.todo[t0] : t0;
        EcWork wk;
        {
            const uint4 *q = (const uint4 *) (a.work + wi);
            const uint4 m0 = q[0], m1 = q[1], m2 = q[2];
            wk.beg_utg = (uint64_t) ecw_uniu(m0.y) << 32 | ecw_uniu(m0.x);
            wk.end_utg = (uint64_t) ecw_uniu(m0.w) << 32 | ecw_uniu(m0.z);
            wk.read = ecw_uniu(m1.x), wk.beg_pos = ecw_uniu(m1.y);
            wk.l = (int32_t) ecw_uniu(m1.z), wk.r = (int32_t) ecw_uniu(m1.w);
            wk.hs16 = ecw_uniu(m2.x), wk.lp = ecw_uniu(m2.y), wk.ln = ecw_uniu(m2.z), wk.pad = 0;
        }
        EcBlockOut o;
        o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0, o.tried = 0, o.n_path = 0, o.wf_steps = 0, o.wf_diag = 0, o.tier = ECF_TIER + (uint32_t) NW;
        const uint64_t tick0 = __builtin_amdgcn_s_memrealtime();
        if (ECW_RARE(wk.l < EC_MIN_ERR_SEQ_LEN)) {
            o.short_block = 1;                         // syncerr.c:502-504
        } else {
            uint32_t st = 0, np = 0;
            if (ECW_RARE(!(ecf_solve_block<NW, CERT>(a.lv, a.rd, wk, sh, a.max_edist, st, np, o.tried, o.n_path, o.wf_steps, o.wf_diag)))) {
                o.flags = 1;
                if (t == 0) a.todo_out[atomicAdd(a.todo_cnt, 1ULL)] = (uint32_t) wi;
            } else {
                o.status = st, o.np = np;
                if (st == EC_SUCCESS && np) {
                    if (ECW_RARE(pool_at + np > pool_end)) {
                        const unsigned long long want = np > ECW_POOL_CHUNK? np : ECW_POOL_CHUNK;
                        if (t == 0) {
                            const unsigned long long off = atomicAdd(a.pool_cursor, want);
                            sh.bc[2] = (int32_t) (uint32_t) off, sh.bc[3] = (int32_t) (uint32_t) (off >> 32);
                        }
                        __syncthreads();
                        pool_at = (uint64_t) ecw_uniu((uint32_t) sh.bc[3]) << 32 | ecw_uniu((uint32_t) sh.bc[2]), pool_end = pool_at + want;
                    }
                    o.path_off = pool_at;
                    if (pool_at + np <= a.pool_cap) for (uint32_t j = (uint32_t) t; j < np; j += 64 * NW) a.path_pool[pool_at + j] = sh.o_path[j];
                    pool_at += np;
                }
            }
        }
        o.ticks = (uint32_t) (__builtin_amdgcn_s_memrealtime() - tick0);
        if (t == 0) a.out[wi] = o;
        __syncthreads();
    }
}

}  // namespace oatk
