// oatk_amd/csrc/ec_heavy.hpp -- the error-block solver for the blocks that are NOT small: a WORKGROUP per block (round 5).
//
// The search is the one of ec_wave.hpp (dfs_search + wf_ed_core, syncerr.c:144-286, levdist.c:75-310), statement for statement; what differs is
// where the wavefront lives and how many lanes work on it.  What the config-1 surrogate showed (profiles/r05a_*): a block inside a
// repeat runs into MAX_DFS_PATH = 10 000 dead ends (syncerr.c:142), and nearly every one of them is a path that has left the read for good, whose
// alignment climbs score by score to the band's edge: ~bw steps on wavefronts of ~bw diagonals that advance a base or two each -- a million
// wavefront steps on hundreds of diagonals per block, every step a handful of instructions per diagonal and then a decision that needs all of
// them.  One wave with the wavefront in LDS (ec_wave.hpp, four rounds of 64 diagonals, an LDS round trip for the entry, the neighbours and the
// store) took ~2 us per step and 2.8 s for the worst block of 200 k reads.  Here
//   * a block has NW waves (256 lanes), every lane owns R diagonals IN REGISTERS: diagonal d lives in slot d + bw + 1, slot s in register
//     s / 256 of lane s % 256 -- a fixed place, so a wavefront that widens or is trimmed moves nothing;
//   * a step is: every lane compares one sixteen-base window per diagonal it owns (two LDS reads), the few diagonals that matched all sixteen
//     are run down by their whole wave 1024 bases at a time; the lowest diagonal that reached an end is found by a ballot per wave and ONE
//     barrier per step (each wave leaves its candidate and its two edge entries in LDS, double-buffered by the step's parity); the next
//     wavefront max(k[d-1], k[d]+1, k[d+1]+1) takes its neighbours through DPP wave shifts, the edge entries covering the seams between waves;
//   * the strings (read segment, consensus) are in LDS as before; DFS frames live in an LDS arena that spills into the block's HBM slab; the two
//     paths and the optimum consensus -- written once per arc, read at a tie -- live in the slab;
//   * everything the search decides is decided by every wave from the same LDS words, so the waves stay in step without a master.
// Results are those of ec_wave.hpp (and of the reference) bit for bit: same order of arcs, same tie rules, same dead-end accounting, same
// lowest-diagonal-first end of a step.
#pragma once
#include "ec_wave.hpp"

namespace oatk {

#define ECH_NEG (-(1 << 28))      // "no diagonal here": survives + 1 and max3 without ever looking like a furthest point
#define ECH_INF 0x7FFFFFFF
#define ECH_NW 4                  // waves per block

struct EchFrame {                 // state at the entry of one DFS level with siblings (syncerr.c:158-171), followed by k[n]
    uint32_t arc_i, arc_end;
    int32_t l0, score, t_end, q_end, n, s_lo, prev_off, depth;
};

// LDS carve-up (32-bit words): [red: 2 x NW x 2][edge: 2 x R x NW x 2][any: 2 x NW][bc: 8] ts cs frames
__host__ __device__ inline uint32_t ech_misc_words(int R) { return (uint32_t) (2 * ECH_NW * 2 + 2 * R * ECH_NW * 2 + 2 * ECH_NW + 8 + 1) & ~1u; }
__host__ __device__ inline uint32_t ech_lds_words(int32_t cap_t, int32_t cap_c, int32_t cap_fl, int R)
{
    return ((ech_misc_words(R) + ecw_words(cap_t) + ecw_words(cap_c) + 1u) & ~1u) + (uint32_t) cap_fl / 4u;
}
// the block's HBM slab: two paths, the optimum consensus, the frames that do not fit the LDS arena
__host__ __device__ inline uint64_t ech_slab_bytes(int32_t cap_c, int32_t cap_path, int32_t cap_fh)
{
    return ((uint64_t) 16 * (uint64_t) cap_path + (uint64_t) 4 * ecw_words(cap_c) + (uint64_t) cap_fh + 63) & ~63ULL;
}

template <int NW> __device__ __forceinline__ void ech_barrier()
{
    if constexpr (NW == 1) ecw_sync(); else __syncthreads();
}
template <int CTRL> __device__ __forceinline__ int32_t ech_dpp(int32_t old, int32_t src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}

struct EchShared {
    int32_t *red, *edge, *any, *bc;
    uint32_t *ts, *cs;
    uint8_t *fl;                  // LDS frame arena
    uint8_t *fh;                  // HBM frame arena (behind it)
    uint32_t *os;
    uint64_t *c_path, *o_path;
    int32_t cap_t, cap_c, cap_path, cap_fl, cap_fh;
    int32_t step_budget;          // wavefront steps after which a block is given up here (0: never): it starts again where several steps share a barrier (ec_fused.hpp)
};

// One wavefront step (levdist.c:156-224, extension mode, no traceback) over the registers k[R]; returns 1 when an end was reached.
// s_lo = slot of the lowest diagonal, n = diagonals; par = the parity of the double-buffered exchange words (flipped here).
template <int NW, int R>
__device__ __forceinline__ int ech_step(const EchShared &sh, int32_t tl, int32_t ql, int32_t bw, int32_t OFF, int32_t (&k)[R], int32_t &s_lo, int32_t &n,
                                        int32_t &t_end, int32_t &q_end, uint32_t &par)
{
    constexpr int T = 64 * NW;
    const int t = (int) threadIdx.x, lane = t & 63;
    const int wave = ecw_uni(t >> 6);
    const uint32_t *ts = sh.ts, *qs = sh.cs;
    int32_t kn[R];
    bool a0[R];
    int32_t first_slot = ECH_INF, first_k = 0;
    t_end = q_end = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        kn[r] = k[r], a0[r] = false;
        if (r * T + T <= s_lo || r * T >= s_lo + n) continue;                  // (uniform) none of this register's slots is in the wavefront
        const int32_t s = r * T + t;
        const bool valid = s >= s_lo && s < s_lo + n;
        int32_t kk = valid? k[r] : 0;
        const int32_t d = valid? s - OFF : 0;
        const bool act0 = valid && !(kk >= tl || kk + d >= ql);
        const int32_t lim = (ql - d < tl? ql - d : tl) - 1;
        {
            const int32_t rem = lim - kk;
            const int32_t p = act0? kk + 1 : 0;                                 // (lanes without work read, harmlessly, the head of the strings)
            const uint32_t x = ecw_win16(ts, p) ^ ecw_win16(qs, act0? p + d : 0);
            int32_t m = x? __builtin_ctz(x) >> 1 : 16;
            m = m < rem? m : rem;
            m = act0 && rem > 0? m : 0;
            kk += m;
            // the diagonals that matched all sixteen (the path that follows the read; every p-th diagonal inside a tandem array): their wave runs them down
            // together, 64 windows = 1024 bases a turn
            uint64_t mb = __ballot(act0 && m == 16 && kk < lim);
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
        const bool reached = act0 && (kk + d == ql - 1 || kk == tl - 1);
        kn[r] = act0? kk : k[r];
        a0[r] = act0;
        const uint64_t rb = __ballot(reached);
        if (rb && first_slot == ECH_INF) {
            const int fl = __builtin_ctzll(rb);
            first_slot = r * T + (wave << 6) + fl;
            first_k = (int32_t) ecw_lane((uint32_t) kk, fl);
        }
    }
    // what the other waves need: this wave's lowest end, and the entries at its two edges (as they are after the extension)
    int32_t *red = sh.red + par * (NW * 2), *edge = sh.edge + par * (R * NW * 2);
    if (NW > 1) {
        if (lane == 0) {
            red[wave * 2] = first_slot, red[wave * 2 + 1] = first_k;
#pragma unroll
            for (int r = 0; r < R; ++r) edge[(r * NW + wave) * 2] = kn[r];
        }
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < R; ++r) edge[(r * NW + wave) * 2 + 1] = kn[r];
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int32_t fs = ecw_uni(red[w * 2]), fk = ecw_uni(red[w * 2 + 1]);
            if (w != wave && fs < first_slot) first_slot = fs, first_k = fk;
        }
    }
    par ^= 1u;
    if (first_slot != ECH_INF) {                                                // a step ends at the LOWEST diagonal that reaches an end; only the diagonals below it are stored (levdist.c:166-180)
#pragma unroll
        for (int r = 0; r < R; ++r) if (a0[r] && r * T + t < first_slot) k[r] = kn[r];
        t_end = first_k, q_end = first_k + (first_slot - OFF);
        return 1;
    }
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
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r * T + T <= ns || r * T >= ns + n + 2) { k[r] = ECH_NEG; continue; }     // (uniform)
        const int32_t s = r * T + t;
        const int32_t c = kn[r];
        int32_t left = ech_dpp<0x138>(ECH_NEG, c);                                 // wave_shr:1 -- lane i takes lane i - 1 (slot s - 1); lane 0 keeps `old`
        int32_t right = ech_dpp<0x130>(ECH_NEG, c);                                // wave_shl:1 -- lane i takes lane i + 1 (slot s + 1); lane 63 keeps `old`
        if (NW > 1 || R > 1) {
            int32_t el = ECH_NEG, er = ECH_NEG;
            if (NW > 1) {
                if (wave > 0) el = edge[(r * NW + wave - 1) * 2 + 1];
                else if (r > 0) el = edge[((r - 1) * NW + NW - 1) * 2 + 1];
                if (wave < NW - 1) er = edge[(r * NW + wave + 1) * 2];
                else if (r < R - 1) er = edge[((r + 1) * NW) * 2];
            } else {
                if (r > 0) el = (int32_t) ecw_lane((uint32_t) kn[r > 0? r - 1 : 0], 63);
                if (r < R - 1) er = (int32_t) ecw_lane((uint32_t) kn[r < R - 1? r + 1 : r], 0);
            }
            left = lane == 0? el : left;
            right = lane == 63? er : right;
        }
        int32_t v = left;
        v = c + 1 > v? c + 1 : v;
        v = right + 1 > v? right + 1 : v;
        k[r] = s >= n_lo && s < n_lo + n_n? v : ECH_NEG;
    }
    s_lo = n_lo, n = n_n;
    return 0;
}

// wf_ed_core on its own over the workgroup solver's step (test entry: include/oatk_hip_ec.h, oatk_hip_debug_wf_ed with variant 1): one workgroup per job,
// strings copied to LDS, the wavefront resumed from one query length to the next exactly as the search resumes it
template <int NW, int R>
__global__ __launch_bounds__(64 * NW) void ech_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                            const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3, int32_t cap_words)
{
    constexpr int T = 64 * NW;
    extern __shared__ uint32_t ech_lds[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x;
    EchShared sh;
    sh.red = (int32_t *) ech_lds, sh.edge = sh.red + 2 * NW * 2, sh.any = sh.edge + 2 * R * NW * 2, sh.bc = sh.any + 2 * NW;
    sh.ts = ech_lds + ech_misc_words(R), sh.cs = sh.ts + cap_words;
    const int32_t tlen = tl[j], band = bw[j];
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];      // (packed with their pad words: ecw_words)
    for (uint64_t i = t; i < nt; i += T) sh.ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += T) sh.cs[i] = qw[qw_off[j] + i];
    const int32_t OFF = (band < 0? tlen : band) + 1;
    int32_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = r * T + t == OFF? -1 : ECH_NEG;               // the caller's initial state: diagonal 0, nothing matched, score 0 (syncerr.c:465-482)
    int32_t s_lo = OFF, n = 1, score = 0, t_end = -1, q_end = -1;
    uint32_t par = 0;
    __syncthreads();
    for (uint64_t s = step_off[j]; s < step_off[j + 1]; ++s) {
        const int32_t ql = step_ql[s];
        for (;;) {
            if (ech_step<NW, R>(sh, tlen, ql, band, OFF, k, s_lo, n, t_end, q_end, par)) break;
            ++score;
            if (band >= 0 && score > band) break;
        }
        if (t == 0) out3[3 * s] = score, out3[3 * s + 1] = t_end + 1, out3[3 * s + 2] = q_end + 1;
    }
}

// Solve one block with the whole workgroup.  Returns false when the block outgrows the carve-up (it is then re-run by the slab tier of ec_wave.hpp).
template <int NW, int R>
__device__ bool ech_solve_block(const EcLive &lv, const EcReads &rd, const EcWork &wk, const EchShared &sh, double max_edist,
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
    if (ECW_RARE(tl > sh.cap_t || 2 * bw + 3 > R * T)) return false;
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
    int32_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = r * T + t == OFF? -1 : ECH_NEG;
    int32_t s_lo = OFF, n = 1;
    uint32_t par = 0, apar = 0;
    if (t == 0) sh.c_path[0] = wk.beg_utg;
    int32_t fsz = 0, top = -1, nfr = 0;
    bool vpend = false;
    uint32_t v_arc = 0;
    int32_t v_depth = 0;

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
#pragma unroll
            for (int r = 0; r < R; ++r) { const int32_t s = r * T + t; if (s >= s_lo && s < s_lo + n) sv[s - s_lo] = k[r]; }
        } else {
            EchFrame *f = (EchFrame *) (sh.fh + (at - sh.cap_fl));
            if (t == 0) *f = hd;
            int32_t *sv = (int32_t *) (f + 1);
#pragma unroll
            for (int r = 0; r < R; ++r) { const int32_t s = r * T + t; if (s >= s_lo && s < s_lo + n) sv[s - s_lo] = k[r]; }
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
            n = ecw_uni(hd.n), s_lo = ecw_uni(hd.s_lo);
            if (in_lds) {
                const int32_t *sv = (const int32_t *) (sh.fl + top + sizeof(EchFrame));
#pragma unroll
                for (int r = 0; r < R; ++r) { const int32_t s = r * T + t; k[r] = s >= s_lo && s < s_lo + n? sv[s - s_lo] : ECH_NEG; }
            } else {
                const int32_t *sv = (const int32_t *) (sh.fh + (top - sh.cap_fl) + sizeof(EchFrame));
#pragma unroll
                for (int r = 0; r < R; ++r) { const int32_t s = r * T + t; k[r] = s >= s_lo && s < s_lo + n? sv[s - s_lo] : ECH_NEG; }
            }
        }
        ++tried;
        if (ECW_RARE(pre_idx != a)) pre = ecw_arc_load(lv.arc, a);
        const uint64_t w = ecw_uniu(pre.a.x);
        const int32_t ls = (int32_t) ecw_uniu(pre.a.y), ext = K - ls;
        const uint32_t w_hs16 = ecw_uniu(pre.a.z), w_mpos = ecw_uniu(pre.a.w), w_lp = ecw_uniu(pre.b.x), w_ln = ecw_uniu(pre.b.y);
        const int32_t t_end0 = t_end;
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
        // a vertex on an unbranched stretch that cannot be the end of the path needs no alignment of its own (ec_wave.hpp)
        if (edist == INT32_MAX && wk.end_utg != EC_NONE && wk.end_utg != w && w_ln == 1 && n_path < EC_MAX_DFS_PATH && c_len - K <= tl + bw && c_len >= bw + 3) {
            vpend = true, v_arc = w_lp, v_depth = depth + 1;
            continue;
        }
        // wf_ed_core (levdist.c:265-310)
        for (;;) {
            ++wf_steps, wf_diag += (uint64_t) n;
            if (ech_step<NW, R>(sh, tl, c_len, bw, OFF, k, s_lo, n, t_end, q_end, par)) break;
            ++score;
            if (ECW_RARE(score > bw)) break;
        }
        if (ECW_RARE(sh.step_budget > 0 && wf_steps > (uint32_t) sh.step_budget)) return false;
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
    }
    ech_barrier<NW>();
    status_out = (uint32_t) status, np_out = (uint32_t) np, tried_out = tried, n_path_out = (uint32_t) n_path, wf_steps_out = wf_steps, wf_diag_out = (uint32_t) (wf_diag >> 6);
    return true;
}

// One workgroup per block, blocks taken one at a time from the list (a.todo, longest first where the host can tell).  EcwArgs as for ec_wave_kernel; a.slabs /
// a.slab_bytes = the workgroups' HBM slabs, a.cap_f = the LDS frame arena, a.os_words = bytes of the slab's frame arena.
template <int NW, int R>
__global__ __launch_bounds__(64 * NW) void ec_heavy_kernel(EcwArgs a)
{
    extern __shared__ uint32_t ech_lds[];
    const int t = (int) threadIdx.x;
    EchShared sh;
    sh.cap_t = a.cap_t, sh.cap_c = a.cap_c, sh.cap_path = a.cap_path, sh.cap_fl = a.cap_f, sh.cap_fh = (int32_t) a.os_words;
    sh.step_budget = a.cap_w;
    sh.red = (int32_t *) ech_lds, sh.edge = sh.red + 2 * NW * 2, sh.any = sh.edge + 2 * R * NW * 2, sh.bc = sh.any + 2 * NW;
    sh.ts = ech_lds + ech_misc_words(R), sh.cs = sh.ts + ecw_words(a.cap_t);
    uint32_t *p = sh.cs + ecw_words(a.cap_c);
    p += (p - ech_lds) & 1;
    sh.fl = (uint8_t *) p;
    uint8_t *slab = a.slabs + (uint64_t) blockIdx.x * a.slab_bytes;
    sh.c_path = (uint64_t *) slab, sh.o_path = sh.c_path + a.cap_path;
    sh.os = (uint32_t *) (sh.o_path + a.cap_path);
    sh.fh = (uint8_t *) (sh.os + ecw_words(a.cap_c));
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
        o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0, o.tried = 0, o.n_path = 0, o.wf_steps = 0, o.wf_diag = 0, o.tier = (NW == 1? 16u : 8u) + (uint32_t) R;
        const uint64_t tick0 = __builtin_amdgcn_s_memrealtime();
        if (ECW_RARE(wk.l < EC_MIN_ERR_SEQ_LEN)) {
            o.short_block = 1;                         // syncerr.c:502-504
        } else {
            uint32_t st = 0, np = 0;
            if (ECW_RARE(!(ech_solve_block<NW, R>(a.lv, a.rd, wk, sh, a.max_edist, st, np, o.tried, o.n_path, o.wf_steps, o.wf_diag)))) {
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
