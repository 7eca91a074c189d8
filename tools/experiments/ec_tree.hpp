// oatk_amd/csrc/ec_tree.hpp -- the error-block solver for blocks whose SEARCH is heavy: the waves of a workgroup share one block's search tree (round 5).
//
// What the config-1 surrogate's heaviest blocks look like (tests/trace/ec_trace.c, profiles/r05c_*): a block inside a tandem array walks a graph of ~50
// vertices for 40 000 arcs and 10 000 dead ends (MAX_DFS_PATH, syncerr.c:142); four steps in five belong to the ~10 000 arcs that die by score -- a
// hundred wavefront steps each on hundreds of diagonals -- and the tree branches at ~1 700 levels with four to seven arcs each.  ec_heavy.hpp put four
// waves on ONE wavefront (a barrier and an LDS exchange per step: ~1 us a step, 1.36 s for the worst block of 200 k reads).  But the alignments of
// sibling arcs start from the same saved state and do not depend on each other (wf_ed_core is a pure function of target, consensus and saved wavefront;
// the search couples them only through dfs_info: the dead-end counter and the optimum).  So here
//   * a workgroup has ECT_W waves and ONE block; every wave runs whole alignments on its own -- the wavefront in registers (ECT_R diagonals per lane,
//     2 bw + 3 <= 64 ECT_R), neighbours through DPP shifts, no barrier and no LDS exchange inside a step -- on its own copy of the consensus in LDS;
//   * the wave that owns a level with several live arcs PUBLISHES it (a header in LDS with an atomic cursor; the saved wavefront in its HBM frame arena).
//     The owner takes the arcs in order and works on them in its own context; an idle wave may take an arc the owner has not reached yet (a THIEF): it
//     copies the consensus and path prefix from the owner, runs the arc's whole subtree as a sub-task -- publishing levels of its own -- and leaves a LOG;
//   * what the search decides through dfs_info is replayed exactly when the owner reaches the arc:
//       - n_path: a sub-task counts dead ends from zero with a budget B that is an upper bound of what the real search would have left; for every dead
//         end it logs pe = its count when the dead end's LEVEL was entered.  Under a smaller real budget r <= B the search evaluates exactly the nodes whose
//         level was entered below r (before the cap the two runs are the same run; after it only siblings of entered levels are evaluated, and their
//         levels were entered earlier still), so the owner keeps the entries with c_join + pe < its own budget and adds their number to its count;
//       - the optimum: every in-band arrival at the sink is logged (score, whether the target advanced, path as arc indices) with its pe, filtered the
//         same way, and the root applies the events in preorder with the reference's rules (syncerr.c:209-252), rebuilding a consensus from its path
//         where two optima have to be compared;
//       - once the root's counter reaches MAX_DFS_PATH everything not merged yet lies to its right in preorder and has no budget left: a flag stops the
//         sub-tasks from entering further levels (they still evaluate the siblings of entered levels, as the reference does).
//     An arc the owner reaches first is simply run in the owner's context: with no thief about, the search is the sequential one, statement for statement.
//   * a sub-task that outgrows its log, its frames or its path gives the arc back (ABORT); the owner runs it itself; a root that outgrows them leaves the
//     block to the next tier (ec_heavy.hpp / the slab tier).
// Results are the reference's bit for bit: status, optimum path, and the final n_path; `tried`, wf_steps and wf_diag count the work that was DONE, which
// includes what thieves did beyond the cap.
#pragma once
#include "../../oatk_amd/csrc/ec_heavy.hpp"

namespace oatk {

#define ECT_W 8                   // waves per workgroup
#define ECT_R 8                   // diagonals per lane
#define ECT_NFH 32                // published levels per wave (headers in LDS)
#define ECT_NFX 1024              // further levels of a wave, deeper than those: headers in its slab, nobody else takes their arcs
#define ECT_HW 16                 // words per level header
#define ECT_TIER 24u
#define ECT_STEAL_ARCS 32         // arcs of a level a thief may take (one bit each in the header's masks)

// control words of a workgroup (LDS)
#define ECT_C_Q0 0
#define ECT_C_Q1 1
#define ECT_C_CAP 2               // the root's counter has reached MAX_DFS_PATH
#define ECT_C_DONE 3
#define ECT_C_NOSTEAL 4
#define ECT_C_TRIED 5
#define ECT_C_STEPS 6
#define ECT_C_DIAG 7
#define ECT_C_STEALS 8
#define ECT_C_OK 9
#define ECT_C_P0 10
#define ECT_C_P1 11
#define ECT_C_ABORTS 12
#define ECT_C_LIGHT 13             // the search has been light so far (few wavefront steps per arc): a sub-task would cost more than it saves
#define ECT_C_NF 16               // [ECT_W] published levels of each wave
#define ECT_C_FLOOR 32            // [ECT_W] the shallowest of a wave's levels a thief may take an arc of: only the few deepest (see ect_solve_block)
#define ECT_C_WAIT 40             // (diagnostics, -DECT_PROF prints them for slow blocks) ticks the root / the owners of sub-tasks spent waiting for a thief
#define ECT_C_WAIT_SUB 41
#define ECT_C_TASK_TICKS 42       // ticks thieves spent inside sub-tasks
#define ECT_C_KEPT 43             // dead-end entries the root kept / dropped when it merged logs
#define ECT_C_DROPPED 44
#define ECT_C_SUBDEAD 45          // dead ends sub-tasks logged
#define ECT_MAXCTX 7              // tasks a wave may have begun and not finished, besides the one it is running (it takes a sub-task while it waits for another wave)
#define ECT_C_REM 48              // [ECT_W][ECT_MAXCTX + 1] what a wave's task at each nesting depth may still count, at most (live: it shrinks as the tasks it hangs below count on)
#define ECT_C_CNT (ECT_C_REM + ECT_W * (ECT_MAXCTX + 1))      // [ECT_W][ECT_MAXCTX + 1] what that task has counted so far (live)
#define ECT_C_CTX (ECT_C_CNT + ECT_W * (ECT_MAXCTX + 1))      // [ECT_W][ECT_MAXCTX][ECT_CTXW] the tasks a wave has set aside
#define ECT_CTXW 20
#define ECT_CTL_WORDS (ECT_C_CTX + ECT_W * ECT_MAXCTX * ECT_CTXW)
// level header (LDS): what a thief needs to find and take an arc, and what the owner needs every time it comes back to the level
#define ECT_H_CUR 0               // next arc nobody has taken (CAS)
#define ECT_H_ARC0 1
#define ECT_H_END 2
#define ECT_H_OFF 3               // the frame in the owner's arena
#define ECT_H_DONE 4              // bit i: arc arc0 + i was run by a thief and its log is complete
#define ECT_H_ABORT 5             // bit i: ... and given back
#define ECT_H_MERGED 6            // arcs merged so far | the owner is inside arc arc0 + merged (bit 31)
#define ECT_H_DEPTH 7              // depth of the level's vertex | nesting depth of the owner's task << 24 (which of its REM / CNT words count for a task that hangs below this level)
#define ECT_H_SEND 8              // arcs below this one may be taken by a thief: none before the owner is back from the level's first arc, then as many as the budget is likely to reach
#define ECT_H_BUDGET 9
#define ECT_H_L0 10
#define ECT_H_KEY0 11             // where the level's first arc lies in preorder (64-bit fixed point; smaller = earlier), and what one arc spans: what a thief goes by
#define ECT_H_KEY1 12
#define ECT_H_KW0 13
#define ECT_H_KW1 14
#define ECT_H_GO 15               // bit i: the owner has arrived at arc arc0 + i and waits for it: a sub-task that held back (see `pace`) goes ahead

struct EctFrame { int32_t score, t_end, q_end, n, s_lo; uint32_t lhi0, lhi1; int32_t pe; };          // followed by k[n] (padded to four), then EctRes[min(arcs, 32)]
struct EctRes { uint32_t dead_top, n_dead, evt_off, evt_words, wave; int32_t limit; uint32_t pad1, pad2; };     // limit: the log is good for budgets up to this one (a task taken on speculation counts up to a small budget of its own)

__host__ __device__ inline uint32_t ect_lds_words(int32_t cap_t, int32_t cap_c)
{
    return (uint32_t) ECT_CTL_WORDS + (uint32_t) (ECT_W * ECT_NFH * ECT_HW) + ecw_words(cap_t) + (uint32_t) ECT_W * ecw_words(cap_c);
}
// a wave's part of the slab: frames, log, the two path arrays
__host__ __device__ inline uint64_t ect_wave_bytes(int32_t cap_path, int32_t cap_fa, int32_t cap_la)
{
    return ((uint64_t) cap_fa + (uint64_t) cap_la + 12ULL * (uint64_t) cap_path + 4ULL * ECT_NFX * ECT_HW + 63) & ~63ULL;
}
// ... and behind the waves' parts the root's optimum: path (vertices, arcs), consensus, and a second consensus to compare it with
__host__ __device__ inline uint64_t ect_slab_bytes(int32_t cap_c, int32_t cap_path, int32_t cap_fa, int32_t cap_la)
{
    return ((uint64_t) ECT_W * ect_wave_bytes(cap_path, cap_fa, cap_la) + 12ULL * (uint64_t) cap_path + 8ULL * ecw_words(cap_c) + 63) & ~63ULL;
}

__device__ __forceinline__ uint32_t ect_vld(const uint32_t *p) { return ecw_uniu(*(const volatile uint32_t *) p); }            // LDS word another wave may write
__device__ __forceinline__ uint32_t ect_gld(const uint32_t *p) { return ecw_uniu(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }   // HBM word another wave wrote
__device__ __forceinline__ void ect_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void ect_acquire() { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// One wavefront step (levdist.c:156-224, extension mode, no traceback) by ONE wave over the registers k[R]: diagonal d lives in slot d + OFF, slot s in
// register s / 64 of lane s % 64; slots outside the wavefront hold ECH_NEG.  Returns 1 when an end was reached.
template <int R>
__device__ __forceinline__ int ect_step(const uint32_t *ts, const uint32_t *qs, int32_t tl, int32_t ql, int32_t bw, int32_t OFF, int32_t (&k)[R], int32_t &s_lo, int32_t &n,
                                        int32_t &t_end, int32_t &q_end)
{
    static_assert(R % 2 == 0, "registers are handled in pairs");
    const int lane = (int) threadIdx.x & 63;
    const int r_lo = s_lo >> 6, r_hi = (s_lo + n - 1) >> 6;                     // (uniform) registers with slots of the wavefront
    int32_t kn[R];
    uint32_t actb = 0, reab = 0;                                                // bit r: this lane's slot of register r was extended / reached an end
    t_end = q_end = -1;
#pragma unroll
    for (int g = 0; g < R; g += 2) {
        kn[g] = k[g], kn[g + 1] = k[g + 1];
        if (g + 1 < r_lo || g > r_hi) continue;
        // two registers side by side: their LDS round trips overlap
        int32_t kk[2], dd[2], lim[2];
        bool act[2], more[2];
        uint32_t x[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int32_t s = (g + u) * 64 + lane;
            const bool valid = s >= s_lo && s < s_lo + n;
            kk[u] = valid? k[g + u] : 0;
            dd[u] = valid? s - OFF : 0;
            act[u] = valid && !(kk[u] >= tl || kk[u] + dd[u] >= ql);
            lim[u] = (ql - dd[u] < tl? ql - dd[u] : tl) - 1;
            const int32_t p = act[u]? kk[u] + 1 : 0;                            // (lanes without work read, harmlessly, the head of the strings)
            x[u] = ecw_win16(ts, p) ^ ecw_win16(qs, act[u]? p + dd[u] : 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int32_t rem = lim[u] - kk[u];
            int32_t m = x[u]? __builtin_ctz(x[u]) >> 1 : 16;
            m = m < rem? m : rem;
            m = act[u] && rem > 0? m : 0;
            kk[u] += m;
            more[u] = act[u] && m == 16 && kk[u] < lim[u];
        }
        if (__ballot(more[0] || more[1])) {
            // diagonals that matched all sixteen (the path that follows the read; every p-th diagonal inside a tandem array): three more windows lane by lane ...
            for (int it = 0; it < 3; ++it) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int32_t p = more[u]? kk[u] + 1 : 0;
                    x[u] = ecw_win16(ts, p) ^ ecw_win16(qs, more[u]? p + dd[u] : 0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int32_t rem = lim[u] - kk[u];
                    int32_t m = x[u]? __builtin_ctz(x[u]) >> 1 : 16;
                    m = m < rem? m : rem;
                    m = more[u]? m : 0;
                    kk[u] += m;
                    more[u] = more[u] && m == 16 && kk[u] < lim[u];
                }
                if (!__ballot(more[0] || more[1])) break;
            }
            // ... and what still goes on is run down by the whole wave, 1024 bases a turn
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint64_t mb = __ballot(more[u]);
                while (mb) {
                    const int l = __builtin_ctzll(mb);
                    mb &= mb - 1;
                    int32_t bk = (int32_t) ecw_lane((uint32_t) kk[u], l);
                    const int32_t bd = (int32_t) ecw_lane((uint32_t) dd[u], l), blim = (int32_t) ecw_lane((uint32_t) lim[u], l);
                    for (;;) {
                        const int32_t rr = blim - bk - (lane << 4);             // bases left from this lane's window on
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
                    kk[u] = lane == l? bk : kk[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kn[g + u] = act[u]? kk[u] : k[g + u];
            actb |= (uint32_t) act[u] << (g + u);
            reab |= (uint32_t) (act[u] && (kk[u] + dd[u] == ql - 1 || kk[u] == tl - 1)) << (g + u);
        }
    }
    if (ECW_RARE(__ballot(reab != 0) != 0)) {                                   // a step ends at the LOWEST diagonal that reaches an end; only the diagonals below it are stored (levdist.c:166-180)
        int32_t first_slot = ECH_INF, first_k = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t rb = __ballot((reab >> r) & 1u);
            if (rb && first_slot == ECH_INF) {
                const int fl = __builtin_ctzll(rb);
                first_slot = r * 64 + fl;
                first_k = (int32_t) ecw_lane((uint32_t) kn[r], fl);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) if (((actb >> r) & 1u) && r * 64 + lane < first_slot) k[r] = kn[r];
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
        if (r * 64 + 64 <= ns || r * 64 >= ns + n + 2) { k[r] = ECH_NEG; continue; }      // (uniform)
        const int32_t s = r * 64 + lane;
        const int32_t c = kn[r];
        int32_t left = ech_dpp<0x138>(ECH_NEG, c);                                 // wave_shr:1 -- lane i takes lane i - 1 (slot s - 1); lane 0 keeps `old`
        int32_t right = ech_dpp<0x130>(ECH_NEG, c);                                // wave_shl:1 -- lane i takes lane i + 1 (slot s + 1); lane 63 keeps `old`
        int32_t el = ECH_NEG, er = ECH_NEG;
        if (r > 0) el = (int32_t) ecw_lane((uint32_t) kn[r > 0? r - 1 : 0], 63);
        if (r < R - 1) er = (int32_t) ecw_lane((uint32_t) kn[r < R - 1? r + 1 : r], 0);
        left = lane == 0? el : left;
        right = lane == 63? er : right;
        int32_t v = left;
        v = c + 1 > v? c + 1 : v;
        v = right + 1 > v? right + 1 : v;
        k[r] = s >= n_lo && s < n_lo + n_n? v : ECH_NEG;
    }
    s_lo = n_lo, n = n_n;
    return 0;
}

// wf_ed_core on its own over ect_step (test entry: oatk_hip_debug_wf_ed_wg with R = 8): one wave per job
template <int R>
__global__ __launch_bounds__(64) void ect_wf_ed_kernel(const uint32_t *tw, const uint64_t *tw_off, const int32_t *tl, const uint32_t *qw, const uint64_t *qw_off,
                                                       const int32_t *bw, const int32_t *step_ql, const uint64_t *step_off, int32_t *out3, int32_t cap_words)
{
    extern __shared__ uint32_t ect_lds[];
    const uint64_t j = blockIdx.x;
    const int t = (int) threadIdx.x;
    uint32_t *ts = ect_lds, *cs = ts + cap_words;
    const int32_t tlen = tl[j], band = bw[j];
    const uint64_t nt = tw_off[j + 1] - tw_off[j], nq = qw_off[j + 1] - qw_off[j];
    for (uint64_t i = t; i < nt; i += 64) ts[i] = tw[tw_off[j] + i];
    for (uint64_t i = t; i < nq; i += 64) cs[i] = qw[qw_off[j] + i];
    const int32_t OFF = (band < 0? tlen : band) + 1;
    int32_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = r * 64 + t == OFF? -1 : ECH_NEG;
    int32_t s_lo = OFF, n = 1, score = 0, t_end = -1, q_end = -1;
    ecw_sync();
    for (uint64_t s = step_off[j]; s < step_off[j + 1]; ++s) {
        const int32_t ql = step_ql[s];
        for (;;) {
            if (ect_step<R>(ts, cs, tlen, ql, band, OFF, k, s_lo, n, t_end, q_end)) break;
            ++score;
            if (band >= 0 && score > band) break;
        }
        if (t == 0) out3[3 * s] = score, out3[3 * s + 1] = t_end + 1, out3[3 * s + 2] = q_end + 1;
    }
}

// The free arc a wave with nothing to run should take: of the levels the other waves have published (their few deepest, ECT_C_FLOOR), the arc that comes first in preorder
// within [w_lo, w_hi) -- what the search itself would try next -- and the deeper level on a tie.  Every lane looks at some headers.  Returns wave * ECT_NFH + slot, or -1.
__device__ __forceinline__ int32_t ect_find_arc(const uint32_t *ctl, const uint32_t *fh, int me, uint64_t w_lo, uint64_t w_hi)
{
    const int lane = (int) threadIdx.x & 63;
    uint32_t bk1 = 0xFFFFFFFFu, bk0 = 0xFFFFFFFFu;
    int32_t bdep = -1, bidx = -1;
    for (int i = lane; i < ECT_W * ECT_NFH; i += 64) {
        const int wv = i / ECT_NFH, sl = i % ECT_NFH;
        if (wv == me || (uint32_t) sl >= *(const volatile uint32_t *) (ctl + ECT_C_NF + wv) || (uint32_t) sl < *(const volatile uint32_t *) (ctl + ECT_C_FLOOR + wv)) continue;
        const volatile uint32_t *h = fh + ((uint32_t) wv * ECT_NFH + (uint32_t) sl) * ECT_HW;
        const uint32_t cur = h[ECT_H_CUR], a0 = h[ECT_H_ARC0], lim = h[ECT_H_SEND];
        if (cur >= a0 && cur < lim) {
            const uint64_t key = ((uint64_t) h[ECT_H_KEY1] << 32 | h[ECT_H_KEY0]) + (uint64_t) (cur - a0) * ((uint64_t) h[ECT_H_KW1] << 32 | h[ECT_H_KW0]);
            const uint32_t k1 = (uint32_t) (key >> 32), k0 = (uint32_t) key;
            const int32_t dep = (int32_t) (h[ECT_H_DEPTH] & 0xFFFFFFu);
            if (key >= w_lo && key < w_hi && (k1 < bk1 || (k1 == bk1 && (k0 < bk0 || (k0 == bk0 && dep > bdep))))) bk1 = k1, bk0 = k0, bdep = dep, bidx = i;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t o1 = (uint32_t) __shfl_xor((int) bk1, o, 64), o0 = (uint32_t) __shfl_xor((int) bk0, o, 64);
        const int32_t odep = __shfl_xor(bdep, o, 64), oidx = __shfl_xor(bidx, o, 64);
        const bool less = o1 < bk1 || (o1 == bk1 && (o0 < bk0 || (o0 == bk0 && (odep > bdep || (odep == bdep && oidx < bidx)))));
        if (oidx >= 0 && (bidx < 0 || less)) bk1 = o1, bk0 = o0, bdep = odep, bidx = oidx;
    }
    return ecw_uni(bidx);
}

struct EctShared {
    uint32_t *ctl, *fh, *ts, *cs_all;                 // LDS
    uint8_t *slab;                                    // the workgroup's HBM slab
    uint64_t wave_bytes;
    int32_t cap_t, cap_c, cap_path, cap_fa, cap_la, cw;      // cw = words of one consensus
    int32_t reach;                                    // levels of a wave, counted from its deepest, whose arcs a thief may take
    int32_t spec_budget;                              // dead ends a sub-task may count when it was taken before its level's owner was back from the level's first arc
};

// append the part of the k-mer of an arc's target that lies beyond the overlap (syncerr.c:186-190; ec_wave.hpp) to a packed consensus of c_len bases -- one wave
__device__ __forceinline__ void ect_append(uint32_t *cs, int32_t c_len, int32_t ext, int32_t ls, int K, const uint8_t *vs, uint32_t mpos, bool w_rev)
{
    const int lane = (int) threadIdx.x & 63;
    const uint32_t pos = mpos >> 1;
    const bool asc = w_rev == (bool) (mpos & 1u);
    const int32_t w0 = c_len >> 4, w1 = (c_len + ext - 1) >> 4;
    for (int32_t wb = w0; wb <= w1; wb += 64) {
        const int32_t wi = wb + lane;
        if (wi > w1) continue;
        const int32_t t0 = (wi << 4) - c_len;
        uint32_t x = asc? ecw_gather16(vs, (int64_t) pos + ls + t0, false) : ecw_gather16(vs, (int64_t) pos + K - 1 - ls - t0, true);
        if (t0 < 0) {
            const uint32_t keep = (1u << ((uint32_t) (-t0) << 1)) - 1u;
            x = (cs[wi] & keep) | (x & ~keep);
        }
        cs[wi] = x;
    }
}

// the consensus a path of arcs spells (at least `upto` bases of it), into an HBM buffer -- rare: two optima have to be compared and one of them came out of a log
__device__ void ect_rebuild(const EcLive &lv, const EcReads &rd, const uint32_t *arcs, int32_t n_arcs, int32_t upto, uint32_t *dst)
{
    int32_t c_len = 0;
    for (int32_t i = 0; i < n_arcs && c_len < upto; ++i) {
        const EcwArcRegs ar = ecw_arc_load(lv.arc, ect_gld(arcs + i));
        const uint32_t w = ecw_uniu(ar.a.x);
        const int32_t ls = (int32_t) ecw_uniu(ar.a.y), ext = rd.K - ls;
        ect_append(dst, c_len, ext, ls, rd.K, rd.hoco_s + ((uint64_t) ecw_uniu(ar.a.z) << 4), ecw_uniu(ar.a.w), (w & 1u) != 0);
        c_len += ext;
        ecw_sync();
    }
}

// One block by the whole workgroup.  Returns false (on every wave) when the block outgrows the carve-up.
template <int R>
__device__ bool ect_solve_block(const EcLive &lv, const EcReads &rd, const EcWork &wk, const EctShared &sh, double max_edist,
                                uint32_t &status_out, uint32_t &np_out, uint32_t &n_path_out, const uint64_t *&o_path_out)
{
    const int lane = (int) threadIdx.x & 63;
    const int me = ecw_uni((int) threadIdx.x >> 6);
    const int K = rd.K;
    const int32_t tl = wk.l;
    int32_t bw = (int32_t) ceil((double) tl * max_edist);
    if (bw < EC_MIN_ERR_BASE) bw = EC_MIN_ERR_BASE;
    const int32_t OFF = bw + 1;
    if (ECW_RARE(tl > sh.cap_t || 2 * bw + 3 > R * 64)) return false;
    uint32_t *ctl = sh.ctl, *fh = sh.fh;
    const uint32_t *ts = sh.ts;
    uint32_t *cs = sh.cs_all + (uint32_t) me * (uint32_t) sh.cw;
    uint8_t *wslab = sh.slab + (uint64_t) me * sh.wave_bytes;
    uint8_t *fa = wslab, *la = wslab + sh.cap_fa;
    uint64_t *c_path = (uint64_t *) (la + sh.cap_la);
    uint32_t *c_arc = (uint32_t *) (c_path + sh.cap_path);
    uint64_t *o_path = (uint64_t *) (sh.slab + (uint64_t) ECT_W * sh.wave_bytes);
    uint32_t *o_arc = (uint32_t *) (o_path + sh.cap_path);
    uint32_t *os = o_arc + sh.cap_path, *es = os + sh.cw;
    {   // target: the read segment, reverse-complemented for a leading block (get_kmer_dna_seq, syncmer.c:1237) -- all the waves together
        const uint8_t *hs = rd.hoco_s + ((uint64_t) wk.hs16 << 4);
        const bool R_ = wk.r != 0;
        constexpr int T = 64 * ECT_W;
        const int t = (int) threadIdx.x;
        for (int32_t wb = 0; (wb << 4) < tl; wb += T) {
            const int32_t wi = wb + t;
            if ((wi << 4) < tl) {
                const int64_t start = R_? (int64_t) wk.beg_pos + tl - 1 - (wi << 4) : (int64_t) wk.beg_pos + (wi << 4);
                sh.ts[wi] = ecw_gather16(hs, start, R_);
            }
        }
        for (int i = t; i < ECT_C_CTX; i += T) if (i >= 2) ctl[i] = 0;
    }
    __syncthreads();

    const uint32_t block_t0 = (uint32_t) __builtin_amdgcn_s_memrealtime();
    // ---- a wave's state: the search it is running (the root's, or a sub-task's) ----
    bool root = false, aborted = false;
    int32_t c = 0, B = 0;                             // dead ends counted by this task, and its budget
    int32_t status = EC_FAILURE, edist = INT32_MAX, s_edist = INT32_MAX, o_len = 0, np = 0, o_na = 0;      // root only (o_na: arcs that spell the optimum consensus -- one more than the path has when its last vertex was dropped, syncerr.c:226-227)
    bool os_valid = false;
    uint32_t la_lo = 0, la_hi = (uint32_t) sh.cap_la; // the free part of this wave's log arena: events grow up from la_lo, dead-end entries down from la_hi
    uint32_t evt_start = 0, evt_pos = 0, dead_top = 0, n_dead = 0;
    int32_t c_len = 0, score = 0, t_end = 0, q_end = 0, s_lo = OFF, n = 1;
    int32_t k[R];
    int32_t fa_top = 0, nfr = 0;
    bool vpend = false;
    uint32_t v_arc = 0;
    int32_t v_depth = 0, v_pe = 0;
    uint32_t tried = 0, wf_steps = 0;
    uint64_t wf_diag = 0;
    uint32_t *myfh = fh + (uint32_t) me * (ECT_NFH * ECT_HW);

    uint32_t *myhx = (uint32_t *) (c_arc + sh.cap_path);      // headers of this wave's levels beyond the LDS table
    int task_owner = 0, task_owner_d = 0;                     // the wave (and the nesting depth of its task) whose level this wave's sub-task hangs below
    int nctx = 0, nfr_base = 0;                               // tasks set aside; this task's first level on the wave's stack of levels
 bool pace = false;                                        // this sub-task was taken on speculation and holds back (below)
    int32_t pace_pe = 0;
    bool hit_own = false;                                     // this sub-task has refused to enter a level because its OWN budget was spent (its log then holds for budgets up to B only)
    int res_ow = 0, res_slot = 0;                             // where this sub-task's result goes: the level, and the arc
    uint32_t res_arc = 0;
    // where this wave's task lies in preorder: [t_lo, t_hi) of a 64-bit line.  A level takes the upper sixteenth of what is left below the levels above it (the deeper
    // level comes FIRST: its arcs are tried before the remaining arcs of the levels above), and shares it among its arcs; a thief's task has its arc's share.  A
    // priority -- and the test for `hangs below the arc I am waiting for` (below): exact, the shares are nested and disjoint; when the line runs out a level's share is
    // empty and nothing is found below it.
    uint64_t t_lo = 0, t_hi = 0;

    auto hdr = [&](int wave, int slot) -> uint32_t * { return fh + ((uint32_t) wave * ECT_NFH + (uint32_t) slot) * ECT_HW; };
    // a field of the header of this wave's level fi: LDS for the first ECT_NFH, the slab behind them
    auto H_LD = [&](int fi, int field) -> uint32_t { return fi < ECT_NFH? ect_vld(myfh + fi * ECT_HW + field) : ect_gld(myhx + (fi - ECT_NFH) * ECT_HW + field); };
    auto H_ST = [&](int fi, int field, uint32_t v) {
        if (lane == 0) {
            if (fi < ECT_NFH) *(volatile uint32_t *) (myfh + fi * ECT_HW + field) = v;
            else __hip_atomic_store(myhx + (fi - ECT_NFH) * ECT_HW + field, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto frame_k_words = [&](int32_t nn) -> int32_t { return (nn + 3) & ~3; };
    // take arc `want` of a level if nobody has (owner), or the next free one below `limit` (thief); 0xFFFFFFFF = none
    auto claim = [&](uint32_t *h, uint32_t limit, bool exact, uint32_t want) -> uint32_t {
        uint32_t r = 0xFFFFFFFFu;
        if (lane == 0) {
            uint32_t old = *(volatile uint32_t *) (h + ECT_H_CUR);
            while (old < limit && (!exact || old == want)) {
                const uint32_t seen = atomicCAS(h + ECT_H_CUR, old, old + 1u);
                if (seen == old) { r = old; break; }
                old = seen;
            }
        }
        return ecw_uniu(r);
    };
    auto push_frame = [&](uint32_t lp, uint32_t ln, int32_t depth, int32_t pe) -> bool {
        const int32_t nres = ln < ECT_STEAL_ARCS? (int32_t) ln : ECT_STEAL_ARCS;
        const int32_t bytes = (int32_t) sizeof(EctFrame) + 4 * frame_k_words(n) + (int32_t) sizeof(EctRes) * nres;
        if (nfr == ECT_NFH + ECT_NFX || fa_top + bytes > sh.cap_fa) return false;
        EctFrame *f = (EctFrame *) (fa + fa_top);
        int32_t *sv = (int32_t *) (f + 1);
        if (lane == 0) f->score = score, f->t_end = t_end, f->q_end = q_end, f->n = n, f->s_lo = s_lo, f->lhi0 = (uint32_t) t_hi, f->lhi1 = (uint32_t) (t_hi >> 32), f->pe = pe;
#pragma unroll
        for (int r = 0; r < R; ++r) { const int32_t s = r * 64 + lane; if (s >= s_lo && s < s_lo + n) sv[s - s_lo] = k[r]; }
        H_ST(nfr, ECT_H_ARC0, lp), H_ST(nfr, ECT_H_END, lp + ln), H_ST(nfr, ECT_H_OFF, (uint32_t) fa_top), H_ST(nfr, ECT_H_DONE, 0), H_ST(nfr, ECT_H_ABORT, 0), H_ST(nfr, ECT_H_MERGED, 0);
        H_ST(nfr, ECT_H_DEPTH, (uint32_t) depth | (uint32_t) nctx << 24), H_ST(nfr, ECT_H_GO, 0), H_ST(nfr, ECT_H_SEND, ln < ECT_STEAL_ARCS? lp + ln : lp + ECT_STEAL_ARCS), H_ST(nfr, ECT_H_BUDGET, (uint32_t) (B - pe)), H_ST(nfr, ECT_H_L0, (uint32_t) c_len);
        {
            // (when the line has run out the level lies NOWHERE: a key above every share, so that no wave takes it for something below the arc it waits for)
            const uint64_t width = (t_hi - t_lo) >> 4, l_lo = width? t_hi - width : 1ULL << 63, kw = width / (uint64_t) ln;

            H_ST(nfr, ECT_H_KEY0, (uint32_t) l_lo), H_ST(nfr, ECT_H_KEY1, (uint32_t) (l_lo >> 32)), H_ST(nfr, ECT_H_KW0, (uint32_t) kw), H_ST(nfr, ECT_H_KW1, (uint32_t) (kw >> 32));
            if (width) t_hi = l_lo;
        }
        ect_release();                                 // (the frame and the header are in place before the cursor says so)
        H_ST(nfr, ECT_H_CUR, lp);
        // Thieves take arcs of a wave's few DEEPEST levels only.  What lies further up is further right in preorder: whole subtrees the search reaches late or -- with
        // MAX_DFS_PATH dead ends behind it -- never, and a wave that has taken one is gone for as long as the rest of the search takes (no task is ever interrupted).
        if (lane == 0) {
            if (nfr < ECT_NFH) *(volatile uint32_t *) (ctl + ECT_C_NF + me) = (uint32_t) (nfr + 1);
            *(volatile uint32_t *) (ctl + ECT_C_FLOOR + me) = (uint32_t) (nfr + 1 > sh.reach? nfr + 1 - sh.reach : 0);
        }
        fa_top += bytes;
        ++nfr;
        return true;
    };
    auto restore_frame = [&](const uint8_t *arena, uint32_t off, int32_t l0) -> int32_t {
        const EctFrame *f = (const EctFrame *) (arena + off);
        const uint32_t fw = ((const uint32_t *) f)[lane & 7];                  // (one load for the header's words, a lane each)
        score = (int32_t) ecw_lane(fw, 0), t_end = (int32_t) ecw_lane(fw, 1), q_end = (int32_t) ecw_lane(fw, 2), n = (int32_t) ecw_lane(fw, 3), s_lo = (int32_t) ecw_lane(fw, 4);
        c_len = l0;
        const int32_t *sv = (const int32_t *) (f + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) { const int32_t s = r * 64 + lane; k[r] = s >= s_lo && s < s_lo + n? sv[s - s_lo] : ECH_NEG; }
        return (int32_t) ecw_lane(fw, 7);
    };
    auto frame_pe = [&](uint32_t off) -> int32_t {
        const uint32_t fw = ((const uint32_t *) (fa + off))[lane & 7];
        return (int32_t) ecw_lane(fw, 7);
    };
    auto frame_res = [&](const uint8_t *arena, uint32_t off, uint32_t idx) -> EctRes * {
        const EctFrame *f = (const EctFrame *) (arena + off);
        const int32_t nn = (int32_t) ect_gld((const uint32_t *) &f->n);
        return (EctRes *) ((uint8_t *) (f + 1) + 4 * frame_k_words(nn)) + idx;
    };
    // what this wave's task may still count: its own budget, and what the task it hangs below has left NOW (that task's count only grows, and whatever it has
    // counted when it reaches this arc it will have counted at least what it has counted by now) -- published for the tasks that hang below this one
    auto remaining = [&]() -> int32_t {
        int32_t lim = B;
        if (!root) { const int32_t up = (int32_t) ect_vld(ctl + ECT_C_REM + task_owner * (ECT_MAXCTX + 1) + task_owner_d); lim = up < lim? up : lim; }
        const int32_t rem = lim - c;
        if (lane == 0) { *(volatile uint32_t *) (ctl + ECT_C_REM + me * (ECT_MAXCTX + 1) + nctx) = (uint32_t) rem; *(volatile uint32_t *) (ctl + ECT_C_CNT + me * (ECT_MAXCTX + 1) + nctx) = (uint32_t) c; }
        return rem;
    };
    // log space of this wave: an event of `words` words, or one dead-end entry
    auto log_room = [&](uint32_t bytes) -> bool { return evt_pos + bytes + 4u * (n_dead + 1u) + 64u <= dead_top; };

    // an in-band arrival at the sink applied to the root's optimum (syncerr.c:209-252).  Inline: the path is c_path / c_arc [0, cn), the consensus cs; from a log: arcs[0, cn - 1)
    auto apply_event = [&](int32_t sc, bool advanced, int32_t cn, int32_t qe, int32_t na, const uint32_t *log_arcs, bool more_to_come) {
        status = EC_SUCCESS;
        if (sc <= edist) {
            if (advanced) s_edist = edist;
            edist = sc;
            if (ECW_RARE(edist == s_edist)) {
                bool diff = qe != o_len;
                if (!diff) {
                    if (!os_valid) { ect_rebuild(lv, rd, o_arc + 1, o_na, o_len, os); os_valid = true; }
                    const uint32_t *cand = cs;
                    if (log_arcs) { ect_rebuild(lv, rd, log_arcs, na, qe, es); cand = es; }
                    ecw_sync();
                    bool d = false;
                    const int32_t nw = (qe + 15) >> 4;
                    for (int32_t wi = lane; wi < nw; wi += 64) {
                        uint32_t x = cand[wi] ^ os[wi];
                        if (wi == nw - 1 && (qe & 15)) x &= (1u << ((qe & 15) << 1)) - 1u;
                        d |= x != 0;
                    }
                    diff = __ballot(d) != 0;
                }
                if (diff) status = EC_AMBISEQ;
                if (status == EC_SUCCESS) {
                    bool pd = cn != np;
                    if (!pd) {
                        bool d = false;
                        for (int32_t i = lane; i < cn; i += 64) {
                            const uint64_t v = log_arcs? (i == 0? wk.beg_utg : (uint64_t) lv.arc[log_arcs[i - 1]].w) : c_path[i];
                            d |= v != o_path[i];
                        }
                        pd = __ballot(d) != 0;
                    }
                    if (pd) status = EC_AMBISNQ;
                }
                ecw_sync();
            }
            o_len = qe;
            for (int32_t i = lane; i <= na; i += 64) {
                if (log_arcs) { if (i < cn) o_path[i] = i == 0? wk.beg_utg : (uint64_t) lv.arc[log_arcs[i - 1]].w; o_arc[i] = i == 0? 0u : log_arcs[i - 1]; }
                else { if (i < cn) o_path[i] = c_path[i]; o_arc[i] = c_arc[i]; }
            }
            np = cn, o_na = na;
            os_valid = false;
            // the optimum consensus is only ever compared with a LATER path's (a tie): when the search ends here nobody reads it
            if (!log_arcs && more_to_come) {
                for (int32_t wi = lane; wi < ((o_len + 15) >> 4); wi += 64) os[wi] = cs[wi];
                os_valid = true;
            }
            ecw_sync();
        } else if (sc < s_edist) {
            s_edist = sc;
        }
    };

    // the log a thief left for arc idx of the level, taken into this task: dead ends counted (and, in a sub-task, logged on), events applied (root) or logged on
    auto merge_res = [&](int fi, uint32_t idx) -> bool {
        const EctRes *rs = frame_res(fa, H_LD(fi, ECT_H_OFF), idx);
        const uint32_t rw = ((const uint32_t *) rs)[lane & 7];
        const uint32_t r_top = ecw_lane(rw, 0), r_nd = ecw_lane(rw, 1), r_eo = ecw_lane(rw, 2), r_ew = ecw_lane(rw, 3), r_wave = ecw_lane(rw, 4);
        const uint8_t *src_la = sh.slab + (uint64_t) r_wave * sh.wave_bytes + sh.cap_fa;
        const int32_t lvl_pe = frame_pe(H_LD(fi, ECT_H_OFF)), c_join = c;
        if (!root && !log_room(4u * r_nd + 4u * r_ew)) return false;
        uint32_t kept = 0;
        for (uint32_t i0 = 0; i0 < r_nd; i0 += 64) {
            const uint32_t i = i0 + (uint32_t) lane;
            int32_t pe = 0;
            bool keep = false;
            if (i < r_nd) {
                pe = *(const int32_t *) (src_la + r_top - 4u * (i + 1u));
                keep = pe < 0 || c_join + pe < B;
            }
            const uint64_t km = __ballot(keep);
            if (!root && keep) {
                const uint32_t at = n_dead + kept + (uint32_t) __builtin_popcountll(km & ((1ULL << lane) - 1ULL));
                *(int32_t *) (la + dead_top - 4u * (at + 1u)) = pe < 0? lvl_pe : c_join + pe;
            }
            kept += (uint32_t) __builtin_popcountll(km);
        }
        if (!root) n_dead += kept;
        else if (lane == 0) { atomicAdd(ctl + ECT_C_KEPT, kept); atomicAdd(ctl + ECT_C_DROPPED, r_nd - kept); }
        c += (int32_t) kept;
        const uint32_t *ev = (const uint32_t *) (src_la + r_eo);
        for (uint32_t at = 0; at < r_ew; ) {
            const uint32_t words = ect_gld(ev + at);
            const int32_t pe = (int32_t) ect_gld(ev + at + 1);
            if (pe < 0 || c_join + pe < B) {
                if (root) {
                    const int32_t cn = (int32_t) ect_gld(ev + at + 4);
                    apply_event((int32_t) ect_gld(ev + at + 2), (ect_gld(ev + at + 3) & 1u) != 0, cn, (int32_t) ect_gld(ev + at + 5), (int32_t) ect_gld(ev + at + 6), ev + at + 7, true);
                    if (lane == 0) atomicAdd(ctl + ECT_C_P0, 1u);
                } else {
                    uint32_t *dst = (uint32_t *) (la + evt_pos);
                    for (uint32_t i = (uint32_t) lane; i < words; i += 64) dst[i] = i == 1? (uint32_t) (pe < 0? lvl_pe : c_join + pe) : ev[at + i];
                    evt_pos += 4u * words;
                }
            }
            at += words;
        }
        ecw_sync();
        return true;
    };

    // the search from where this wave's state stands, until its stack is empty (dfs_search, syncerr.c:144-286, iteratively)
    // ... returns 1 when the next arc in order is with a thief that has not finished (the wave looks for something to do below that arc meanwhile, see the loop at the end)
    auto run = [&]() -> int {
        EcwArcRegs pre;
        pre.a = make_uint4(0, 0, 0, 0), pre.b = make_uint2(0, 0);
        uint32_t pre_idx = 0xFFFFFFFFu;
        while (nfr > nfr_base || vpend) {
            uint32_t a;
            int32_t depth, lvl_pe;
            if (ECW_LIKELY(vpend)) {                   // carry on where the search stands: nothing to restore
                vpend = false;
                a = v_arc, depth = v_depth, lvl_pe = v_pe;
            } else {
                const int fi = nfr - 1;
                const bool shared = fi < ECT_NFH;      // (a level beyond the LDS table is this wave's alone)
                uint32_t *h = myfh + (shared? fi : 0) * ECT_HW;
                const uint32_t arc0 = H_LD(fi, ECT_H_ARC0), arc_end = H_LD(fi, ECT_H_END);
                uint32_t mg = H_LD(fi, ECT_H_MERGED);
                if (mg >> 31) { mg = (mg & 0x7FFFFFFFu) + 1u; H_ST(fi, ECT_H_MERGED, mg); }        // back from the arc this wave ran itself
                uint32_t take = 0xFFFFFFFFu;
                if (ECW_RARE(aborted)) {
                    // unwinding: nobody takes another arc of this level; the thieves that are inside it finish (their logs are dropped with this task's)
                    if (shared) {
                        uint32_t old = 0;
                        if (lane == 0) { old = atomicMax(h + ECT_H_CUR, arc_end); atomicOr(h + ECT_H_GO, 0xFFFFFFFFu); }      // (and nobody holds back)
                        old = ecw_uniu(old);
                        const uint32_t claimed = (old < arc_end? old : arc_end) - arc0;
                        for (uint32_t idx = mg; idx < claimed; ++idx)
                            while (!((ect_vld(h + ECT_H_DONE) >> idx) & 1u)) { __builtin_amdgcn_s_sleep(4); if (ECW_RARE((uint32_t) __builtin_amdgcn_s_memrealtime() - block_t0 > 3000000000u)) __builtin_trap(); }
                        ect_acquire();
                    }
                } else if (!shared) {
                    if (arc0 + mg < arc_end) take = arc0 + mg, H_ST(fi, ECT_H_CUR, take + 1u);
                } else {
                    for (;;) {
                        const uint32_t cur = ect_vld(h + ECT_H_CUR);
                        const uint32_t claimed = (cur < arc_end? cur : arc_end) - arc0;
                        if (mg < claimed) {             // a thief has (or had) the next arc in order
                            if (!((ect_vld(h + ECT_H_DONE) >> mg) & 1u)) { if (lane == 0) atomicOr(h + ECT_H_GO, 1u << mg); return 1; }
                            ect_acquire();
                            if ((ect_vld(h + ECT_H_ABORT) >> mg) & 1u) { take = arc0 + mg; break; }           // given back: this wave runs it
                            {   // taken on speculation, and the search has more budget left than the sub-task allowed itself: its log does not say what lies beyond
                                const EctRes *rs = frame_res(fa, H_LD(fi, ECT_H_OFF), mg);
                                const int32_t lim = (int32_t) ecw_lane(((const uint32_t *) rs)[lane & 7], 5);
                                if (lim != INT32_MAX && remaining() > lim) { if (lane == 0) atomicAdd(ctl + ECT_C_P1, 1u); take = arc0 + mg; break; }
                            }
                            if (!merge_res(fi, mg)) { aborted = true; break; }
                            if (root && c >= EC_MAX_DFS_PATH && lane == 0) *(volatile uint32_t *) (ctl + ECT_C_CAP) = 1u;
                            ++mg;
                            H_ST(fi, ECT_H_MERGED, mg);
                            continue;
                        }
                        if (arc0 + mg >= arc_end) break;                       // level exhausted
                        take = claim(h, arc_end, true, arc0 + mg);
                        if (take != 0xFFFFFFFFu) break;
                    }
                    if (aborted) continue;
                }
                if (take == 0xFFFFFFFFu) {             // return to the nearest level with siblings left
                    --nfr;
                    fa_top = (int32_t) H_LD(fi, ECT_H_OFF);
                    {   // (the line this level had is free again)
                        const EctFrame *f = (const EctFrame *) (fa + fa_top);
                        const uint32_t fw = ((const uint32_t *) f)[lane & 7];
                        t_hi = (uint64_t) ecw_lane(fw, 6) << 32 | ecw_lane(fw, 5);
                    }
                    if (lane == 0) {
                        if (shared) *(volatile uint32_t *) (ctl + ECT_C_NF + me) = (uint32_t) nfr;
                        *(volatile uint32_t *) (ctl + ECT_C_FLOOR + me) = (uint32_t) (nfr > sh.reach? nfr - sh.reach : 0);
                    }
                    continue;
                }
                // restore the state this level was entered with (syncerr.c:277-284)
                lvl_pe = restore_frame(fa, H_LD(fi, ECT_H_OFF), (int32_t) H_LD(fi, ECT_H_L0));
                H_ST(fi, ECT_H_MERGED, mg | 0x80000000u);
                a = take, depth = (int32_t) (H_LD(fi, ECT_H_DEPTH) & 0xFFFFFFu);
            }
            if (pace) {
                // Taken on speculation: the owner is still below the level's first arc, and whether the search ever tries this one nobody knows -- below MAX_DFS_PATH dead ends most
                // subtrees of a tandem array are never entered.  What is known: the level's arcs lead to subtrees of much the same size.  So this sub-task counts no further than
                // what the owner's task has counted below the level so far (plus a few): it keeps pace, loses at most as much as the first arc's subtree cost when that one takes
                // the search to its end, and is nearly done when the owner arrives -- which is when it is told to go ahead.  It never refuses a level on this account (it WAITS), so its
                // log is that of its full budget.
                const uint32_t *oh = hdr(res_ow, res_slot);
                const uint32_t gobit = 1u << (res_arc - ect_vld(oh + ECT_H_ARC0));
                for (;;) {
                    if (ect_vld(oh + ECT_H_GO) & gobit) { pace = false; break; }
                    if (ect_vld(ctl + ECT_C_CAP)) break;
                    const int32_t seen = (int32_t) ect_vld(ctl + ECT_C_CNT + task_owner * (ECT_MAXCTX + 1) + task_owner_d) - pace_pe;
                    if (c < seen + 4) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (ECW_RARE((uint32_t) __builtin_amdgcn_s_memrealtime() - block_t0 > 3000000000u)) __builtin_trap();
                }
            }
            ++tried;
            if (ECW_RARE(pre_idx != a)) pre = ecw_arc_load(lv.arc, a);
            const uint64_t w = ecw_uniu(pre.a.x);
            const int32_t ls = (int32_t) ecw_uniu(pre.a.y), ext = K - ls;
            const uint32_t w_hs16 = ecw_uniu(pre.a.z), w_mpos = ecw_uniu(pre.a.w), w_lp = ecw_uniu(pre.b.x), w_ln = ecw_uniu(pre.b.y);
            const int32_t t_end0 = t_end;
            if (ECW_RARE(depth + 2 > sh.cap_path || c_len + ext > sh.cap_c)) { aborted = true; continue; }
            int32_t cn = depth + 2;                    // entries in c_path
            // the arc most likely to be tried next: the first one out of w (in flight during the gather and the alignment)
            pre_idx = 0xFFFFFFFFu;
            if (ECW_LIKELY(w_ln)) pre = ecw_arc_load(lv.arc, w_lp), pre_idx = w_lp;
            ect_append(cs, c_len, ext, ls, K, rd.hoco_s + ((uint64_t) w_hs16 << 4), w_mpos, (w & 1ULL) != 0);
            c_len += ext;
            if (lane == 0) c_path[depth + 1] = w, c_arc[depth + 1] = a;
            ecw_sync();
            const bool may_enter = remaining() > 0 && !ect_vld(ctl + ECT_C_CAP);      // the callee would return at once otherwise (syncerr.c:146-148)
            // a vertex on an unbranched stretch that cannot be the end of the path needs no alignment of its own (ec_wave.hpp) -- while no optimum exists,
            // which only the root can know
            if (root && edist == INT32_MAX && wk.end_utg != EC_NONE && wk.end_utg != w && w_ln == 1 && may_enter && c_len - K <= tl + bw && c_len >= bw + 3) {
                vpend = true, v_arc = w_lp, v_depth = depth + 1, v_pe = c;
                continue;
            }
            // wf_ed_core (levdist.c:265-310)
            for (;;) {
                ++wf_steps, wf_diag += (uint64_t) n;
                if (ect_step<R>(ts, cs, tl, c_len, bw, OFF, k, s_lo, n, t_end, q_end)) break;
                ++score;
                if (ECW_RARE(score > bw)) break;
            }
            t_end += 1, q_end += 1;
            if (root && lane == 0) *(volatile uint32_t *) (ctl + ECT_C_LIGHT) = wf_steps <= 6u * tried? 1u : 0u;        // (sub-tasks pay when an arc costs many steps)
            const int32_t ql = c_len;
            const int32_t sc = score + tl - t_end;     // syncerr.c:209
            const bool goes_on = score <= bw && ql - K <= tl + bw && ((wk.end_utg != EC_NONE && wk.end_utg != w) || t_end < tl);
            if (sc <= bw && (wk.end_utg == EC_NONE || wk.end_utg == w)) {
                if (wk.end_utg == EC_NONE && q_end < ql) --cn;
                if (root) {
                    apply_event(sc, t_end > t_end0, cn, q_end, depth + 1, nullptr, nfr > 0 || (goes_on && may_enter && w_ln > 0));
                } else {
                    const uint32_t words = 7u + (uint32_t) (depth + 1);
                    if (ECW_RARE(!log_room(4u * words))) { aborted = true; if (lane == 0) *(volatile uint32_t *) (ctl + ECT_C_NOSTEAL) = 1u; continue; }
                    uint32_t *dst = (uint32_t *) (la + evt_pos);
                    if (lane == 0) dst[0] = words, dst[1] = (uint32_t) lvl_pe, dst[2] = (uint32_t) sc, dst[3] = t_end > t_end0? 1u : 0u, dst[4] = (uint32_t) cn, dst[5] = (uint32_t) q_end, dst[6] = (uint32_t) (depth + 1);
                    for (int32_t i = lane; i < depth + 1; i += 64) dst[7 + i] = c_arc[i + 1];
                    evt_pos += 4u * words;
                    ecw_sync();
                }
            }
            if (goes_on && !may_enter && !root && B - c <= 0) hit_own = true;
            if (goes_on) {
                if (may_enter) {
                    if (ECW_LIKELY(w_ln == 1)) vpend = true, v_arc = w_lp, v_depth = depth + 1, v_pe = c;
                    else if (w_ln > 1 && !push_frame(w_lp, w_ln, depth + 1, c)) { aborted = true; continue; }       // (no arcs: the callee's loop does not run)
                }
            } else {
                if (!root) {
                    if (ECW_RARE(!log_room(4u))) { aborted = true; if (lane == 0) *(volatile uint32_t *) (ctl + ECT_C_NOSTEAL) = 1u; continue; }
                    if (lane == 0) *(int32_t *) (la + dead_top - 4u * (n_dead + 1u)) = lvl_pe;
                    ++n_dead;
                }
                ++c;
                if (root && c >= EC_MAX_DFS_PATH && lane == 0) *(volatile uint32_t *) (ctl + ECT_C_CAP) = 1u;
            }
        }
        return 0;
    };
    auto flush_stats = [&]() {
        if (lane == 0) { atomicAdd(ctl + ECT_C_TRIED, tried); atomicAdd(ctl + ECT_C_STEPS, wf_steps); atomicAdd(ctl + ECT_C_DIAG, (uint32_t) (wf_diag >> 6)); }
        tried = 0, wf_steps = 0, wf_diag = 0;
    };

    // ---- wave 0 runs the search itself (the root); the others take arcs of published levels until it is done.  A wave whose task has to wait for a thief looks for an arc
    //      BELOW the one it waits for (its share of the line, [w_lo, w_hi)), sets its task aside and runs that first: what it takes on is finished, merged and forgotten before
    //      the arc it waits for is, so levels, frames and log space stack up and come down in order, and its consensus and path up to the level it waits at stay what they are
    //      (everything below that level spells the same prefix) ----
    bool have_task = false, waiting = false;
    if (me == 0) {
        root = true, B = EC_MAX_DFS_PATH, c = 0, t_lo = 0, t_hi = 1ULL << 63;
#pragma unroll
        for (int r = 0; r < R; ++r) k[r] = r * 64 + lane == OFF? -1 : ECH_NEG;
        if (lane == 0) c_path[0] = wk.beg_utg, c_arc[0] = 0;
        evt_start = evt_pos = la_lo, dead_top = la_hi, n_dead = 0;
        (void) remaining();
        if (ECW_LIKELY(wk.ln == 1)) vpend = true, v_arc = wk.lp, v_depth = 0, v_pe = 0;
        else if (wk.ln > 1 && !push_frame(wk.lp, wk.ln, 0, 0)) aborted = true;
        have_task = true;
    }
    for (;;) {
        if (have_task && !waiting) {
            const uint64_t task_t0 = __builtin_amdgcn_s_memrealtime();
            const int must_wait = run();
            if (!root && lane == 0) atomicAdd(ctl + ECT_C_TASK_TICKS, (uint32_t) (__builtin_amdgcn_s_memrealtime() - task_t0));
            if (must_wait) {
                waiting = true;
                la_lo = evt_pos, la_hi = dead_top - 4u * n_dead;          // (log space for what this wave takes on meanwhile: above this task's, given back when it goes on)
                continue;
            }
            flush_stats();
            if (root) {
                if (lane == 0) ctl[ECT_C_OK] = aborted? 0u : 1u;
                ect_release();
                if (lane == 0) *(volatile uint32_t *) (ctl + ECT_C_DONE) = 1u;
                break;
            }
            {   // a sub-task is finished: its log to the level it was taken from
                uint32_t *h = hdr(res_ow, res_slot);
                const uint8_t *ofa = sh.slab + (uint64_t) res_ow * sh.wave_bytes;
                const uint32_t bit = res_arc - ect_vld(h + ECT_H_ARC0);
                EctRes *rs = frame_res(ofa, ect_vld(h + ECT_H_OFF), bit);
                if (!aborted) {
                    if (lane == 0) rs->dead_top = dead_top, rs->n_dead = n_dead, rs->evt_off = evt_start, rs->evt_words = (evt_pos - evt_start) >> 2, rs->wave = (uint32_t) me, rs->limit = hit_own? B : INT32_MAX;
                    la_lo = evt_pos, la_hi = dead_top - 4u * n_dead;
                    if (lane == 0) atomicAdd(ctl + ECT_C_SUBDEAD, n_dead);
                } else {
                    la_lo = evt_start, la_hi = dead_top;
                }
                ect_release();
                if (lane == 0) {
                    if (aborted) { atomicOr(h + ECT_H_ABORT, 1u << bit); atomicAdd(ctl + ECT_C_ABORTS, 1u); }
                    atomicAdd(ctl + ECT_C_STEALS, 1u);
                }
                ect_release();
                if (lane == 0) atomicOr(h + ECT_H_DONE, 1u << bit);
            }
            if (nctx > 0) {                            // back to the task that was set aside (it looks at its level again)
                --nctx;
                const uint32_t *cx = ctl + ECT_C_CTX + ((uint32_t) me * ECT_MAXCTX + (uint32_t) nctx) * ECT_CTXW;
                root = ect_vld(cx + 0) != 0, c = (int32_t) ect_vld(cx + 1), B = (int32_t) ect_vld(cx + 2), task_owner = (int) ect_vld(cx + 3), task_owner_d = (int) ect_vld(cx + 4);
                t_lo = (uint64_t) ect_vld(cx + 6) << 32 | ect_vld(cx + 5), t_hi = (uint64_t) ect_vld(cx + 8) << 32 | ect_vld(cx + 7);
                evt_start = ect_vld(cx + 9), evt_pos = ect_vld(cx + 10), dead_top = ect_vld(cx + 11), n_dead = ect_vld(cx + 12);
                nfr_base = (int) ect_vld(cx + 13), aborted = ect_vld(cx + 14) != 0, res_ow = (int) ect_vld(cx + 15), res_slot = (int) ect_vld(cx + 16), res_arc = ect_vld(cx + 17), hit_own = (ect_vld(cx + 18) & 1u) != 0, pace = (ect_vld(cx + 18) & 2u) != 0, pace_pe = (int32_t) ect_vld(cx + 19);
                have_task = true, waiting = true;      // (it waits until the arc it waited for is done; what this wave has just finished lay below that arc)
            } else {
                have_task = false, waiting = false;
            }
            continue;
        }
        // nothing to run: no task at all, or a task that waits at its deepest level for the arc `mg` of it
        uint64_t w_lo = 0, w_hi = ~0ULL;
        if (!have_task) {
            if (ect_vld(ctl + ECT_C_DONE)) break;
        } else {
            const uint32_t *h = myfh + (nfr - 1) * ECT_HW;             // (a level a thief has an arc of is one of the first ECT_NFH)
            const uint32_t mg = ect_vld(h + ECT_H_MERGED) & 0x7FFFFFFFu;
            if ((ect_vld(h + ECT_H_DONE) >> mg) & 1u) { waiting = false; continue; }
            const uint64_t kw = (uint64_t) ect_vld(h + ECT_H_KW1) << 32 | ect_vld(h + ECT_H_KW0);
            w_lo = ((uint64_t) ect_vld(h + ECT_H_KEY1) << 32 | ect_vld(h + ECT_H_KEY0)) + (uint64_t) mg * kw, w_hi = w_lo + kw;
        }
        uint32_t got = 0xFFFFFFFFu;
        int ow = 0, slot = 0;
        if (!ect_vld(ctl + ECT_C_NOSTEAL) && !ect_vld(ctl + ECT_C_LIGHT) && (!have_task || (nctx < ECT_MAXCTX && w_hi > w_lo))) {
            const int32_t best = ect_find_arc(ctl, fh, me, w_lo, w_hi);
            if (best >= 0) {
                ow = best / ECT_NFH, slot = best % ECT_NFH;
                uint32_t *h = hdr(ow, slot);
                got = claim(h, ect_vld(h + ECT_H_SEND), false, 0);
                // (the cursor may have moved on to another arc of that level between the look and the claim: still below the arc this wave waits for -- a level's arcs all
                //  lie in the level's share of the line, and that lies inside the awaited arc's share or outside it)
            }
        }
        if (got == 0xFFFFFFFFu) {
            const uint32_t w0 = (uint32_t) __builtin_amdgcn_s_memrealtime();
            if (ECW_RARE(w0 - block_t0 > 3000000000u)) __builtin_trap();      // thirty seconds on one block: the waves wait for each other (a bug) -- fail, do not hang
            if (have_task) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(16);
            if (have_task && lane == 0) atomicAdd(ctl + (root? ECT_C_WAIT : ECT_C_WAIT_SUB), (uint32_t) __builtin_amdgcn_s_memrealtime() - w0);
            continue;
        }
        // a sub-task: arc `got` of that level with everything below it
        ect_acquire();
        if (have_task) {                               // the waiting task is set aside
            uint32_t *cx = ctl + ECT_C_CTX + ((uint32_t) me * ECT_MAXCTX + (uint32_t) nctx) * ECT_CTXW;
            if (lane == 0) {
                cx[0] = root? 1u : 0u, cx[1] = (uint32_t) c, cx[2] = (uint32_t) B, cx[3] = (uint32_t) task_owner, cx[4] = (uint32_t) task_owner_d;
                cx[5] = (uint32_t) t_lo, cx[6] = (uint32_t) (t_lo >> 32), cx[7] = (uint32_t) t_hi, cx[8] = (uint32_t) (t_hi >> 32);
                cx[9] = evt_start, cx[10] = evt_pos, cx[11] = dead_top, cx[12] = n_dead;
                cx[13] = (uint32_t) nfr_base, cx[14] = aborted? 1u : 0u, cx[15] = (uint32_t) res_ow, cx[16] = (uint32_t) res_slot, cx[17] = res_arc, cx[18] = (hit_own? 1u : 0u) | (pace? 2u : 0u), cx[19] = (uint32_t) pace_pe;
            }
            ++nctx;
            ecw_sync();
        }
        {
            uint32_t *h = hdr(ow, slot);
            const uint8_t *ofa = sh.slab + (uint64_t) ow * sh.wave_bytes;
            root = false, aborted = false, hit_own = false, c = 0, B = (int32_t) ect_vld(h + ECT_H_BUDGET), task_owner = ow, task_owner_d = (int) (ect_vld(h + ECT_H_DEPTH) >> 24);
            // On speculation: the owner is still below the level's first arc and nobody knows whether the search ever tries this one (below MAX_DFS_PATH dead ends most subtrees of
            // a tandem array are never entered).  The sub-task counts up to a small budget of its own -- what it logs is what the search does with that much or less left -- so
            // that a wave is never gone for long on something that may be for nothing; the owner runs the arc itself when it arrives with more.
            const bool speculative = (ect_vld(h + ECT_H_MERGED) & 0x7FFFFFFFu) == 0;
            if (speculative && B > sh.spec_budget) B = sh.spec_budget;
            res_ow = ow, res_slot = slot, res_arc = got;
            {
                const uint64_t kw = (uint64_t) ect_vld(h + ECT_H_KW1) << 32 | ect_vld(h + ECT_H_KW0);
                t_lo = ((uint64_t) ect_vld(h + ECT_H_KEY1) << 32 | ect_vld(h + ECT_H_KEY0)) + (uint64_t) (got - ect_vld(h + ECT_H_ARC0)) * kw, t_hi = t_lo + kw;
            }
            pace_pe = restore_frame(ofa, ect_vld(h + ECT_H_OFF), (int32_t) ect_vld(h + ECT_H_L0));
            pace = speculative;
            const int32_t depth = (int32_t) (ect_vld(h + ECT_H_DEPTH) & 0xFFFFFFu);
            {   // the owner's consensus and path up to the level
                const uint32_t *ocs = sh.cs_all + (uint32_t) ow * (uint32_t) sh.cw;
                for (int32_t wi = lane; wi <= (c_len >> 4); wi += 64) cs[wi] = *(const volatile uint32_t *) (ocs + wi);
                const uint64_t *ocp = (const uint64_t *) (ofa + sh.cap_fa + sh.cap_la);
                const uint32_t *oca = (const uint32_t *) (ocp + sh.cap_path);
                for (int32_t i = lane; i <= depth; i += 64) c_path[i] = ocp[i], c_arc[i] = oca[i];
            }
            evt_start = evt_pos = la_lo, dead_top = la_hi, n_dead = 0;
            nfr_base = nfr;
            vpend = true, v_arc = got, v_depth = depth, v_pe = -1;
            (void) remaining();
            have_task = true, waiting = false;
            ecw_sync();
        }
    }
    __syncthreads();
    const bool ok = ect_vld(ctl + ECT_C_OK) != 0;
    if (me == 0) status_out = (uint32_t) status, np_out = (uint32_t) np, n_path_out = (uint32_t) c, o_path_out = o_path;
    return ok;
}

// One workgroup per block, blocks taken one at a time from the list.  EcwArgs as for ec_heavy_kernel; a.cap_f = bytes of a wave's frame arena, a.os_words = bytes of
// its log arena.
template <int R>
__global__ __launch_bounds__(64 * ECT_W) void ec_tree_kernel(EcwArgs a)
{
    extern __shared__ uint32_t ect_lds[];
    const int t = (int) threadIdx.x;
    EctShared sh;
    sh.cap_t = a.cap_t, sh.cap_c = a.cap_c, sh.cap_path = a.cap_path, sh.cap_fa = a.cap_f, sh.cap_la = (int32_t) a.os_words, sh.cw = (int32_t) ecw_words(a.cap_c);
    sh.ctl = ect_lds, sh.fh = sh.ctl + ECT_CTL_WORDS, sh.ts = sh.fh + ECT_W * ECT_NFH * ECT_HW, sh.cs_all = sh.ts + ecw_words(a.cap_t);
    sh.slab = a.slabs + (uint64_t) blockIdx.x * a.slab_bytes;
    sh.wave_bytes = ect_wave_bytes(a.cap_path, a.cap_f, (int32_t) a.os_words);
    sh.reach = a.cap_w > 0? a.cap_w : 1 << 20;
    sh.spec_budget = a.batch > 1? a.batch : 0x7FFFFFFF;
    const uint64_t total = a.todo? a.n_todo : a.n_work;
    uint64_t pool_at = 0, pool_end = 0;
    for (int i = t; i < ECT_C_CTX; i += 64 * ECT_W) sh.ctl[i] = 0;
    __syncthreads();
    for (;;) {
        if (t == 0) {
            const unsigned long long t0 = atomicAdd(a.next, 1ULL);
            sh.ctl[ECT_C_Q0] = (uint32_t) t0, sh.ctl[ECT_C_Q1] = (uint32_t) (t0 >> 32);
        }
        __syncthreads();
        const uint64_t t0 = (uint64_t) ect_vld(sh.ctl + ECT_C_Q1) << 32 | ect_vld(sh.ctl + ECT_C_Q0);
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
        o.status = EC_FAILURE, o.np = 0, o.path_off = 0, o.flags = 0, o.short_block = 0, o.tried = 0, o.n_path = 0, o.wf_steps = 0, o.wf_diag = 0, o.tier = ECT_TIER;
        const uint64_t tick0 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();                               // (every wave has the queue's word before the block's prologue clears the control words)
        if (ECW_RARE(wk.l < EC_MIN_ERR_SEQ_LEN)) {
            o.short_block = 1;                         // syncerr.c:502-504
        } else {
            uint32_t st = 0, np = 0, n_path = 0;
            const uint64_t *o_path = nullptr;
            if (ECW_RARE(!(ect_solve_block<R>(a.lv, a.rd, wk, sh, a.max_edist, st, np, n_path, o_path)))) {
                o.flags = 1;
                if (t == 0) a.todo_out[atomicAdd(a.todo_cnt, 1ULL)] = (uint32_t) wi;
            } else if (t < 64) {
                o.status = st, o.np = np, o.n_path = n_path;
                o.tried = ect_vld(sh.ctl + ECT_C_TRIED), o.wf_steps = ect_vld(sh.ctl + ECT_C_STEPS), o.wf_diag = ect_vld(sh.ctl + ECT_C_DIAG);
                { const uint32_t st_n = ect_vld(sh.ctl + ECT_C_STEALS); o.tier = ECT_TIER | (st_n < 0xFFFFFFu? st_n : 0xFFFFFFu) << 8; }          // (diagnostic: sub-tasks run by thieves)
                if (st == EC_SUCCESS && np) {
                    if (ECW_RARE(pool_at + np > pool_end)) {
                        const unsigned long long want = np > ECW_POOL_CHUNK? np : ECW_POOL_CHUNK;
                        unsigned long long off = 0;
                        if (t == 0) off = atomicAdd(a.pool_cursor, want);
                        pool_at = ecw_uni64(off), pool_end = pool_at + want;
                    }
                    o.path_off = pool_at;
                    if (pool_at + np <= a.pool_cap) for (uint32_t j = (uint32_t) t; j < np; j += 64) a.path_pool[pool_at + j] = o_path[j];
                    pool_at += np;
                }
            }
        }
#ifdef ECT_PROF                                          // (development builds: the diagonals' column of tools/ec_effort.py shows the ticks the root spent waiting, in units of 64)
        if (t == 0 && !o.flags && !o.short_block) o.wf_diag = sh.ctl[ECT_C_WAIT];
#endif
        o.ticks = (uint32_t) (__builtin_amdgcn_s_memrealtime() - tick0);
        if (t == 0) a.out[wi] = o;
        __syncthreads();
    }
}

}  // namespace oatk
