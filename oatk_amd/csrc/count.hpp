// oatk_amd/csrc/count.hpp -- syncmer count / ID assignment on the device.
//
// Replaces `collect_syncmer_from_reads` + `process_kmer_cluster` (syncmer.c:1397-1451, :1270-1393).
// The reference sorts 128-bit (hash, sid<<32|idx<<1|rev) records, walks equal-hash groups, splits true
// 64-bit collisions by comparing the k-mer sequences, and hands out dense IDs in that order.  Here:
//   1. place_records: scan records are unordered but carry (sid, ordinal), and the per-read counts are
//      known, so each record has an exact slot scm_off[read] + ordinal.  Scattering them there yields the
//      per-read arrays (m_pos, s_mer, hash) AND a sequence already ordered by the low 64 key bits.
//   2. one stable 64-bit radix sort of (hash -> slot) finishes the 128-bit order.
//   3. mark_heads / verify_group: equal-hash neighbours are compared base-for-base with their group head
//      (one wave per record, one 32-base word per lane) -- the collision check the reference performs.
//   4. split_collisions: groups that really hold different k-mers (never seen in practice; forced in tests
//      by masking hash bits) are partitioned in first-seen order by one lane per group.
//   5. IDs = prefix sum of new-cluster flags; coverage, CSR occurrence lists, s-mer consistency
//      (syncmer.c:1365-1376) and the per-read k_mer = id << 1 rewrite (:1378) are one pass each.
#pragma once
#include "common.hpp"

namespace oatk {

// shard regions -> dense record arrays (shard i's records land at prefix[i] ...)
struct CompactArgs {
    const uint64_t *raw_lo, *raw_smer;
    const uint32_t *raw_mpos;
    const uint32_t *shard_cnt;
    const uint64_t *shard_prefix;
    uint32_t region_cap;
    uint64_t *rec_lo, *rec_smer;
    uint32_t *rec_mpos;
};

__global__ __launch_bounds__(256) void compact_records_kernel(CompactArgs a)
{
    const uint32_t sh = blockIdx.x, n = a.shard_cnt[sh];
    const size_t src = (size_t) sh * a.region_cap;
    const uint64_t dst = a.shard_prefix[sh];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        a.rec_lo[dst + i] = a.raw_lo[src + i];
        a.rec_smer[dst + i] = a.raw_smer[src + i];
        a.rec_mpos[dst + i] = a.raw_mpos[src + i];
    }
}

struct PlaceArgs {
    const uint64_t *rec_hash, *rec_lo, *rec_smer;
    const uint32_t *rec_mpos;
    uint32_t n_rec;
    uint64_t sid0;
    const uint64_t *scm_off;      // exclusive prefix of n_scm over reads
    uint64_t hash_mask;           // debug knob (tests force collisions); ~0 in production
    uint64_t *pos_hash, *pos_lo, *pos_smer;
    uint32_t *pos_mpos;
    uint64_t *key_hash;           // masked copy used as the sort key
    uint32_t *iota;
};

__global__ void place_records_kernel(PlaceArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    uint64_t lo = a.rec_lo[i];
    uint64_t rd = (lo >> 32) - a.sid0;
    uint64_t p = a.scm_off[rd] + ((uint32_t) lo >> 1);
    a.pos_hash[p] = a.rec_hash[i];
    a.key_hash[p] = a.rec_hash[i] & a.hash_mask;
    a.pos_lo[p] = lo;
    a.pos_smer[p] = a.rec_smer[i];
    a.pos_mpos[p] = a.rec_mpos[i];
    a.iota[p] = (uint32_t) p;
}

// 32 oriented k-mer bases [32*wd, 32*wd+32) MSB-first, zero beyond K, straight from the hoco string in HBM
__device__ __forceinline__ uint64_t kmer_word_global(const uint32_t *hs32, uint32_t pos, uint32_t rev, int K, int wd)
{
    int32_t t = rev? (int32_t) pos + K - 32 - 32 * wd : (int32_t) pos + 32 * wd;
    int nb = K - 32 * wd;
    nb = nb > 32? 32 : nb;
    int32_t wi = t >> 4;
    uint32_t sh = ((uint32_t) t & 15u) * 2u;
    uint32_t w0, w1, w2;
    if (wi >= 0) {                // one 12-byte load (the lanes of a comparison sit 8 bytes apart: three dword loads look up every cache line three times)
        struct __attribute__((packed, aligned(4))) W3 { uint32_t a, b, c; };
        const W3 v = *(const W3 *) (hs32 + wi);
        w0 = __builtin_bswap32(v.a), w1 = __builtin_bswap32(v.b), w2 = __builtin_bswap32(v.c);
    } else {                      // (the first bases of a slab's first read, reverse strand: what lies in front is masked)
        w0 = 0u;
        w1 = wi + 1 >= 0? __builtin_bswap32(hs32[wi + 1]) : 0u;
        w2 = wi + 2 >= 0? __builtin_bswap32(hs32[wi + 2]) : 0u;
    }
    uint64_t hi = (uint64_t) w0 << 32 | w1;
    uint64_t V = sh? (hi << sh) | ((uint64_t) w2 >> (32u - sh)) : hi;
    if (rev) V = revcomp32(V);
    if (nb < 32) V &= ~0ULL << (64 - 2 * nb);
    return V;
}

struct GroupArgs {
    const uint64_t *sorted_key;   // masked hash, ascending
    const uint32_t *perm;         // sorted index -> slot
    uint32_t n_rec;
    const uint64_t *pos_lo;
    const uint32_t *pos_mpos;
    const uint8_t *hoco_s;
    const uint64_t *off;
    uint64_t sid0;
    int K;
    uint32_t *head;               // 1 where a new equal-hash group starts
    uint32_t *head_idx;           // sorted index of the group head (filled by a max-scan)
    uint32_t *newclus;            // 1 where a new syncmer (cluster) starts; starts as a copy of head
    uint32_t *flags;              // [0] some group holds different k-mers, [1] s-mer mismatch, [2] too many clusters
    uint64_t *loc;                // where the k-mer of sorted record i lives: (32-bit word index of its read's hoco string) << 32 | pos << 1 | rev
    // what else the sort permutation is followed for, fetched while it is followed the first time: the occurrence word (sid | index | strand)
    // and the s-mer of every sorted record -- finish_heads and check_smer then read them in order instead of gathering them again
    const uint64_t *pos_smer;
    uint64_t *occ_sorted, *smer_sorted;
    // ... and fetched as ONE 32-byte record per slot (pack_slots_kernel): occurrence word, s-mer, k-mer locator.  Three arrays indexed by the same
    // random slot cost three 64-byte fetches and the locator a fourth, dependent one (the read's offset); packed side by side in slot order -- a
    // streaming pass -- they cost one
    uint4 *slot_rec;
};

__global__ void pack_slots_kernel(GroupArgs a)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_rec) return;
    const uint64_t lo = a.pos_lo[p], sm = a.pos_smer[p];
    const uint64_t loc = ((a.off[(lo >> 32) - a.sid0] >> 4) << 32) | a.pos_mpos[p];      // (slots are in read order: the offsets stream too)
    a.slot_rec[2 * (size_t) p] = make_uint4((uint32_t) lo, (uint32_t) (lo >> 32), (uint32_t) sm, (uint32_t) (sm >> 32));
    a.slot_rec[2 * (size_t) p + 1] = make_uint4((uint32_t) loc, (uint32_t) (loc >> 32), 0u, 0u);
}

// The sort of the k-mer hashes looks at their top 40 bits only (five radix passes instead of eight: 42 M records at config 3, 0.36 ms per pass).
// Two DIFFERENT hashes share 40 bits about n_distinct^2 / 2^41 times per batch -- a handful of runs -- and then their records sit interleaved,
// in slot order.  These two kernels finish the order there: every position where the hash changes inside a run of equal top bits reports to
// the run's start, and the smallest such position of a run sorts it (stable insertion sort of keys and permutation; runs are a few coverages
// long).  A run longer than OATK_SORT_REPAIR_MAX sets flags[3] and the host sorts again on all 64 bits.
#define OATK_SORT_LOW_BITS 24
#define OATK_SORT_REPAIR_MAX 4096u
__device__ __forceinline__ bool sort_repair_boundary(const uint64_t *key, uint32_t n, uint32_t i, uint32_t &start, uint32_t *flags)
{
    if (i == 0 || i >= n) return false;
    const uint64_t k = key[i], kp = key[i - 1];
    if ((k >> OATK_SORT_LOW_BITS) != (kp >> OATK_SORT_LOW_BITS) || k == kp) return false;
    uint32_t s = i - 1;
    while (s > 0 && (key[s - 1] >> OATK_SORT_LOW_BITS) == (k >> OATK_SORT_LOW_BITS)) {
        --s;
        if (i - s > OATK_SORT_REPAIR_MAX) { flags[3] = 1u; return false; }
    }
    start = s;
    return true;
}
__global__ void sort_repair_find_kernel(const uint64_t *key, uint32_t n, uint32_t *owner, uint32_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s;
    if (sort_repair_boundary(key, n, i, s, flags)) atomicMin(&owner[s], i);       // owner[] starts as all ones
    // (nothing is taken on trust: rocPRIM's small-size paths were seen to leave the top bits of a bit-range sort out of order -- tools/ubench/sort_repair_test.hip,
    //  2000 - 100 000 keys -- and any inversion sends the count to the sort on all 64 bits)
    if (i > 0 && i < n && (key[i] >> OATK_SORT_LOW_BITS) < (key[i - 1] >> OATK_SORT_LOW_BITS)) flags[3] = 1u;
}
__global__ void sort_repair_kernel(uint64_t *key, uint32_t *perm, uint32_t n, uint32_t *owner, uint32_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s;
    if (!sort_repair_boundary(key, n, i, s, flags) || owner[s] != i) return;     // (only a run's owner writes, and nobody reads what it writes: the others leave at once)
    const uint64_t top = key[i] >> OATK_SORT_LOW_BITS;
    uint32_t e = i + 1;
    while (e < n && (key[e] >> OATK_SORT_LOW_BITS) == top) {
        ++e;
        if (e - s > OATK_SORT_REPAIR_MAX) { flags[3] = 1u; return; }
    }
    for (uint32_t a = i; a < e; ++a) {                    // [s, i) holds one hash: sorted already
        const uint64_t kk = key[a];
        const uint32_t pp = perm[a];
        uint32_t b = a;
        while (b > s && key[b - 1] > kk) { key[b] = key[b - 1], perm[b] = perm[b - 1]; --b; }
        key[b] = kk, perm[b] = pp;
    }
}

// The sort is a library call whose paths differ by size and version: what comes back is CHECKED to be the input, permuted and in order.  Two independent
// 64-bit sums over a mix of every (hash, slot) pair are taken of the sort's input and of its output (after the repair); equal sums = the same multiset of
// pairs (up to 2^-128), so perm is a permutation and every hash sits beside its own slot; the order is compared directly.
// flags[8..11] hold the input's sums, flags[12..15] the output's, flags[4] an inversion.  Grid-stride with a fixed grid and one pair of atomics per
// workgroup: a first version with a pair per WAVE of a one-thread-per-record kernel cost 16 ms at config 3 -- 1.3 M atomics on two addresses.
__device__ __forceinline__ void pair_sums(uint64_t key, uint32_t slot, uint64_t &s0, uint64_t &s1)
{
    uint64_t x = key ^ ((uint64_t) slot * 0x9E3779B97F4A7C15ULL);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 32;
    s0 = x;
    x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29; x += key;
    s1 = x;
}
// perm == nullptr: the pairs are (key[i], i) -- the sort's input; otherwise (key[i], perm[i]) and the keys are checked to ascend
__global__ __launch_bounds__(256) void pair_sum_kernel(const uint64_t *key, const uint32_t *perm, uint32_t n, uint32_t *flags, int at)
{
    __shared__ uint64_t part[2][4];
    uint64_t s0 = 0, s1 = 0;
    bool inv = false;
    // (four independent elements per turn: the loads of a turn are in flight together)
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint64_t k[4], kp[4];
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t i = i0 + u * stride;
            const bool in = i < n;
            k[u] = in? key[i] : 0, v[u] = in? (perm? perm[i] : (uint32_t) i) : 0, kp[u] = in && perm && i? key[i - 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * stride >= n) continue;
            uint64_t a, b;
            pair_sums(k[u], v[u], a, b);
            s0 += a, s1 += b;
            if (k[u] < kp[u]) inv = true;
        }
    }
    #pragma unroll
    for (int d = 32; d > 0; d >>= 1) s0 += __shfl_xor(s0, d), s1 += __shfl_xor(s1, d);
    if ((threadIdx.x & 63u) == 0) part[0][threadIdx.x >> 6] = s0, part[1][threadIdx.x >> 6] = s1;
    if (inv) flags[4] = 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *) (flags + at), (unsigned long long) (part[0][0] + part[0][1] + part[0][2] + part[0][3]));
        atomicAdd((unsigned long long *) (flags + at + 2), (unsigned long long) (part[1][0] + part[1][1] + part[1][2] + part[1][3]));
    }
}

// round 4: the heads alone (sorted keys only).  With the ids known from a scan of these flags BEFORE the verification -- on the assumption, checked
// afterwards, that no hash group holds two k-mers -- one pass through the permutation both gathers what verify_group needs and writes what
// finish_heads_kernel wrote in a second pass (gather_finish_kernel); a collision falls back to the two passes below.
__global__ void heads_only_kernel(GroupArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    const uint32_t h = i == 0 || a.sorted_key[i] != a.sorted_key[i - 1];
    a.head[i] = h;
    a.newclus[i] = h;
    a.head_idx[i] = h? i : 0u;
}

__global__ void mark_heads_kernel(GroupArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    uint32_t h = i == 0 || a.sorted_key[i] != a.sorted_key[i - 1];
    a.head[i] = h;
    a.newclus[i] = h;
    a.head_idx[i] = h? i : 0u;
    // the gathers verify_group would otherwise chain in front of every k-mer read, done here once, a lane per record
    const uint32_t p = a.perm[i];
    const uint4 r0 = a.slot_rec[2 * (size_t) p], r1 = a.slot_rec[2 * (size_t) p + 1];
    a.loc[i] = (uint64_t) r1.y << 32 | r1.x;
    a.occ_sorted[i] = (uint64_t) r0.y << 32 | r0.x, a.smer_sorted[i] = (uint64_t) r0.w << 32 | r0.z;
}
// after a split of colliding groups the permutation has changed inside them: fetch again
__global__ void regather_kernel(GroupArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    const uint32_t p = a.perm[i];
    const uint4 r0 = a.slot_rec[2 * (size_t) p], r1 = a.slot_rec[2 * (size_t) p + 1];
    a.loc[i] = (uint64_t) r1.y << 32 | r1.x;
    a.occ_sorted[i] = (uint64_t) r0.y << 32 | r0.x, a.smer_sorted[i] = (uint64_t) r0.w << 32 | r0.z;
}

// Is the k-mer of every sorted record that is not a group head identical to its head's?  Half a wave (32 lanes x 32 bases cover k <= 1024
// in one step) takes a STRIP of eight consecutive records.  The kernel is a chain of dependent gathers -- head flag and head index, the two
// locators, the k-mer words -- and its rate is (bytes in flight) / (HBM latency): with two records per half wave and a wave that retires
// after them it moved 1.5 TB/s at full occupancy (9.0 ms at config 3).  A strip pays the first two levels once for eight records (lane r of
// the half wave fetches record r's, the values travel by shuffle) and has the sixteen k-mer word loads of the strip in flight together.
#ifndef OATK_VG_STRIP
#define OATK_VG_STRIP 8
#endif
#ifndef OATK_VFY_WAVES
#define OATK_VFY_WAVES 1
#endif
__global__ __launch_bounds__(256, OATK_VFY_WAVES) void verify_group_kernel(GroupArgs a, uint32_t *bad_head)
{
    const uint32_t hl = threadIdx.x & 31, half0 = threadIdx.x & 32;       // lane in the half wave; first lane of the half wave within the wave
    const uint32_t i0 = (uint32_t) OATK_VG_STRIP * (blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5));
    // level 1 + 2: lane r < 8 of the half wave owns record i0 + r
    uint32_t my_h = 0;
    uint64_t my_lp = 0, my_lq = 0;
    bool my_live = false;
    if (hl < OATK_VG_STRIP) {
        const uint32_t i = i0 + hl;
        my_live = i < a.n_rec && !a.head[i];
        if (my_live) {
            my_h = a.head_idx[i];
            my_lp = a.loc[i];
            my_lq = a.loc[my_h];
        }
    }
    const uint64_t live_mask = (__ballot(my_live) >> half0) & ((1ULL << OATK_VG_STRIP) - 1ULL);
    if (live_mask == 0) return;                                           // (a strip of heads: singletons, most of the error k-mers)
    const uint32_t *hs32 = (const uint32_t *) a.hoco_s;
    const int nw = (a.K + 31) / 32;
    uint32_t diff = 0;                                                    // bit r: this lane saw record r differ from its head
    // A strip usually lies inside ONE group (a solid syncmer occurs once per read that covers it: tens of records at HiFi depth): then its
    // records share their head, whose k-mer is fetched and realigned once instead of eight times.  Decided for the wave (both strips).
    const uint32_t first_live = (uint32_t) __builtin_ctzll(live_mask);
    const uint64_t lq0 = (uint64_t) __shfl((long long) my_lq, (int) (half0 + first_live));
    const bool one_head = __ballot(my_live && my_lq != lq0) == 0;
    for (int w0 = 0; w0 < nw; w0 += 32) {                                 // (uniform trip count: every lane takes part in the shuffles, k < 225 idles lanes in the loads only)
        const int wd = w0 + (int) hl;
        const bool in = wd < nw;
        uint64_t p[OATK_VG_STRIP], q[OATK_VG_STRIP];
        if (one_head) {
            const uint64_t q0 = in? kmer_word_global(hs32 + (lq0 >> 32), (uint32_t) lq0 >> 1, (uint32_t) lq0 & 1u, a.K, wd) : 0;
#pragma unroll
            for (int r = 0; r < OATK_VG_STRIP; ++r) {                     // every load of the strip is issued before the first comparison
                const uint64_t lp = (uint64_t) __shfl((long long) my_lp, (int) (half0 + r));
                const bool live = in && ((live_mask >> r) & 1u);
                p[r] = live? kmer_word_global(hs32 + (lp >> 32), (uint32_t) lp >> 1, (uint32_t) lp & 1u, a.K, wd) : q0;
            }
#pragma unroll
            for (int r = 0; r < OATK_VG_STRIP; ++r) diff |= (uint32_t) (p[r] != q0) << r;
        } else {
#pragma unroll
            for (int r = 0; r < OATK_VG_STRIP; ++r) {
                const uint64_t lp = (uint64_t) __shfl((long long) my_lp, (int) (half0 + r)), lq = (uint64_t) __shfl((long long) my_lq, (int) (half0 + r));
                const bool live = in && ((live_mask >> r) & 1u);
                p[r] = live? kmer_word_global(hs32 + (lp >> 32), (uint32_t) lp >> 1, (uint32_t) lp & 1u, a.K, wd) : 0;
                q[r] = live? kmer_word_global(hs32 + (lq >> 32), (uint32_t) lq >> 1, (uint32_t) lq & 1u, a.K, wd) : 0;
            }
#pragma unroll
            for (int r = 0; r < OATK_VG_STRIP; ++r) diff |= (uint32_t) (p[r] != q[r]) << r;
        }
    }
#pragma unroll
    for (int r = 0; r < OATK_VG_STRIP; ++r) {
        const uint64_t bad = (__ballot((diff >> r) & 1u) >> half0) & 0xFFFFFFFFULL;
        if (bad && hl == (uint32_t) r) a.flags[0] = 1u, bad_head[my_h] = 1u;   // lane r holds record r's head index
    }
}

// one lane per colliding group: first-seen clustering, then a stable partition of the group's slice of perm
#define OATK_MAX_SPLIT 16
__global__ void split_collisions_kernel(GroupArgs a, const uint32_t *bad_head, uint32_t *perm_rw, uint32_t *tag, uint32_t *tmp_perm)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec || !a.head[i] || !bad_head[i]) return;
    uint32_t e = i + 1;
    while (e < a.n_rec && !a.head[e]) ++e;
    const int nw = (a.K + 31) / 32;
    uint32_t rep[OATK_MAX_SPLIT], nclus = 0;
    for (uint32_t t = i; t < e; ++t) {
        uint32_t p = perm_rw[t];
        uint64_t lo_p = a.pos_lo[p];
        const uint32_t *hs_p = (const uint32_t *) (a.hoco_s + (a.off[(lo_p >> 32) - a.sid0] >> 2));
        uint32_t mp = a.pos_mpos[p], c;
        for (c = 0; c < nclus; ++c) {
            uint32_t q = perm_rw[rep[c]];
            uint64_t lo_q = a.pos_lo[q];
            const uint32_t *hs_q = (const uint32_t *) (a.hoco_s + (a.off[(lo_q >> 32) - a.sid0] >> 2));
            uint32_t mq = a.pos_mpos[q];
            bool same = true;
            for (int wd = 0; wd < nw && same; ++wd)
                same = kmer_word_global(hs_p, mp >> 1, mp & 1u, a.K, wd) == kmer_word_global(hs_q, mq >> 1, mq & 1u, a.K, wd);
            if (same) break;
        }
        if (c == nclus) {
            if (nclus == OATK_MAX_SPLIT) { a.flags[2] = 1u; return; }
            rep[nclus++] = t;
        }
        tag[t] = c;
    }
    // stable partition by cluster; clusters keep first-seen order (syncmer.c:1324-1334, :1353-1360)
    uint32_t w = i;
    for (uint32_t c = 0; c < nclus; ++c) {
        bool first = true;
        for (uint32_t t = i; t < e; ++t)
            if (tag[t] == c) {
                tmp_perm[w] = perm_rw[t];
                a.newclus[w] = first? 1u : 0u;
                first = false;
                ++w;
            }
    }
    for (uint32_t t = i; t < e; ++t) perm_rw[t] = tmp_perm[t];
}

struct FinishArgs {
    const uint32_t *perm;
    const uint32_t *newclus;
    const uint32_t *clus_id;      // inclusive scan of newclus, minus one
    uint32_t n_rec;
    const uint64_t *sorted_key, *smer_sorted;     // smer_sorted: the s-mer of every sorted record (mark_heads_kernel)
    uint64_t *scm_h, *scm_s;
    uint64_t *scm_occ_off;        // [n_scm + 1]
    uint64_t *scm_occ;            // [n_rec]: written by mark_heads_kernel (GroupArgs::occ_sorted)
    const uint64_t *loc;          // k-mer locator of every sorted record (mark_heads_kernel)
    uint64_t *scm_loc;            // [n_scm] where a syncmer's k-mer can be read: the locator of its first occurrence (what the EC graph's vertices need)
    uint64_t *pos_kid;            // id << 1 per slot
    uint32_t *flags;
};

__global__ void finish_heads_kernel(FinishArgs a, uint32_t n_scm)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    uint32_t p = a.perm[i], id = a.clus_id[i];
    a.pos_kid[p] = (uint64_t) id << 1;
    if (a.newclus[i]) {
        a.scm_h[id] = a.sorted_key[i];
        a.scm_s[id] = a.smer_sorted[i];
        a.scm_occ_off[id] = i;
        a.scm_loc[id] = a.loc[i];
    }
    if (i == a.n_rec - 1) a.scm_occ_off[n_scm] = a.n_rec;
}

// mark_heads_kernel's gathers and finish_heads_kernel's writes in one pass (clus_id1 = id + 1, straight from the inclusive scan of the head flags)
__global__ void gather_finish_kernel(GroupArgs g, FinishArgs a, const uint32_t *clus_id1, uint32_t n_scm)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    const uint32_t p = g.perm[i], id = clus_id1[i] - 1u;
    const uint4 r0 = g.slot_rec[2 * (size_t) p], r1 = g.slot_rec[2 * (size_t) p + 1];
    const uint64_t loc = (uint64_t) r1.y << 32 | r1.x, smer = (uint64_t) r0.w << 32 | r0.z;
    g.loc[i] = loc;
    g.occ_sorted[i] = (uint64_t) r0.y << 32 | r0.x, g.smer_sorted[i] = smer;
    a.pos_kid[p] = (uint64_t) id << 1;
    if (g.newclus[i]) {
        a.scm_h[id] = a.sorted_key[i];
        a.scm_s[id] = smer;
        a.scm_occ_off[id] = i;
        a.scm_loc[id] = loc;
    }
    if (i == a.n_rec - 1) a.scm_occ_off[n_scm] = a.n_rec;
}

// identical k-mers must carry the same s-mer (fatal in the reference, syncmer.c:1370-1376)
__global__ void check_smer_kernel(FinishArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rec) return;
    uint32_t id = a.clus_id[i];
    if (a.smer_sorted[i] != a.scm_s[id]) a.flags[1] = 1u;
}

__global__ void cov_kernel(const uint64_t *occ_off, uint32_t *cov, uint32_t n_scm)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_scm) cov[i] = (uint32_t) (occ_off[i + 1] - occ_off[i]);
}

}  // namespace oatk
