/*
 * oracle/consensus.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates scg_syncmer_consensus (syncasm.c:888-1003) on flat arrays, split in the two parts the device path splits it in:
 *   orc_consensus_rl      per-position run-length totals of a syncmer over its non-corrected occurrences (:949-988), in the
 *                         syncmer's FORWARD orientation, the number of such occurrences, and the first of them (:912-923)
 *   orc_consensus_string  the string the reference appends for (rev, beg, hoco_seq) from those (:899-910, :925-947, :990-1001)
 * The reverse orientation needs nothing of its own: index t of the reverse request is index K-1-t of the forward totals.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static uint32_t base_at(const uint8_t *hs, uint64_t p) { return (hs[p >> 2] >> (((p & 3) ^ 3) << 1)) & 3; }

void orc_consensus_rl(const orc_reads_view_t *v, uint64_t n_occ, const uint64_t *occ, int K, uint64_t *tot_rl, uint32_t *m_seq, uint64_t *first_occ)
{
    uint64_t i, j;
    memset(tot_rl, 0, sizeof(uint64_t) * (size_t) K);
    *m_seq = 0, *first_occ = UINT64_MAX;
    for (i = 0; i < n_occ; ++i) {
        const uint64_t rd = (occ[i] >> 32) - v->sid0, idx = (occ[i] >> 1) & 0x7FFFFFFFULL, at = v->scm_off[rd] + idx;
        if (v->k_mer[at] & 1) continue;                       /* error-corrected entries carry no position (:958-959) */
        const uint64_t p = v->m_pos[at] >> 1, r = v->m_pos[at] & 1;
        const uint8_t *ho_rl = v->ho_rl + v->rl_off[rd];
        const uint32_t *lrl = v->ho_l_rl + v->lrl_off[rd];
        uint64_t k = 0;
        for (j = 0; j < p; ++j) if (ho_rl[j] == 255) ++k;     /* long runs before the k-mer (:969-972) */
        for (j = 0; j < (uint64_t) K; ++j) {
            uint32_t rl = ho_rl[p + j];
            if (rl == 255) rl = lrl[k++];
            tot_rl[r? (uint64_t) K - 1 - j : j] += rl;
        }
        if (*first_occ == UINT64_MAX) *first_occ = occ[i];
        ++*m_seq;
    }
}

int64_t orc_consensus_string(const orc_reads_view_t *v, const uint64_t *tot_rl, uint32_t m_seq, uint64_t first_occ, int K, int rev, int64_t beg,
                             int hoco_seq, char *out)
{
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    int64_t bl = beg < 0? -beg : 0, o = 0, i;
    while (beg < 0) out[o++] = 'N', ++beg;
    const int64_t l = K - beg;
    bl += l;
    if (first_occ == UINT64_MAX) {                             /* every occurrence was corrected away (:926-932) */
        for (i = 0; i < l; ++i) out[o++] = 'N';
        return bl;
    }
    const uint64_t rd = (first_occ >> 32) - v->sid0, idx = (first_occ >> 1) & 0x7FFFFFFFULL, at = v->scm_off[rd] + idx;
    const uint64_t p = v->m_pos[at] >> 1, r = (v->m_pos[at] & 1) ^ (uint64_t) rev;
    const uint8_t *hs = v->hoco_s + v->hs_off[rd];
    for (i = 0; i < l; ++i) {
        /* index i of the requested orientation; forward index t = beg + i (rev = 0) or its mirror */
        const uint32_t c = r? 3u ^ base_at(hs, p + (uint64_t) (l - 1 - i)) : base_at(hs, p + (uint64_t) (beg + i));
        out[o++] = nt[c];
        if (!hoco_seq) {
            const int64_t t = rev? K - 1 - (beg + i) : beg + i;
            const long b = lround((double) tot_rl[t] / m_seq);
            long q;
            for (q = 0; q < b; ++q) out[o++] = nt[c];
            bl += b;
        }
    }
    return bl;
}
