/*
 * oatk_amd/csrc/host/stat_host.c -- host side of the drop-in boundary for sr_db_stat (syncmer.c:867-1028).
 *
 * The device tabulates (oatk_hip_stat): how many s-mers / k-mers occur c times, c = 1 .. 1000+, and the sum of the distances
 * between syncmers adjacent on a read.  Here: the averages (kh_ctab_stat :619-646 -- its double sum adds integers below 2^53, so one
 * division of exact totals gives the same double), the peak finder (ha_analyze_count :768-864) and the nine lines sr_db_stat prints.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oatk_hip_stat.h"
#include "oatk_syncasm.h"
#include "host_internal.h"

#define LOWEST_CUT 5                      /* syncmer.c:754 */

/*
 * The peak finder behind "peak_hom / peak_het" (what ha_analyze_count, syncmer.c:768-864, decides; its verbose plots are not reproduced).
 * In words: skip the error tail (counts fall from depth max(start, 5) on), take the tallest depth after the dip as the main peak, then look for
 * one shoulder on each side: the tallest local maximum between dip and main peak (walking down from the peak) and the tallest one above it
 * (walking up).  A shoulder counts when it reaches 5 % of the main peak and the valley between the two drops to 95 % of the shoulder or lower;
 * one on the right must also sit below 2.5 x the main depth.  A right shoulder makes the main peak the heterozygous one.
 */
typedef struct { int at; int64_t height; } peak_t;

/* tallest local maximum strictly between the depths `from` and `to`, met first when walking from `from` towards `to` */
static peak_t tallest_between(const int64_t *cnt, int from, int to)
{
    peak_t best = {-1, -1};
    const int step = to > from? 1 : -1;
    for (int d = from + step; step > 0? d < to : d > to; d += step)
        if (cnt[d] >= cnt[d - 1] && cnt[d] >= cnt[d + 1] && cnt[d] > best.height) best.at = d, best.height = cnt[d];
    return best;
}

/* does the shoulder stand clear of the main peak?  (the comparisons are made in double, as the reference makes them) */
static int stands_clear(const int64_t *cnt, peak_t shoulder, peak_t main_peak)
{
    const int lo = shoulder.at < main_peak.at? shoulder.at : main_peak.at, hi = shoulder.at < main_peak.at? main_peak.at : shoulder.at;
    int64_t valley = main_peak.height;
    for (int d = lo + 1; d < hi; ++d)
        if (cnt[d] < valley) valley = cnt[d];
    return !(shoulder.height < main_peak.height * 0.05 || valley > shoulder.height * 0.95);
}

static int analyze_count(int n_cnt, int start_cnt, const int64_t *cnt, int *peak_het)
{
    assert(n_cnt > start_cnt);
    *peak_het = -1;
    int dip = cnt[1] > 0? 1 : 2;
    if (dip < start_cnt) dip = start_cnt;
    while (dip + 1 < n_cnt && cnt[dip + 1] <= cnt[dip]) ++dip;         /* the end of the falling error tail */
    if (dip == n_cnt - 1) return -1;                                   /* it never rises again: low coverage */
    peak_t top = {dip + 1, cnt[dip + 1]};
    for (int d = dip + 2; d < n_cnt; ++d)
        if (cnt[d] > top.height) top.at = d, top.height = cnt[d];
    peak_t left = tallest_between(cnt, top.at, dip), right = tallest_between(cnt, top.at, n_cnt - 1);
    if (left.at >= 0 && !stands_clear(cnt, left, top)) left.at = -1;
    if (right.at >= 0 && (!stands_clear(cnt, right, top) || right.at > top.at * 2.5)) right.at = -1;
    if (right.at > 0) {
        *peak_het = top.at;
        return right.at;
    }
    if (left.at > 0) *peak_het = left.at;
    return top.at;
}

/* the peak finder alone, for a histogram cnt[0 .. OATK_STAT_MAX_DEPTH] (include/oatk_syncasm.h) */
void oatk_stat_peaks(const int64_t *cnt, int *peak_hom, int *peak_het)
{
    int het = 0;
    *peak_hom = analyze_count(OATK_STAT_MAX_DEPTH + 1, LOWEST_CUT, cnt, &het);
    *peak_het = het;
}

int oatk_sr_db_stat(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, FILE *fo, int verbose)
{
    oatk_stat_raw_t *raw = (oatk_stat_raw_t *) calloc(1, sizeof(oatk_stat_raw_t));
    int rc = oatk_hip_stat(ctx, raw);
    (void) verbose;                                                /* the histogram plots of verbose > 1 are not reproduced */
    if (!rc) rc = oatk_host_stat_report(sr_db, raw, fo);
    free(raw);
    return rc;
}

int oatk_host_stat_report(oatk_sr_db_t *sr_db, const oatk_stat_raw_t *raw, FILE *fo)
{
    oatk_sr_stat_t *st = sr_db->stats;
    if (!st) st = sr_db->stats = (oatk_sr_stat_t *) calloc(1, sizeof(oatk_sr_stat_t));
    const uint64_t m = raw->n_syncmers, n = raw->n_reads;
    if (m == 0) {
        if (fo) fprintf(fo, "[M::%s] empty syncmer collection\n", "sr_db_stat");
        return OATK_OK;
    }
    const double dist = (double) raw->sum_dist / (int) raw->n_dist;          /* 0/0 = NaN where the reference divides 0.0 by 0 too */
    const int smeru = (int) raw->smer_unique, kmeru = (int) raw->kmer_unique;
    /* no singletons at all: the reference reports a stale variable (kh_ctab_stat :637-643); the device call worked out its value */
    const int smer1 = raw->smer_cnt[1]? (int) raw->smer_cnt[1] : (int) raw->smer_no_singleton;
    const int kmer1 = raw->kmer_cnt[1]? (int) raw->kmer_cnt[1] : (int) raw->kmer_no_singleton;
    const double smera = (double) m / smeru, kmera = (double) m / kmeru;
    int s_het = 0, k_het = 0;
    const int s_hom = analyze_count(OATK_STAT_MAX_DEPTH + 1, LOWEST_CUT, raw->smer_cnt, &s_het);
    const int k_hom = analyze_count(OATK_STAT_MAX_DEPTH + 1, LOWEST_CUT, raw->kmer_cnt, &k_het);
    if (fo) {
        const char *f = "sr_db_stat";
        fprintf(fo, "[M::%s] number syncmers collected: %lu\n", f, (unsigned long) m);
        fprintf(fo, "[M::%s] number syncmers per read: %.3f\n", f, (double) m / n);
        fprintf(fo, "[M::%s] average kmer space: %.3f\n", f, dist);
        fprintf(fo, "[M::%s] number uniqe smer: %d; singletons: %d (%.3f%%)\n", f, smeru, smer1, (double) smer1 * 100 / smeru);
        fprintf(fo, "[M::%s] average smer count: %.3f\n", f, smera);
        fprintf(fo, "[M::%s] smer peak_hom: %d; peak_het: %d\n", f, s_hom, s_het);
        fprintf(fo, "[M::%s] number uniqe kmer: %d; singletons: %d (%.3f%%)\n", f, kmeru, kmer1, (double) kmer1 * 100 / kmeru);
        fprintf(fo, "[M::%s] average kmer count: %.3f\n", f, kmera);
        fprintf(fo, "[M::%s] kmer peak_hom: %d; peak_het: %d\n", f, k_hom, k_het);
    }
    st->syncmer_n = m, st->syncmer_per_read = (double) m / n, st->syncmer_avg_dist = dist;
    st->smer_unique = smeru, st->smer_singleton = smer1, st->smer_avg_cnt = smera, st->smer_peak_hom = s_hom, st->smer_peak_het = s_het;
    st->kmer_unique = kmeru, st->kmer_singleton = kmer1, st->kmer_avg_cnt = kmera, st->kmer_peak_hom = k_hom, st->kmer_peak_het = k_het;
    return OATK_OK;
}
