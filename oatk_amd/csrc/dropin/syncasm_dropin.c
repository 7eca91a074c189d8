/*
 * oatk_amd/csrc/dropin/syncasm_dropin.c -- the reference's hot-path symbols, served from the MI355X (include/oatk_dropin.h).
 *
 * Six functions with the reference's names and signatures; each tries the device (liboatk_host.so over liboatk_hip.so) and
 * otherwise runs the maintainer's ORIGINAL body, which the build keeps reachable as orig_<name> (objcopy --redefine-sym on the
 * reference's object files).  The struct types are the layout-compatible mirrors of include/oatk_syncasm.h, so this file needs
 * none of the reference's headers.  There is no arithmetic of the hot path in here: only which path runs, and the bookkeeping
 * of what is resident on the device.
 *
 *   run_syncasm.c:81    sr_read                     -> files' text to the device, records + scan there          (syncmer.c:487)
 *   run_syncasm.c:88    sr_db_stat                  -> device histograms, host peak finder                      (syncmer.c:867)
 *   run_syncasm.c:103   collect_syncmer_from_reads  -> device count                                             (syncmer.c:1397)
 *   run_syncasm.c:109   make_syncmer_graph(.,.,0,0.) -> an EMPTY graph goes out: the graph the correction runs against is built on
 *                       the device inside read_error_correction, so scg_consensus(hoco) at :117 has nothing to do and
 *                       scg_destroy at :132 frees the placeholder                                                (syncasm.c:203)
 *   run_syncasm.c:124   read_error_correction       -> device EC graph + correction + table refresh             (syncerr.c:819)
 *   run_syncasm.c:138   make_syncmer_graph(c, a)    -> device arc counting + filters + asmg_finalize order       (syncasm.c:203)
 *   run_syncasm.c:221.. scg_read_alignment          -> device chaining of every read                            (alignment.c:596)
 *   syncasm.c:888, :477 scg_syncmer_consensus / calc_syncmer_overlap consult oatk_hook_cons / oatk_hook_ovl first
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "oatk_dropin.h"
#include "oatk_hip_ec.h"
#include "oatk_multi.h"
#include "oatk_syncasm.h"

/* sstream.h:46-51 */
typedef struct { uint64_t n_seq; char **files; int n_files, n; void *s; } dropin_sstream_t;

/* ---- the original bodies (the reference's own definitions under their build-time names) ---- */
void orig_sr_read(dropin_sstream_t *s_stream, oatk_sr_db_t *sr_db, size_t mD, int n_threads);
void orig_sr_db_stat(oatk_sr_db_t *sr_db, FILE *fo, int verbose);
oatk_syncmer_db_t *orig_collect_syncmer_from_reads(oatk_sr_db_t *sr_db);
oatk_scg_t *orig_make_syncmer_graph(oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f);
void orig_read_error_correction(oatk_sr_db_t *sr_db, oatk_scg_t *g, double max_edist, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c,
                                double max_arc_f, int n_threads, FILE *fo, int verbose);
void orig_scg_read_alignment(oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int n_threads, int for_unzip);
/* reference functions used as they are */
void scg_consensus(oatk_sr_db_t *sr_db, oatk_scg_t *scg, int hoco_seq, int save_seq, FILE *fo);          /* syncasm.c:716 */
void scg_destroy(oatk_scg_t *g);                                                                         /* syncasm.c:69 */

int64_t (*oatk_hook_cons)(void *, void *, int, int64_t, void *, int) = 0;
int (*oatk_hook_ovl)(void *, uint64_t, void *, uint64_t, const int32_t **, const uint32_t **, int *) = 0;

enum { F_READ, F_STAT, F_COLLECT, F_GRAPH, F_EC, F_ALIGN, F_CONS, F_OVL, F_COUNT_ };
static const char *F_NAME[F_COUNT_] = {"sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction",
                                       "scg_read_alignment", "scg_syncmer_consensus", "calc_syncmer_overlap"};

static struct {
    int init, enabled, log;
    oatk_hip_ctx *ctx;
    int multi_one_device;          /* ... all of them on one GPU */
    oatk_multi *multi;             /* OATK_DEVICES names several handles: the reads are spread over them (include/oatk_multi.h); ctx is then its handle 0 */
    int corrected;                 /* the resident chains are read_error_correction's */
    int resident;                  /* the device batch mirrors sr_db (its reads, and every rewrite of their chains since) */
    int counted;                   /* ... and the table collect_syncmer_from_reads returned */
    oatk_scg_t *placeholder;       /* the empty graph handed out for the (0, 0.) call */
    oatk_sr_db_t *sr_db;
    oatk_syncmer_db_t *scm_db;
    oatk_consensus_t *cons;
    oatk_overlap_t *ovl;
    uint64_t served[2 * F_COUNT_];
    double secs[2 * F_COUNT_];
    double t_start, t_ready;       /* first call into this file; device context ready */
} D;

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static void note(int f, int original, double t0, const char *why)
{
    const double dt = now() - t0;
    D.served[2 * f + original] += 1, D.secs[2 * f + original] += dt;
    if (D.log && f < F_CONS)
        fprintf(stderr, "[M::oatk_dropin] +%.3f s %s: %s, %.3f s%s%s\n", t0 - D.t_start, F_NAME[f], original? "original body" : "MI355X", dt, why? " -- " : "", why? why : "");
}

static void hooks_off(void)
{
    oatk_hook_cons = 0, oatk_hook_ovl = 0;
    if (D.cons) oatk_consensus_destroy(D.cons);
    if (D.ovl) oatk_overlap_destroy(D.ovl);
    D.cons = 0, D.ovl = 0;
}

static void at_exit(void)
{
    const double t_exit = now();
    hooks_off();
    if (D.log) {
        int f;
        fprintf(stderr, "[M::oatk_dropin] device context ready after %.3f s; exit handlers reached at +%.3f s\n", D.t_ready - D.t_start, t_exit - D.t_start);
        fprintf(stderr, "[M::oatk_dropin] %-28s %10s %10s %10s %10s\n", "function", "MI355X", "seconds", "original", "seconds");
        for (f = 0; f < F_COUNT_; ++f)
            fprintf(stderr, "[M::oatk_dropin] %-28s %10lu %10.3f %10lu %10.3f\n", F_NAME[f], (unsigned long) D.served[2 * f], D.secs[2 * f],
                    (unsigned long) D.served[2 * f + 1], D.secs[2 * f + 1]);
    }
    if (D.multi) oatk_multi_destroy(D.multi);
    else if (D.ctx) oatk_hip_destroy(D.ctx);
    D.ctx = 0, D.multi = 0;
    if (D.log) fprintf(stderr, "[M::oatk_dropin] device context released in %.3f s\n", now() - t_exit);
}

static void init(void)
{
    if (D.init) return;
    D.init = 1;
    D.t_start = now();
    const char *e = getenv("OATK_DROPIN");
    D.enabled = !(e && (e[0] == '0' || e[0] == 'n' || e[0] == 'N'));
    e = getenv("OATK_DROPIN_LOG");
    D.log = e && e[0] && e[0] != '0';
    if (D.enabled) {
        /* OATK_DEVICES=0,1,2,3: one handle per listed ordinal, the reads spread over them (an ordinal may repeat: handles then share a GPU) */
        int dev[64], nd = 0;
        e = getenv("OATK_DEVICES");
        while (e && *e && nd < 64) {
            char *end = 0;
            const long v = strtol(e, &end, 10);
            if (end == e) break;
            dev[nd++] = (int) v;
            e = *end == ','? end + 1 : end;
            if (*end != ',') break;
        }
        if (nd > 1) {
            int i, same = 1;
            for (i = 1; i < nd; ++i) same &= dev[i] == dev[0];
            D.multi_one_device = same;
            D.multi = oatk_multi_create(dev, nd);
            if (D.multi) D.ctx = oatk_multi_ctx(D.multi, 0);
            else fprintf(stderr, "[W::oatk_dropin] OATK_DEVICES: the %d handles or their communicators could not be made: every call runs the original body\n", nd);
        } else {
            e = getenv("OATK_DEVICE");
            D.ctx = oatk_hip_create(nd == 1? dev[0] : (e? atoi(e) : 0));
            if (!D.ctx) fprintf(stderr, "[W::oatk_dropin] no usable MI355X (gfx950) device: every call runs the original body\n");
        }
    }
    e = getenv("OATK_DROPIN_ARENA");
    oatk_host_set_arena(!(e && e[0] == '0'));                            /* this build owns sr_destroy / sr_db_clean / sr_db_destroy (below) */
    D.t_ready = now();
    atexit(at_exit);
}

void oatk_dropin_counts(uint64_t *out16)
{
    memcpy(out16, D.served, sizeof(D.served));
}

static const char *why_not(int rc)
{
    static char buf[512];
    snprintf(buf, sizeof(buf), "device path declined (code %d: %s)", rc, D.multi? oatk_multi_last_error(D.multi) : (D.ctx? oatk_hip_last_error(D.ctx) : "no device"));
    return buf;
}

/* ------------------------------- sr_destroy / sr_db_clean / sr_db_destroy / syncmer_db_clean / syncmer_db_destroy ------------------------------- */
/* syncmer.c:1047-1084 with one difference: a member array that lives in an arena goes with its arena (include/oatk_syncasm.h) */

void sr_destroy(oatk_sr_t *sr) { oatk_sr_destroy(sr); }
void sr_db_clean(oatk_sr_db_t *sr_db) { oatk_sr_db_clean(sr_db); }
void sr_db_destroy(oatk_sr_db_t *sr_db)
{
    if (!sr_db) return;
    const double t0 = now();
    const size_t n = sr_db->n;
    oatk_sr_db_clean(sr_db);
    free(sr_db);
    if (D.init && D.log && n) fprintf(stderr, "[M::oatk_dropin] +%.3f s sr_db_destroy: %lu reads, %.3f s\n", t0 - D.t_start, (unsigned long) n, now() - t0);
}
void syncmer_db_clean(oatk_syncmer_db_t *scm_db) { oatk_syncmer_db_clean(scm_db); }          /* syncmer.c:1094-1110 */
void syncmer_db_destroy(oatk_syncmer_db_t *scm_db)
{
    const double t0 = now();
    const size_t n = scm_db? scm_db->n : 0;
    oatk_syncmer_db_destroy(scm_db);
    if (D.init && D.log && n) fprintf(stderr, "[M::oatk_dropin] +%.3f s syncmer_db_destroy: %lu syncmers, %.3f s\n", t0 - D.t_start, (unsigned long) n, now() - t0);
}

/* ---------------------------------------------------------------- sr_read ---------------------------------------------------------------- */

void sr_read(dropin_sstream_t *s_stream, oatk_sr_db_t *sr_db, size_t mD, int n_threads)
{
    init();
    const double t0 = now();
    const char *why = 0;
    hooks_off();
    D.resident = D.counted = D.corrected = 0, D.sr_db = 0, D.scm_db = 0, D.placeholder = 0;
    if (!D.ctx) why = D.enabled? "no device" : "OATK_DROPIN=0";
    else if (s_stream->n_seq != 0 || s_stream->n != 0) why = "the stream was already read from";
    else if (sr_db->k > oatk_hip_max_k()) why = "k beyond the device scan's window";
    if (!why) {
        oatk_host_set_threads(n_threads);                                /* the struct filling uses as many host threads as the caller grants (-t) */
        oatk_sr_db_clean(sr_db);                                         /* syncmer.c:494-495: k and s stay */
        if (!D.multi || D.multi_one_device) {                            /* device memory in pieces, taken from the driver ahead of the need (include/oatk_hip.h); over several
                                                                          * DEVICES the buffers stay hipMalloc's: RCCL has never been run on anything else here */
            uint64_t text = 0;
            int i;
            for (i = 0; i < s_stream->n_files; ++i) {
                struct stat sb;
                const size_t ln = strlen(s_stream->files[i]);
                if (stat(s_stream->files[i], &sb) == 0 && S_ISREG(sb.st_mode))
                    text += (uint64_t) sb.st_size * (ln > 3 && strcmp(s_stream->files[i] + ln - 3, ".gz") == 0? 4u : 1u);
            }
            if (mD && text > (uint64_t) mD) text = (uint64_t) mD;
            (void) oatk_hip_mem_pool(D.ctx, text + text + text / 2 + (2ull << 30));      /* the batch is ~1.6 bytes per base, scan pieces, count, correction and alignment on top */
        }
        const int rc = D.multi? oatk_multi_sr_read_files_capped(D.multi, sr_db, s_stream->files, s_stream->n_files, (uint64_t) mD)
                              : oatk_sr_read_files_capped(D.ctx, sr_db, s_stream->files, s_stream->n_files, (uint64_t) mD);
        if (rc == OATK_OK) {
            s_stream->n_seq = sr_db->n;                                  /* what sstream_read would have counted (sstream.c:87) */
            D.resident = 1, D.sr_db = sr_db;
            note(F_READ, 0, t0, 0);
            return;
        }
        why = why_not(rc);
    }
    oatk_sr_db_clean(sr_db);                                             /* the original's own sr_db_clean (syncmer.c:494) would free() arena members of an earlier device fill */
    orig_sr_read(s_stream, sr_db, mD, n_threads);
    note(F_READ, 1, t0, why);
}

/* -------------------------------------------------------------- sr_db_stat --------------------------------------------------------------- */

void sr_db_stat(oatk_sr_db_t *sr_db, FILE *fo, int verbose)
{
    init();
    const double t0 = now();
    const char *why = 0;
    if (!D.resident || sr_db != D.sr_db) why = "no resident batch";
    else if (verbose > 1) why = "the verbose histogram plots are the original's";
    else if (D.multi && D.counted && !D.corrected) why = "between the count and the correction the handles' k-mer keys are their own ids";
    if (!why) {
        const int rc = D.multi? oatk_multi_sr_db_stat(D.multi, sr_db, fo, verbose) : oatk_sr_db_stat(D.ctx, sr_db, fo, verbose);
        if (rc == OATK_OK) { note(F_STAT, 0, t0, 0); return; }
        why = why_not(rc);
    }
    orig_sr_db_stat(sr_db, fo, verbose);
    note(F_STAT, 1, t0, why);
}

/* ------------------------------------------------------ collect_syncmer_from_reads ------------------------------------------------------- */

oatk_syncmer_db_t *collect_syncmer_from_reads(oatk_sr_db_t *sr_db)
{
    init();
    const double t0 = now();
    const char *why = 0;
    if (!D.resident || sr_db != D.sr_db) why = "no resident batch";
    if (!why) {
        int rc = 0;
        oatk_syncmer_db_t *db = D.multi? oatk_multi_collect_syncmer_from_reads(D.multi, sr_db, &rc)
                                       : oatk_collect_syncmer_from_reads(D.ctx, sr_db, &rc);     /* "identical kmers have different smers" exits, as in the original */
        if (rc == OATK_OK) {
            D.counted = 1, D.scm_db = db;
            note(F_COLLECT, 0, t0, 0);
            return db;                                                   /* NULL when there are no syncmers (syncmer.c:1414-1417) */
        }
        why = why_not(rc);
    }
    D.resident = 0;                                                      /* the table is the host's from here on: the device has no ids for it */
    oatk_syncmer_db_t *db = orig_collect_syncmer_from_reads(sr_db);
    note(F_COLLECT, 1, t0, why);
    return db;
}

/* ---------------------------------------------------------- make_syncmer_graph ----------------------------------------------------------- */

static int cmp_u128(const void *a, const void *b)
{
    const unsigned __int128 x = *(const unsigned __int128 *) a, y = *(const unsigned __int128 *) b;
    return (x > y) - (x < y);
}

/* the syncmer -> unitig index of scg_t (syncasm.h:56-62; what scg_scm_utg_index, syncasm.c:116, leaves): one entry
 * scm_id[49] | scm_rev[1] | utg_id[42] | utg_pos[36] per syncmer of every live vertex, ascending, and per syncmer id a pointer to its first */
static void index_syncmers(oatk_scg_t *g)
{
    const oatk_asmg_t *u = g->utg_asmg;
    uint64_t i, j, n = 0, ns = g->scm_db->n;
    for (i = 0; i < u->n_vtx; ++i) if (!u->vtx[i].del) n += u->vtx[i].n;
    if (n == 0) return;
    unsigned __int128 *e = (unsigned __int128 *) malloc(sizeof(unsigned __int128) * n);
    void **idx = (void **) malloc(sizeof(void *) * (ns + 1));
    if (!e || !idx) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(EXIT_FAILURE); }
    int sorted = 1;
    for (i = 0, n = 0; i < u->n_vtx; ++i) {
        if (u->vtx[i].del) continue;
        for (j = 0; j < u->vtx[i].n; ++j, ++n) {
            e[n] = (unsigned __int128) u->vtx[i].a[j] << 78 | (unsigned __int128) i << 36 | j;
            if (n && e[n] < e[n - 1]) sorted = 0;
        }
    }
    if (!sorted) qsort(e, n, sizeof(unsigned __int128), cmp_u128);
    for (i = 0, j = 0; i <= ns; ++i) {                                  /* idx[i] = first entry whose syncmer id is >= i */
        while (j < n && (uint64_t) (e[j] >> 79) < i) ++j;
        idx[i] = e + j;
    }
    g->scm_u = e, g->idx_u = idx;
}

static int64_t hook_cons(void *sr_db, void *scm, int rev, int64_t beg, void *c_seq, int hoco_seq)
{
    const int64_t l = oatk_scg_syncmer_consensus(D.cons, (const oatk_sr_db_t *) sr_db, (uint64_t) ((oatk_syncmer_t *) scm - D.scm_db->a), rev, beg,
                                                 (oatk_kstring_t *) c_seq, hoco_seq);
    D.served[2 * F_CONS + (l < 0)] += 1;
    return l;
}

static int hook_ovl(void *m1, uint64_t rc1, void *m2, uint64_t rc2, const int32_t **dist, const uint32_t **cnt, int *tail)
{
    const oatk_syncmer_t *base = D.scm_db->a;
    int n = oatk_overlap_lookup(D.ovl, (uint64_t) ((oatk_syncmer_t *) m1 - base) << 1 | rc1, (uint64_t) ((oatk_syncmer_t *) m2 - base) << 1 | rc2, dist, cnt, tail);
    if (D.multi && n == 0) n = -1;                                      /* several handles tabulate the pairs between graph vertices only: anything else is the original walk's */
    D.served[2 * F_OVL + (n < 0)] += 1;
    return n;
}

oatk_scg_t *make_syncmer_graph(oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f)
{
    init();
    const double t0 = now();
    const char *why = 0;
    if (!scm_db || scm_db->n == 0) return orig_make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f);       /* syncasm.c:205 */
    if (!D.resident || !D.counted || sr_db != D.sr_db || scm_db != D.scm_db) why = "no resident batch";
    if (!why && min_k_cov == 0 && min_a_cov_f == 0.) {
        /* run_syncasm.c:109: the graph of ALL syncmers exists only to be corrected against.  It is built on the device when the correction
         * runs; the caller gets a graph without vertices, on which scg_consensus (:117) is a no-op and which scg_destroy (:132) frees. */
        oatk_scg_t *g = (oatk_scg_t *) calloc(1, sizeof(oatk_scg_t));
        g->scm_db = scm_db;
        g->utg_asmg = (oatk_asmg_t *) calloc(1, sizeof(oatk_asmg_t));
        D.placeholder = g;
        note(F_GRAPH, 0, t0, "placeholder: the EC graph is built on the device by read_error_correction");
        return g;
    }
    if (!why && D.multi && !D.corrected) why = "with several handles the graph of uncorrected reads is the original's (--no-read-ec)";
    if (!why) {
        int rc = 0;
        oatk_asmg_t *a = D.multi? oatk_multi_make_syncmer_asmg(D.multi, scm_db, min_k_cov, min_a_cov_f, &rc)
                                : oatk_make_syncmer_asmg(D.ctx, scm_db, min_k_cov, min_a_cov_f, &rc);
        if (rc == OATK_OK && a) {
            oatk_scg_t *g = (oatk_scg_t *) calloc(1, sizeof(oatk_scg_t));
            g->scm_db = scm_db, g->utg_asmg = a;
            index_syncmers(g);                                           /* syncasm.c:296 */
            /* from here on scg_consensus runs four times over this table (run_syncasm.c:164-303): its sums and distance tables come
             * from the device in one fetch each */
            hooks_off();
            int r1 = 0, r2 = 0;
            D.cons = D.multi? oatk_multi_consensus_fetch(D.multi, min_k_cov, sr_db->k, &r1) : oatk_consensus_fetch(D.ctx, min_k_cov, sr_db->k, &r1);
            D.ovl = D.multi? oatk_multi_overlap_fetch(D.multi, min_k_cov, &r2) : oatk_overlap_fetch(D.ctx, &r2);
            if (D.cons && !r1) oatk_hook_cons = hook_cons;
            if (D.ovl && !r2) oatk_hook_ovl = hook_ovl;
            D.placeholder = 0;
            note(F_GRAPH, 0, t0, 0);
            return g;
        }
        why = why_not(rc);
    }
    oatk_scg_t *g = orig_make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f);
    note(F_GRAPH, 1, t0, why);
    return g;
}

/* --------------------------------------------------------- read_error_correction --------------------------------------------------------- */

static void ec_report(const uint64_t *st, int verbose)
{
    const char *f = "read_error_correction";                            /* syncerr.c:902-907 and the verbose block behind it */
    fprintf(stderr, "[M::%s] Error Correction Summary Results\n", f);
    fprintf(stderr, "[M::%s] total number of error blocks : %ld\n", f, (long) (st[0] + st[5] + st[10]));
    fprintf(stderr, "[M::%s]                - uncorrected : %ld\n", f, (long) (st[1] + st[6]));
    fprintf(stderr, "[M::%s]                  - corrected : %ld\n", f, (long) (st[2] + st[7]));
    fprintf(stderr, "[M::%s]             - ambiguous seqs : %ld\n", f, (long) (st[3] + st[8]));
    fprintf(stderr, "[M::%s]             - ambiguous path : %ld\n", f, (long) (st[4] + st[9]));
    if (verbose) {
        static const char *lab[11] = {"error blocks in the tail end", "               - uncorrected", "                 - corrected", "            - ambiguous seqs",
                                      "            - ambiguous path", "  error blocks in the middle", "               - uncorrected", "                 - corrected",
                                      "            - ambiguous seqs", "            - ambiguous path", "     error blocks overlapped"};
        int i;
        for (i = 0; i < 11; ++i) fprintf(stderr, "[M::%s] %s : %ld\n", f, lab[i], (long) st[i]);
    }
}

void read_error_correction(oatk_sr_db_t *sr_db, oatk_scg_t *g, double max_edist, uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c,
                           double max_arc_f, int n_threads, FILE *fo, int verbose)
{
    init();
    const double t0 = now();
    const char *why = 0;
    const int placeholder = g && g == D.placeholder;
    oatk_scg_t *real = 0;                                               /* the host graph, when the placeholder has to be replaced after all */
    uint64_t st[12];
    hooks_off();                                                        /* the chains are about to change */
    if (!D.resident || !D.counted || sr_db != D.sr_db || !g || g->scm_db != D.scm_db) why = "no resident batch";
    else if (fo) why = "the corrected reads are to be written out (debug build): the original does that";
    if (!why) {
        oatk_host_set_threads(n_threads);
        int rc = D.multi? (placeholder? oatk_multi_read_error_correction(D.multi, sr_db, g->scm_db, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, st) : OATK_E_ARG)
                        : oatk_read_error_correction(D.ctx, sr_db, g->scm_db, placeholder? 0 : g->utg_asmg, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, st);
        if (rc == OATK_E_SPLIT && placeholder && !D.multi) {
            /* the device refuses to ORDER this graph (duplicate arcs of a long tandem repeat, an arc with dozens of distances): the original
             * builds it, the correction itself still runs on the device against that graph */
            if (D.log) fprintf(stderr, "[M::oatk_dropin] read_error_correction: %s; graph from the original make_syncmer_graph + scg_consensus\n", why_not(rc));
            real = orig_make_syncmer_graph(sr_db, g->scm_db, 0, 0.);
            scg_consensus(sr_db, real, 1, 1, 0);
            rc = oatk_read_error_correction(D.ctx, sr_db, g->scm_db, real->utg_asmg, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, st);
        }
        if (rc == OATK_OK) {
            ec_report(st, verbose);
            D.corrected = 1, D.placeholder = 0;                         /* (the caller's scg_destroy frees it: a later graph may reuse its address) */
            if (real) scg_destroy(real);
            note(F_EC, 0, t0, real? "device correction against the original's graph" : 0);
            return;
        }
        why = why_not(rc);
    }
    D.resident = 0, D.placeholder = 0;                                  /* the host rewrites the chains: the device batch is stale */
    if (placeholder && !real) {
        real = orig_make_syncmer_graph(sr_db, g->scm_db, 0, 0.);
        scg_consensus(sr_db, real, 1, 1, 0);
    }
    oatk_sr_db_own_chains(sr_db);                                       /* the original reallocs k_mer / m_pos / s_mer (syncerr.c:604-608) */
    oatk_syncmer_db_own_mpos(g->scm_db);                                /* ... and frees and mallocs every syncmer's occurrence list (:789-790) */
    orig_read_error_correction(sr_db, real? real : g, max_edist, err_mer_c, max_err_c, err_arc_c, max_arc_f, n_threads, fo, verbose);
    if (real) scg_destroy(real);
    note(F_EC, 1, t0, why);
}

/* ---------------------------------------------------------- scg_read_alignment ----------------------------------------------------------- */

void scg_read_alignment(oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int n_threads, int for_unzip)
{
    init();
    const double t0 = now();
    const char *why = 0;
    if (!D.resident || !D.counted || sr_db != D.sr_db || !g || g->scm_db != D.scm_db) why = "no resident batch";
    else if (!g->idx_u) why = "the graph carries no syncmer index";
    else if (D.multi && !D.corrected) why = "with several handles the uncorrected chains on the devices carry the handles' own ids (--no-read-ec)";
    if (!why) {
        uint64_t n_skipped = 0;
        const int rc = D.multi? oatk_multi_scg_read_alignment(D.multi, sr_db, ra_v, g, for_unzip, &n_skipped)
                              : oatk_scg_read_alignment(D.ctx, sr_db, ra_v, g, for_unzip, &n_skipped, 0);     /* all or nothing: ra_v untouched on refusal */
        if (rc == OATK_OK) { note(F_ALIGN, 0, t0, 0); return; }
        why = n_skipped? "reads beyond the device aligner's per-read limits" : why_not(rc);
    }
    orig_scg_read_alignment(sr_db, ra_v, g, n_threads, for_unzip);
    note(F_ALIGN, 1, t0, why);
}
