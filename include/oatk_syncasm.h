/*
 * include/oatk_syncasm.h -- host mirror of the reference's hot-path data structures and entry points.
 *
 * The structs below are LAYOUT-COMPATIBLE with the reference's (same member order, types and widths), because
 * the rest of oatk -- make_syncmer_graph, scg_consensus, read_error_correction's caller, pathfinder_minicircle --
 * keeps consuming them unchanged.  They are declared under oatk_-prefixed names so this header can be
 * included next to the reference's own syncmer.h in one translation unit (tests do exactly that through the shim and
 * hand these objects to the reference's functions).
 *
 *   oatk_sr_t          <->  sr_t          syncmer.h:48-70
 *   oatk_sr_db_t       <->  sr_db_t       syncmer.h:79-84   (kvec: n, m, a; then k, s, stats)
 *   oatk_syncmer_t     <->  syncmer_t     syncmer.h:86-96
 *   oatk_syncmer_db_t  <->  syncmer_db_t  syncmer.h:99-114
 *
 * Every array handed out is malloc-family memory: the reference frees members with free() (syncmer.c:1047-1110).
 */
#ifndef OATK_SYNCASM_H
#define OATK_SYNCASM_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "oatk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t sid;          /* read id = index in input order */
    char *sname;
    uint32_t hoco_l;       /* homopolymer-compressed length */
    uint8_t *hoco_s;       /* 2-bit bases, 4 per byte, MSB first; ambiguous bases stored as A */
    uint8_t *ho_rl;        /* min(run, 256) - 1 per hoco position */
    uint32_t *ho_l_rl;     /* run - 1 for runs > 255, in position order (NULL if none) */
    uint32_t *n_nucl;      /* raw coordinates of ambiguous bases (NULL if none) */
    uint32_t n;            /* syncmers on the read */
    uint32_t *m_pos;       /* hoco position << 1 | strand */
    uint64_t *s_mer;       /* canonical s-mer << 1 | open/close-strand bit */
    uint64_t *k_mer;       /* k-mer hash after the scan; syncmer id << 1 | corrected after the count */
} oatk_sr_t;

typedef struct {
    uint64_t syncmer_n;
    double syncmer_per_read, syncmer_avg_dist, smer_avg_cnt, kmer_avg_cnt;
    int smer_unique, smer_singleton, smer_peak_hom, smer_peak_het;
    int kmer_unique, kmer_singleton, kmer_peak_hom, kmer_peak_het;
} oatk_sr_stat_t;

typedef struct {
    size_t n, m;
    oatk_sr_t *a;
    int k, s;
    oatk_sr_stat_t *stats;
} oatk_sr_db_t;

typedef struct {
    uint64_t h, s;         /* k-mer hash, s-mer code */
    uint32_t cov:31, del:1;
    uint64_t *m_pos;       /* occurrences: sid << 32 | index on read << 1 | strand */
} oatk_syncmer_t;

typedef struct {
    size_t n, m;
    oatk_syncmer_t *a;
    uint16_t *c;
    uint64_t *h;
} oatk_syncmer_db_t;

/* Host threads the adaptors below use to fill the structs (default: up to 16 of the online cores): pass the caller's n_threads, as the
 * reference's own functions take it (syncmer.c:487, syncerr.c:819).  oatk_par_run calls fn(arg, tid, n) on n threads and joins them. */
void oatk_host_set_threads(int n);
int oatk_host_threads(void);
typedef void (*oatk_par_fn)(void *arg, int tid, int n_threads);
void oatk_par_run(oatk_par_fn fn, void *arg);
void oatk_par_run_n(oatk_par_fn fn, void *arg, int n_threads);

/* malloc'ed and initialised like sr_db_init (syncmer.c:1060-1067) */
oatk_sr_db_t *oatk_sr_db_new(int k, int s);

/* Scan a packed read stream on the device (oatk_hip_scan_host) and fill sr_db->a[0..n_reads) from the resident
 * results -- the body of sr_read (syncmer.c:487) after the reader loop.  sr_db must be initialised (k, s set, n = 0).
 * names[i] (may be NULL) is adopted as sname.  Returns an OATK_* code; the resident batch stays in ctx for the count. */
int oatk_sr_read_packed(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint8_t *seq, const uint64_t *off, const uint32_t *len,
                        uint64_t n_reads, uint64_t seq_bytes, char **names);

/* sr_db->a[first .. first + n_reads) from the scan resident in ctx, whose read 0 is read `first` of the database (a piece of a batch that is
 * assembled with oatk_hip_scan_append); sr_db->a must have room, sr_db->n is raised to first + n_reads */
int oatk_sr_db_fill_range(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, uint64_t first, const uint64_t *off, uint64_t n_reads, char **names);
/* the second half of the above for a scan that is already resident (off[i] = offset of read i in the packed stream that was scanned) */
int oatk_sr_db_fill_resident(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, const uint64_t *off, uint64_t n_reads, char **names);
/* sr_read (syncmer.c:487) for files (plain or gzip'ed FASTA / four-line FASTQ), without kseq: text to the device (oatk_ingest_files), record
 * scan and syncmer scan there, sr_db filled from the resident results, snames cut out of the headers */
int oatk_sr_read_files(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, char **files, int n_files);
/* the same with sr_read's data cap (-D, syncmer.c:537-541): reading stops behind the read that takes the total of raw bases to m_data (0 = no cap) */
int oatk_sr_read_files_capped(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t m_data);
/* the device half of the above for text that is already in host memory: streamed through the device in windows (upload of window i + 1 beside
 * the record scan and syncmer scan of window i), the scanned pieces assembled in ctx; nothing is copied back.  pinned != 0: the text is
 * page-locked and goes over PCIe as it lies.  window = 0: default. */
int oatk_scan_text(oatk_hip_ctx *ctx, const uint8_t *text, uint64_t n_bytes, int pinned, int k, int s, uint64_t window, uint64_t *n_reads);
/* Test hook: the text window (bytes) oatk_sr_read_files streams the input in; 0 = default (768 MiB).  Results never depend on it. */
void oatk_host_debug_window(uint64_t bytes);

/* collect_syncmer_from_reads (syncmer.c:1397): count on the device, build syncmer_db_t, rewrite sr->k_mer to id << 1.
 * Returns NULL when there are no syncmers (syncmer.c:1414-1417) or on error (*rc set). */
oatk_syncmer_db_t *oatk_collect_syncmer_from_reads(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, int *rc);

/* ---- assembly graph as the reference builds it (graph.h:39-63), layout-compatible; only read here ---- */
typedef struct {
    uint64_t v, w;         /* oriented vertices: id << 1 | rev */
    uint64_t ln, ls;       /* overlap in syncmers / in consensus bases */
    uint32_t cov:30, del:1, comp:1;
    uint64_t link_id;
} oatk_asmg_arc_t;

typedef struct {
    uint64_t n;
    uint64_t *a;
    char *seq;
    uint64_t len;
    uint32_t cov:30, del:1, circ:1;
} oatk_asmg_vtx_t;

typedef struct {
    uint64_t n_vtx, m_vtx;
    oatk_asmg_vtx_t *vtx;
    uint64_t n_arc, m_arc;
    oatk_asmg_arc_t *arc;
    uint64_t *idx_p, *idx_n;
} oatk_asmg_t;

/* read_error_correction (syncerr.c:819) on the device: `asmg` is scg->utg_asmg of the EC graph the reference built with
 * make_syncmer_graph(sr_db, scm_db, 0, 0.) + scg_consensus(hoco) (run_syncasm.c:109-117).  Rewrites every read's
 * k_mer / m_pos / s_mer / n and the syncmer table's cov / del / m_pos exactly like the reference, marks the graph's
 * deleted vertices and arcs like find_error_syncmers(..., del_err = 1), and returns the block statistics in stats12
 * (layout of include/oatk_hip_ec.h).  The batch must still be resident in ctx (scan + count done on it).
 * asmg == NULL: the EC graph is built on the device too (oatk_hip_ec_graph) -- the caller then skips make_syncmer_graph,
 * scg_consensus and scg_destroy of run_syncasm.c:109-132 altogether; only the reads and the syncmer table are written back. */
int oatk_read_error_correction(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, oatk_asmg_t *asmg, double max_edist,
                               uint32_t err_mer_c, uint32_t max_err_c, uint32_t err_arc_c, double max_arc_f, uint64_t *stats12);

/* make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299, run_syncasm.c:138) on the device, from the batch resident
 * in ctx (after oatk_read_error_correction: the corrected reads): returns what the reference stores in scg->utg_asmg -- one vertex per
 * surviving syncmer, filtered arcs in (v, w) order with symmetry flags and link ids, the index -- allocated so that the reference's
 * asmg_destroy frees it, and updates scm_db->a[i].del like syncasm.c:228.  NULL with *rc == 0 when the table is empty (syncasm.c:205).
 * The caller wraps it: scg->scm_db = scm_db; scg->utg_asmg = g; scg_scm_utg_index(scg). */
oatk_asmg_t *oatk_make_syncmer_asmg(oatk_hip_ctx *ctx, oatk_syncmer_db_t *scm_db, uint32_t min_k_cov, double min_a_cov_f, int *rc);
void oatk_asmg_destroy(oatk_asmg_t *g);

/* sstream_open + the sstream_read loop (sstream.c:70-103, syncmer.c:519-543) without kseq: the files (plain or gzip'ed FASTA / four-line
 * FASTQ, read one after the other) are inflated into host memory and their TEXT is handed to the device reader
 * (include/oatk_hip_ingest.h), which leaves the packed read stream resident: follow with oatk_hip_scan_ingested. */
int oatk_ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads);

/* sr_db_stat (syncmer.c:867): the two sorts and the tabulation on the device (include/oatk_hip_stat.h), the peak finder and the report
 * here; fills sr_db->stats (allocated if NULL) and prints the reference's nine lines to fo (may be NULL).  Works on the batch resident
 * in ctx at whatever stage it is -- after sr_read (run_syncasm.c:88) or after read_error_correction (:131). */
int oatk_sr_db_stat(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, FILE *fo, int verbose);
/* the peak finder of sr_db_stat alone (what ha_analyze_count, syncmer.c:768-864, returns for MAX_DEPTH + 1 = 1001 bins and LOWEST_CUT 5):
 * cnt[c] = how many distinct s-mers / k-mers occur c times, c = 0 .. 1000 (the last bin collects everything deeper) */
void oatk_stat_peaks(const int64_t *cnt, int *peak_hom, int *peak_het);

/* ---- scg_syncmer_consensus (syncasm.c:888-1003) served from the device ----
 * oatk_consensus_fetch runs oatk_hip_consensus (include/oatk_hip_cons.h) on the resident batch -- after the count, or after the
 * error correction -- and copies the per-syncmer arrays to the host; oatk_scg_syncmer_consensus then appends to c_seq exactly what
 * the reference's function appends for (syncmer id, rev, beg, hoco_seq) and returns what it returns, or -1 when the syncmer was not
 * prepared (coverage below min_cov or deleted): the caller then runs its own routine.  oatk_kstring_t is kstring_t (kstring.h). */
typedef struct { size_t l, m; char *s; } oatk_kstring_t;
typedef struct {
    uint64_t n_scm, n_sel;
    int k;
    uint32_t *slot;        /* [n_scm] index into the arrays below, ~0 = not prepared */
    uint32_t *rl;          /* [n_sel * k] rounded mean run length per forward hoco position */
    uint32_t *m_seq;       /* [n_sel] occurrences that took part */
    uint64_t *first;       /* [n_sel] the first of them (sid << 32 | idx << 1 | rev), ~0 if none */
} oatk_consensus_t;
oatk_consensus_t *oatk_consensus_fetch(oatk_hip_ctx *ctx, uint32_t min_cov, int k, int *rc);
void oatk_consensus_destroy(oatk_consensus_t *c);
int64_t oatk_scg_syncmer_consensus(const oatk_consensus_t *cs, const oatk_sr_db_t *sr_db, uint64_t scm_id, int rev, int64_t beg,
                                   oatk_kstring_t *c_seq, int hoco_seq);

/* calc_syncmer_overlap (syncasm.c:477-582) and scg_unitig_consensus (:1004-1046) served from the device's pair tables
 * (include/oatk_hip_cons.h: oatk_hip_overlap_hist).  oatk_overlap_fetch builds the tables of ALL adjacent pairs from the batch resident in
 * ctx (after oatk_read_error_correction: the corrected reads) and copies them to the host.  A maintainer can either keep the reference's
 * own khashl table and refill it -- oatk_overlap_lookup gives the distinct distances of m1 -> m2 in the order the reference's walk would
 * insert them, their counts, and whether one more put of an existing key is due -- or use the replica here:
 * oatk_calc_syncmer_overlap returns what the reference returns, `hm` being the table a caller keeps across calls (NULL = fresh), whose
 * size carries over exactly like the reference's (kh_clear keeps the buckets, so the tie-break of a later pair depends on it).
 * oatk_scg_unitig_consensus appends the unitig's sequence to c_seq and returns its length, or -1 if one of its syncmers has no prepared
 * consensus (c_seq is then partial: reset it and run the original routine). */
typedef struct {
    uint64_t n_pairs, n_entries;
    uint64_t *key;         /* [n_pairs] canonical oriented pairs, ascending: v << 32 | w with v <= w, else the complementary pair's */
    uint64_t *off;         /* [n_pairs + 1] */
    int32_t *dist;         /* [n_entries] distinct distances in first-appearance order */
    uint32_t *cnt;         /* [n_entries] */
    uint8_t *tail;         /* [n_pairs] the walk's last add_ovl_count call was a repeat */
} oatk_overlap_t;
typedef struct { uint32_t bits, count; uint32_t *used; int32_t *keys, *vals; } oatk_ovl_table_t;     /* khashl<int,int>, identity hash */
oatk_overlap_t *oatk_overlap_fetch(oatk_hip_ctx *ctx, int *rc);
void oatk_overlap_destroy(oatk_overlap_t *o);
int oatk_overlap_lookup(const oatk_overlap_t *o, uint64_t v, uint64_t w, const int32_t **dist, const uint32_t **cnt, int *tail_repeat);
oatk_ovl_table_t *oatk_ovl_table_new(void);
void oatk_ovl_table_destroy(oatk_ovl_table_t *h);
int oatk_calc_syncmer_overlap(const oatk_overlap_t *o, uint64_t v, uint64_t w, oatk_ovl_table_t *hm);
int64_t oatk_scg_unitig_consensus(const oatk_consensus_t *cs, const oatk_overlap_t *o, const oatk_sr_db_t *sr_db, const uint64_t *v, uint64_t n,
                                  oatk_kstring_t *c_seq, int hoco_seq);

/* scg_read_alignment (alignment.c:596-691) on the device (include/oatk_hip_align.h): `g` is the reference's scg_t with its unitig graph
 * and syncmer -> unitig index as they are at the time of the call, `ra_v` the previous alignments (read when for_unzip, replaced by the
 * new ones).  The reads aligned are the chains resident in ctx, which must be the whole of sr_db.  Layout-compatible mirrors of syncasm.h:51-80.
 * *n_skipped reads (indices in *skipped, malloc'ed, no particular order) exceeded the device routine's per-read limits (160 unitig hits,
 * 128 fragments, 6 equally good predecessors, 48 fragments in a chain) and got no alignment: the caller runs the original routine for
 * them (none on HiFi data so far).  With skipped == NULL the call is all or nothing: if any read was skipped it returns OATK_E_SPLIT
 * (*n_skipped set) and leaves ra_v as it was, so the caller can run the original routine on the untouched state. */
typedef struct {
    oatk_syncmer_db_t *scm_db;
    oatk_asmg_t *utg_asmg;
    void *scm_u;           /* uint128_t *: scm_id[49] | scm_rev[1] | utg_id[42] | utg_pos[36] */
    void **idx_u;          /* uint128_t **: [n_scm + 1] into scm_u */
} oatk_scg_t;
typedef struct { uint64_t uid, u_beg, u_end; uint32_t s_beg, s_end; } oatk_ra_frg_t;
typedef struct { uint64_t sid; uint32_t n; oatk_ra_frg_t *a; double s; } oatk_scg_ra_t;
typedef struct { size_t n, m; oatk_scg_ra_t *a; } oatk_scg_ra_v;
int oatk_scg_read_alignment(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip, uint64_t *n_skipped,
                            uint32_t **skipped);

/* same destructors as the reference (syncmer.c:1047-1110) for objects that are not handed to it */
void oatk_sr_db_clean(oatk_sr_db_t *sr_db);

/* ---- arenas: for a program that owns the reference's destroy functions as well (include/oatk_dropin.h) -------------------------------------
 * By default every member array handed out is its own malloc'ed block, because the reference frees and reallocs them one by one (sr_destroy
 * syncmer.c:1047-1058; read_error_correction syncerr.c:604-608): 7 blocks per read.  With oatk_host_set_arena(1) the reads of a piece share ONE
 * block (and so do the chains read_error_correction rewrites); then ONLY these functions may free or replace members:
 *   oatk_sr_db_clean / oatk_sr_destroy   sr_db_clean / sr_destroy   (and oatk_syncmer_db_clean / _destroy for the occurrence lists of the syncmer table)
 *   oatk_sr_member_free(p)               free() for a member that may live in an arena
 *   oatk_sr_db_own_chains(sr_db)         k_mer / m_pos / s_mer of every read back into blocks of their own, before handing the database to code
 *                                        that reallocs them (the reference's own read_error_correction) */
void oatk_host_set_arena(int on);
int oatk_host_arena(void);
void *oatk_host_arena_alloc(size_t bytes, const void *owner);      /* released with the owner's oatk_sr_db_clean / oatk_syncmer_db_clean */
void oatk_host_arena_adopt(void *block, size_t bytes, const void *owner);     /* a malloc'ed block becomes an arena of `owner` */
void oatk_syncmer_db_clean(oatk_syncmer_db_t *scm_db);             /* syncmer_db_clean, arena-aware like oatk_syncmer_db_destroy */
void oatk_syncmer_db_own_mpos(oatk_syncmer_db_t *scm_db);          /* every syncmer's m_pos back into a block of its own (before the reference's update_syncmer_db) */
void oatk_sr_member_free(void *p);
void oatk_sr_destroy(oatk_sr_t *sr);
void oatk_sr_db_own_chains(oatk_sr_db_t *sr_db);
void oatk_syncmer_db_destroy(oatk_syncmer_db_t *scm_db);

#ifdef __cplusplus
}
#endif
#endif
