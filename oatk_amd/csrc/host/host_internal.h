/*
 * oatk_amd/csrc/host/host_internal.h -- the second halves of the host adaptors (include/oatk_syncasm.h), shared between the one-handle entry
 * points and the N-handle ones of multi_host.c (include/oatk_multi.h): each adaptor is "run the device step, then turn what is resident into the
 * reference's structs"; with reads sharded over several handles the device step is a collective and the second half is the same.
 */
#ifndef OATK_HOST_INTERNAL_H
#define OATK_HOST_INTERNAL_H

#include "oatk_hip_stat.h"
#include "oatk_syncasm.h"

/* syncmer_db_t from the flat table (collect_syncmer_from_reads, syncmer.c:1419-1444) and every read's k_mer rewritten from `kid` (all reads' ids in
 * read order, :1378).  With arenas `occ` becomes the table's storage (adopted); everything else stays the caller's. */
oatk_syncmer_db_t *oatk_host_build_syncmer_db(oatk_sr_db_t *sr_db, uint64_t n_scm, uint64_t n_occ, const uint64_t *h, const uint64_t *s, const uint32_t *cov,
                                              const uint64_t *occ_off, uint64_t **occ, const uint64_t *kid);
/* what read_error_correction leaves in the reads (syncerr.c:600-612) and update_syncmer_db in the table (:769-814), from flat arrays: new_n per read,
 * chains concatenated in read order, the refreshed table.  With arenas the chain arrays and `occ` are adopted (the pointers are cleared). */
void oatk_host_ec_write_back(oatk_sr_db_t *sr_db, oatk_syncmer_db_t *scm_db, const uint32_t *new_n, uint64_t **new_k, uint32_t **new_m, uint64_t **new_s,
                             const uint32_t *cov, const uint8_t *del, const uint64_t *occ_off, uint64_t **occ);
/* asmg_t from the graph resident in ctx (OATK_BUF_AG_*), scm_db->a[i].del updated like syncasm.c:228 */
oatk_asmg_t *oatk_host_asmg_from_resident(oatk_hip_ctx *ctx, oatk_syncmer_db_t *scm_db, uint64_t nv, uint64_t na, int *rc);
oatk_consensus_t *oatk_host_consensus_from_resident(oatk_hip_ctx *ctx, int k, int *rc);
oatk_overlap_t *oatk_host_overlap_from_resident(oatk_hip_ctx *ctx, uint64_t n_pairs, uint64_t n_entries, int *rc);
/* sr_db_stat's arithmetic, report and sr_db->stats from the raw tabulation */
int oatk_host_stat_report(oatk_sr_db_t *sr_db, const oatk_stat_raw_t *raw, FILE *fo);
/* scg_read_alignment with the reads spread over n handles: handle r holds reads [first[r], first[r + 1]) */
int oatk_host_read_alignment_n(oatk_hip_ctx **ctx, const uint64_t *first, int n, oatk_sr_db_t *sr_db, oatk_scg_ra_v *ra_v, oatk_scg_t *g, int for_unzip,
                               uint64_t *n_skipped, uint32_t **skipped);
/* sr_read (syncmer.c:487) for files with the reads spread over n handles by position in the input: first[0 .. n] receives the read ranges */
int oatk_host_sr_read_files_n(oatk_hip_ctx **ctx, int n, oatk_sr_db_t *sr_db, char **files, int n_files, uint64_t *first, uint64_t m_data);


typedef struct { uint8_t *p, *end; } oatk_name_bump_t;
char *oatk_host_name_dup(const uint8_t *src, size_t len, oatk_name_bump_t *b, const void *owner);
int oatk_host_threads_granted(void);
void oatk_host_set_threads_internal(int n);

/* a gzip'ed file (one member, several, BGZF) as a stream of inflated bytes (gzsrc.c) */
typedef struct oatk_gzsrc oatk_gzsrc_t;
oatk_gzsrc_t *oatk_gzsrc_open(const char *path, int n_threads, int *rc);
int64_t oatk_gzsrc_read(oatk_gzsrc_t *g, uint8_t *dst, uint64_t cap);
uint64_t oatk_gzsrc_tell_in(const oatk_gzsrc_t *g);
uint64_t oatk_gzsrc_size_in(const oatk_gzsrc_t *g);
uint64_t oatk_gzsrc_members_on_many_threads(const oatk_gzsrc_t *g);      /* (statistics, tests) */
int oatk_gzsrc_kind(const oatk_gzsrc_t *g);          /* 1 plain member(s), 2 BGZF, 3 not a regular file (zlib's gzread) */
void oatk_gzsrc_close(oatk_gzsrc_t *g);
/* one member's deflate data inflated on many threads (host/gzpar.c) */
typedef struct oatk_gzpar oatk_gzpar_t;
oatk_gzpar_t *oatk_gzpar_open(const uint8_t *deflate, uint64_t n_in, int n_threads);
int64_t oatk_gzpar_read(oatk_gzpar_t *p, uint8_t *dst, uint64_t cap);
int oatk_gzpar_done(const oatk_gzpar_t *p);
uint64_t oatk_gzpar_in_used(const oatk_gzpar_t *p);
uint32_t oatk_gzpar_crc(const oatk_gzpar_t *p);
uint64_t oatk_gzpar_total(const oatk_gzpar_t *p);
void oatk_gzpar_close(oatk_gzpar_t *p);

#endif
