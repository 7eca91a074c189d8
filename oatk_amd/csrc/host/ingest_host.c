/*
 * oatk_amd/csrc/host/ingest_host.c -- files to the device reader (include/oatk_hip_ingest.h), in place of sstream_open / sstream_read
 * (sstream.c:70-103): several files are read one after the other as one stream of records; plain or gzip'ed (zlib's gzread handles
 * both, exactly as the reference's gzdopen does, sstream.c:50).  The host only moves bytes: inflate (serial per stream on any
 * hardware) and one copy of the text to the device, where the records are found.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "oatk_hip_ingest.h"
#include "oatk_syncasm.h"

/* one plain file that ends in a newline needs no copy at all on the host: it is mapped and handed to the device reader as it lies in the page
 * cache.  Returns 1 when it applied (*rc set), 0 when the general path must run. */
static int ingest_one_mapped(oatk_hip_ctx *ctx, const char *path, uint64_t *n_reads, uint8_t **text, size_t *text_len, int *mapped, int *rc)
{
    struct stat sb;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return 0;
    if (fstat(fd, &sb) != 0 || sb.st_size < 3) { close(fd); return 0; }
    uint8_t *m = (uint8_t *) mmap(0, (size_t) sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 0;
    if ((m[0] == 0x1f && m[1] == 0x8b) || m[sb.st_size - 1] != '\n') { munmap(m, (size_t) sb.st_size); return 0; }
    uint64_t used = 0;
    *rc = oatk_hip_ingest_host(ctx, m, (uint64_t) sb.st_size, OATK_FMT_AUTO, 1, n_reads, &used);
    if (text && !*rc) *text = m, *text_len = (size_t) sb.st_size, *mapped = 1;
    else munmap(m, (size_t) sb.st_size);
    return 1;
}

static int ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads, uint8_t **text, size_t *text_len, int *mapped)
{
    if (mapped) *mapped = 0;
    if (n_files == 1) {
        int rc = 0, dummy = 0;
        if (ingest_one_mapped(ctx, files[0], n_reads, text, text_len, mapped? mapped : &dummy, &rc)) return rc;
    }
    size_t cap = (size_t) 1 << 26, len = 0;
    {   /* size the buffer for the files as they lie (right for plain files, a start for gzip'ed ones): no regrowing copies of gigabytes */
        int j;
        for (j = 0; j < n_files; ++j) { struct stat sb; if (stat(files[j], &sb) == 0 && sb.st_size > 0) cap += (size_t) sb.st_size + 2; }
    }
    uint8_t *buf = (uint8_t *) malloc(cap);
    int i, fmt = OATK_FMT_AUTO;
    if (!buf) return OATK_E_NOMEM;
    for (i = 0; i < n_files; ++i) {
        /* a plain file is read directly (gzread would copy it through zlib's buffers at a third of the speed); two magic bytes tell */
        {
            FILE *pf = fopen(files[i], "rb");
            unsigned char mg[2] = {0, 0};
            size_t nm = pf? fread(mg, 1, 2, pf) : 0;
            if (pf && !(nm == 2 && mg[0] == 0x1f && mg[1] == 0x8b)) {
                rewind(pf);
                for (;;) {
                    if (cap - len < ((size_t) 1 << 24)) {
                        cap += cap / 2;
                        uint8_t *nb = (uint8_t *) realloc(buf, cap);
                        if (!nb) { fclose(pf); free(buf); return OATK_E_NOMEM; }
                        buf = nb;
                    }
                    const size_t got = fread(buf + len, 1, cap - len - 1, pf);
                    if (got == 0) break;
                    len += got;
                }
                fclose(pf);
                if (len && buf[len - 1] != '\n') buf[len++] = '\n';
                continue;
            }
            if (pf) fclose(pf);
        }
        gzFile fp = gzopen(files[i], "r");
        if (!fp) { fprintf(stderr, "[E::%s] fail to open file \"%s\"\n", __func__, files[i]); free(buf); return OATK_E_ARG; }   /* sstream.c:46-49 */
        (void) gzbuffer(fp, 1 << 20);
        for (;;) {
            if (cap - len < ((size_t) 1 << 24)) {
                cap += cap / 2;
                uint8_t *nb = (uint8_t *) realloc(buf, cap);
                if (!nb) { gzclose(fp); free(buf); return OATK_E_NOMEM; }
                buf = nb;
            }
            const size_t want = cap - len > ((size_t) 1 << 30)? (size_t) 1 << 30 : cap - len;
            const int got = gzread(fp, buf + len, (unsigned) want);
            if (got < 0) { gzclose(fp); free(buf); return OATK_E_ARG; }
            if (got == 0) break;
            len += (size_t) got;
        }
        gzclose(fp);
        /* kseq starts every file at a header line (sstream.c:91-97 opens a fresh kseq): a file that does not end in a newline must not
         * glue its last line to the next file's first header */
        if (len && buf[len - 1] != '\n') buf[len++] = '\n';
    }
    uint64_t used = 0;
    const int rc = oatk_hip_ingest_host(ctx, buf, len, fmt, 1, n_reads, &used);
    if (text && !rc) *text = buf, *text_len = len;
    else free(buf);
    return rc;
}

int oatk_ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads)
{
    return ingest_files(ctx, files, n_files, n_reads, 0, 0, 0);
}

/* sr_read (syncmer.c:487) for files, entirely through the device: text -> records -> scan, then sr_db filled from the resident results.
 * Read names (kseq's name: the header up to the first white space) are cut out of the text here. */
int oatk_sr_read_files(oatk_hip_ctx *ctx, oatk_sr_db_t *sr_db, char **files, int n_files)
{
    uint8_t *text = 0;
    size_t text_len = 0;
    uint64_t n = 0, b = 0, i;
    int mapped = 0;
    int rc = ingest_files(ctx, files, n_files, &n, &text, &text_len, &mapped);
    if (rc) return rc;
    rc = oatk_hip_scan_ingested(ctx, 0, sr_db->k, sr_db->s);
    if (rc) { if (mapped) munmap(text, text_len); else free(text); return rc; }
    uint64_t *off = 0, *hdr = 0;
    char **names = 0;
    if (n) {
        const void *d = 0;
        rc = oatk_hip_buffer(ctx, OATK_BUF_INGEST_OFF, &d, &b);
        if (!rc) { off = (uint64_t *) malloc(b? b : 1); rc = oatk_hip_d2h(ctx, off, d, b); }
        if (!rc) rc = oatk_hip_buffer(ctx, OATK_BUF_INGEST_HDR, &d, &b);
        if (!rc) { hdr = (uint64_t *) malloc(b? b : 1); rc = oatk_hip_d2h(ctx, hdr, d, b); }
        if (!rc) {
            names = (char **) malloc(sizeof(char *) * n);
            for (i = 0; i < n; ++i) {
                size_t p = (size_t) hdr[i] + 1, e = p;                       /* behind '>' / '@' */
                while (e < text_len && text[e] != ' ' && text[e] != '\t' && text[e] != '\n' && text[e] != '\r') ++e;
                names[i] = (char *) malloc(e - p + 1);
                memcpy(names[i], text + p, e - p);
                names[i][e - p] = 0;
            }
            rc = oatk_sr_db_fill_resident(ctx, sr_db, off, n, names);
        }
    }
    if (mapped) munmap(text, text_len); else free(text);
    free(off); free(hdr); free(names);
    return rc;
}
