/*
 * oatk_amd/csrc/host/ingest_host.c -- files to the device reader (include/oatk_hip_ingest.h), in place of sstream_open / sstream_read
 * (sstream.c:70-103): several files are read one after the other as one stream of records; plain or gzip'ed (zlib's gzread handles
 * both, exactly as the reference's gzdopen does, sstream.c:50).  The host only moves bytes: inflate (serial per stream on any
 * hardware) and one copy of the text to the device, where the records are found.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "oatk_hip_ingest.h"
#include "oatk_syncasm.h"

int oatk_ingest_files(oatk_hip_ctx *ctx, char **files, int n_files, uint64_t *n_reads)
{
    size_t cap = (size_t) 1 << 26, len = 0;
    uint8_t *buf = (uint8_t *) malloc(cap);
    int i, fmt = OATK_FMT_AUTO;
    if (!buf) return OATK_E_NOMEM;
    for (i = 0; i < n_files; ++i) {
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
    free(buf);
    return rc;
}
