/* tests/c/ingest_estimate_test.c -- the room the streamed reader asks for ahead of the reads (oatk_amd/csrc/host/ingest_estimate.h), without a device.
 * Test infrastructure (tests/test_host_ingest_estimate.py builds and runs it). */
#include <assert.h>
#include <stdio.h>
#include "ingest_estimate.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void)
{
    const uint64_t MB = 1 << 20, GB = 1ULL << 30;
    /* a plain file: 768 MB of text from 768 MB of input, 30 GB in all, 51 k reads in the window -> ~2 M reads, a little generously */
    CHECK(oatk_est_trust(0, 768 * MB, 768 * MB));
    uint64_t m = oatk_est_reads(0, 51200, 0, 0, 768 * MB, 30 * GB, 1, 0, 1ULL << 40);
    CHECK(m > 2048000 && m < 2048000 * 1.06 + 2048);
    /* a .gz: 768 MB of text from 200 MB of input */
    CHECK(oatk_est_trust(20, 200 * MB, 768 * MB));
    m = oatk_est_reads(0, 51200, 0, 20, 200 * MB, 925 * MB, 1, 0, 1ULL << 40);
    CHECK(m > 51200 * 4 && m < 51200 * 5);
    /* round 5: the source stood still (twenty bytes of header "taken" for a window of text), or moved by a few bytes: no extrapolation, the array doubles */
    CHECK(!oatk_est_trust(20, 20, 768 * MB));
    CHECK(!oatk_est_trust(20, 40, 768 * MB));
    CHECK(!oatk_est_trust(40, 20, 768 * MB));                                   /* (a position that went backwards) */
    m = oatk_est_reads(0, 51200, 0, 20, 20, 925 * MB, 0, 0, 1ULL << 40);
    CHECK(m == 51200);
    m = oatk_est_reads(51200, 51000, 51200, 20, 20, 925 * MB, 0, 0, 1ULL << 40);
    CHECK(m == 102400);                                                        /* doubling where that is more than what is needed */
    m = oatk_est_reads(51200, 60000, 51200, 20, 20, 925 * MB, 0, 0, 1ULL << 40);
    CHECK(m == 111200);
    /* ... and what that cost, had it been believed: 46 million windows' worth of reads -- cut down to what is needed by the cap on new entries */
    m = oatk_est_reads(0, 51200, 0, 0, 20, 925 * MB, 1, 0, 100000000);
    CHECK(m == 51200);
    m = oatk_est_reads(0, 51200, 0, 0, 20, 925 * MB, 1, 0, 0);                  /* (no cap known: 2.6 x 10^12 reads are beyond anything -- what is needed) */
    CHECK(m == 51200);
    m = oatk_est_reads(0, 51200, 0, 0, 20000, 925 * MB, 1, 0, 0);               /* (no cap known and a number one could believe: the estimate as it is -- the trust test is what guards) */
    CHECK(m > 2000000000ULL && m < (1ULL << 36));
    /* the last window, a capped read: exactly what is needed */
    CHECK(oatk_est_reads(1000, 500, 1000, 0, 10 * MB, 20 * MB, 1, 1, 1ULL << 40) == 1500);
    /* an estimate below what is already here (the file's end is nearer than the position says) never shrinks the array below the need */
    CHECK(oatk_est_reads(1000000, 500, 1000, 900 * MB, 925 * MB, 925 * MB, 1, 0, 1ULL << 40) >= 1000500);
    /* absurd numbers do not overflow into a small array */
    m = oatk_est_reads(0, 1ULL << 30, 0, 0, 1, UINT64_MAX, 1, 0, 0);
    CHECK(m == (1ULL << 30));                                                   /* ... nor into one whose size in bytes wraps */
    /* the device batch's factor: one handle, a quarter of the input taken -> ~4; eight handles -> the share, plus a piece */
    double s = oatk_est_scale(0, 250 * MB, 1000 * MB, 1);
    CHECK(s > 4.0 && s < 4.2);
    s = oatk_est_scale(0, 25 * MB, 1000 * MB, 8);
    CHECK(s > 6.0 && s < 6.3);
    s = oatk_est_scale(900 * MB, 925 * MB, 1000 * MB, 8);                       /* the last handle: what is left, not its share */
    CHECK(s > 5.0 && s < 5.2);
    printf("ok\n");
    return 0;
}
