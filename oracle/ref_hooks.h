/*
 * oracle/ref_hooks.h -- TEST INFRASTRUCTURE ONLY.  Force-included when `make ref_hooked` compiles the reference's syncasm.c: two function
 * pointers that the build recipe (oracle/Makefile, a sed script on the way into gcc -- no patched source is ever written to disk) wires into
 * scg_syncmer_consensus and calc_syncmer_overlap exactly where INTEGRATION.md 3b / 3b' tell a maintainer to put the calls.  NULL hooks
 * leave the reference's behaviour untouched.
 */
#ifndef REFX_HOOKS_H
#define REFX_HOOKS_H
#include <stdint.h>
/* returns the consensus length, or < 0: run the original body */
extern int64_t (*refx_hook_cons)(void *sr_db, void *scm, int rev, int64_t beg, void *c_seq, int hoco_seq);
/* returns the number of distinct distances of m1 -> m2 (first-appearance order, counts, "the walk ended on a repeat"), or < 0: run the original walk */
extern int (*refx_hook_ovl)(void *m1, uint64_t rc1, void *m2, uint64_t rc2, const int32_t **dist, const uint32_t **cnt, int *tail);
#endif
