/*
 * oatk_amd/csrc/host/par.c -- the few host threads the adaptors use to fill the reference's structs (one malloc + memcpy per member array
 * per read is what sr_destroy's free() calls force, syncmer.c:1047-1058; done by one thread it costs more than the whole device pipeline).
 * The count follows the caller's n_threads (the CLI's -t), as the reference's own pthreads do (syncmer.c:487, syncerr.c:819).
 */
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>

#include "oatk_syncasm.h"

static int g_threads = 0, g_threads_raw = 0;

/* more than 16 threads only get in each other's way here (measured on a 2 x 64-core host: page-cache reads and first-touch page faults stop
 * scaling, and a thread is started per piece), whatever the caller grants */
void oatk_host_set_threads(int n) { g_threads = n > 0? (n > 16? 16 : n) : 0; g_threads_raw = n > 0? n : 0; }

/* the adaptors' own adjustments (half the threads fill structs while the other half reads the file): what the caller granted is kept */
void oatk_host_set_threads_internal(int n) { g_threads = n > 0? (n > 16? 16 : n) : 0; }

/* what the caller granted, uncapped (at most 64): inflating BGZF members scales with the cores, unlike the struct filling above */
int oatk_host_threads_granted(void)
{
    if (g_threads_raw > 0) return g_threads_raw > 64? 64 : g_threads_raw;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1? 1 : (n > 64? 64 : (int) n);
}

int oatk_host_threads(void)
{
    if (g_threads > 0) return g_threads;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1? 1 : (n > 16? 16 : (int) n);
}

typedef struct { oatk_par_fn fn; void *arg; int tid, n; } par_t;

static void *par_entry(void *p)
{
    par_t *t = (par_t *) p;
    t->fn(t->arg, t->tid, t->n);
    return 0;
}

void oatk_par_run(oatk_par_fn fn, void *arg) { oatk_par_run_n(fn, arg, oatk_host_threads()); }

void oatk_par_run_n(oatk_par_fn fn, void *arg, int n)
{
    if (n > 256) n = 256;
    if (n <= 1) { fn(arg, 0, 1); return; }
    pthread_t th[256];
    par_t job[256];
    int i, started = 1;
    for (i = 1; i < n; ++i) {
        job[i].fn = fn, job[i].arg = arg, job[i].tid = started, job[i].n = n;
        if (pthread_create(&th[started], 0, par_entry, &job[i]) != 0) break;      /* fewer threads: the slices below must still cover everything */
        ++started;
    }
    if (started != n) {                     /* could not start them all: join what runs (they were told n), then do the missing slices here */
        int t;
        fn(arg, 0, n);
        for (t = started; t < n; ++t) fn(arg, t, n);
        for (t = 1; t < started; ++t) pthread_join(th[t], 0);
        return;
    }
    fn(arg, 0, n);
    for (i = 1; i < n; ++i) pthread_join(th[i], 0);
}
