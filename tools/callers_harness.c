/* Caller threads for tools/bench_latency.py: T native threads, each issuing one-query mmidx_search calls on ONE index
 * (what a JVM's reader threads do through the JNI shim) -- measured without the Python interpreter lock in the way.
 * Built on the fly with gcc; talks to the library through the function pointer it is given. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef int (*search_fn)(void *, int, int64_t, const double *, int32_t *, double *, int32_t *);

struct job {
    search_fn fn;
    void *h;
    int k, D, calls, t, errors;
    const double *Q;
    int64_t nQ;
    pthread_barrier_t *bar;
};

static void *worker(void *p) {
    struct job *j = (struct job *)p;
    int32_t *oi = (int32_t *)malloc((size_t)j->k * 4);
    double *od = (double *)malloc((size_t)j->k * 8);
    int32_t oc = 0;
    pthread_barrier_wait(j->bar);
    for (int i = 0; i < j->calls; i++) {
        const int64_t q = ((int64_t)j->t * j->calls + i) % j->nQ;
        if (j->fn(j->h, j->k, 1, j->Q + q * j->D, oi, od, &oc)) j->errors++;
    }
    pthread_barrier_wait(j->bar);
    free(oi);
    free(od);
    return 0;
}

/* returns the wall time in seconds of threads x calls one-query searches; *errors = failed calls */
double run_callers(search_fn fn, void *h, int k, int D, const double *Q, int64_t nQ, int threads, int calls, int *errors) {
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, (unsigned)threads + 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)calloc((size_t)threads, sizeof(struct job));
    for (int t = 0; t < threads; t++) {
        jobs[t] = (struct job){fn, h, k, D, calls, t, 0, Q, nQ, &bar};
        pthread_create(&th[t], 0, worker, &jobs[t]);
    }
    struct timespec a, b;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &a);
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &b);
    *errors = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], 0);
        *errors += jobs[t].errors;
    }
    free(th);
    free(jobs);
    pthread_barrier_destroy(&bar);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
