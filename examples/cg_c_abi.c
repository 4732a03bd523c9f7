/* Test06CSRSolveCG through the C ABI alone: no Python, no torch, no Legion.
 *
 * Builds the reference's 1-D negative Laplacian (N rows, CSR, rects {lo,hi} inclusive, int64
 * columns; src/ExampleSystems.cpp:382-389, 448-466), b = 1, x0 = 0, and runs the reference's CG
 * sequence (src/CGSolver.hpp:32-55) with one lsk_* call per leaf task; every scalar stays on the
 * device (lsk_dot_* results feed lsk_axpy_dev_* / lsk_xpay_dev_* through the get_alpha rule).
 * Prints the residual-norm-squared history; for N = 100 that is the reference's
 * EXPECTED_CG_OUTPUT (test_all.py:130-133): 100 4900 4704 4512 4324 4140 3960 3784 3612 3444 3280.
 *
 *   gcc -O2 -I include examples/cg_c_abi.c -L legionsolvers_amd -llsk -Wl,-rpath,'$ORIGIN/../legionsolvers_amd' -o examples/cg_c_abi
 *   examples/cg_c_abi [N] [iterations] [trace]
 *
 * With a third argument the steps after the first are ONE recorded step replayed (lsk_stream_begin_capture /
 * lsk_trace_replay: what rt->begin_trace / end_trace around solver.step() are in the reference's benchmark,
 * test/BenchmarkStencil.cpp:219-222): the step works between two fixed scalar slots and ends by moving the
 * new r.r into the old one's place; the history is copied out after every replay.  Same output.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lsk.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != LSK_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, lsk_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100;
    const int iterations = argc > 2 ? atoi(argv[2]) : 10;
    const int traced = argc > 3;
    if (n < 1 || iterations < 0) return 2;
    const int64_t nnz = 3 * n - 2;

    /* host copy of the matrix in the reference's layout */
    int64_t *rects = malloc(sizeof(int64_t) * 2 * n), *col = malloc(sizeof(int64_t) * nnz);
    double *val = malloc(sizeof(double) * nnz);
    int64_t k = 0;
    for (int64_t r = 0; r < n; ++r) {
        rects[2 * r] = k;
        if (r > 0) { col[k] = r - 1; val[k++] = -1.0; }
        col[k] = r; val[k++] = 2.0;
        if (r + 1 < n) { col[k] = r + 1; val[k++] = -1.0; }
        rects[2 * r + 1] = k - 1;
    }

    CHECK(lsk_init(0));
    lsk_stream_t s;
    CHECK(lsk_stream_create(&s));
    void *d_rects, *d_col, *d_val, *d_x, *d_b, *d_p, *d_q, *d_r, *d_scal, *d_work;
    const size_t vec = sizeof(double) * (size_t)n, ws = lsk_dot_workspace_bytes(n);
    CHECK(lsk_malloc(&d_rects, sizeof(int64_t) * 2 * (size_t)n));
    CHECK(lsk_malloc(&d_col, sizeof(int64_t) * (size_t)nnz));
    CHECK(lsk_malloc(&d_val, sizeof(double) * (size_t)nnz));
    CHECK(lsk_malloc(&d_x, vec));
    CHECK(lsk_malloc(&d_b, vec));
    CHECK(lsk_malloc(&d_p, vec));
    CHECK(lsk_malloc(&d_q, vec));
    CHECK(lsk_malloc(&d_r, vec));
    CHECK(lsk_malloc(&d_scal, sizeof(double) * (size_t)(iterations + 5)));
    CHECK(lsk_malloc(&d_work, ws));
    CHECK(lsk_memcpy_h2d(d_rects, rects, sizeof(int64_t) * 2 * (size_t)n, s));
    CHECK(lsk_memcpy_h2d(d_col, col, sizeof(int64_t) * (size_t)nnz, s));
    CHECK(lsk_memcpy_h2d(d_val, val, sizeof(double) * (size_t)nnz, s));

    double *x = d_x, *b = d_b, *p = d_p, *q = d_q, *r = d_r;
    double *rr = d_scal;            /* rr[i]: residual norm squared after i steps */
    double *pq = rr + iterations + 1, *minus_one = pq + 1;
    const double neg = -1.0;
    CHECK(lsk_memcpy_h2d(minus_one, &neg, sizeof(double), s));

    /* CGSolver's constructor: P = R = b, rr[0] = R.R */
    CHECK(lsk_fill_f64(n, 1.0, b, s));
    CHECK(lsk_fill_f64(n, 0.0, x, s));
    CHECK(lsk_copy_f64(n, b, p, s));
    CHECK(lsk_copy_f64(n, b, r, s));
    CHECK(lsk_dot_f64(n, r, r, rr, d_work, ws, s));
    /* one step from the r.r in *old to the r.r in *new_ */
#define CG_STEP(old, new_)                                                                        \
    do {                                                                                          \
        /* Q = A P (beta = 0 replaces the reference's zero fill + accumulate) */                  \
        CHECK(lsk_csr_spmv_f64(n, nnz, d_rects, 0, d_col, d_val, p, 0, n - 1, q, 0, s));          \
        CHECK(lsk_dot_f64(n, p, q, pq, d_work, ws, s));                                           \
        /* x += (rr/pq) P;  R += (-1 * rr / pq) Q;  rr' = R.R;  P = R + (rr'/rr) P */             \
        CHECK(lsk_axpy_dev_f64(n, 2, (old), pq, NULL, NULL, p, x, s));                            \
        CHECK(lsk_axpy_dev_f64(n, 3, minus_one, (old), pq, NULL, q, r, s));                       \
        CHECK(lsk_dot_f64(n, r, r, (new_), d_work, ws, s));                                       \
        CHECK(lsk_xpay_dev_f64(n, 2, (new_), (old), NULL, NULL, r, p, s));                        \
    } while (0)
    if (!traced || iterations < 2) {
        for (int it = 0; it < iterations; ++it) CG_STEP(rr + it, rr + it + 1);
    } else {
        /* the first step as it is (a kernel's first launch on a device is not recordable); then one step
         * between the fixed slots cur and next_, ending with cur <- next_, recorded once and replayed */
        double *cur = minus_one + 1, *next_ = minus_one + 2;
        CG_STEP(rr, rr + 1);
        CHECK(lsk_memcpy_d2d(cur, rr + 1, sizeof(double), s));
        CHECK(lsk_stream_synchronize(s));
        lsk_trace_t *trace = NULL;
        CHECK(lsk_stream_begin_capture(s));
        CG_STEP(cur, next_);
        CHECK(lsk_copy_f64(1, next_, cur, s));
        CHECK(lsk_stream_end_capture(s, &trace));
        for (int it = 1; it < iterations; ++it) {
            CHECK(lsk_trace_replay(trace, s));
            CHECK(lsk_memcpy_d2d(rr + it + 1, cur, sizeof(double), s));
        }
        CHECK(lsk_stream_synchronize(s));
        CHECK(lsk_trace_destroy(trace));
    }
    double *hist = malloc(sizeof(double) * (size_t)(iterations + 1));
    CHECK(lsk_memcpy_d2h(hist, rr, sizeof(double) * (size_t)(iterations + 1), s));
    CHECK(lsk_stream_synchronize(s));
    for (int i = 0; i <= iterations; ++i) printf("%g\n", hist[i]);

    void *all[] = {d_rects, d_col, d_val, d_x, d_b, d_p, d_q, d_r, d_scal, d_work};
    for (unsigned i = 0; i < sizeof(all) / sizeof(all[0]); ++i) CHECK(lsk_free(all[i]));
    CHECK(lsk_stream_destroy(s));
    free(rects); free(col); free(val); free(hist);
    return 0;
}
