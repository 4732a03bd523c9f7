/*
 * oracle.c -- CPU restatement of the reference's Krylov inner loop.  TEST INFRASTRUCTURE:
 * it is the checker the HIP path is compared against (tests/, smoke) and the CPU baseline
 * bench.py times; nothing the product ships calls it.
 *
 * Every function follows one body of the reference (path:line relative to the reference
 * checkout) with Legion's accessors replaced by plain pointers.  Loops are sequential and in
 * the reference's order so that floating-point results are the reference CPU path's, bit for
 * bit; compile with -ffp-contract=off so that `a += b * c` stays a multiply and an add.
 *
 * The functions ending in _mt are the same arithmetic split over row blocks with OpenMP
 * ("one worker per shard"): used only for CPU-baseline timing.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>

/* Worker threads of the _mt functions (CPU-baseline timing). */
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oracle_get_max_threads(void) { return omp_get_max_threads(); }

/* ---- get_alpha: src/LegionUtilities.cpp:72-97 ---------------------------------------- */
#define GET_ALPHA(SUF, T)                                                     \
    int oracle_get_alpha_##SUF(int nf, const T *f, T *alpha) {                \
        switch (nf) {                                                         \
        case 0: *alpha = (T)1; return 0;                                      \
        case 1: *alpha = f[0]; return 0;                                      \
        case 2: *alpha = f[0] / f[1]; return 0;                               \
        case 3: *alpha = (f[0] * f[1]) / f[2]; return 0;                      \
        case 4: *alpha = (f[0] * f[1]) / (f[2] * f[3]); return 0;             \
        default: return -1; /* assert(false) in the reference */              \
        }                                                                     \
    }
GET_ALPHA(f32, float)
GET_ALPHA(f64, double)

static inline float fma_f32(float a, float b, float c) { return fmaf(a, b, c); }
static inline double fma_f64(double a, double b, double c) { return fma(a, b, c); }

/* ---- BLAS-1: src/LinearAlgebraTasks.cpp ----------------------------------------------- */
#define BLAS1(SUF, T)                                                                  \
    /* ScalTask::task_body :37-43  x[p] = alpha * x[p] */                              \
    void oracle_scal_##SUF(int64_t n, T alpha, T *x) {                                 \
        for (int64_t i = 0; i < n; ++i) x[i] = alpha * x[i];                           \
    }                                                                                  \
    /* AxpyTask::task_body :80-87  y[p] = fma(alpha, x[p], y[p]) */                    \
    void oracle_axpy_##SUF(int64_t n, T alpha, const T *x, T *y) {                     \
        for (int64_t i = 0; i < n; ++i) y[i] = fma_##SUF(alpha, x[i], y[i]);           \
    }                                                                                  \
    /* XpayTask::task_body :124-131  y[p] = fma(alpha, y[p], x[p]) */                  \
    void oracle_xpay_##SUF(int64_t n, T alpha, const T *x, T *y) {                     \
        for (int64_t i = 0; i < n; ++i) y[i] = fma_##SUF(alpha, y[i], x[i]);           \
    }                                                                                  \
    /* DotTask::task_body :166-174  result += v[p] * w[p], left to right from 0 */     \
    T oracle_dot_##SUF(int64_t n, const T *v, const T *w) {                            \
        T result = (T)0;                                                               \
        for (int64_t i = 0; i < n; ++i) result += v[i] * w[i];                         \
        return result;                                                                 \
    }                                                                                  \
    /* xpay over an N-D dense rect through strided accessors, as xpay_kernel's         \
     * Pitches::unflatten + AffineAccessor do (src/LinearAlgebraTasks.cu:118-131). */  \
    void oracle_xpay_nd_##SUF(int dim, const int64_t *extent, T alpha, const T *x,     \
                              const int64_t *xs, T *y, const int64_t *ys) {            \
        int64_t e[3] = {1, 1, 1}, sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0};                \
        for (int d = 0; d < dim; ++d) {                                                \
            e[3 - dim + d] = extent[d];                                                \
            sx[3 - dim + d] = xs[d];                                                   \
            sy[3 - dim + d] = ys[d];                                                   \
        }                                                                              \
        for (int64_t a = 0; a < e[0]; ++a)                                             \
            for (int64_t b = 0; b < e[1]; ++b)                                         \
                for (int64_t c = 0; c < e[2]; ++c) {                                   \
                    const int64_t xo = a * sx[0] + b * sx[1] + c * sx[2];              \
                    const int64_t yo = a * sy[0] + b * sy[1] + c * sy[2];              \
                    y[yo] = fma_##SUF(alpha, y[yo], x[xo]);                            \
                }                                                                      \
    }
BLAS1(f32, float)
BLAS1(f64, double)

/* ---- scalar tasks: src/UtilityTasks.cpp:34-99 ------------------------------------------ */
#define SCALAR_OPS(SUF, T, SQRT)                                                       \
    T oracle_scalar_op_##SUF(int op, T a, T b) {                                       \
        switch (op) {                                                                  \
        case 0: return -a;               /* NegateScalarTask   :35-39 */               \
        case 1: return a + b;            /* AddScalarTask      :43-48 */               \
        case 2: return a - b;            /* SubtractScalarTask :52-57 */               \
        case 3: return a * b;            /* MultiplyScalarTask :61-66 */               \
        case 4: return a / b;            /* DivideScalarTask   :70-75 */               \
        case 5: return SQRT(a);          /* SqrtScalarTask     :79-84 */               \
        case 6: return (T)1 / SQRT(a);   /* RSqrtScalarTask    :88-93 */               \
        default: return a;                                                             \
        }                                                                              \
    }
SCALAR_OPS(f32, float, sqrtf)
SCALAR_OPS(f64, double, sqrt)

/* ---- matvecs ---------------------------------------------------------------------------- */
#define MATVEC(SUF, T)                                                                 \
    /* COOMatvecTask::task_body, src/COOMatrixTasks.cpp:67-74:                         \
     * for k ascending: if col in x-domain and row in y-domain: y[row] += val*x[col].  \
     * x[0] is global column x_lo, y[0] global row y_lo. */                            \
    void oracle_coo_matvec_##SUF(int64_t nnz, const int64_t *row, const int64_t *col,  \
                                 const T *val, const T *x, int64_t x_lo, int64_t x_hi, \
                                 T *y, int64_t y_lo, int64_t y_hi) {                   \
        for (int64_t k = 0; k < nnz; ++k) {                                            \
            const int64_t r = row[k], c = col[k];                                      \
            if (c >= x_lo && c <= x_hi && r >= y_lo && r <= y_hi)                      \
                y[r - y_lo] += val[k] * x[c - x_lo];                                   \
        }                                                                              \
    }                                                                                  \
    /* CSRMatvecTask::task_body, src/CSRMatrixTasks.cpp:73-91, literally: for every    \
     * kernel point k (ascending) search all row rects for the one containing k (the   \
     * LAST match wins, as `row = cur_row` is overwritten), then accumulate.           \
     * O(nnz*rows): small sizes only. */                                               \
    void oracle_csr_matvec_literal_##SUF(int64_t rows, int64_t nnz, const int64_t *rects, \
                                         int64_t k_base, const int64_t *col, const T *val, \
                                         const T *x, int64_t x_lo, int64_t x_hi, T *y) { \
        for (int64_t kk = 0; kk < nnz; ++kk) {                                         \
            const int64_t k = k_base + kk;                                             \
            int64_t found = -1;                                                        \
            for (int64_t r = 0; r < rows; ++r)                                         \
                if (rects[2 * r] <= k && k <= rects[2 * r + 1]) found = r;             \
            if (found < 0) continue; /* assert(row_found) in the reference */          \
            const int64_t c = col[kk];                                                 \
            if (c >= x_lo && c <= x_hi) y[found] += val[kk] * x[c - x_lo];             \
        }                                                                              \
    }                                                                                  \
    /* The same sums row by row in O(nnz): for non-overlapping rects every y[r]        \
     * receives the same products in the same (ascending k) order as the literal       \
     * loop, so results are identical.  beta = 0 starts each row from 0 (what the      \
     * reference gets after SquarePlanner::matvec's zero_fill, SquarePlanner.hpp:341), \
     * beta = 1 accumulates onto y. */                                                 \
    void oracle_csr_matvec_##SUF(int64_t rows, int64_t nnz, const int64_t *rects,      \
                                 int64_t k_base, const int64_t *col, const T *val,     \
                                 const T *x, int64_t x_lo, int64_t x_hi, T *y, int beta) { \
        for (int64_t r = 0; r < rows; ++r) {                                           \
            T acc = beta ? y[r] : (T)0;                                                \
            int64_t lo = rects[2 * r], hi = rects[2 * r + 1];                          \
            if (lo < k_base) lo = k_base;                                              \
            if (hi > k_base + nnz - 1) hi = k_base + nnz - 1;                          \
            for (int64_t k = lo; k <= hi; ++k) {                                       \
                const int64_t c = col[k - k_base];                                     \
                if (c >= x_lo && c <= x_hi) acc += val[k - k_base] * x[c - x_lo];      \
            }                                                                          \
            y[r] = acc;                                                                \
        }                                                                              \
    }                                                                                  \
    void oracle_csr_matvec_mt_##SUF(int64_t rows, int64_t nnz, const int64_t *rects,   \
                                    int64_t k_base, const int64_t *col, const T *val,  \
                                    const T *x, int64_t x_lo, int64_t x_hi, T *y, int beta) { \
        _Pragma("omp parallel for schedule(static)")                                   \
        for (int64_t r = 0; r < rows; ++r) {                                           \
            T acc = beta ? y[r] : (T)0;                                                \
            int64_t lo = rects[2 * r], hi = rects[2 * r + 1];                          \
            if (lo < k_base) lo = k_base;                                              \
            if (hi > k_base + nnz - 1) hi = k_base + nnz - 1;                          \
            for (int64_t k = lo; k <= hi; ++k) {                                       \
                const int64_t c = col[k - k_base];                                     \
                if (c >= x_lo && c <= x_hi) acc += val[k - k_base] * x[c - x_lo];      \
            }                                                                          \
            y[r] = acc;                                                                \
        }                                                                              \
    }                                                                                  \
    /* One CG step of CGSolver::step (src/CGSolver.hpp:46-55) on one system with the   \
     * vector ops split into `shards` row blocks, each block's dot summed left to right \
     * and the block sums folded in block order (Legion's REDOP_SUM over the colour    \
     * space).  rr_old in, returns rr_new.  Timing harness for the CPU baseline. */    \
    T oracle_cg_step_mt_##SUF(int64_t rows, int64_t nnz, const int64_t *rects,         \
                              const int64_t *col, const T *val, T *sol, T *p, T *q,    \
                              T *r, T rr_old, int shards, T *partial) {                \
        oracle_csr_matvec_mt_##SUF(rows, nnz, rects, 0, col, val, p, 0, rows - 1, q, 0); \
        _Pragma("omp parallel for schedule(static)")                                   \
        for (int s = 0; s < shards; ++s) {                                             \
            const int64_t b = rows * s / shards, e = rows * (s + 1) / shards;          \
            partial[s] = oracle_dot_##SUF(e - b, p + b, q + b);                        \
        }                                                                              \
        T pq = partial[0];                                                             \
        for (int s = 1; s < shards; ++s) pq = pq + partial[s];                         \
        const T f2[2] = {rr_old, pq};                                                  \
        T alpha, nalpha;                                                               \
        oracle_get_alpha_##SUF(2, f2, &alpha);                                         \
        const T f3[3] = {(T)-1, rr_old, pq};                                           \
        oracle_get_alpha_##SUF(3, f3, &nalpha);                                        \
        _Pragma("omp parallel for schedule(static)")                                   \
        for (int s = 0; s < shards; ++s) {                                             \
            const int64_t b = rows * s / shards, e = rows * (s + 1) / shards;          \
            oracle_axpy_##SUF(e - b, alpha, p + b, sol + b);                           \
            oracle_axpy_##SUF(e - b, nalpha, q + b, r + b);                            \
            partial[s] = oracle_dot_##SUF(e - b, r + b, r + b);                        \
        }                                                                              \
        T rr = partial[0];                                                             \
        for (int s = 1; s < shards; ++s) rr = rr + partial[s];                         \
        const T g2[2] = {rr, rr_old};                                                  \
        T beta;                                                                        \
        oracle_get_alpha_##SUF(2, g2, &beta);                                          \
        _Pragma("omp parallel for schedule(static)")                                   \
        for (int s = 0; s < shards; ++s) {                                             \
            const int64_t b = rows * s / shards, e = rows * (s + 1) / shards;          \
            oracle_xpay_##SUF(e - b, beta, r + b, p + b);                              \
        }                                                                              \
        return rr;                                                                     \
    }
MATVEC(f32, float)
MATVEC(f64, double)

/* ---- convertGlobalRowptrToLocalIndPtr: src/CuSPARSEHelpers.hpp:20-30 ------------------- */
void oracle_rects_to_indptr(int64_t rows, const int64_t *rects, int64_t *indptr) {
    for (int64_t i = 0; i < rows; ++i) indptr[i] = rects[2 * i] - rects[0];
    if (rows > 0) indptr[rows] = rects[2 * (rows - 1) + 1] + 1 - rects[0];
}

/* ---- stencil generators: src/StencilGenerator.cpp:129-242, 379-517 ---------------------
 * Row-major walk over the grid [0,extent) (bounds.lo = 0 as in every caller), offsets
 * already sorted with compare_row_major (src/StencilGenerator.hpp:174-183); an entry is
 * written where point+offset stays in bounds; kernel_index counts them; row rect =
 * {row_begin, kernel_index-1} (:494-511).  The reference's bulk-slice fast-forward
 * (:471-492) only skips work outside the piece and does not change what is written.
 * Pass col == NULL to count only.  Returns the number of entries. */
static int64_t stencil_walk(int dim, const int64_t *extent, int noff, const int64_t *off,
                            int64_t *row, int64_t *col, int64_t *rects, int *which) {
    int64_t e[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) e[3 - dim + d] = extent[d];
    int64_t k = 0;
    for (int64_t a = 0; a < e[0]; ++a)
        for (int64_t b = 0; b < e[1]; ++b)
            for (int64_t c = 0; c < e[2]; ++c) {
                const int64_t p[3] = {a, b, c};
                const int64_t lin = (a * e[1] + b) * e[2] + c; /* linearize_row_major */
                const int64_t row_begin = k;
                for (int o = 0; o < noff; ++o) {
                    int64_t s[3] = {a, b, c};
                    int in = 1;
                    for (int d = 0; d < dim; ++d) {
                        const int dd = 3 - dim + d;
                        s[dd] = p[dd] + off[o * dim + d];
                        if (s[dd] < 0 || s[dd] >= e[dd]) in = 0;
                    }
                    if (in) {
                        if (col) {
                            col[k] = (s[0] * e[1] + s[1]) * e[2] + s[2];
                            which[k] = o;
                            if (row) row[k] = lin;
                        }
                        ++k;
                    }
                }
                if (rects) {
                    rects[2 * lin] = row_begin;
                    rects[2 * lin + 1] = k - 1;
                }
            }
    return k;
}

int64_t oracle_stencil_size(int dim, const int64_t *extent, int noff, const int64_t *off) {
    return stencil_walk(dim, extent, noff, off, NULL, NULL, NULL, NULL);
}

/* which[k] = index of the (sorted) offset that produced entry k: the caller maps it to the
 * coefficient in the entry type it wants. */
int64_t oracle_stencil_fill(int dim, const int64_t *extent, int noff, const int64_t *off,
                            int64_t *row, int64_t *col, int64_t *rects, int *which) {
    return stencil_walk(dim, extent, noff, off, row, col, rects, which);
}

/* ---- task ids: src/TaskBaseClasses.hpp -------------------------------------------------- */
enum {
    ORIGIN = 500000, /* LibraryOptions.hpp:25 */
    NUM_META = 1,    /* TaskIDs.hpp:9-14 */
    NUM_ENTRY = 3,   /* TypeList<float,double,void>, TaskBaseClasses.hpp:43-54 */
    NUM_INDEX = 4,   /* TypeList<int,unsigned,long long,void>, :17-38 */
    MAX_DIM = 3      /* LibraryOptions.hpp:45 */
};
#define IDX0 1
#define IDX1 (NUM_INDEX * IDX0)
#define IDX2 (NUM_INDEX * IDX1)
#define IDX3 (NUM_INDEX * IDX2)
#define DIM0 1
#define DIM1 (MAX_DIM * DIM0)
#define DIM2 (MAX_DIM * DIM1)
#define DIM3 (MAX_DIM * DIM2)
#define BLOCK_SIZE (NUM_ENTRY * DIM3 * IDX3) /* :88-90 */

int64_t oracle_task_id_t(int block, int t) { /* :100-103 */
    return ORIGIN + NUM_META + (int64_t)BLOCK_SIZE * block + t;
}
int64_t oracle_task_id_di(int block, int n, int i) { /* :161-165 */
    return ORIGIN + NUM_META + (int64_t)BLOCK_SIZE * block + MAX_DIM * i + (n - 1);
}
int64_t oracle_task_id_tdi(int block, int t, int n, int i) { /* :236-242 */
    return ORIGIN + NUM_META + (int64_t)BLOCK_SIZE * block + NUM_ENTRY * MAX_DIM * i +
           NUM_ENTRY * (n - 1) + t;
}
int64_t oracle_task_id_tdddiii(int block, int t, int n1, int n2, int n3, int i1, int i2,
                               int i3) { /* :330-350 */
    return ORIGIN + NUM_META + (int64_t)BLOCK_SIZE * block +
           (int64_t)IDX3 * NUM_ENTRY * DIM2 * (n1 - 1) + (int64_t)IDX3 * NUM_ENTRY * DIM1 * (n2 - 1) +
           (int64_t)IDX3 * NUM_ENTRY * DIM0 * (n3 - 1) + (int64_t)IDX2 * NUM_ENTRY * i1 +
           (int64_t)IDX1 * NUM_ENTRY * i2 + (int64_t)IDX0 * NUM_ENTRY * i3 + t;
}
