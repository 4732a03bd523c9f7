/*
 * lsk.h -- C ABI of the MI355X-native Krylov inner loop (SpMV + BLAS-1) for LegionSolvers.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an `int`
 * status (0 = LSK_OK) and enqueues its work on the HIP stream passed in; nothing here
 * synchronises the host unless its name ends in `_host`.  Device pointers are raw HBM
 * addresses (hipMalloc, a torch tensor's data_ptr, or a Legion/Realm GPU_FB_MEM instance).
 *
 * The entry points are exactly what the GPU (TOC_PROC) variants of the reference's leaf
 * tasks bind; each comment cites the reference interface it replaces (paths are relative to
 * the reference checkout).
 *
 * Index conventions follow the reference: kernel indices `k`, column indices and row
 * indices are GLOBAL 64-bit coordinates; a shard is handed pointers to the first element
 * of its piece plus the global coordinate of that element.
 */
#ifndef LSK_H_INCLUDED
#define LSK_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes ---- */
#define LSK_OK 0
#define LSK_ERR_HIP (-1)       /* a HIP runtime call failed (see lsk_last_error)        */
#define LSK_ERR_ARG (-2)       /* invalid argument (null pointer, negative size, ...)    */
#define LSK_ERR_ALIGN (-3)     /* a pointer does not meet the documented alignment       */
#define LSK_ERR_UNSUPPORTED (-4)
#define LSK_ERR_COMM (-5)      /* an RCCL call failed                                    */
#define LSK_ERR_NO_DEVICE (-6) /* no HIP device visible                                  */

/* hipStream_t, passed opaquely so that this header needs no HIP include. */
typedef void *lsk_stream_t;

/* ------------------------------------------------------------- shim (runtime) ------- */
/* Replaces src/CUDAUtilities.{hpp,cpp} and src/CudaLibs.{hpp,cu}.                       */

/* Library version, "lsk <major>.<minor> gfx950". */
const char *lsk_version(void);

/* Message of the last failing call on this host thread ("" if none).
 * Replaces the fprintf in check_cuda/check_cublas/check_cusparse
 * (src/CUDAUtilities.cpp:13-63). */
const char *lsk_last_error(void);

/* If nonzero, a failing call prints its message to stderr and abort()s, which is the
 * reference's error behaviour (print + assert(false), src/CUDAUtilities.cpp:13-27).
 * Default 0: print nothing, return the status code. */
void lsk_set_abort_on_error(int enable);

/* Kernel-selection options, for A/B timing of kernel forms; every setting computes the same
 * result.  Keys: "csr.variant" (0 workgroup tiles | 1 wave tiles, default 1),
 * "csr.nontemporal" (0 | 1: stream col/val past the caches, default 0), "xcd_remap" (c > 0:
 * each XCD takes runs of c consecutive tiles; c < 0: one contiguous eighth of the tiles per
 * XCD; 0: workgroup id = tile id, the default; kept PER HOST THREAD, so that the value a thread
 * sets before its launch, or a tuner flips while it times candidates, never reaches launches made
 * from another thread), "csr.log_rows_per_wave" (-1 = from
 * the mean row length, else 0..8), "csr.max_units" (16-byte units per lane the wave-tile form
 * keeps in flight: 0 = chosen from the mean tile size, 4 or 8), "blas1.form" (0 grid-strided | 1 contiguous chunk per
 * workgroup, default 1), "blas1.nontemporal" (whether the BLAS-1 streams go past the caches: 1 | 0 |
 * -1 = for vectors of 128 MiB and more, the default), "coo.variant" (0 workgroup chunks through LDS | 1 wave-wide segmented
 * scan, default 1), "comm.skip_single_rank" (1, the default: lsk_allreduce_sum_* / lsk_allgather_* on a
 * one-rank communicator answer without calling RCCL | 0: they enqueue the RCCL collective all the
 * same), "sliced.form" (sliced-pattern product: 1, the default: every wave stages the windows of x of its
 * own rows and the block meets only to write y | 0 block-staged windows, one barrier before the sums),
 * "sliced.dma" (1, the default: the wave-staged form sends the windows from memory straight into LDS | 0
 * through registers), "sliced.runs" (1, the default: it reads lists of runs of three run by run | 0 entry
 * by entry), "sliced.pair_form" (lsk_csr_sliced_spmv2_*: 1 wave-staged, one vector after the other | 0
 * both vectors' windows in LDS at once | -1, the default: 1 where that form serves the structure), "debug.sync" (0, the
 * default | 1, also LSK_DEBUG_SYNC=1 in the environment: every launch is followed by a synchronisation of the device
 * and an error check, so an asynchronous fault is reported by the call that caused it: CHECK_CUDA_STREAM of the
 * reference's debug builds, src/CUDAUtilities.cpp:110-122; not while a stream records).  No counterpart in the reference
 * (cuSPARSE picks its own algorithm behind CUSPARSE_SPMV_ALG_DEFAULT).  All other options
 * are process-wide and read at launch time without synchronisation: set them between launches. */
int lsk_set_option(const char *key, int value);
int lsk_get_option(const char *key, int *value);

/* Number of visible HIP devices. */
int lsk_device_count(int *count);

/* Make `device` current and create its per-device library state (reduction scratch).
 * Idempotent.  Replaces LoadCUDALibsTask::task_body (src/CudaLibs.cu:11-20), which warms
 * the stream and the cuBLAS/cuSPARSE handles of one GPU; there are no handles any more. */
int lsk_init(int device);

/* hipMalloc / hipFree wrappers: replace Legion::DeferredBuffer(..., GPU_FB_MEM, nullptr, 16)
 * for callers without Legion (src/CSRMatrixTasks.cu:133-137, src/CuSPARSEHelpers.hpp:73-75).
 * Returned memory is at least 256-byte aligned. */
int lsk_malloc(void **ptr, size_t bytes);
int lsk_free(void *ptr);
/* Free and total bytes of the current device's memory (hipMemGetInfo): what a caller needs to decide whether a
 * set-up-time copy, or the blocks by which it steers where its vectors land in HBM, still fit.  The reference
 * leaves that to Legion's memory sizes (-ll:fsize, bench_all_ls.py:171-181). */
int lsk_mem_info(size_t *free_bytes, size_t *total_bytes);

/* Stream helpers: replace CUDALibraryContext::get_cuda_stream / CUDAStreamView
 * (src/CUDAUtilities.cpp:66-75,110-122).  Streams are created at default priority. */
int lsk_stream_create(lsk_stream_t *stream);
int lsk_stream_destroy(lsk_stream_t stream);
int lsk_stream_synchronize(lsk_stream_t stream);
/* Synchronise the stream and report any pending asynchronous error: CHECK_CUDA_STREAM
 * (src/CUDAUtilities.hpp:25-29), which the reference runs at task exit in debug builds. */
int lsk_stream_check(lsk_stream_t stream);

/* Traces.  The reference's benchmark wraps every `-pt` solver steps in a Legion trace
 * (rt->begin_trace / rt->end_trace, test/BenchmarkStencil.cpp:219-222,237-240): the first pass records the
 * task launches, later passes replay them without the analysis.  Here the counterpart is a HIP graph:
 * between lsk_stream_begin_capture and lsk_stream_end_capture the leaf calls on `stream` are recorded,
 * NOT executed; lsk_trace_replay enqueues the whole recording on a stream with one call.  What may be
 * recorded: every leaf that only enqueues kernels (BLAS-1 with device-resident futures, the scalar
 * tasks, lsk_sum_partials_*, the products; lsk_memcpy_d2d) once it has run at least once on the device
 * outside a recording; not the calls that read a result on the host, allocate, synchronise or time
 * (lsk_dot_host_*, create / update / tune / info calls, lsk_malloc, lsk_stream_synchronize).  A HIP
 * call that fails inside a recording (an allocation, say) spoils it: lsk_stream_end_capture then returns
 * an error and no trace, and the stream may stay unusable: destroy it (a call that this library refuses
 * for its arguments enqueued nothing and spoils nothing).  The recording refers to the device memory its calls were given; the caller keeps that
 * memory, and what it holds between replays is the caller's business (a solver step that reads a
 * scalar of the step before needs it where the recorded kernels look for it).  One thread records
 * on one stream at a time; other threads' launches on other streams go on meanwhile. */
typedef struct lsk_trace lsk_trace_t;
int lsk_stream_begin_capture(lsk_stream_t stream);
int lsk_stream_end_capture(lsk_stream_t stream, lsk_trace_t **trace);
/* *recording = 1 while `stream` is between the two calls above (0 for the default stream).  The calls of this
 * library that wait for the device (lsk_stream_synchronize, lsk_stream_check, lsk_dot_host_*, the tune calls)
 * ask the same question and answer LSK_ERR_UNSUPPORTED before enqueueing anything, which leaves the recording
 * whole; a caller that times or reads results lazily (a placement timed on the first product) asks here first.
 * No counterpart in the reference: Legion itself knows when it is recording a trace. */
int lsk_stream_is_recording(lsk_stream_t stream, int *recording);
int lsk_trace_replay(const lsk_trace_t *trace, lsk_stream_t stream);
/* out2[0] = kernel and copy nodes of the recording, out2[1] = replays so far. */
int lsk_trace_info(const lsk_trace_t *trace, int64_t *out2);
int lsk_trace_destroy(lsk_trace_t *trace);

int lsk_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, lsk_stream_t stream);
int lsk_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, lsk_stream_t stream);
int lsk_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, lsk_stream_t stream);

/* ------------------------------------------------------------- get_alpha ------------ */
/* alpha from 0..4 resolved futures, in the reference's exact operation order
 * (src/LegionUtilities.cpp:72-97): 0 -> 1; 1 -> f0; 2 -> f0/f1; 3 -> (f0*f1)/f2;
 * 4 -> (f0*f1)/(f2*f3).  Host function. nf outside 0..4 returns LSK_ERR_ARG. */
int lsk_get_alpha_f32(int nf, const float *f, float *alpha);
int lsk_get_alpha_f64(int nf, const double *f, double *alpha);

/* ------------------------------------------------------------- BLAS-1, host alpha --- */
/* x <- alpha * x.            ScalTask::cuda_task_body (src/LinearAlgebraTasks.cu:14-56),
 *                            CPU semantics src/LinearAlgebraTasks.cpp:37-43.            */
int lsk_scal_f32(int64_t n, float alpha, float *x, lsk_stream_t stream);
int lsk_scal_f64(int64_t n, double alpha, double *x, lsk_stream_t stream);

/* y <- fma(alpha, x, y).     AxpyTask::cuda_task_body (src/LinearAlgebraTasks.cu:59-113),
 *                            CPU semantics src/LinearAlgebraTasks.cpp:80-87.            */
int lsk_axpy_f32(int64_t n, float alpha, const float *x, float *y, lsk_stream_t stream);
int lsk_axpy_f64(int64_t n, double alpha, const double *x, double *y, lsk_stream_t stream);

/* y <- fma(alpha, y, x).     xpay_kernel + XpayTask::cuda_task_body
 *                            (src/LinearAlgebraTasks.cu:118-176), CPU semantics
 *                            src/LinearAlgebraTasks.cpp:124-131.                        */
int lsk_xpay_f32(int64_t n, float alpha, const float *x, float *y, lsk_stream_t stream);
int lsk_xpay_f64(int64_t n, double alpha, const double *x, double *y, lsk_stream_t stream);

/* N-dimensional xpay over a dense rect with per-array element strides, the case the
 * reference's xpay_kernel handles through Pitches::unflatten + affine accessors
 * (src/Pitches.hpp:27-113, src/LinearAlgebraTasks.cu:118-131).  dim in 1..3;
 * extent[d] points along dimension d; x_stride/y_stride in elements. */
int lsk_xpay_nd_f32(int dim, const int64_t *extent, float alpha, const float *x,
                    const int64_t *x_stride, float *y, const int64_t *y_stride,
                    lsk_stream_t stream);
int lsk_xpay_nd_f64(int dim, const int64_t *extent, double alpha, const double *x,
                    const int64_t *x_stride, double *y, const int64_t *y_stride,
                    lsk_stream_t stream);

/* x[i] <- value / dst <- src: the IndexFillLauncher / IndexCopyLauncher the reference's
 * PartitionedVector issues (src/PartitionedVector.cpp:151-191). */
int lsk_fill_f32(int64_t n, float value, float *x, lsk_stream_t stream);
int lsk_fill_f64(int64_t n, double value, double *x, lsk_stream_t stream);
int lsk_copy_f32(int64_t n, const float *src, float *dst, lsk_stream_t stream);
int lsk_copy_f64(int64_t n, const double *src, double *dst, lsk_stream_t stream);

/* ------------------------------------------------------------- dot / nrm2 ----------- */
/* Bytes of device workspace lsk_dot_* / lsk_nrm2_* need for a vector of n elements. */
size_t lsk_dot_workspace_bytes(int64_t n);

/* *result_dev <- sum_i v[i]*w[i], left on the device; deterministic (fixed partial tree).
 * DotTask::cuda_task_body (src/LinearAlgebraTasks.cu:179-238) minus its stream sync.
 * `workspace` may be NULL: then the per-device scratch made by lsk_init is used, and
 * calls that overlap on one GPU (different streams) would race on it.
 * n == 0 writes 0 (the reference returns 0 for an empty domain, .cu:213-214). */
int lsk_dot_f32(int64_t n, const float *v, const float *w, float *result_dev,
                void *workspace, size_t workspace_bytes, lsk_stream_t stream);
int lsk_dot_f64(int64_t n, const double *v, const double *w, double *result_dev,
                void *workspace, size_t workspace_bytes, lsk_stream_t stream);

/* Same, returning the value to the host after synchronising the stream: the reference's
 * exact task-body contract (returns ENTRY_T by value, src/LinearAlgebraTasks.cu:222-237). */
int lsk_dot_host_f32(int64_t n, const float *v, const float *w, float *result_host,
                     lsk_stream_t stream);
int lsk_dot_host_f64(int64_t n, const double *v, const double *w, double *result_host,
                     lsk_stream_t stream);

/* lsk_dot_f64 for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  Same kernels, same order of
 * additions and so the same bits as lsk_dot_f64; the partial sums live in a static device array
 * inside the library (32 KiB per GPU, held until the process ends), whose address on the current
 * device is looked up once (lsk_init does it; otherwise the first call on that device).  Calls no
 * hipMalloc / hipFree and no allocating routine.  n above LSK_DOT_NOALLOC_MAX_N is refused with
 * LSK_ERR_ARG.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU
 * (different streams or host threads) race; calls on one stream do not. */
#define LSK_DOT_NOALLOC_MAX_N (INT64_C(1) << 34)
int lsk_dot_noalloc_f64(int64_t n, const double *v, const double *w, double *result_dev,
                        lsk_stream_t stream);

/* *result_dev <- sqrt(sum_i v[i]^2) in one pass over v.  The reference has no nrm2 task:
 * its solvers compute dot(v,v) followed by SqrtScalarTask (src/GMRESSolver.hpp:101-103,
 * src/UtilityTasks.cpp:78-84); this fuses the two. */
int lsk_nrm2_f32(int64_t n, const float *v, float *result_dev, void *workspace,
                 size_t workspace_bytes, lsk_stream_t stream);
int lsk_nrm2_f64(int64_t n, const double *v, double *result_dev, void *workspace,
                 size_t workspace_bytes, lsk_stream_t stream);

/* ------------------------------------------------- device-resident scalars ---------- */
/* A Legion::Future<ENTRY_T> becomes one ENTRY_T in HBM written by a stream-ordered kernel,
 * so a dot result can feed the next axpy without a host round trip (the reference blocks
 * in Future::get_result inside get_alpha, src/LegionUtilities.cpp:77-92). */

/* Scalar task bodies (src/UtilityTasks.cpp:34-99) as one-thread kernels. */
#define LSK_SOP_NEG 0   /* out = -a        NegateScalarTask   */
#define LSK_SOP_ADD 1   /* out = a + b     AddScalarTask      */
#define LSK_SOP_SUB 2   /* out = a - b     SubtractScalarTask */
#define LSK_SOP_MUL 3   /* out = a * b     MultiplyScalarTask */
#define LSK_SOP_DIV 4   /* out = a / b     DivideScalarTask   */
#define LSK_SOP_SQRT 5  /* out = sqrt(a)   SqrtScalarTask     */
#define LSK_SOP_RSQRT 6 /* out = 1/sqrt(a) RSqrtScalarTask    */
#define LSK_SOP_COPY 7  /* out = a                            */
int lsk_scalar_op_f32(int op, const float *a, const float *b, float *out, lsk_stream_t stream);
int lsk_scalar_op_f64(int op, const double *a, const double *b, double *out, lsk_stream_t stream);

/* out <- in[0] + in[1] + ... + in[count-1], added left to right: folds the per-piece dot
 * results of one index launch as Legion's REDOP_SUM future-map reduction does
 * (src/PartitionedVector.cpp:353-357). */
int lsk_sum_scalars_f32(int count, const float *in, float *out, lsk_stream_t stream);
int lsk_sum_scalars_f64(int count, const double *in, double *out, lsk_stream_t stream);

/* BLAS-1 with alpha = get_alpha(nf futures) evaluated on the device from f0..f3
 * (device pointers; unused ones NULL), same rule and operation order as lsk_get_alpha_*. */
int lsk_scal_dev_f32(int64_t n, int nf, const float *f0, const float *f1, const float *f2,
                     const float *f3, float *x, lsk_stream_t stream);
int lsk_scal_dev_f64(int64_t n, int nf, const double *f0, const double *f1, const double *f2,
                     const double *f3, double *x, lsk_stream_t stream);
int lsk_axpy_dev_f32(int64_t n, int nf, const float *f0, const float *f1, const float *f2,
                     const float *f3, const float *x, float *y, lsk_stream_t stream);
int lsk_axpy_dev_f64(int64_t n, int nf, const double *f0, const double *f1, const double *f2,
                     const double *f3, const double *x, double *y, lsk_stream_t stream);
int lsk_xpay_dev_f32(int64_t n, int nf, const float *f0, const float *f1, const float *f2,
                     const float *f3, const float *x, float *y, lsk_stream_t stream);
int lsk_xpay_dev_f64(int64_t n, int nf, const double *f0, const double *f1, const double *f2,
                     const double *f3, const double *x, double *y, lsk_stream_t stream);

/* Fused axpy and squared norm: y <- fma(alpha, x, y) exactly as lsk_axpy_* / lsk_axpy_dev_*, and
 * partials[0 .. *count) <- per-workgroup shares of sum_i y_new[i]^2 (`capacity` is the room in
 * `partials`: 2048 slots are always enough; vectors of 128 MiB and more use up to LSK_MAX_PARTIALS =
 * 65536 when they are there, a few per cent faster, and the count -- hence the order of the sum --
 * follows the room offered; *count is written on the host).  lsk_sum_partials_* turns them into
 * the dot product y.y when the caller wants it.  Saves the second pass over y that AxpyTask
 * followed by DotTask(v, v) makes (src/CGSolver.hpp:49-51: r <- r - alpha q; r.r).  No
 * counterpart in the reference, whose tasks are launched one by one. */
#define LSK_MAX_PARTIALS 65536
int lsk_axpy_sqnorm_f32(int64_t n, float alpha, const float *x, float *y, float *partials,
                        int64_t capacity, int *count, lsk_stream_t stream);
int lsk_axpy_sqnorm_f64(int64_t n, double alpha, const double *x, double *y, double *partials,
                        int64_t capacity, int *count, lsk_stream_t stream);
int lsk_axpy_sqnorm_dev_f32(int64_t n, int nf, const float *f0, const float *f1, const float *f2,
                            const float *f3, const float *x, float *y, float *partials,
                            int64_t capacity, int *count, lsk_stream_t stream);
int lsk_axpy_sqnorm_dev_f64(int64_t n, int nf, const double *f0, const double *f1, const double *f2,
                            const double *f3, const double *x, double *y, double *partials,
                            int64_t capacity, int *count, lsk_stream_t stream);

/* Fused axpy and dot with a third vector: y <- fma(alpha, x, y) as lsk_axpy_dev_*, and
 * partials[0 .. *count) <- per-workgroup shares of sum_i y_new[i] * w[i] (w == y gives
 * lsk_axpy_sqnorm_dev_*; w may be x).  GMRES's modified Gram-Schmidt sweep w' <- w' - h_k v_k followed
 * by h_{k+1} = v_{k+1}.w' (src/GMRESSolver.hpp:92-100) in 4 vector passes instead of 5.  No
 * counterpart in the reference. */
int lsk_axpy_dot_dev_f32(int64_t n, int nf, const float *f0, const float *f1, const float *f2,
                         const float *f3, const float *x, float *y, const float *w, float *partials,
                         int64_t capacity, int *count, lsk_stream_t stream);
int lsk_axpy_dot_dev_f64(int64_t n, int nf, const double *f0, const double *f1, const double *f2,
                         const double *f3, const double *x, double *y, const double *w, double *partials,
                         int64_t capacity, int *count, lsk_stream_t stream);

/* Fused axpy and xpay over one read of p: z <- fma(alpha, p, z) and p <- fma(beta, p, r), both from
 * the old p, alpha = get_alpha(a0..) and beta = get_alpha(b0..) evaluated on the device.  CG's
 * x <- x + alpha p (src/CGSolver.hpp:48) deferred to its p <- r + beta p (:54), between which x and p
 * are not touched; element for element the same operations as lsk_axpy_dev_* and lsk_xpay_dev_*.
 * The three vectors must be distinct.  No counterpart in the reference. */
int lsk_axpy_xpay_dev_f32(int64_t n, int nfa, const float *a0, const float *a1, const float *a2,
                          const float *a3, int nfb, const float *b0, const float *b1, const float *b2,
                          const float *b3, const float *r, float *p, float *z, lsk_stream_t stream);
int lsk_axpy_xpay_dev_f64(int64_t n, int nfa, const double *a0, const double *a1, const double *a2,
                          const double *a3, int nfb, const double *b0, const double *b1, const double *b2,
                          const double *b3, const double *r, double *p, double *z, lsk_stream_t stream);

/* Several dot products with one vector, and several updates of one vector, in one pass each: for
 * GMRES's orthogonalisation in its CLASSICAL Gram-Schmidt form, where every h_k = v_k . w comes from
 * the same w (the reference runs the modified form, h_k and w <- w - h_k v_k alternating, two passes
 * over w per basis vector: src/GMRESSolver.hpp:92-100; the classical form is different arithmetic,
 * an option of the caller, never a silent replacement).  No counterpart in the reference.
 *   lsk_multi_dot_*: results_dev[j] <- sum_i v[j][i] * w[i] for j < k (k <= 64; v: HOST array of k
 *     device pointers), deterministic; the vectors go in groups of 8 per launch, so w is read
 *     ceil(k / 8) times instead of k.  workspace: lsk_multi_dot_workspace_bytes(k) bytes of device
 *     memory (required).
 *   lsk_multi_axpy_*: y <- fma(c_{k-1}, x[k-1], ... fma(c_0, x[0], y)) with c_j = coef_dev[j], or
 *     -coef_dev[j] when negate != 0: element for element the chain of AxpyTasks y <- y + c_j x[j],
 *     j ascending (bit-identical to k calls of lsk_axpy_dev_*).  partials != NULL: also partials[0 ..
 *     *count) <- per-workgroup shares of sum(y_new^2), as lsk_axpy_sqnorm_* (same rule for the room). */
size_t lsk_multi_dot_workspace_bytes(int k);
int lsk_multi_dot_f32(int64_t n, int k, const float *const *v, const float *w, float *results_dev,
                      void *workspace, size_t workspace_bytes, lsk_stream_t stream);
int lsk_multi_dot_f64(int64_t n, int k, const double *const *v, const double *w, double *results_dev,
                      void *workspace, size_t workspace_bytes, lsk_stream_t stream);
int lsk_multi_axpy_f32(int64_t n, int k, const float *const *x, const float *coef_dev, int negate,
                       float *y, float *partials, int64_t capacity, int *count, lsk_stream_t stream);
int lsk_multi_axpy_f64(int64_t n, int k, const double *const *x, const double *coef_dev, int negate,
                       double *y, double *partials, int64_t capacity, int *count, lsk_stream_t stream);

/* *result_dev <- partials[0] + ... + partials[count-1], in an order fixed by `count` alone (so
 * bit-reproducible): the second stage of the fused forms (lsk_axpy_sqnorm_*, lsk_csr_spmv_dot_*).
 * Counts above 65536 take two kernel stages through `workspace` (lsk_dot_workspace_bytes(0)
 * bytes; NULL = the library's own per-device scratch, then not concurrently with other calls
 * that use it). */
int lsk_sum_partials_f32(int64_t count, const float *partials, float *result_dev, void *workspace,
                         size_t workspace_bytes, lsk_stream_t stream);
int lsk_sum_partials_f64(int64_t count, const double *partials, double *result_dev, void *workspace,
                         size_t workspace_bytes, lsk_stream_t stream);

/* ------------------------------------------------------------- CSR SpMV ------------- */
/* y[r] <- beta*y[r] + sum_{k in rects[r]} val[k] * x[col[k]]   for the shard's rows,
 * beta in {0, 1}.
 *
 * Replaces CSRMatvecTask::cuda_task_body (src/CSRMatrixTasks.cu:14-156: rect->indptr
 * rebasing kernel + cusparseCreateCsr + cusparseSpMV, beta = 0) and keeps the CPU body's
 * semantics (src/CSRMatrixTasks.cpp:73-91: accumulate, entries whose column lies outside
 * the input domain are skipped), selected by `beta`.
 *
 *  rows        rows in this shard (volume of the output piece)
 *  nnz         kernel entries in this shard (volume of the kernel piece)
 *  rects       rows pairs {lo, hi}, hi inclusive, GLOBAL kernel indices: the reference's
 *              Legion::Rect<1,long long> rowptr field (16 bytes per row, 16-byte aligned)
 *  k_base      global kernel index of col[0] / val[0]  (kernel_domain.lo())
 *  col, val    the shard's kernel piece: column ids (global) and entries
 *  x           the ghost instance of the input vector; x[0] is global column x_lo and the
 *              instance covers x_lo..x_hi inclusive (input_domain.bounds)
 *  y           first row of the shard's output piece
 *
 * Rows need not tile the piece contiguously (any {lo,hi} inside the piece is handled);
 * contiguous row blocks take the fast streaming path.  Empty rows have hi == lo - 1. */
int lsk_csr_spmv_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                     const int64_t *col, const float *val, const float *x, int64_t x_lo,
                     int64_t x_hi, float *y, int beta, lsk_stream_t stream);
int lsk_csr_spmv_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                     const int64_t *col, const double *val, const double *x, int64_t x_lo,
                     int64_t x_hi, double *y, int beta, lsk_stream_t stream);

/* Fused product and dot: y as lsk_csr_spmv_*, and partials[0 .. *count) <- per-wave shares of
 * sum_r y_new[r] * w[r], where w[0] pairs with the piece's first row (w = the piece of the
 * vector the caller will dot y with; for CG, p itself: p.Ap, src/CGSolver.hpp:46-47).  The shares
 * are formed from values still in registers, so the DotTask's pass over y and w is saved.
 * `capacity` must be at least lsk_csr_spmv_dot_partials(rows, nnz, sizeof(T)) (which follows
 * the same tile-size rule as the launch, options included); *count is written on the host.
 * lsk_sum_partials_* gives the dot product.  Always the wave-tile form.  No counterpart in the
 * reference. */
int64_t lsk_csr_spmv_dot_partials(int64_t rows, int64_t nnz, int elem_size);
int lsk_csr_spmv_dot_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                         const int64_t *col, const float *val, const float *x, int64_t x_lo,
                         int64_t x_hi, float *y, int beta, const float *w, float *partials,
                         int64_t capacity, int64_t *count, lsk_stream_t stream);
int lsk_csr_spmv_dot_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                         const int64_t *col, const double *val, const double *x, int64_t x_lo,
                         int64_t x_hi, double *y, int beta, const double *w, double *partials,
                         int64_t capacity, int64_t *count, lsk_stream_t stream);

/* Set-up-time tuning of the tile -> XCD placement for one CSR piece: runs the SpMV above (beta = 0,
 * y is overwritten with A x) a few times per candidate value of the "xcd_remap" option, timed with
 * HIP events on `stream`, and returns the fastest candidate in *remap_out and its milliseconds
 * per launch in *ms_out (either may be null).  The option itself is left as it was: the caller
 * stores the value with the matrix and sets it before its launches.  Which placement wins depends
 * on how the piece's columns repeat from row block to row block (for a linearised stencil, on
 * the plane size), so it is measured, not derived.  Blocks until the measurements are done.
 * No counterpart in the reference (cuSPARSE's algorithm choice is internal). */
int lsk_csr_spmv_tune_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, const float *val, const float *x, int64_t x_lo,
                          int64_t x_hi, float *y, lsk_stream_t stream, int *remap_out, float *ms_out);
int lsk_csr_spmv_tune_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, const double *val, const double *x, int64_t x_lo,
                          int64_t x_hi, double *y, lsk_stream_t stream, int *remap_out, float *ms_out);

/* Compressed structure: the set-up-time replacement for the per-call rebasing of
 * src/CuSPARSEHelpers.hpp:9-31,71-80 (the reference converts its rects to 0-based offsets and
 * hands cuSPARSE 64-bit indices on EVERY call; here the conversion is done once and kept by the
 * caller beside the matrix piece).
 *   lsk_csr_compress_cols_u32: col32[k] = col[k] - x_lo for columns inside x_lo..x_hi, 0xffffffff
 *     for the others (which every product skips, as the 64-bit form skips them).  The window must
 *     span fewer than 2^32 - 1 columns, else LSK_ERR_UNSUPPORTED.
 *   lsk_csr_compress_rows_u32: rowoff32[r] = rects[r].lo - k_base for r < rows, rowoff32[rows] =
 *     rects[rows-1].hi + 1 - k_base (rows + 1 values: offsets into the piece's col/val arrays, the
 *     classic CSR row pointer).  Needs rects that are back to back, ascending and inside the piece,
 *     and nnz < 2^32; else LSK_ERR_UNSUPPORTED (blocks until the check has run).
 * In both cases a caller that gets LSK_ERR_UNSUPPORTED stays with lsk_csr_spmv_* (that answer never
 * aborts, whatever lsk_set_abort_on_error says).
 * lsk_csr_spmv_c32_* is lsk_csr_spmv_* on those arrays: `x` points at the window's first
 * element, `x_extent` = x_hi - x_lo + 1 of the compression; a run of rows [a, b) of the piece is
 * multiplied by passing rowoff32 + a, rows = b - a, the same col32 / val, and as nnz the number of
 * entries of those rows (rowoff32[b] - rowoff32[a]; it only sizes the tiles, by nnz / rows).
 * It reads 4 bytes of column and 4 of row structure instead of 8 and 16; tiles and the order of
 * additions are the same, so y is bit-identical to lsk_csr_spmv_*.  _dot_ and _tune_ as the
 * 64-bit entry points of those names. */
int lsk_csr_compress_cols_u32(int64_t nnz, const int64_t *col, int64_t x_lo, int64_t x_hi,
                              uint32_t *col32, lsk_stream_t stream);
int lsk_csr_compress_rows_u32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                              uint32_t *rowoff32, lsk_stream_t stream);
int lsk_csr_spmv_c32_f32(int64_t rows, int64_t nnz, const uint32_t *rowoff32, const uint32_t *col32,
                         const float *val, const float *x, int64_t x_extent, float *y, int beta,
                         lsk_stream_t stream);
int lsk_csr_spmv_c32_f64(int64_t rows, int64_t nnz, const uint32_t *rowoff32, const uint32_t *col32,
                         const double *val, const double *x, int64_t x_extent, double *y, int beta,
                         lsk_stream_t stream);
int lsk_csr_spmv_c32_dot_f32(int64_t rows, int64_t nnz, const uint32_t *rowoff32,
                             const uint32_t *col32, const float *val, const float *x,
                             int64_t x_extent, float *y, int beta, const float *w, float *partials,
                             int64_t capacity, int64_t *count, lsk_stream_t stream);
int lsk_csr_spmv_c32_dot_f64(int64_t rows, int64_t nnz, const uint32_t *rowoff32,
                             const uint32_t *col32, const double *val, const double *x,
                             int64_t x_extent, double *y, int beta, const double *w, double *partials,
                             int64_t capacity, int64_t *count, lsk_stream_t stream);
int lsk_csr_spmv_c32_tune_f32(int64_t rows, int64_t nnz, const uint32_t *rowoff32,
                              const uint32_t *col32, const float *val, const float *x,
                              int64_t x_extent, float *y, lsk_stream_t stream, int *remap_out,
                              float *ms_out);
int lsk_csr_spmv_c32_tune_f64(int64_t rows, int64_t nnz, const uint32_t *rowoff32,
                              const uint32_t *col32, const double *val, const double *x,
                              int64_t x_extent, double *y, lsk_stream_t stream, int *remap_out,
                              float *ms_out);

/* Rows far longer than the mean (a power-law row-length distribution).  The reference hands every
 * matrix to cusparseSpMV(CUSPARSE_SPMV_ALG_DEFAULT), which balances any distribution inside the
 * library (src/CSRMatrixTasks.cu:117-150).  lsk_csr_spmv_* / lsk_csr_spmv_c32_* give every wave a fixed
 * number of rows and stream a row that is too long for a wave's LDS buffer through that buffer, which
 * is right for any matrix but leaves a row of 10^5 .. 10^6 entries to ONE wave.  For such pieces:
 *   lsk_csr_long_rows_create: from a piece's rowoff32 (rows + 1 offsets, lsk_csr_compress_rows_u32) the
 *     list of rows longer than `threshold` entries (0: the library's 8192; at least 2048), each cut into
 *     chunks of 8192 entries.  LSK_ERR_UNSUPPORTED (never aborts) when no row is that long: the caller
 *     stays with lsk_csr_spmv_c32_*.  Blocks.
 *   lsk_csr_spmv_c32_split_*: lsk_csr_spmv_c32_* for the rows row_begin .. row_begin + rows - 1 of that
 *     piece (rowoff32 and y those of row_begin; col32, val, x, nnz as there) with the listed rows left
 *     out of the tile kernel and multiplied one workgroup per chunk, a row's chunk sums added in chunk
 *     order by one more small kernel: the same bits on every call, whatever the hardware schedules.
 *     All other rows get exactly what lsk_csr_spmv_c32_* gives them.  plan = NULL: lsk_csr_spmv_c32_*.
 *     The chunk sums live in the plan: products through one plan are ordered on one stream.
 *   lsk_csr_long_rows_info: out4 = {listed rows, chunks, threshold, entries per chunk}. */
typedef struct lsk_csr_long_rows lsk_csr_long_rows_t;
int lsk_csr_long_rows_create(int64_t rows, const uint32_t *rowoff32, int64_t threshold, lsk_stream_t stream,
                             lsk_csr_long_rows_t **out);
int lsk_csr_long_rows_info(const lsk_csr_long_rows_t *plan, int64_t *out4);
int lsk_csr_long_rows_destroy(lsk_csr_long_rows_t *plan);
int lsk_csr_spmv_c32_split_f32(const lsk_csr_long_rows_t *plan, int64_t row_begin, int64_t rows, int64_t nnz,
                               const uint32_t *rowoff32, const uint32_t *col32, const float *val, const float *x,
                               int64_t x_extent, float *y, int beta, lsk_stream_t stream);
int lsk_csr_spmv_c32_split_f64(const lsk_csr_long_rows_t *plan, int64_t row_begin, int64_t rows, int64_t nnz,
                               const uint32_t *rowoff32, const uint32_t *col32, const double *val, const double *x,
                               int64_t x_extent, double *y, int beta, lsk_stream_t stream);

/* indptr[i] = rects[i].lo - rects[0].lo for i < rows; indptr[rows] = rects[rows-1].hi + 1
 * - rects[0].lo.  Bit-exact restatement of convertGlobalRowptrToLocalIndPtr
 * (src/CuSPARSEHelpers.hpp:9-31) for callers that want a classic CSR offset array; the
 * SpMV above does not need it. */
int lsk_rects_to_indptr_i64(int64_t rows, const int64_t *rects, int64_t *indptr,
                            lsk_stream_t stream);

/* ------------------------------------------------------------- COO SpMV ------------- */
/* y[row[k]] += val[k] * x[col[k]] for every k of the piece with row[k] in y_lo..y_hi and
 * col[k] in x_lo..x_hi (global, inclusive).  y[0] is global row y_lo, x[0] global column
 * x_lo.  Accumulates (beta = 1): the caller zero-fills y first, as SquarePlanner::matvec
 * does (src/SquarePlanner.hpp:340-357).
 *
 * Replaces COOMatvecTask::cuda_task_body (src/COOMatrixTasks.cu:12-146: cusparseCreateCoo +
 * cusparseSpMV on shifted x and y pointers) with the CPU body's guards
 * (src/COOMatrixTasks.cpp:67-74).  Any entry order is accepted; row-sorted input runs
 * fastest (runs of equal rows are summed on chip before one atomic add). */
int lsk_coo_spmv_f32(int64_t nnz, const int64_t *row, const int64_t *col, const float *val,
                     const float *x, int64_t x_lo, int64_t x_hi, float *y, int64_t y_lo,
                     int64_t y_hi, lsk_stream_t stream);
int lsk_coo_spmv_f64(int64_t nnz, const int64_t *row, const int64_t *col, const double *val,
                     const double *x, int64_t x_lo, int64_t x_hi, double *y, int64_t y_lo,
                     int64_t y_hi, lsk_stream_t stream);

/* The same on a set-up-time 32-bit copy of the index fields (lsk_csr_compress_cols_u32 serves for
 * both: row32 = row - row_origin over the caller's row window, col32 = col - x_lo over the x window,
 * 0xffffffff = outside).  `x` points at the x window's first element, `y` at the row
 * row_origin + y_first; entries with row32 outside y_first..y_last or col32 >= x_extent are
 * dropped.  16 instead of 24 (fp64) bytes per entry; same kernel, same run sums. */
int lsk_coo_spmv_c32_f32(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const float *val,
                         const float *x, int64_t x_extent, float *y, int64_t y_first, int64_t y_last,
                         lsk_stream_t stream);
int lsk_coo_spmv_c32_f64(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const double *val,
                         const double *x, int64_t x_extent, double *y, int64_t y_first, int64_t y_last,
                         lsk_stream_t stream);

/* x[i] <- a uniform value in [0, 1) that depends on `seed` and the GLOBAL element index
 * first_index + i only: one splitmix64 step of seed + (index + 1) * 0x9e3779b97f4a7c15, its top 53
 * bits / 2^53 (f64) or top 24 bits / 2^24 (f32).  Replaces RandomFillTask (src/UtilityTasks.cpp:
 * 128-166: a std::mt19937 seeded from std::random_device per task, serial, different on every run)
 * with something reproducible and independent of the partition. */
int lsk_random_fill_f32(int64_t n, uint64_t seed, int64_t first_index, float *x, lsk_stream_t stream);
int lsk_random_fill_f64(int64_t n, uint64_t seed, int64_t first_index, double *x, lsk_stream_t stream);

/* Column-blocked entry order for matrices whose x gathers have no locality (unstructured COO):
 * order[i] = the entry that comes i-th when the entries are sorted by column block
 * col32 / block_cols (stable: inside a block they keep their order, e.g. rows ascending).  With
 * the fields gathered through `order` (lsk_gather_*: dst[i] = src[order[i]]) every run of entries
 * reads one block of x, which stays in L2 when block_cols * sizeof(T) is a few MiB; the product
 * kernel and its results (up to the order of the atomic additions) are unchanged.  nnz < 2^31.
 * Set-up time: allocates and frees its scratch, returns when done.  No counterpart in the
 * reference. */
int lsk_coo_block_order_u32(int64_t nnz, const uint32_t *col32, int64_t x_extent, int64_t block_cols,
                            uint32_t *order, lsk_stream_t stream);
int lsk_gather_u32(int64_t n, const uint32_t *src, const uint32_t *order, uint32_t *dst, lsk_stream_t stream);
int lsk_gather_f32(int64_t n, const float *src, const uint32_t *order, float *dst, lsk_stream_t stream);
int lsk_gather_f64(int64_t n, const double *src, const uint32_t *order, double *dst, lsk_stream_t stream);

/* The binned COO product, for matrices whose columns have no locality, so that the x gathers of
 * lsk_coo_spmv_* (a cache line through L2 per entry), not the entry stream, bound the product.
 * Computes what COOMatvecTask computes (src/COOMatrixTasks.cpp:58-76, src/COOMatrixTasks.cu:12-43)
 * for the entries inside the windows, on a set-up-time copy ordered by (block of columns, tile of
 * rows), in two streaming passes: (1) a workgroup holds a block of x in LDS and writes the products
 * val * x[col] of a run of the block's entries to a workspace; (2) a workgroup holds a tile of y in
 * LDS and adds the tile's segments of the workspace into it.  No gathers from memory, no global
 * atomics; 3 s + 4 bytes per entry cross HBM as contiguous streams.  Sums agree with the CPU
 * loop to rounding (the order of the LDS additions is the hardware's, as with the reference's
 * atomicAdd kernel).
 *
 * create: row32 / col32 / val as for lsk_coo_spmv_c32_* (any entry order; entries with row32
 * outside y_first..y_last or col32 >= x_extent, or 0xffffffff, are left out).  block_cols /
 * tile_rows: 0 for the library's choice (blocks of 128 KiB of x; the largest tiles that make a
 * whole number of rounds over the CUs), else multiples of 8 up to 20472 (fp32 blocks: 40952).  The
 * tile of y is held and summed in fp64 for both types and rounded once at the end.  Set-up time: sorts, allocates the copy (padded entries x (2 s + 8) bytes, the workspace
 * included), returns when done.  LSK_ERR_UNSUPPORTED (never an abort): nnz >= 2^31, more than
 * 2^24 (block, tile) pairs, or no entry inside the windows.
 * spmv: y <- A x (beta = 0) or y <- y + A x (beta = 1); `x` points at the x window's first
 * element, `y` at row y_first.  Calls on one handle share its workspace: they must not overlap.
 * update_values: the caller rewrote val in place (same rows and columns); one pass.
 * info: out8 = entries inside the windows, padded entries, block_cols, tile_rows, blocks, tiles,
 * workgroups of pass 1, non-empty segments.  No counterpart in the reference. */
typedef struct lsk_coo_binned lsk_coo_binned_t;
int lsk_coo_binned_create_f32(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const float *val,
                              int64_t y_first, int64_t y_last, int64_t x_extent, int block_cols, int tile_rows,
                              lsk_stream_t stream, lsk_coo_binned_t **out);
int lsk_coo_binned_create_f64(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const double *val,
                              int64_t y_first, int64_t y_last, int64_t x_extent, int block_cols, int tile_rows,
                              lsk_stream_t stream, lsk_coo_binned_t **out);
int lsk_coo_binned_spmv_f32(const lsk_coo_binned_t *m, const float *x, float *y, int beta, lsk_stream_t stream);
int lsk_coo_binned_spmv_f64(const lsk_coo_binned_t *m, const double *x, double *y, int beta, lsk_stream_t stream);
/* The same product in two calls, for a caller whose x arrives in parts (one process per GPU: the columns
 * a rank owns are there, the others are on their way; under Legion the READ_ONLY ghost requirement makes
 * the whole of x arrive before the task starts, src/COOMatrix.cpp:179-188).  phase 0: the first pass for
 * the blocks of columns that lie wholly inside col_lo .. col_hi (counted from the x window's first
 * column), reading x only there; phase 1: the first pass for all other blocks, then the second pass, which
 * writes y.  Phase 0 then phase 1 on one stream = lsk_coo_binned_spmv_* (the same products from the first
 * pass; the second pass adds them in the hardware's order, as always). */
int lsk_coo_binned_spmv_phase_f32(const lsk_coo_binned_t *m, const float *x, float *y, int beta, int phase,
                                  int64_t col_lo, int64_t col_hi, lsk_stream_t stream);
int lsk_coo_binned_spmv_phase_f64(const lsk_coo_binned_t *m, const double *x, double *y, int beta, int phase,
                                  int64_t col_lo, int64_t col_hi, lsk_stream_t stream);
int lsk_coo_binned_update_values_f32(lsk_coo_binned_t *m, int64_t nnz, const float *val, lsk_stream_t stream);
int lsk_coo_binned_update_values_f64(lsk_coo_binned_t *m, int64_t nnz, const double *val, lsk_stream_t stream);
int lsk_coo_binned_info(const lsk_coo_binned_t *m, int64_t *out8);
int lsk_coo_binned_destroy(lsk_coo_binned_t *m);

/* The transposed product (CSRRmatvecTask / COORmatvecTask: registered by the reference,
 * src/Initialize.cpp:464-479, with bodies that are assert(false) on CPU and GPU alike,
 * src/CSRMatrixTasks.cpp:95-100, src/COOMatrixTasks.cpp:78-83, src/CSRMatrixTasks.cu:159-164) is
 * served without atomics: at set-up the entries are sorted by column once, which gives the CSR
 * form of the transpose in the 32-bit structure of lsk_csr_spmv_c32_*; every y <- A^T x is then
 * that kernel, deterministic and at the same rate as y <- A x.  The sort is
 * lsk_coo_block_order_u32 with block_cols = 1 (stable: inside a column the rows ascend) followed
 * by lsk_gather_*; the two helpers below supply what a CSR matrix lacks for it.
 *
 * lsk_csr_entry_rows_u32: row32[k - k_base] <- r for every entry k of rect r (0-based in `rects`,
 * arguments as lsk_csr_spmv_*), 0xffffffff for an entry inside no rect; rows < 2^32 - 1.
 * lsk_sorted_key_offsets_u32: offsets[c] <- the number of keys[0..n) below c, for c = 0..nkeys
 * (nkeys + 1 values; keys ascending): the row offsets of the transposed structure, with
 * offsets[nkeys] = the entries whose column lies inside the window (the marker 0xffffffff sorts
 * behind them). */
int lsk_csr_entry_rows_u32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, uint32_t *row32,
                           lsk_stream_t stream);
int lsk_sorted_key_offsets_u32(int64_t n, const uint32_t *keys, int64_t nkeys, uint32_t *offsets,
                               lsk_stream_t stream);

/* ------------------------------------------------ CSR product on sliced row patterns --- */
/* A set-up-time re-layout for CSR matrices whose rows repeat a few column patterns (every
 * linearised stencil: col = row + one of a few lists of distances; at most 64 patterns of at most
 * 32 entries).  lsk_csr_sliced_create_* gives every row the id of its pattern, verifies every entry
 * against the pattern table, and copies the values into blocks of 256 R consecutive rows, R = 16 /
 * sizeof(T) (512 rows of fp64, 1024 of fp32), stored so that one lane loads the j-th values of its R
 * rows with one 16-byte access (short rows padded with zeros); lsk_csr_sliced_spmv_* then computes
 * rows row_begin..row_end-1 of y <- A x + beta y from that copy alone: no column array and no row
 * offsets are read, the windows of x a block's list reads are staged once through LDS, and y leaves
 * in bursts of 4 KiB.  Every row is summed by one lane in ascending entry order with separate
 * multiply and add, which is bit for bit the loop of the reference's CPU task
 * (src/CSRMatrixTasks.cpp:73-91); the reference's GPU task hands the arrays to cuSPARSE instead
 * (src/CSRMatrixTasks.cu:117-150).
 *
 * rows, nnz, rects, k_base, col, val as lsk_csr_spmv_*; row_first = global index of rects' first
 * row; x_lo..x_hi = the window of the input vector every product will be given (x[0] = element
 * x_lo); y[0] = the first row.  The structure holds a COPY of the values: lsk_csr_sliced_update_values_*
 * brings it up to date when they change.
 * create answers LSK_ERR_UNSUPPORTED (never an abort) when the matrix has no such structure
 * (more patterns, longer rows, a column outside the window, rects outside the piece, or padding
 * that would cost more than the column array saves): stay with lsk_csr_spmv_c32_* / lsk_csr_spmv_*.
 * Set-up time: allocates, synchronises.  lsk_csr_sliced_info: out[0] = padded entries (the values
 * one product reads), out[1] = blocks, out[2] = blocks whose rows fit no common distance list (every
 * row looks its own up), out[3] = patterns, out[4] = waves (quarter blocks) in which some row has
 * only an ordered part of its block's list (they read a 32-bit mask per row), out[5] = rows per
 * block, out[6] = bytes of LDS a launch uses, out[7] = 1 if lsk_csr_sliced_spmv2_* serves the structure
 * (every block on one staged list), else 0, out[8] = 1 if every list that serves a block is runs of exactly
 * three consecutive distances (27-point stencils); out holds 9 values.
 * Option "xcd_remap" places the workgroups as for lsk_csr_spmv_*. */
typedef struct lsk_csr_sliced lsk_csr_sliced_t;
int lsk_csr_sliced_create_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const float *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out);
int lsk_csr_sliced_create_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const double *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out);
/* The structure holds a copy of the entry values, whereas the reference's task reads the region on
 * every call (src/CSRMatrixTasks.cu:88-118).  A caller that rewrites the entries in place (same rects,
 * same columns, new values) calls this before its next product: it copies `val` into the slices again
 * (the last set-up kernel alone: one read of val, one write of the slices, stream-ordered, no
 * allocation, no synchronisation).  rows, nnz, rects, k_base as given to create; a change of the
 * STRUCTURE needs destroy + create. */
int lsk_csr_sliced_update_values_f32(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects,
                                     int64_t k_base, const float *val, lsk_stream_t stream);
int lsk_csr_sliced_update_values_f64(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects,
                                     int64_t k_base, const double *val, lsk_stream_t stream);
int lsk_csr_sliced_spmv_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                            int beta, lsk_stream_t stream);
int lsk_csr_sliced_spmv_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                            int beta, lsk_stream_t stream);
/* The same product, and partials[0 .. *count) <- one sum per block of y_new[r] * w[r] over the
 * block's rows inside row_begin..row_end-1 (w[0] pairs with the structure's first row, like y),
 * formed from values still in registers and added in a fixed order whatever the placement;
 * lsk_sum_partials_* gives the dot product (CG's p.Ap, src/CGSolver.hpp:46-47, without a pass over y
 * and w).  capacity >= lsk_csr_sliced_dot_partials(m, row_begin, row_end), the number of blocks those
 * rows touch; *count (that number) is written on the host. */
/* Two products with one pass over the values: y_a <- A x_a (+ beta y_a), y_b <- A x_b (+ beta y_b), for
 * callers that step two systems on one matrix, as the reference's BenchmarkStencil does
 * (test/BenchmarkStencil.cpp:181-199: sol/rhs and sol1/rhs1 on one matrix; its planner launches one
 * CSRMatvecTask per system).  Each result is, bit for bit, what lsk_csr_sliced_spmv_* gives for it
 * alone; the values are read once, so the pair moves `padded entries * s + 4 rows * s` bytes instead
 * of twice `padded entries * s + 2 rows * s`.  Both x windows and both y pieces have the layout of
 * the single product.  The _dot form leaves the shares of y_a . w_a in partials_a and of y_b . w_b in
 * partials_b (*count each, as lsk_csr_sliced_spmv_dot_*).  LSK_ERR_UNSUPPORTED (never an abort) for a
 * structure with blocks that do not run on one staged list: launch the two products one by one. */
int lsk_csr_sliced_spmv2_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *xa,
                             const float *xb, float *ya, float *yb, int beta, lsk_stream_t stream);
int lsk_csr_sliced_spmv2_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *xa,
                             const double *xb, double *ya, double *yb, int beta, lsk_stream_t stream);
int lsk_csr_sliced_spmv2_dot_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *xa,
                                 const float *xb, float *ya, float *yb, int beta, const float *wa, const float *wb,
                                 float *partials_a, float *partials_b, int64_t capacity, int64_t *count,
                                 lsk_stream_t stream);
int lsk_csr_sliced_spmv2_dot_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *xa,
                                 const double *xb, double *ya, double *yb, int beta, const double *wa, const double *wb,
                                 double *partials_a, double *partials_b, int64_t capacity, int64_t *count,
                                 lsk_stream_t stream);
int64_t lsk_csr_sliced_dot_partials(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end);
int lsk_csr_sliced_spmv_dot_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                                int beta, const float *w, float *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream);
int lsk_csr_sliced_spmv_dot_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                                int beta, const double *w, double *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream);
/* Times the whole product (x: the window, y: rows values, overwritten) for a few values of option
 * "xcd_remap" and reports the fastest and its milliseconds per launch, as lsk_csr_spmv_tune_* does.
 * Callers that remake the structure a few times and keep the fastest copy (where the driver puts an
 * array in HBM moves the product's time by several per cent) compare these milliseconds. */
int lsk_csr_sliced_tune_f32(const lsk_csr_sliced_t *m, const float *x, float *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms);
int lsk_csr_sliced_tune_f64(const lsk_csr_sliced_t *m, const double *x, double *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms);
/* The same for the two-vector product lsk_csr_sliced_spmv2_* (ya <- A xa, yb <- A xb, both overwritten): with
 * two windows of x competing for each XCD's L2 it prefers runs of neighbouring blocks on one XCD where the
 * one-vector product does not care (27-point 512^3 fp64: 6.45 ms with placement 0, 5.55 ms with 64), so a
 * caller that pairs products keeps a placement of its own for them.  LSK_ERR_UNSUPPORTED where
 * lsk_csr_sliced_spmv2_* answers that.  No counterpart in the reference. */
int lsk_csr_sliced_tune2_f32(const lsk_csr_sliced_t *m, const float *xa, const float *xb, float *ya, float *yb,
                             lsk_stream_t stream, int *xcd_remap, float *ms);
int lsk_csr_sliced_tune2_f64(const lsk_csr_sliced_t *m, const double *xa, const double *xb, double *ya, double *yb,
                             lsk_stream_t stream, int *xcd_remap, float *ms);
int lsk_csr_sliced_info(const lsk_csr_sliced_t *m, int64_t *out);
/* 1 when entry j of a block's list has ONE value in every row of every block on that list, checked bit by
 * bit against the copy's values at create and again at every update_values (a constant-coefficient stencil,
 * as every matrix of test/BenchmarkStencil.cpp:35-124 is); 0 otherwise.  With option "sliced.uniform" = 1
 * (default 0) the products of such a structure take the values as scalars from a 64 x 32 table and do not
 * read the value array: rows * 2 s bytes per product instead of padded entries * s + rows * 2 s, same
 * multiplications and additions in the same order, same bits.  After update_values the check runs behind
 * the refill on that stream and nobody waits for it: products launched before its verdict is in read the
 * value array; this call does wait for it.  No counterpart in the reference. */
int lsk_csr_sliced_uniform_values(const lsk_csr_sliced_t *m);
int lsk_csr_sliced_destroy(lsk_csr_sliced_t *m);

/* ------------------------------------------------ set-up helpers (device side) ------ */
/* out[0] = min idx[i], out[1] = max idx[i] (out = {INT64_MAX, INT64_MIN} when n == 0):
 * the bounding interval of create_partition_by_image over a column field
 * (src/CSRMatrix.cpp:112-132, src/COOMatrix.cpp), i.e. the ghost range of a shard. */
int lsk_index_bounds_i64(int64_t n, const int64_t *idx, int64_t *out_minmax,
                         lsk_stream_t stream);

/* Which rows of a CSR piece read columns outside lo..hi: out[0] <- the largest row index (0-based
 * in the piece) that has a column below lo, or -1; out[1] <- the smallest row index that has a
 * column above hi, or `rows`.  Rows strictly between them read only lo..hi.  With lo..hi = the
 * interval of x a process owns, those are the rows whose product does not wait for the ghost
 * cells: the caller launches them while the halo is still in flight.  What Legion does implicitly
 * by starting a matvec point task as soon as ITS ghost copy has landed
 * (src/CSRMatrix.cpp:202-211).  Arguments as lsk_csr_spmv_*. */
int lsk_csr_halo_rows_i64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, int64_t lo, int64_t hi, int64_t *out,
                          lsk_stream_t stream);

/* Linearised stencil matrices, bit-identical to FillLinearizedCSRStencilTask /
 * FillLinearizedCOOStencilTask with IndexOrder::ROW_MAJOR (src/StencilGenerator.cpp:129-242,
 * 379-517): grid bounds [0,extent[d]) in `dim` (1..3) dimensions, row id = row-major
 * linearisation, `noffsets` (<= 32) offsets already sorted lexicographically with their
 * coefficients; an entry exists where point+offset stays inside the grid.
 *
 * lsk_stencil_count: counts[r - row_lo] = entries of row r, for r in row_lo..row_hi.
 * lsk_stencil_fill_csr: given rects (global k, from a prefix sum of the counts) writes
 *   col[k - k_base], val[k - k_base] for the rows row_lo..row_hi.
 * lsk_stencil_fill_coo: same, plus row[k - k_base]. */
int lsk_stencil_count(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                      int64_t row_lo, int64_t row_hi, int64_t *counts, lsk_stream_t stream);
int lsk_stencil_fill_csr_f32(int dim, const int64_t *extent, int noffsets,
                             const int64_t *offsets, const float *coeffs, int64_t row_lo,
                             int64_t row_hi, const int64_t *rects, int64_t k_base,
                             int64_t *col, float *val, lsk_stream_t stream);
int lsk_stencil_fill_csr_f64(int dim, const int64_t *extent, int noffsets,
                             const int64_t *offsets, const double *coeffs, int64_t row_lo,
                             int64_t row_hi, const int64_t *rects, int64_t k_base,
                             int64_t *col, double *val, lsk_stream_t stream);
int lsk_stencil_fill_coo_f32(int dim, const int64_t *extent, int noffsets,
                             const int64_t *offsets, const float *coeffs, int64_t row_lo,
                             int64_t row_hi, const int64_t *rects, int64_t k_base,
                             int64_t *row, int64_t *col, float *val, lsk_stream_t stream);
int lsk_stencil_fill_coo_f64(int dim, const int64_t *extent, int noffsets,
                             const int64_t *offsets, const double *coeffs, int64_t row_lo,
                             int64_t row_hi, const int64_t *rects, int64_t k_base,
                             int64_t *row, int64_t *col, double *val, lsk_stream_t stream);

/* out[i] = in[0] + ... + in[i] (in-place allowed), n < 2^31: the running entry count the serial
 * fill tasks carry from row to row (src/StencilGenerator.cpp:494-511), as one device scan.  Set-up
 * time: allocates and frees its scratch and returns when the scan has finished. */
int lsk_inclusive_sum_i64(int64_t n, const int64_t *in, int64_t *out, lsk_stream_t stream);

/* rects[i] = {base + prefix[i], base + prefix[i] + counts[i] - 1} where prefix is the
 * exclusive prefix sum of counts given as `prefix_incl[i] = counts[0]+...+counts[i]`
 * (the row_begin/row_end rule of src/StencilGenerator.cpp:494-511). */
int lsk_rects_from_counts_i64(int64_t rows, const int64_t *counts, const int64_t *prefix_incl,
                              int64_t base, int64_t *rects, lsk_stream_t stream);

/* ------------------------------------------------------------- task ids ------------- */
/* Task-id arithmetic of src/TaskBaseClasses.hpp (origin 500000, NUM_META_TASK_IDS = 1,
 * block size 3*27*64 = 5184; entry index float 0 / double 1; index-type index int 0 /
 * unsigned 1 / long long 2).  Host functions, bit-exact. */
int64_t lsk_task_id_t(int block_id, int entry_index);                 /* :100-103 */
int64_t lsk_task_id_di(int block_id, int dim, int index_type_index);  /* :161-165 */
int64_t lsk_task_id_tdi(int block_id, int entry_index, int dim, int index_type_index); /* :236-242 */
int64_t lsk_task_id_tdddiii(int block_id, int entry_index, int n1, int n2, int n3, int i1,
                            int i2, int i3);                          /* :330-350 */

/* ------------------------------------------------------------- RCCL layer ----------- */
/* One communicator per process (one process or one Legion GPU processor per MI355X).  The
 * reference has no NCCL/MPI call sites: the dot all-reduce is Legion's REDOP_SUM future
 * map (src/PartitionedVector.cpp:353-357) and the halo gather is the READ_ONLY ghost
 * partition requirement (src/CSRMatrix.cpp:202-211); these calls do both on the task's
 * stream over xGMI. */
typedef struct lsk_comm *lsk_comm_t;
#define LSK_UNIQUE_ID_BYTES 128
int lsk_comm_unique_id(void *id_out /* LSK_UNIQUE_ID_BYTES */);
int lsk_comm_init_rank(lsk_comm_t *comm, int nranks, int rank, const void *id);
int lsk_comm_destroy(lsk_comm_t comm);
/* Gives the communicator up without waiting for what is enqueued on it (ncclCommAbort): for a rank
 * that found a peer missing and wants its own streams to drain.  The handle is dead afterwards. */
int lsk_comm_abort(lsk_comm_t comm);
int lsk_comm_rank(lsk_comm_t comm, int *rank, int *nranks);
/* In-place sum over ranks of `count` scalars at buf (device). */
int lsk_allreduce_sum_f32(lsk_comm_t comm, float *buf, int count, lsk_stream_t stream);
int lsk_allreduce_sum_f64(lsk_comm_t comm, double *buf, int count, lsk_stream_t stream);
/* Point-to-point halo exchange, grouped into one RCCL group: for i < nsend send
 * send_count[i] elements at send_ptr[i] to send_peer[i]; same for receives.  A peer may be the
 * calling rank itself (the i-th such send lands in the i-th such receive). */
int lsk_halo_exchange_f32(lsk_comm_t comm, int nsend, const int *send_peer,
                          const float *const *send_ptr, const int64_t *send_count, int nrecv,
                          const int *recv_peer, float *const *recv_ptr,
                          const int64_t *recv_count, lsk_stream_t stream);
int lsk_halo_exchange_f64(lsk_comm_t comm, int nsend, const int *send_peer,
                          const double *const *send_ptr, const int64_t *send_count, int nrecv,
                          const int *recv_peer, double *const *recv_ptr,
                          const int64_t *recv_count, lsk_stream_t stream);
/* recv[rank*count .. ] <- every rank's `count` elements at send (x all-gather for
 * unstructured matrices whose ghost range is all of x). */
int lsk_allgather_f32(lsk_comm_t comm, const float *send, float *recv, int64_t count,
                      lsk_stream_t stream);
int lsk_allgather_f64(lsk_comm_t comm, const double *send, double *recv, int64_t count,
                      lsk_stream_t stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* LSK_H_INCLUDED */
