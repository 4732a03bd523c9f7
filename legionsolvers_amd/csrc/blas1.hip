// BLAS-1 leaf kernels for gfx950: scal / axpy / xpay / fill / copy, dot / nrm2, the scalar
// "future" ops, and the N-D strided xpay.  All are HBM-bound streams: 16 bytes per lane per
// access, four independent accesses in flight per lane, grid capped at 8 workgroups per CU
// and grid-strided.
//
// Arithmetic follows the reference's CPU leaf tasks (src/LinearAlgebraTasks.cpp:37-43,
// 80-87, 124-131): scal is one multiply, axpy is fma(alpha, x, y), xpay is fma(alpha, y, x),
// so element results are bit-identical to the CPU path.
#include <mutex>

#include "lsk_common.hpp"

// Partial sums of lsk_dot_noalloc_f64: part of the library's image on every GPU it runs on (32 KiB
// each, held until the process ends), so that entry point never asks the driver for memory.  In
// host code this name is only a handle for hipGetSymbolAddress; kernels are given the address that
// call returns for the current device (noalloc_scratch below), never the name.
static __device__ __align__(256) unsigned char scratch[lsk::kScratchBytes];

namespace {

using lsk::kBlock;

enum Op { OP_SCAL = 0, OP_AXPY = 1, OP_XPAY = 2, OP_FILL = 3, OP_COPY = 4 };

// alpha either by value (nf < 0) or as 0..4 device-resident futures combined with the
// get_alpha rule (src/LegionUtilities.cpp:72-97).
template <typename T>
struct Alpha {
    int nf;
    T host;
    const T *f0, *f1, *f2, *f3;
};

template <typename T>
__device__ __forceinline__ T resolve_alpha(const Alpha<T> &a) {
    switch (a.nf) {
    case 0: return static_cast<T>(1);
    case 1: return *a.f0;
    case 2: return *a.f0 / *a.f1;
    case 3: return (*a.f0 * *a.f1) / *a.f2;
    case 4: return (*a.f0 * *a.f1) / (*a.f2 * *a.f3);
    default: return a.host;
    }
}

// 16-byte vectors as the compiler's own vector types (the nontemporal builtins take no others).
typedef double double2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <typename T> struct Vec;
template <> struct Vec<double> { using type = double2v; static constexpr int N = 2; };
template <> struct Vec<float> { using type = float4v; static constexpr int N = 4; };

// Loads and stores of the streams.  NT: nontemporal (streamed past the caches).  On vectors far
// larger than the caches that is worth 7-10 % (fp64, 2^27 elements: dot 0.340 -> 0.305 ms, axpy 0.604
// -> 0.542 ms, and only with BOTH the loads and the stores nontemporal); on vectors the Infinity Cache
// can hold between two kernels (128 MB) the plain forms are as fast or faster.
template <bool NT, typename E> __device__ __forceinline__ E ld(const E *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename E> __device__ __forceinline__ void st(E *p, E v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T, int OP>
__device__ __forceinline__ T apply(T alpha, T x, T y) {
    if constexpr (OP == OP_SCAL) return alpha * y;
    if constexpr (OP == OP_AXPY) return fma_t(alpha, x, y);
    if constexpr (OP == OP_XPAY) return fma_t(alpha, y, x);
    if constexpr (OP == OP_FILL) return alpha;
    return x; // OP_COPY
}

template <typename T, int OP>
__device__ __forceinline__ double2v apply_v(T alpha, double2v x, double2v y) {
    return double2v{apply<double, OP>(alpha, x.x, y.x), apply<double, OP>(alpha, x.y, y.y)};
}
template <typename T, int OP>
__device__ __forceinline__ float4v apply_v(T alpha, float4v x, float4v y) {
    return float4v{apply<float, OP>(alpha, x.x, y.x), apply<float, OP>(alpha, x.y, y.y),
                   apply<float, OP>(alpha, x.z, y.z), apply<float, OP>(alpha, x.w, y.w)};
}

constexpr bool reads_x(int op) { return op == OP_AXPY || op == OP_XPAY || op == OP_COPY; }
constexpr bool reads_y(int op) { return op == OP_SCAL || op == OP_AXPY || op == OP_XPAY; }

// E = element type of one access: the 16-byte vector when every pointer is 16-byte aligned,
// else the scalar.  x and y are not declared restrict: y == x is a legal call.
template <typename T, int OP, typename E, int EN, bool NT = false>
__global__ __launch_bounds__(kBlock) void ew_kernel(int64_t n, Alpha<T> a, const T *x, T *y, int form) {
    const T alpha = resolve_alpha(a);
    const int64_t ne = n / EN;
    const E *xe = reinterpret_cast<const E *>(x);
    E *ye = reinterpret_cast<E *>(y);
    // form 1: a workgroup sweeps contiguous chunks of 4 x 256 accesses (its four accesses per
    // lane are 4 KiB apart); form 0: every access of the grid is one grid-stride apart.
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    const int64_t inner = form ? kBlock : gstride;
    const int64_t outer = 4 * gstride;
    int64_t i = form ? int64_t(blockIdx.x) * (4 * kBlock) + threadIdx.x
                     : int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0);

    for (; i + 3 * inner < ne; i += outer) {
        E xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (reads_x(OP)) xv[u] = ld<NT>(xe + i + u * inner); else xv[u] = E{};
            if constexpr (reads_y(OP)) yv[u] = ld<NT>(ye + i + u * inner); else yv[u] = E{};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (EN == 1) st<NT>(ye + i + u * inner, apply<T, OP>(alpha, xv[u], yv[u]));
            else st<NT>(ye + i + u * inner, apply_v<T, OP>(alpha, xv[u], yv[u]));
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = i + u * inner;
        if (j < ne) {
            E xv{}, yv{};
            if constexpr (reads_x(OP)) xv = ld<NT>(xe + j);
            if constexpr (reads_y(OP)) yv = ld<NT>(ye + j);
            if constexpr (EN == 1) st<NT>(ye + j, apply<T, OP>(alpha, xv, yv));
            else st<NT>(ye + j, apply_v<T, OP>(alpha, xv, yv));
        }
    }
    if (EN > 1 && first) { // the n % EN trailing elements
        for (int64_t j = ne * EN; j < n; ++j) {
            T xs = T(0), ys = T(0);
            if constexpr (reads_x(OP)) xs = x[j];
            if constexpr (reads_y(OP)) ys = y[j];
            y[j] = apply<T, OP>(alpha, xs, ys);
        }
    }
}

// Whether a stream over n elements uses the nontemporal accesses: option "blas1.nontemporal" (1, 0,
// or -1 = for vectors of 128 MiB and more: two of them no longer fit the 256 MiB last-level cache together, so
// what a kernel leaves behind does not serve the next one.  Measured (tools/exp/nt_threshold_exp.py): CG on the 27-point
// stencil steps 13 % faster with the streaming form at 128 MiB vectors, 9 % at 250 MiB, 2-3 % at 54-86 MiB, equally fast
// at 31 MiB; BiCGStab on the random COO matrix with 119 MiB vectors 1.5 % SLOWER (the binned product finds less of x in
// the cache).
inline bool stream_past_caches(int64_t n, size_t elem_bytes) {
    const int opt = lsk::options().blas1_nontemporal;
    if (opt >= 0) return opt != 0;
    return static_cast<double>(n) * static_cast<double>(elem_bytes) >= 128.0 * 1024.0 * 1024.0;
}

// Workgroups of an element-wise stream over `accesses` accesses (4 per lane and trip).  Vectors that no
// cache holds (`past_caches`) get one 16 KiB chunk per workgroup, however many that makes: measured on
// 1 GiB vectors, 65536 workgroups of one trip each run axpy 4.7 %, axpy + xpay 3.7 % and a copy 4 %
// faster than 2048 workgroups looping (tools/blas1_shapes.py); smaller vectors keep the capped grid.
inline int stream_grid(int64_t accesses, bool past_caches = false) {
    int64_t blocks = lsk::ceil_div(accesses, int64_t(kBlock) * 4);
    if (blocks < 1) blocks = 1;
    const int64_t cap = past_caches ? int64_t(1) << 30 : lsk::kMaxGridBlocks;
    if (blocks > cap) blocks = cap;
    return static_cast<int>(blocks);
}

template <typename T, int OP>
int launch_ew(const char *name, int64_t n, Alpha<T> a, const T *x, T *y, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length %lld", name, (long long)n);
    if (n == 0) return LSK_OK; // "If there are no points to process, exit." (.cu:40-41)
    if (!y || (reads_x(OP) && !x)) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    if (a.nf > 4) return lsk::fail(LSK_ERR_ARG, "%s: %d futures (0..4 allowed)", name, a.nf);
    const T *fs[4] = {a.f0, a.f1, a.f2, a.f3};
    for (int i = 0; i < a.nf; ++i)
        if (!fs[i]) return lsk::fail(LSK_ERR_ARG, "%s: future %d is null", name, i);
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const bool vec = lsk::aligned16(y) && (!reads_x(OP) || lsk::aligned16(x));
    hipStream_t s = lsk::as_stream(stream);
    const int form = lsk::options().blas1_form;
    if (vec && stream_past_caches(n, sizeof(T))) {
        ew_kernel<T, OP, V, VN, true><<<stream_grid(n / VN + 1, true), kBlock, 0, s>>>(n, a, x, y, form);
    } else if (vec) {
        ew_kernel<T, OP, V, VN><<<stream_grid(n / VN + 1), kBlock, 0, s>>>(n, a, x, y, form);
    } else {
        ew_kernel<T, OP, T, 1><<<stream_grid(n), kBlock, 0, s>>>(n, a, x, y, form);
    }
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

template <typename T>
Alpha<T> host_alpha(T v) { return Alpha<T>{-1, v, nullptr, nullptr, nullptr, nullptr}; }

template <typename T>
Alpha<T> dev_alpha(int nf, const T *f0, const T *f1, const T *f2, const T *f3) {
    return Alpha<T>{nf < 0 ? 5 : nf, T(0), f0, f1, f2, f3}; // nf < 0 is rejected as > 4
}

// ------------------------------------------------------------------ dot / nrm2 -------

enum Finish { FIN_NONE = 0, FIN_SQRT = 1 };

template <typename T>
__device__ __forceinline__ T finish(T v, int fin) {
    if (fin == FIN_SQRT) return sqrt(v);
    return v;
}

// Sum of the block's values, valid in thread 0.  Order: butterfly inside each wave, then
// waves 0..3 left to right; fixed for a given launch shape, so results are reproducible.
template <typename T>
__device__ __forceinline__ T block_sum(T v) {
    __shared__ T s_wave[kBlock / lsk::kWave];
    v = lsk::wave_sum(v);
    const int lane = threadIdx.x % lsk::kWave, wave = threadIdx.x / lsk::kWave;
    if (lane == 0) s_wave[wave] = v;
    __syncthreads();
    T total = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / lsk::kWave; ++w) total += s_wave[w];
    }
    return total;
}

template <typename T> __device__ __forceinline__ T dot_v(double2v a, double2v b, T acc) {
    acc = fma_t(a.x, b.x, acc);
    return fma_t(a.y, b.y, acc);
}
template <typename T> __device__ __forceinline__ T dot_v(float4v a, float4v b, T acc) {
    acc = fma_t(a.x, b.x, acc);
    acc = fma_t(a.y, b.y, acc);
    acc = fma_t(a.z, b.z, acc);
    return fma_t(a.w, b.w, acc);
}
template <typename T> __device__ __forceinline__ T dot_v(T a, T b, T acc) { return fma_t(a, b, acc); }

// SAME: w is v (dot(v,v), nrm2): the vector is read once.
template <typename T, typename E, int EN, bool SAME, bool NT = false>
__global__ __launch_bounds__(kBlock) void dot_kernel(int64_t n, const T *__restrict__ v,
                                                     const T *__restrict__ w,
                                                     T *__restrict__ out, int fin, int form) {
    const int64_t ne = n / EN;
    const E *ve = reinterpret_cast<const E *>(v);
    const E *we = reinterpret_cast<const E *>(w);
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    const int64_t inner = form ? kBlock : gstride;
    const int64_t outer = 4 * gstride;
    int64_t i = form ? int64_t(blockIdx.x) * (4 * kBlock) + threadIdx.x
                     : int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (; i + 3 * inner < ne; i += outer) {
        E a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = ld<NT>(ve + i + u * inner);
            if constexpr (!SAME) b[u] = ld<NT>(we + i + u * inner);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = dot_v<T>(a[u], SAME ? a[u] : b[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = i + u * inner;
        if (j < ne) {
            E a = ld<NT>(ve + j);
            E b = a;
            if constexpr (!SAME) b = ld<NT>(we + j);
            acc[u] = dot_v<T>(a, b, acc[u]);
        }
    }
    if (EN > 1 && first) {
        for (int64_t j = ne * EN; j < n; ++j) acc[1] = fma_t(v[j], SAME ? v[j] : w[j], acc[1]);
    }
    const T total = block_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (threadIdx.x == 0) out[blockIdx.x] = (gridDim.x == 1) ? finish(total, fin) : total;
}

// Second stage: one workgroup folds the per-block partials in a fixed order.
template <typename T>
__global__ __launch_bounds__(kBlock) void fold_kernel(int count, const T *__restrict__ partials,
                                                      T *__restrict__ out, int fin) {
    T acc = T(0);
    for (int i = threadIdx.x; i < count; i += kBlock) acc += partials[i];
    const T total = block_sum(acc);
    if (threadIdx.x == 0) *out = finish(total, fin);
}

// The same for counts that one small workgroup would take too long over and a first stage of its own is
// too much for (a kernel more on a solver step's critical path): 1024 threads, each adds its share of the
// partials left to right with eight loads in flight, then butterfly inside each wave and waves 0..15 left to
// right.  The order depends on the count alone.
constexpr int kWideFold = 1024;
template <typename T>
__global__ __launch_bounds__(kWideFold) void fold_wide_kernel(int count, const T *__restrict__ partials,
                                                              T *__restrict__ out, int fin) {
    __shared__ T s_wave[kWideFold / lsk::kWave];
    T acc = T(0);
    int i = threadIdx.x;
    for (; i + 7 * kWideFold < count; i += 8 * kWideFold) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[i + u * kWideFold];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < count; i += kWideFold) acc += partials[i];
    acc = lsk::wave_sum(acc);
    const int lane = threadIdx.x % lsk::kWave, wave = threadIdx.x / lsk::kWave;
    if (lane == 0) s_wave[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T total = T(0);
#pragma unroll
        for (int w = 0; w < kWideFold / lsk::kWave; ++w) total += s_wave[w];
        *out = finish(total, fin);
    }
}

template <typename T>
__global__ void store_scalar_kernel(T *out, T value) { *out = value; }

template <typename T>
int launch_dot(const char *name, int64_t n, const T *v, const T *w, T *result, void *workspace,
               size_t workspace_bytes, int fin, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length %lld", name, (long long)n);
    if (!result) return lsk::fail(LSK_ERR_ARG, "%s: null result", name);
    hipStream_t s = lsk::as_stream(stream);
    if (n == 0) { // empty domain -> 0 (src/LinearAlgebraTasks.cu:213-214)
        store_scalar_kernel<T><<<1, 1, 0, s>>>(result, T(0));
        LSK_LAUNCH_CHECK();
        return LSK_OK;
    }
    if (!v || !w) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const bool same = (v == w);
    const int form = lsk::options().blas1_form;
    const bool vec = lsk::aligned16(v) && lsk::aligned16(w);
    const int blocks = vec ? stream_grid(n / VN + 1) : stream_grid(n);
    T *partials = result;
    if (blocks > 1) {
        const size_t need = size_t(blocks) * sizeof(T);
        if (workspace) {
            if (workspace_bytes < need)
                return lsk::fail(LSK_ERR_ARG, "%s: workspace of %zu bytes, %zu needed", name,
                                 workspace_bytes, need);
            partials = static_cast<T *>(workspace);
        } else {
            lsk::DeviceState *st = nullptr;
            int rc = lsk::device_state(&st);
            if (rc != LSK_OK) return rc;
            partials = static_cast<T *>(st->scratch);
        }
    }
    if (vec && stream_past_caches(n, sizeof(T))) {
        if (same) dot_kernel<T, V, VN, true, true><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
        else dot_kernel<T, V, VN, false, true><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
    } else if (vec) {
        if (same) dot_kernel<T, V, VN, true><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
        else dot_kernel<T, V, VN, false><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
    } else {
        if (same) dot_kernel<T, T, 1, true><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
        else dot_kernel<T, T, 1, false><<<blocks, kBlock, 0, s>>>(n, v, w, partials, fin, form);
    }
    LSK_LAUNCH_CHECK();
    if (blocks > 1) {
        fold_kernel<T><<<1, kBlock, 0, s>>>(blocks, partials, result, fin);
        LSK_LAUNCH_CHECK();
    }
    return LSK_OK;
}

// ------------------------------------------------- fused axpy + squared norm, sums -------
//
// y <- fma(alpha, x, y) as the axpy above, and out[block] <- this block's share of sum(y_new^2):
// the values are in registers anyway, so the reduction that usually follows an axpy in a Krylov
// step (CG's r <- r - a q; rr = r.r) costs no second pass over y.  The partials are summed by
// sum_partials when (and if) the caller wants the norm.
// OTHER: out[block] <- share of sum(y_new * w) for a third vector w instead (GMRES's modified
// Gram-Schmidt: w' <- w' - h v_k, then v_{k+1}.w').
template <typename T, typename E, int EN, bool OTHER = false, bool NT = false>
__global__ __launch_bounds__(kBlock) void axpy_sqnorm_kernel(int64_t n, Alpha<T> a, const T *x, T *y,
                                                             T *__restrict__ out, int form,
                                                             const T *w = nullptr) {
    const T alpha = resolve_alpha(a);
    const int64_t ne = n / EN;
    const E *xe = reinterpret_cast<const E *>(x);
    const E *we = reinterpret_cast<const E *>(w);
    E *ye = reinterpret_cast<E *>(y);
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    const int64_t inner = form ? kBlock : gstride;
    const int64_t outer = 4 * gstride;
    int64_t i = form ? int64_t(blockIdx.x) * (4 * kBlock) + threadIdx.x
                     : int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (; i + 3 * inner < ne; i += outer) {
        E xv[4], yv[4], wv[OTHER ? 4 : 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = ld<NT>(xe + i + u * inner);
            yv[u] = ld<NT>(ye + i + u * inner);
            if constexpr (OTHER) wv[u] = ld<NT>(we + i + u * inner);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            E r;
            if constexpr (EN == 1) r = apply<T, OP_AXPY>(alpha, xv[u], yv[u]);
            else r = apply_v<T, OP_AXPY>(alpha, xv[u], yv[u]);
            st<NT>(ye + i + u * inner, r);
            if constexpr (OTHER) acc[u] = dot_v<T>(r, wv[u], acc[u]);
            else acc[u] = dot_v<T>(r, r, acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = i + u * inner;
        if (j < ne) {
            E r;
            if constexpr (EN == 1) r = apply<T, OP_AXPY>(alpha, ld<NT>(xe + j), ld<NT>(ye + j));
            else r = apply_v<T, OP_AXPY>(alpha, ld<NT>(xe + j), ld<NT>(ye + j));
            st<NT>(ye + j, r);
            if constexpr (OTHER) acc[u] = dot_v<T>(r, ld<NT>(we + j), acc[u]);
            else acc[u] = dot_v<T>(r, r, acc[u]);
        }
    }
    if (EN > 1 && first) {
        for (int64_t j = ne * EN; j < n; ++j) {
            const T r = apply<T, OP_AXPY>(alpha, x[j], y[j]);
            y[j] = r;
            acc[1] = fma_t(r, OTHER ? w[j] : r, acc[1]);
        }
    }
    const T total = block_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (threadIdx.x == 0) out[blockIdx.x] = total;
}

constexpr int64_t kMaxPartials = LSK_MAX_PARTIALS; // partial sums a fused update leaves at most (LSK_MAX_PARTIALS)

template <typename T>
int launch_axpy_sqnorm(const char *name, int64_t n, Alpha<T> a, const T *x, T *y, T *partials,
                       int64_t capacity, int *count, lsk_stream_t stream, const T *w = nullptr) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length %lld", name, (long long)n);
    if (!count) return lsk::fail(LSK_ERR_ARG, "%s: null count", name);
    *count = 0;
    if (n == 0) return LSK_OK;
    if (!x || !y || !partials) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    if (x == y) return lsk::fail(LSK_ERR_ARG, "%s: x and y must differ", name);
    if (a.nf > 4) return lsk::fail(LSK_ERR_ARG, "%s: %d futures (0..4 allowed)", name, a.nf);
    const T *fs[4] = {a.f0, a.f1, a.f2, a.f3};
    for (int i = 0; i < a.nf; ++i)
        if (!fs[i]) return lsk::fail(LSK_ERR_ARG, "%s: future %d is null", name, i);
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    if (w == y) w = nullptr; // the vector's own squared norm
    const bool vec = lsk::aligned16(y) && lsk::aligned16(x) && (!w || lsk::aligned16(w));
    const bool nt = vec && stream_past_caches(n, sizeof(T));
    // Vectors that no cache holds run faster the more workgroups share them (one 16 KiB chunk each: 3-5 % over
    // 2048 looping workgroups, tools/blas1_shapes.py); every workgroup leaves one partial sum, so the room the
    // caller offers decides: 2048 slots are always enough, up to kMaxPartials are used when they are there.
    const int least = vec ? stream_grid(n / VN + 1) : stream_grid(n);
    if (capacity < least)
        return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %d needed", name, (long long)capacity, least);
    int blocks = least;
    if (nt) {
        const int64_t most = std::min<int64_t>(std::min<int64_t>(stream_grid(n / VN + 1, true), capacity), kMaxPartials);
        if (most > blocks) blocks = static_cast<int>(most);
    }
    hipStream_t s = lsk::as_stream(stream);
    const int form = lsk::options().blas1_form;
    if (w) {
        if (nt) axpy_sqnorm_kernel<T, V, VN, true, true><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form, w);
        else if (vec) axpy_sqnorm_kernel<T, V, VN, true><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form, w);
        else axpy_sqnorm_kernel<T, T, 1, true><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form, w);
    } else if (nt) axpy_sqnorm_kernel<T, V, VN, false, true><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form);
    else if (vec) axpy_sqnorm_kernel<T, V, VN><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form);
    else axpy_sqnorm_kernel<T, T, 1><<<blocks, kBlock, 0, s>>>(n, a, x, y, partials, form);
    LSK_LAUNCH_CHECK();
    *count = blocks;
    return LSK_OK;
}

// Two updates that read the same vector in one pass: z <- fma(a, p, z) and p <- fma(b, p, r), both
// from the OLD p (CG's x <- x + alpha p followed by p <- r + beta p, src/CGSolver.hpp:48,54, where the
// first does not depend on anything computed in between).  5 instead of 6 vector passes.
template <typename T, typename E, int EN, bool NT = false>
__global__ __launch_bounds__(kBlock) void axpy_xpay_kernel(int64_t n, Alpha<T> a, Alpha<T> b, const T *r, T *p,
                                                           T *z, int form) {
    const T alpha = resolve_alpha(a), beta = resolve_alpha(b);
    const int64_t ne = n / EN;
    const E *re = reinterpret_cast<const E *>(r);
    E *pe = reinterpret_cast<E *>(p);
    E *ze = reinterpret_cast<E *>(z);
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    const int64_t inner = form ? kBlock : gstride;
    const int64_t outer = 4 * gstride;
    int64_t i = form ? int64_t(blockIdx.x) * (4 * kBlock) + threadIdx.x
                     : int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
    auto both = [&](const E &rv, const E &pv, const E &zv, E &p_out, E &z_out) {
        if constexpr (EN == 1) {
            z_out = apply<T, OP_AXPY>(alpha, pv, zv);
            p_out = apply<T, OP_XPAY>(beta, rv, pv);
        } else {
            z_out = apply_v<T, OP_AXPY>(alpha, pv, zv);
            p_out = apply_v<T, OP_XPAY>(beta, rv, pv);
        }
    };
    for (; i + 3 * inner < ne; i += outer) {
        E rv[4], pv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rv[u] = ld<NT>(re + i + u * inner);
            pv[u] = ld<NT>(pe + i + u * inner);
            zv[u] = ld<NT>(ze + i + u * inner);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            E po, zo;
            both(rv[u], pv[u], zv[u], po, zo);
            st<NT>(ze + i + u * inner, zo);
            st<NT>(pe + i + u * inner, po);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = i + u * inner;
        if (j < ne) {
            E po, zo;
            both(ld<NT>(re + j), ld<NT>(pe + j), ld<NT>(ze + j), po, zo);
            st<NT>(ze + j, zo);
            st<NT>(pe + j, po);
        }
    }
    if (EN > 1 && first) {
        for (int64_t j = ne * EN; j < n; ++j) {
            const T pv = p[j];
            z[j] = apply<T, OP_AXPY>(alpha, pv, z[j]);
            p[j] = apply<T, OP_XPAY>(beta, r[j], pv);
        }
    }
}

template <typename T>
int launch_axpy_xpay(const char *name, int64_t n, Alpha<T> a, Alpha<T> b, const T *r, T *p, T *z,
                     lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length %lld", name, (long long)n);
    if (n == 0) return LSK_OK;
    if (!r || !p || !z) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    if (r == p || p == z || r == z) return lsk::fail(LSK_ERR_ARG, "%s: the three vectors must differ", name);
    for (const Alpha<T> *c : {&a, &b}) {
        if (c->nf > 4) return lsk::fail(LSK_ERR_ARG, "%s: %d futures (0..4 allowed)", name, c->nf);
        const T *fs[4] = {c->f0, c->f1, c->f2, c->f3};
        for (int i = 0; i < c->nf; ++i)
            if (!fs[i]) return lsk::fail(LSK_ERR_ARG, "%s: future %d is null", name, i);
    }
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    const bool vec = lsk::aligned16(r) && lsk::aligned16(p) && lsk::aligned16(z);
    const int blocks = vec ? stream_grid(n / VN + 1, stream_past_caches(n, sizeof(T))) : stream_grid(n);
    hipStream_t s = lsk::as_stream(stream);
    const int form = lsk::options().blas1_form;
    if (vec && stream_past_caches(n, sizeof(T))) axpy_xpay_kernel<T, V, VN, true><<<blocks, kBlock, 0, s>>>(n, a, b, r, p, z, form);
    else if (vec) axpy_xpay_kernel<T, V, VN><<<blocks, kBlock, 0, s>>>(n, a, b, r, p, z, form);
    else axpy_xpay_kernel<T, T, 1><<<blocks, kBlock, 0, s>>>(n, a, b, r, p, z, form);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// First stage of a long sum: block b adds the partials b*per .. (b+1)*per-1 left to right per
// thread, then across the block in block_sum's fixed order.
template <typename T>
__global__ __launch_bounds__(kBlock) void sum_kernel(int64_t count, int64_t per, const T *__restrict__ in,
                                                     T *__restrict__ out) {
    const int64_t lo = int64_t(blockIdx.x) * per;
    const int64_t hi = lo + per < count ? lo + per : count;
    T acc = T(0);
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) acc += in[i];
    const T total = block_sum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = total;
}

// *result <- sum of `count` partial sums, in an order fixed by count alone: up to 8192 by one workgroup of 256,
// up to kWideFoldMax by one of 1024, beyond that in two stages.
constexpr int64_t kWideFoldMax = 65536;
template <typename T>
int launch_sum_partials(const char *name, int64_t count, const T *partials, T *result, void *workspace,
                        size_t workspace_bytes, lsk_stream_t stream) {
    if (count < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative count", name);
    if (!result) return lsk::fail(LSK_ERR_ARG, "%s: null result", name);
    hipStream_t s = lsk::as_stream(stream);
    if (count == 0) {
        store_scalar_kernel<T><<<1, 1, 0, s>>>(result, T(0));
        LSK_LAUNCH_CHECK();
        return LSK_OK;
    }
    if (!partials) return lsk::fail(LSK_ERR_ARG, "%s: null partials", name);
    if (count > 4 * lsk::kMaxGridBlocks && count <= kWideFoldMax) {
        fold_wide_kernel<T><<<1, kWideFold, 0, s>>>(static_cast<int>(count), partials, result, FIN_NONE);
    } else if (count > 4 * lsk::kMaxGridBlocks) { // two stages
        const size_t need = size_t(lsk::kMaxGridBlocks) * sizeof(T);
        T *stage = static_cast<T *>(workspace);
        if (stage) {
            if (workspace_bytes < need)
                return lsk::fail(LSK_ERR_ARG, "%s: workspace of %zu bytes, %zu needed", name, workspace_bytes, need);
        } else {
            lsk::DeviceState *st = nullptr;
            int rc = lsk::device_state(&st);
            if (rc != LSK_OK) return rc;
            stage = static_cast<T *>(st->scratch);
        }
        const int64_t per = lsk::ceil_div(count, int64_t(lsk::kMaxGridBlocks));
        const int blocks = static_cast<int>(lsk::ceil_div(count, per));
        sum_kernel<T><<<blocks, kBlock, 0, s>>>(count, per, partials, stage);
        LSK_LAUNCH_CHECK();
        fold_kernel<T><<<1, kBlock, 0, s>>>(blocks, stage, result, FIN_NONE);
    } else {
        fold_kernel<T><<<1, kBlock, 0, s>>>(static_cast<int>(count), partials, result, FIN_NONE);
    }
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// ------------------------------------------------- several dot products / updates in one pass ---
//
// GMRES orthogonalises w against v_0 .. v_j.  The reference's modified Gram-Schmidt sweep
// (src/GMRESSolver.hpp:92-100) alternates h_k = v_k . w and w <- w - h_k v_k, two passes over w per
// basis vector.  In the classical variant every h_k comes from the same w, so all of them can be
// formed in ONE pass over w (multi_dot), and all updates applied in one more (multi_axpy): a group of
// up to kGroup vectors per launch, w read once per group.
constexpr int kGroup = 8;
template <typename T> struct PtrGroup { const T *p[kGroup]; };

// out[j * gridDim.x + block] <- this block's share of sum_i v_j[i] * w[i], j < k.
template <typename T, typename E, int EN, bool NT>
__global__ __launch_bounds__(kBlock) void multi_dot_kernel(int64_t n, PtrGroup<T> v, int k, const T *__restrict__ w,
                                                           T *__restrict__ out) {
    __shared__ T s_wave[kGroup][kBlock / lsk::kWave];
    const int64_t ne = n / EN;
    const E *we = reinterpret_cast<const E *>(w);
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    T acc[kGroup];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) acc[j] = T(0);
    // two accesses of w per lane in flight, and with each the k accesses of the v_j
    int64_t i = int64_t(blockIdx.x) * (2 * kBlock) + threadIdx.x;
    for (; i < ne; i += 2 * gstride) {
        const bool second = i + kBlock < ne;
        E wv[2], vv[2][kGroup];
        wv[0] = ld<NT>(we + i);
        if (second) wv[1] = ld<NT>(we + i + kBlock);
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            if (j < k) {
                const E *ve = reinterpret_cast<const E *>(v.p[j]);
                vv[0][j] = ld<NT>(ve + i);
                if (second) vv[1][j] = ld<NT>(ve + i + kBlock);
            }
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            if (j < k) {
                acc[j] = dot_v<T>(vv[0][j], wv[0], acc[j]);
                if (second) acc[j] = dot_v<T>(vv[1][j], wv[1], acc[j]);
            }
    }
    if (EN > 1 && blockIdx.x == 0 && threadIdx.x == 0) { // the n % EN trailing elements
        for (int64_t t = ne * EN; t < n; ++t)
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                if (j < k) acc[j] = fma_t(v.p[j][t], w[t], acc[j]);
    }
    const int lane = threadIdx.x % lsk::kWave, wave = threadIdx.x / lsk::kWave;
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
        const T total = lsk::wave_sum(acc[j]);
        if (lane == 0) s_wave[j][wave] = total;
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < k) {
        T total = T(0);
#pragma unroll
        for (int q = 0; q < kBlock / lsk::kWave; ++q) total += s_wave[threadIdx.x][q];
        out[int64_t(threadIdx.x) * gridDim.x + blockIdx.x] = total;
    }
}

// results[j] <- the sum of row j of partials (count values each, rows `stride` apart), one workgroup
// per row, in fold_kernel's fixed order.
template <typename T>
__global__ __launch_bounds__(kBlock) void multi_fold_kernel(int count, int64_t stride, const T *__restrict__ partials,
                                                            T *__restrict__ results) {
    const T *row = partials + int64_t(blockIdx.x) * stride;
    T acc = T(0);
    for (int i = threadIdx.x; i < count; i += kBlock) acc += row[i];
    const T total = block_sum(acc);
    if (threadIdx.x == 0) results[blockIdx.x] = total;
}

// w <- fma(c_{k-1}, v_{k-1}, ... fma(c_1, v_1, fma(c_0, v_0, w))), c_j = coef[j] (negated when
// `negate`): element for element the chain of AxpyTasks w <- w + c_j v_j, j ascending.  SQ: out[block]
// <- this block's share of sum(w_new^2) as well.
template <typename T, typename E, int EN, bool NT, bool SQ>
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(int64_t n, PtrGroup<T> v, int k, const T *__restrict__ coef,
                                                            int negate, T *w, T *__restrict__ out) {
    T c[kGroup];
#pragma unroll
    for (int j = 0; j < kGroup; ++j) c[j] = j < k ? (negate ? -coef[j] : coef[j]) : T(0);
    const int64_t ne = n / EN;
    E *we = reinterpret_cast<E *>(w);
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    T acc = T(0);
    int64_t i = int64_t(blockIdx.x) * (2 * kBlock) + threadIdx.x;
    for (; i < ne; i += 2 * gstride) {
        const bool second = i + kBlock < ne;
        E wv[2], vv[2][kGroup];
        wv[0] = ld<NT>(we + i);
        if (second) wv[1] = ld<NT>(we + i + kBlock);
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            if (j < k) {
                const E *ve = reinterpret_cast<const E *>(v.p[j]);
                vv[0][j] = ld<NT>(ve + i);
                if (second) vv[1][j] = ld<NT>(ve + i + kBlock);
            }
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            if (j < k) {
                if constexpr (EN == 1) {
                    wv[0] = apply<T, OP_AXPY>(c[j], vv[0][j], wv[0]);
                    if (second) wv[1] = apply<T, OP_AXPY>(c[j], vv[1][j], wv[1]);
                } else {
                    wv[0] = apply_v<T, OP_AXPY>(c[j], vv[0][j], wv[0]);
                    if (second) wv[1] = apply_v<T, OP_AXPY>(c[j], vv[1][j], wv[1]);
                }
            }
        st<NT>(we + i, wv[0]);
        if (second) st<NT>(we + i + kBlock, wv[1]);
        if constexpr (SQ) {
            acc = dot_v<T>(wv[0], wv[0], acc);
            if (second) acc = dot_v<T>(wv[1], wv[1], acc);
        }
    }
    if (EN > 1 && blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t t = ne * EN; t < n; ++t) {
            T r = w[t];
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                if (j < k) r = fma_t(c[j], v.p[j][t], r);
            w[t] = r;
            if constexpr (SQ) acc = fma_t(r, r, acc);
        }
    }
    if constexpr (SQ) {
        const T total = block_sum(acc);
        if (threadIdx.x == 0) out[blockIdx.x] = total;
    }
}

inline int multi_grid(int64_t accesses) {
    int64_t blocks = lsk::ceil_div(accesses, int64_t(kBlock) * 2);
    if (blocks < 1) blocks = 1;
    if (blocks > lsk::kMaxGridBlocks) blocks = lsk::kMaxGridBlocks;
    return static_cast<int>(blocks);
}

constexpr int kMaxMulti = 64; // vectors per call

template <typename T>
int check_multi(const char *name, int64_t n, int k, const T *const *v, const T *w) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length %lld", name, (long long)n);
    if (k < 0 || k > kMaxMulti) return lsk::fail(LSK_ERR_ARG, "%s: %d vectors (0..%d allowed)", name, k, kMaxMulti);
    if (k > 0 && !v) return lsk::fail(LSK_ERR_ARG, "%s: null vector list", name);
    if (n > 0) {
        if (!w) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
        for (int j = 0; j < k; ++j)
            if (!v[j]) return lsk::fail(LSK_ERR_ARG, "%s: vector %d is null", name, j);
    }
    return LSK_OK;
}

template <typename T>
int launch_multi_dot(const char *name, int64_t n, int k, const T *const *v, const T *w, T *results, void *workspace,
                     size_t workspace_bytes, lsk_stream_t stream) {
    int rc = check_multi<T>(name, n, k, v, w);
    if (rc != LSK_OK) return rc;
    if (k == 0) return LSK_OK;
    if (!results) return lsk::fail(LSK_ERR_ARG, "%s: null results", name);
    hipStream_t s = lsk::as_stream(stream);
    if (n == 0) {
        for (int j = 0; j < k; ++j) store_scalar_kernel<T><<<1, 1, 0, s>>>(results + j, T(0));
        LSK_LAUNCH_CHECK();
        return LSK_OK;
    }
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    bool vec = lsk::aligned16(w);
    for (int j = 0; j < k; ++j) vec = vec && lsk::aligned16(v[j]);
    const int blocks = vec ? multi_grid(n / VN + 1) : multi_grid(n);
    const size_t need = size_t(k) * size_t(blocks) * sizeof(T);
    if (!workspace || workspace_bytes < need)
        return lsk::fail(LSK_ERR_ARG, "%s: workspace of %zu bytes, %zu needed", name, workspace ? workspace_bytes : size_t(0),
                         need);
    T *partials = static_cast<T *>(workspace);
    const bool nt = vec && stream_past_caches(n, sizeof(T));
    for (int j0 = 0; j0 < k; j0 += kGroup) {
        const int kk = k - j0 < kGroup ? k - j0 : kGroup;
        PtrGroup<T> g{};
        for (int j = 0; j < kk; ++j) g.p[j] = v[j0 + j];
        T *out = partials + size_t(j0) * blocks;
        if (nt) multi_dot_kernel<T, V, VN, true><<<blocks, kBlock, 0, s>>>(n, g, kk, w, out);
        else if (vec) multi_dot_kernel<T, V, VN, false><<<blocks, kBlock, 0, s>>>(n, g, kk, w, out);
        else multi_dot_kernel<T, T, 1, false><<<blocks, kBlock, 0, s>>>(n, g, kk, w, out);
        LSK_LAUNCH_CHECK();
    }
    multi_fold_kernel<T><<<k, kBlock, 0, s>>>(blocks, blocks, partials, results);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

template <typename T>
int launch_multi_axpy(const char *name, int64_t n, int k, const T *const *v, const T *coef, int negate, T *w, T *partials,
                      int64_t capacity, int *count, lsk_stream_t stream) {
    int rc = check_multi<T>(name, n, k, v, w);
    if (rc != LSK_OK) return rc;
    if (count) *count = 0;
    if (partials && !count) return lsk::fail(LSK_ERR_ARG, "%s: partials without a count", name);
    if (n == 0 || (k == 0 && !partials)) return LSK_OK;
    if (k > 0 && !coef) return lsk::fail(LSK_ERR_ARG, "%s: null coefficients", name);
    for (int j = 0; j < k; ++j)
        if (v[j] == w) return lsk::fail(LSK_ERR_ARG, "%s: vector %d is the destination", name, j);
    using V = typename Vec<T>::type;
    constexpr int VN = Vec<T>::N;
    bool vec = lsk::aligned16(w);
    for (int j = 0; j < k; ++j) vec = vec && lsk::aligned16(v[j]);
    int blocks = vec ? multi_grid(n / VN + 1) : multi_grid(n);
    if (partials && capacity < blocks)
        return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %d needed", name, (long long)capacity, blocks);
    hipStream_t s = lsk::as_stream(stream);
    const bool nt = vec && stream_past_caches(n, sizeof(T));
    // (vectors that no cache holds: one trip per workgroup, as the plain element-wise streams -- as far as the
    // caller has room for the partial sums, where it wants them: see launch_axpy_sqnorm)
    if (nt) {
        int64_t most = std::min<int64_t>(lsk::ceil_div(n / VN + 1, int64_t(kBlock) * 2), int64_t(1) << 30);
        if (partials) most = std::min<int64_t>(std::min<int64_t>(most, capacity), kMaxPartials);
        if (most > blocks) blocks = static_cast<int>(most);
    }
    const int groups = k == 0 ? 1 : (k + kGroup - 1) / kGroup;
    for (int gi = 0; gi < groups; ++gi) {
        const int j0 = gi * kGroup;
        const int kk = k - j0 < kGroup ? k - j0 : kGroup;
        PtrGroup<T> g{};
        for (int j = 0; j < kk; ++j) g.p[j] = v[j0 + j];
        const bool sq = partials && gi == groups - 1; // the norm of the final values only
        const T *cf = coef ? coef + j0 : nullptr;
#define LSK_MULTI_AXPY(E, EN, NT)                                                                            \
    do {                                                                                                     \
        if (sq) multi_axpy_kernel<T, E, EN, NT, true><<<blocks, kBlock, 0, s>>>(n, g, kk, cf, negate, w, partials); \
        else multi_axpy_kernel<T, E, EN, NT, false><<<blocks, kBlock, 0, s>>>(n, g, kk, cf, negate, w, partials);   \
    } while (0)
        if (nt) LSK_MULTI_AXPY(V, VN, true);
        else if (vec) LSK_MULTI_AXPY(V, VN, false);
        else LSK_MULTI_AXPY(T, 1, false);
#undef LSK_MULTI_AXPY
        LSK_LAUNCH_CHECK();
    }
    if (partials) *count = blocks;
    return LSK_OK;
}

template <typename T>
int dot_host(const char *name, int64_t n, const T *v, const T *w, T *result_host,
             lsk_stream_t stream) {
    if (!result_host) return lsk::fail(LSK_ERR_ARG, "%s: null result", name);
    *result_host = T(0);
    if (n == 0) return LSK_OK;
    LSK_NOT_WHILE_RECORDING(name, stream);
    lsk::DeviceState *st = nullptr;
    int rc = lsk::device_state(&st);
    if (rc != LSK_OK) return rc;
    // Last slot of the scratch holds the result; the partials use the slots before it.
    T *result_dev = reinterpret_cast<T *>(static_cast<char *>(st->scratch) + lsk::kScratchBytes) - 1;
    rc = launch_dot<T>(name, n, v, w, result_dev, st->scratch, lsk::kScratchBytes - 16, FIN_NONE,
                       stream);
    if (rc != LSK_OK) return rc;
    hipStream_t s = lsk::as_stream(stream);
    LSK_HIP(hipMemcpyAsync(result_host, result_dev, sizeof(T), hipMemcpyDeviceToHost, s));
    LSK_HIP(hipStreamSynchronize(s)); // the reference's cudaStreamSynchronize, .cu:235
    return LSK_OK;
}

// ------------------------------------------------------------------ allocation-free dot --
//
// Address of `scratch` on the current device: every GPU holds its own copy of the array at its own
// address, looked up once per device (here or from lsk_init) and kept.
constexpr int kNoAllocMaxDevices = 64;
void *g_noalloc_scratch[kNoAllocMaxDevices] = {};
std::mutex g_noalloc_mutex;

int noalloc_scratch(void **out) {
    int dev = -1;
    LSK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kNoAllocMaxDevices)
        return lsk::fail(LSK_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, kNoAllocMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_scratch[dev]) {
        void *ptr = nullptr;
        LSK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return lsk::fail(LSK_ERR_HIP, "hipGetSymbolAddress gave a null address for the dot scratch");
        g_noalloc_scratch[dev] = ptr;
    }
    *out = g_noalloc_scratch[dev];
    return LSK_OK;
}

// ------------------------------------------------------------------ scalar ops -------

template <typename T>
__global__ void scalar_op_kernel(int op, const T *a, const T *b, T *out) {
    const T x = *a;
    T r;
    switch (op) {
    case LSK_SOP_NEG: r = -x; break;
    case LSK_SOP_ADD: r = x + *b; break;
    case LSK_SOP_SUB: r = x - *b; break;
    case LSK_SOP_MUL: r = x * *b; break;
    case LSK_SOP_DIV: r = x / *b; break;
    case LSK_SOP_SQRT: r = sqrt(x); break;
    case LSK_SOP_RSQRT: r = static_cast<T>(1) / sqrt(x); break; // as RSqrtScalarTask, not rsqrt()
    default: r = x; break;
    }
    *out = r;
}

template <typename T>
int launch_scalar_op(int op, const T *a, const T *b, T *out, lsk_stream_t stream) {
    if (op < LSK_SOP_NEG || op > LSK_SOP_COPY)
        return lsk::fail(LSK_ERR_ARG, "lsk_scalar_op: unknown op %d", op);
    const bool binary = (op >= LSK_SOP_ADD && op <= LSK_SOP_DIV);
    if (!a || !out || (binary && !b)) return lsk::fail(LSK_ERR_ARG, "lsk_scalar_op: null operand");
    scalar_op_kernel<T><<<1, 1, 0, lsk::as_stream(stream)>>>(op, a, b, out);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

template <typename T>
__global__ void sum_scalars_kernel(int count, const T *in, T *out) {
    T acc = in[0];
    for (int i = 1; i < count; ++i) acc = acc + in[i];
    *out = acc;
}

template <typename T>
int launch_sum_scalars(int count, const T *in, T *out, lsk_stream_t stream) {
    if (count < 1) return lsk::fail(LSK_ERR_ARG, "lsk_sum_scalars: count %d < 1", count);
    if (!in || !out) return lsk::fail(LSK_ERR_ARG, "lsk_sum_scalars: null operand");
    sum_scalars_kernel<T><<<1, 1, 0, lsk::as_stream(stream)>>>(count, in, out);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// ------------------------------------------------------------------ N-D xpay ---------

struct NdShape {
    int64_t extent[3];
    int64_t xs[3];
    int64_t ys[3];
};

// One element per lane in row-major order of the (reordered) extents; the host orders the
// dimensions so that the innermost one has the smallest y stride.
template <typename T>
__global__ __launch_bounds__(kBlock) void xpay_nd_kernel(int64_t volume, NdShape sh, T alpha,
                                                         const T *x, T *y) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < volume; i += stride) {
        const int64_t c2 = i % sh.extent[2];
        const int64_t r = i / sh.extent[2];
        const int64_t c1 = r % sh.extent[1];
        const int64_t c0 = r / sh.extent[1];
        const int64_t xo = c0 * sh.xs[0] + c1 * sh.xs[1] + c2 * sh.xs[2];
        const int64_t yo = c0 * sh.ys[0] + c1 * sh.ys[1] + c2 * sh.ys[2];
        y[yo] = fma_t(alpha, y[yo], x[xo]);
    }
}

template <typename T>
int launch_xpay_nd(int dim, const int64_t *extent, T alpha, const T *x, const int64_t *x_stride,
                   T *y, const int64_t *y_stride, lsk_stream_t stream) {
    if (dim < 1 || dim > 3) return lsk::fail(LSK_ERR_ARG, "lsk_xpay_nd: dim %d outside 1..3", dim);
    if (!extent || !x_stride || !y_stride) return lsk::fail(LSK_ERR_ARG, "lsk_xpay_nd: null shape");
    NdShape sh;
    int64_t volume = 1;
    // Leading unit dimensions pad the shape to 3-D.
    for (int d = 0; d < 3; ++d) {
        const int src = d - (3 - dim);
        sh.extent[d] = src >= 0 ? extent[src] : 1;
        sh.xs[d] = src >= 0 ? x_stride[src] : 0;
        sh.ys[d] = src >= 0 ? y_stride[src] : 0;
        if (sh.extent[d] < 0) return lsk::fail(LSK_ERR_ARG, "lsk_xpay_nd: negative extent");
        volume *= sh.extent[d];
    }
    if (volume == 0) return LSK_OK;
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "lsk_xpay_nd: null vector");
    // Order dimensions by decreasing y stride (simple 3-element sort) for coalesced lanes.
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sh.ys[b] > sh.ys[a]) {
                int64_t t;
                t = sh.extent[a]; sh.extent[a] = sh.extent[b]; sh.extent[b] = t;
                t = sh.xs[a]; sh.xs[a] = sh.xs[b]; sh.xs[b] = t;
                t = sh.ys[a]; sh.ys[a] = sh.ys[b]; sh.ys[b] = t;
            }
    xpay_nd_kernel<T><<<stream_grid(volume), kBlock, 0, lsk::as_stream(stream)>>>(volume, sh, alpha, x, y);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // namespace

namespace lsk {
int noalloc_scratch_init() {
    void *ptr = nullptr;
    return noalloc_scratch(&ptr);
}
} // namespace lsk

extern "C" {

int lsk_dot_noalloc_f64(int64_t n, const double *v, const double *w, double *result_dev, lsk_stream_t stream) {
    if (n > LSK_DOT_NOALLOC_MAX_N)
        return lsk::fail(LSK_ERR_ARG, "lsk_dot_noalloc_f64: %lld elements, at most %lld are taken", (long long)n,
                         (long long)LSK_DOT_NOALLOC_MAX_N);
    void *partials = nullptr;
    const int rc = noalloc_scratch(&partials);
    if (rc != LSK_OK) return rc;
    // Every slot the fold reads has been written by this call's first stage, so nothing is cleared.
    return launch_dot<double>("lsk_dot_noalloc_f64", n, v, w, result_dev, partials, lsk::kScratchBytes, FIN_NONE,
                              stream);
}

size_t lsk_multi_dot_workspace_bytes(int k) {
    return size_t(k < 0 ? 0 : k) * size_t(lsk::kMaxGridBlocks) * sizeof(double);
}

size_t lsk_dot_workspace_bytes(int64_t n) {
    (void)n;
    return size_t(lsk::kMaxGridBlocks) * sizeof(double);
}

#define LSK_BLAS1(SUFFIX, T)                                                                      \
    int lsk_scal_##SUFFIX(int64_t n, T alpha, T *x, lsk_stream_t s) {                             \
        return launch_ew<T, OP_SCAL>("lsk_scal_" #SUFFIX, n, host_alpha<T>(alpha), nullptr, x, s);\
    }                                                                                             \
    int lsk_axpy_##SUFFIX(int64_t n, T alpha, const T *x, T *y, lsk_stream_t s) {                 \
        return launch_ew<T, OP_AXPY>("lsk_axpy_" #SUFFIX, n, host_alpha<T>(alpha), x, y, s);      \
    }                                                                                             \
    int lsk_xpay_##SUFFIX(int64_t n, T alpha, const T *x, T *y, lsk_stream_t s) {                 \
        return launch_ew<T, OP_XPAY>("lsk_xpay_" #SUFFIX, n, host_alpha<T>(alpha), x, y, s);      \
    }                                                                                             \
    int lsk_fill_##SUFFIX(int64_t n, T value, T *x, lsk_stream_t s) {                             \
        return launch_ew<T, OP_FILL>("lsk_fill_" #SUFFIX, n, host_alpha<T>(value), nullptr, x, s);\
    }                                                                                             \
    int lsk_copy_##SUFFIX(int64_t n, const T *src, T *dst, lsk_stream_t s) {                      \
        return launch_ew<T, OP_COPY>("lsk_copy_" #SUFFIX, n, host_alpha<T>(T(0)), src, dst, s);   \
    }                                                                                             \
    int lsk_scal_dev_##SUFFIX(int64_t n, int nf, const T *f0, const T *f1, const T *f2,           \
                              const T *f3, T *x, lsk_stream_t s) {                                \
        return launch_ew<T, OP_SCAL>("lsk_scal_dev_" #SUFFIX, n, dev_alpha<T>(nf, f0, f1, f2, f3),\
                                     nullptr, x, s);                                              \
    }                                                                                             \
    int lsk_axpy_dev_##SUFFIX(int64_t n, int nf, const T *f0, const T *f1, const T *f2,           \
                              const T *f3, const T *x, T *y, lsk_stream_t s) {                    \
        return launch_ew<T, OP_AXPY>("lsk_axpy_dev_" #SUFFIX, n, dev_alpha<T>(nf, f0, f1, f2, f3),\
                                     x, y, s);                                                    \
    }                                                                                             \
    int lsk_xpay_dev_##SUFFIX(int64_t n, int nf, const T *f0, const T *f1, const T *f2,           \
                              const T *f3, const T *x, T *y, lsk_stream_t s) {                    \
        return launch_ew<T, OP_XPAY>("lsk_xpay_dev_" #SUFFIX, n, dev_alpha<T>(nf, f0, f1, f2, f3),\
                                     x, y, s);                                                    \
    }                                                                                             \
    int lsk_xpay_nd_##SUFFIX(int dim, const int64_t *extent, T alpha, const T *x,                 \
                             const int64_t *xs, T *y, const int64_t *ys, lsk_stream_t s) {        \
        return launch_xpay_nd<T>(dim, extent, alpha, x, xs, y, ys, s);                            \
    }                                                                                             \
    int lsk_dot_##SUFFIX(int64_t n, const T *v, const T *w, T *r, void *ws, size_t wsb,           \
                         lsk_stream_t s) {                                                        \
        return launch_dot<T>("lsk_dot_" #SUFFIX, n, v, w, r, ws, wsb, FIN_NONE, s);               \
    }                                                                                             \
    int lsk_nrm2_##SUFFIX(int64_t n, const T *v, T *r, void *ws, size_t wsb, lsk_stream_t s) {    \
        return launch_dot<T>("lsk_nrm2_" #SUFFIX, n, v, v, r, ws, wsb, FIN_SQRT, s);              \
    }                                                                                             \
    int lsk_dot_host_##SUFFIX(int64_t n, const T *v, const T *w, T *r, lsk_stream_t s) {          \
        return dot_host<T>("lsk_dot_host_" #SUFFIX, n, v, w, r, s);                               \
    }                                                                                             \
    int lsk_scalar_op_##SUFFIX(int op, const T *a, const T *b, T *out, lsk_stream_t s) {          \
        return launch_scalar_op<T>(op, a, b, out, s);                                             \
    }                                                                                             \
    int lsk_sum_scalars_##SUFFIX(int count, const T *in, T *out, lsk_stream_t s) {                \
        return launch_sum_scalars<T>(count, in, out, s);                                          \
    }                                                                                             \
    int lsk_axpy_sqnorm_##SUFFIX(int64_t n, T alpha, const T *x, T *y, T *partials,               \
                                 int64_t capacity, int *count, lsk_stream_t s) {                  \
        return launch_axpy_sqnorm<T>("lsk_axpy_sqnorm_" #SUFFIX, n, host_alpha<T>(alpha), x, y,   \
                                     partials, capacity, count, s);                               \
    }                                                                                             \
    int lsk_axpy_sqnorm_dev_##SUFFIX(int64_t n, int nf, const T *f0, const T *f1, const T *f2,    \
                                     const T *f3, const T *x, T *y, T *partials,                  \
                                     int64_t capacity, int *count, lsk_stream_t s) {              \
        return launch_axpy_sqnorm<T>("lsk_axpy_sqnorm_dev_" #SUFFIX, n,                           \
                                     dev_alpha<T>(nf, f0, f1, f2, f3), x, y, partials, capacity,  \
                                     count, s);                                                   \
    }                                                                                             \
    int lsk_axpy_dot_dev_##SUFFIX(int64_t n, int nf, const T *f0, const T *f1, const T *f2,       \
                                  const T *f3, const T *x, T *y, const T *w, T *partials,         \
                                  int64_t capacity, int *count, lsk_stream_t s) {                 \
        if (!w && n > 0) return lsk::fail(LSK_ERR_ARG, "lsk_axpy_dot_dev_" #SUFFIX ": null w");   \
        return launch_axpy_sqnorm<T>("lsk_axpy_dot_dev_" #SUFFIX, n,                              \
                                     dev_alpha<T>(nf, f0, f1, f2, f3), x, y, partials, capacity,  \
                                     count, s, w);                                                \
    }                                                                                             \
    int lsk_axpy_xpay_dev_##SUFFIX(int64_t n, int nfa, const T *a0, const T *a1, const T *a2, const T *a3, \
                                   int nfb, const T *b0, const T *b1, const T *b2, const T *b3,   \
                                   const T *r, T *p, T *z, lsk_stream_t s) {                      \
        return launch_axpy_xpay<T>("lsk_axpy_xpay_dev_" #SUFFIX, n, dev_alpha<T>(nfa, a0, a1, a2, a3), \
                                   dev_alpha<T>(nfb, b0, b1, b2, b3), r, p, z, s);                \
    }                                                                                             \
    int lsk_multi_dot_##SUFFIX(int64_t n, int k, const T *const *v, const T *w, T *results, void *ws, \
                               size_t wsb, lsk_stream_t s) {                                      \
        return launch_multi_dot<T>("lsk_multi_dot_" #SUFFIX, n, k, v, w, results, ws, wsb, s);    \
    }                                                                                             \
    int lsk_multi_axpy_##SUFFIX(int64_t n, int k, const T *const *x, const T *coef, int negate,   \
                                T *y, T *partials, int64_t capacity, int *count, lsk_stream_t s) { \
        return launch_multi_axpy<T>("lsk_multi_axpy_" #SUFFIX, n, k, x, coef, negate, y, partials, \
                                    capacity, count, s);                                          \
    }                                                                                             \
    int lsk_sum_partials_##SUFFIX(int64_t count, const T *partials, T *result, void *ws,          \
                                  size_t wsb, lsk_stream_t s) {                                   \
        return launch_sum_partials<T>("lsk_sum_partials_" #SUFFIX, count, partials, result, ws,   \
                                      wsb, s);                                                    \
    }

LSK_BLAS1(f32, float)
LSK_BLAS1(f64, double)
#undef LSK_BLAS1

} // extern "C"
