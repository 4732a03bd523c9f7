// Device-side set-up helpers: the linearised stencil generators (what
// FillLinearizedCSRStencilTask / FillLinearizedCOOStencilTask write, row-major order,
// src/StencilGenerator.cpp:129-242,379-517), the row-rect builder and the image bounds of a
// column field.  Integer work only; results are bit-identical to the reference's serial
// per-piece walk, which starts from the grid origin for every piece and dominates set-up
// time at 512^3 -- here every row is generated independently.
#include <algorithm>
#include <climits>

#include <hipcub/hipcub.hpp>

#include "lsk_common.hpp"

namespace {

using lsk::kBlock;
constexpr int kMaxOffsets = 32;

// Grid padded to 3-D with leading unit extents; offsets padded with leading zeros.
struct Stencil {
    long long extent[3];
    int noffsets;
    long long off[kMaxOffsets][3];
};

template <typename T>
struct Coeffs {
    T c[kMaxOffsets];
};

__device__ __forceinline__ void decode_row(const Stencil &st, long long r, long long p[3]) {
    p[2] = r % st.extent[2];
    const long long q = r / st.extent[2];
    p[1] = q % st.extent[1];
    p[0] = q / st.extent[1];
}

__device__ __forceinline__ bool shifted_inside(const Stencil &st, const long long p[3], int o) {
    bool in = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const long long s = p[d] + st.off[o][d];
        in = in && s >= 0 && s < st.extent[d];
    }
    return in;
}

// linearize_row_major(point + offset) - linearize_row_major(point)
__device__ __forceinline__ long long linear_shift(const Stencil &st, int o) {
    return (st.off[o][0] * st.extent[1] + st.off[o][1]) * st.extent[2] + st.off[o][2];
}

__global__ __launch_bounds__(kBlock) void stencil_count_kernel(Stencil st, long long row_lo,
                                                               long long row_hi,
                                                               long long *__restrict__ counts) {
    const long long stride = static_cast<long long>(gridDim.x) * kBlock;
    for (long long r = row_lo + static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
         r <= row_hi; r += stride) {
        long long p[3];
        decode_row(st, r, p);
        long long n = 0;
        for (int o = 0; o < st.noffsets; ++o) n += shifted_inside(st, p, o) ? 1 : 0;
        counts[r - row_lo] = n;
    }
}

template <typename T, bool COO>
__global__ __launch_bounds__(kBlock) void stencil_fill_kernel(
    Stencil st, Coeffs<T> cf, long long row_lo, long long row_hi,
    const long long *__restrict__ rects, long long k_base, long long *__restrict__ row_out,
    long long *__restrict__ col, T *__restrict__ val) {
    const long long stride = static_cast<long long>(gridDim.x) * kBlock;
    for (long long r = row_lo + static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
         r <= row_hi; r += stride) {
        long long p[3];
        decode_row(st, r, p);
        long long k = rects[2 * (r - row_lo)] - k_base;
        for (int o = 0; o < st.noffsets; ++o) {
            if (shifted_inside(st, p, o)) {
                col[k] = r + linear_shift(st, o);
                val[k] = cf.c[o];
                if constexpr (COO) row_out[k] = r;
                ++k;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void rects_from_counts_kernel(
    long long rows, const long long *__restrict__ counts, const long long *__restrict__ prefix_incl,
    long long base, long long *__restrict__ rects) {
    const long long stride = static_cast<long long>(gridDim.x) * kBlock;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < rows;
         i += stride) {
        const long long end = base + prefix_incl[i]; // one past the row's last entry
        rects[2 * i] = end - counts[i];
        rects[2 * i + 1] = end - 1;
    }
}

__global__ void bounds_init_kernel(long long *out) {
    out[0] = LLONG_MAX;
    out[1] = LLONG_MIN;
}

__global__ __launch_bounds__(kBlock) void index_bounds_kernel(long long n,
                                                              const long long *__restrict__ idx,
                                                              long long *__restrict__ out) {
    __shared__ long long s_min[kBlock / lsk::kWave], s_max[kBlock / lsk::kWave];
    long long lo = LLONG_MAX, hi = LLONG_MIN;
    const long long stride = static_cast<long long>(gridDim.x) * kBlock;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        const long long v = idx[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int off = lsk::kWave / 2; off > 0; off >>= 1) {
        const long long olo = __shfl_xor(lo, off, lsk::kWave);
        const long long ohi = __shfl_xor(hi, off, lsk::kWave);
        lo = olo < lo ? olo : lo;
        hi = ohi > hi ? ohi : hi;
    }
    const int lane = threadIdx.x % lsk::kWave, wave = threadIdx.x / lsk::kWave;
    if (lane == 0) {
        s_min[wave] = lo;
        s_max[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / lsk::kWave; ++w) {
            lo = s_min[w] < lo ? s_min[w] : lo;
            hi = s_max[w] > hi ? s_max[w] : hi;
        }
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}

int grid_for(long long items) {
    long long blocks = lsk::ceil_div(items, kBlock);
    if (blocks < 1) blocks = 1;
    if (blocks > lsk::kMaxGridBlocks * 4) blocks = lsk::kMaxGridBlocks * 4;
    return static_cast<int>(blocks);
}

int make_stencil(const char *name, int dim, const int64_t *extent, int noffsets,
                 const int64_t *offsets, Stencil *st) {
    if (dim < 1 || dim > 3) return lsk::fail(LSK_ERR_ARG, "%s: dim %d outside 1..3", name, dim);
    if (!extent || (noffsets > 0 && !offsets)) return lsk::fail(LSK_ERR_ARG, "%s: null shape", name);
    if (noffsets < 0 || noffsets > kMaxOffsets)
        return lsk::fail(LSK_ERR_ARG, "%s: %d offsets (0..%d allowed)", name, noffsets, kMaxOffsets);
    const int pad = 3 - dim;
    for (int d = 0; d < 3; ++d) {
        st->extent[d] = d < pad ? 1 : extent[d - pad];
        if (st->extent[d] < 1) return lsk::fail(LSK_ERR_ARG, "%s: extent < 1", name);
    }
    st->noffsets = noffsets;
    for (int o = 0; o < noffsets; ++o)
        for (int d = 0; d < 3; ++d) st->off[o][d] = d < pad ? 0 : offsets[o * dim + (d - pad)];
    // The generator walks the offsets in sorted (lexicographic) order; the caller passes
    // them sorted so that col ids ascend within a row exactly as the reference writes them.
    for (int o = 1; o < noffsets; ++o) {
        bool greater = false;
        for (int d = 0; d < 3; ++d) {
            if (st->off[o - 1][d] < st->off[o][d]) break;
            if (st->off[o - 1][d] > st->off[o][d]) { greater = true; break; }
        }
        if (greater) return lsk::fail(LSK_ERR_ARG, "%s: offsets not sorted", name);
    }
    return LSK_OK;
}

template <typename T, bool COO>
int launch_fill(const char *name, int dim, const int64_t *extent, int noffsets,
                const int64_t *offsets, const T *coeffs, int64_t row_lo, int64_t row_hi,
                const int64_t *rects, int64_t k_base, int64_t *row, int64_t *col, T *val,
                lsk_stream_t stream) {
    Stencil st;
    int rc = make_stencil(name, dim, extent, noffsets, offsets, &st);
    if (rc != LSK_OK) return rc;
    if (row_hi < row_lo) return LSK_OK;
    const long long total = st.extent[0] * st.extent[1] * st.extent[2];
    if (row_lo < 0 || row_hi >= total)
        return lsk::fail(LSK_ERR_ARG, "%s: rows %lld..%lld outside the grid", name,
                         (long long)row_lo, (long long)row_hi);
    if (!coeffs || !rects || !col || !val || (COO && !row))
        return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    Coeffs<T> cf;
    for (int o = 0; o < kMaxOffsets; ++o) cf.c[o] = o < noffsets ? coeffs[o] : T(0);
    stencil_fill_kernel<T, COO><<<grid_for(row_hi - row_lo + 1), kBlock, 0, lsk::as_stream(stream)>>>(
        st, cf, row_lo, row_hi, reinterpret_cast<const long long *>(rects), k_base,
        reinterpret_cast<long long *>(row), reinterpret_cast<long long *>(col), val);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// Which rows of a CSR piece read columns outside lo..hi?  out[0] <- largest row index with a
// column below lo (-1 if none), out[1] <- smallest row index with a column above hi (rows if
// none).  One thread per row; rects are clipped to the piece.
__global__ void halo_rows_init_kernel(long long *out, long long rows) {
    out[0] = -1;
    out[1] = rows;
}

__global__ __launch_bounds__(kBlock) void halo_rows_kernel(long long rows, long long nnz,
                                                           const long long *__restrict__ rects,
                                                           long long k_base,
                                                           const long long *__restrict__ col,
                                                           long long lo, long long hi,
                                                           long long *__restrict__ out) {
    long long last_below = -1, first_above = rows;
    const long long stride = static_cast<long long>(gridDim.x) * kBlock;
    for (long long r = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride) {
        long long k0 = rects[2 * r] - k_base, k1 = rects[2 * r + 1] - k_base;
        k0 = k0 < 0 ? 0 : k0;
        k1 = k1 > nnz - 1 ? nnz - 1 : k1;
        bool below = false, above = false;
        for (long long k = k0; k <= k1; ++k) {
            const long long c = col[k];
            below = below || c < lo;
            above = above || c > hi;
        }
        if (below) last_below = r; // r ascends per thread
        if (above && r < first_above) first_above = r;
    }
#pragma unroll
    for (int off = lsk::kWave / 2; off > 0; off >>= 1) {
        const long long ob = __shfl_xor(last_below, off, lsk::kWave);
        const long long oa = __shfl_xor(first_above, off, lsk::kWave);
        last_below = ob > last_below ? ob : last_below;
        first_above = oa < first_above ? oa : first_above;
    }
    if (threadIdx.x % lsk::kWave == 0) {
        if (last_below >= 0) atomicMax(&out[0], last_below);
        if (first_above < rows) atomicMin(&out[1], first_above);
    }
}

// Uniform [0, 1) value of global element `index` under `seed`: one splitmix64 step of
// seed + (index + 1) * golden ratio, top 53 (double) or 24 (float) bits.  A function of the global
// index alone, so a vector's contents do not depend on how it is partitioned.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void random_fill_kernel(int64_t n, unsigned long long seed, int64_t first,
                                                             T *__restrict__ x) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        const unsigned long long r =
            splitmix64(seed + static_cast<unsigned long long>(first + i + 1) * 0x9e3779b97f4a7c15ull);
        if constexpr (sizeof(T) == 8) x[i] = static_cast<T>(r >> 11) * (1.0 / 9007199254740992.0);
        else x[i] = static_cast<T>(r >> 40) * (1.0f / 16777216.0f);
    }
}

template <typename T>
int launch_random_fill(const char *name, int64_t n, uint64_t seed, int64_t first, T *x, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length", name);
    if (n == 0) return LSK_OK;
    if (!x) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    int blocks = grid_for(n);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    random_fill_kernel<T><<<blocks, kBlock, 0, lsk::as_stream(stream)>>>(n, seed, first, x);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// Column-block id of every entry (col32 / block_cols; the "outside the window" marker goes to the
// last block) and its position, as the key / value pairs of the stable sort below.
__global__ __launch_bounds__(kBlock) void block_keys_kernel(int64_t nnz, const uint32_t *__restrict__ col32,
                                                            uint32_t block_cols, uint32_t last_block,
                                                            uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ pos) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t k = int64_t(blockIdx.x) * kBlock + threadIdx.x; k < nnz; k += stride) {
        const uint32_t c = col32[k];
        keys[k] = c == 0xffffffffu ? last_block : c / block_cols;
        pos[k] = static_cast<uint32_t>(k);
    }
}

// Row (counted from the first rect) of every kernel entry -- the row field a COO copy of the same
// matrix would hold; an entry inside no rect keeps the marker 0xffffffff.
__global__ __launch_bounds__(kBlock) void mark_all_kernel(int64_t n, uint32_t *__restrict__ out) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t k = int64_t(blockIdx.x) * kBlock + threadIdx.x; k < n; k += stride) out[k] = 0xffffffffu;
}
__global__ __launch_bounds__(kBlock) void entry_rows_kernel(int64_t rows, int64_t nnz,
                                                            const long long *__restrict__ rects, long long k_base,
                                                            uint32_t *__restrict__ out) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride) {
        long long lo = rects[2 * r] - k_base, hi = rects[2 * r + 1] - k_base;
        if (lo < 0) lo = 0;
        if (hi >= nnz) hi = nnz - 1;
        for (long long k = lo; k <= hi; ++k) out[k] = static_cast<uint32_t>(r);
    }
}

// out[c] = number of keys below c, for c = 0 .. nkeys, in an ascending key array: the row offsets of
// the CSR structure whose rows are the keys.
__global__ __launch_bounds__(kBlock) void sorted_key_offsets_kernel(int64_t n, const uint32_t *__restrict__ keys,
                                                                    int64_t nkeys, uint32_t *__restrict__ out) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t c = int64_t(blockIdx.x) * kBlock + threadIdx.x; c <= nkeys; c += stride) {
        int64_t lo = 0, hi = n; // first position whose key is >= c
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (static_cast<int64_t>(keys[mid]) < c) lo = mid + 1;
            else hi = mid;
        }
        out[c] = static_cast<uint32_t>(lo);
    }
}

template <typename E>
__global__ __launch_bounds__(kBlock) void gather_kernel(int64_t n, const E *__restrict__ src,
                                                        const uint32_t *__restrict__ order, E *__restrict__ dst) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t k = int64_t(blockIdx.x) * kBlock + threadIdx.x; k < n; k += stride) dst[k] = src[order[k]];
}

template <typename E>
int launch_gather(const char *name, int64_t n, const E *src, const uint32_t *order, E *dst, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative length", name);
    if (n == 0) return LSK_OK;
    if (!src || !order || !dst) return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    if (static_cast<const void *>(src) == static_cast<const void *>(dst))
        return lsk::fail(LSK_ERR_ARG, "%s: cannot permute in place", name);
    int blocks = grid_for(n);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    gather_kernel<E><<<blocks, kBlock, 0, lsk::as_stream(stream)>>>(n, src, order, dst);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_csr_halo_rows_i64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, int64_t lo, int64_t hi, int64_t *out,
                          lsk_stream_t stream) {
    if (rows < 0 || nnz < 0) return lsk::fail(LSK_ERR_ARG, "lsk_csr_halo_rows_i64: negative size");
    if (!out) return lsk::fail(LSK_ERR_ARG, "lsk_csr_halo_rows_i64: null output");
    hipStream_t s = lsk::as_stream(stream);
    halo_rows_init_kernel<<<1, 1, 0, s>>>(reinterpret_cast<long long *>(out), rows);
    LSK_LAUNCH_CHECK();
    if (rows == 0 || nnz == 0) return LSK_OK;
    if (!rects || !col) return lsk::fail(LSK_ERR_ARG, "lsk_csr_halo_rows_i64: null input");
    int blocks = grid_for(rows);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    halo_rows_kernel<<<blocks, kBlock, 0, s>>>(rows, nnz, reinterpret_cast<const long long *>(rects), k_base,
                                               reinterpret_cast<const long long *>(col), lo, hi,
                                               reinterpret_cast<long long *>(out));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}


int lsk_stencil_count(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                      int64_t row_lo, int64_t row_hi, int64_t *counts, lsk_stream_t stream) {
    Stencil st;
    int rc = make_stencil("lsk_stencil_count", dim, extent, noffsets, offsets, &st);
    if (rc != LSK_OK) return rc;
    if (row_hi < row_lo) return LSK_OK;
    const long long total = st.extent[0] * st.extent[1] * st.extent[2];
    if (row_lo < 0 || row_hi >= total)
        return lsk::fail(LSK_ERR_ARG, "lsk_stencil_count: rows outside the grid");
    if (!counts) return lsk::fail(LSK_ERR_ARG, "lsk_stencil_count: null output");
    stencil_count_kernel<<<grid_for(row_hi - row_lo + 1), kBlock, 0, lsk::as_stream(stream)>>>(
        st, row_lo, row_hi, reinterpret_cast<long long *>(counts));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

int lsk_stencil_fill_csr_f32(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                             const float *coeffs, int64_t row_lo, int64_t row_hi,
                             const int64_t *rects, int64_t k_base, int64_t *col, float *val,
                             lsk_stream_t stream) {
    return launch_fill<float, false>("lsk_stencil_fill_csr_f32", dim, extent, noffsets, offsets,
                                     coeffs, row_lo, row_hi, rects, k_base, nullptr, col, val, stream);
}

int lsk_stencil_fill_csr_f64(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                             const double *coeffs, int64_t row_lo, int64_t row_hi,
                             const int64_t *rects, int64_t k_base, int64_t *col, double *val,
                             lsk_stream_t stream) {
    return launch_fill<double, false>("lsk_stencil_fill_csr_f64", dim, extent, noffsets, offsets,
                                      coeffs, row_lo, row_hi, rects, k_base, nullptr, col, val, stream);
}

int lsk_stencil_fill_coo_f32(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                             const float *coeffs, int64_t row_lo, int64_t row_hi,
                             const int64_t *rects, int64_t k_base, int64_t *row, int64_t *col,
                             float *val, lsk_stream_t stream) {
    return launch_fill<float, true>("lsk_stencil_fill_coo_f32", dim, extent, noffsets, offsets,
                                    coeffs, row_lo, row_hi, rects, k_base, row, col, val, stream);
}

int lsk_stencil_fill_coo_f64(int dim, const int64_t *extent, int noffsets, const int64_t *offsets,
                             const double *coeffs, int64_t row_lo, int64_t row_hi,
                             const int64_t *rects, int64_t k_base, int64_t *row, int64_t *col,
                             double *val, lsk_stream_t stream) {
    return launch_fill<double, true>("lsk_stencil_fill_coo_f64", dim, extent, noffsets, offsets,
                                     coeffs, row_lo, row_hi, rects, k_base, row, col, val, stream);
}

int lsk_random_fill_f32(int64_t n, uint64_t seed, int64_t first_index, float *x, lsk_stream_t stream) {
    return launch_random_fill<float>("lsk_random_fill_f32", n, seed, first_index, x, stream);
}
int lsk_random_fill_f64(int64_t n, uint64_t seed, int64_t first_index, double *x, lsk_stream_t stream) {
    return launch_random_fill<double>("lsk_random_fill_f64", n, seed, first_index, x, stream);
}

int lsk_coo_block_order_u32(int64_t nnz, const uint32_t *col32, int64_t x_extent, int64_t block_cols,
                            uint32_t *order, lsk_stream_t stream) {
    if (nnz < 0 || x_extent < 0) return lsk::fail(LSK_ERR_ARG, "lsk_coo_block_order_u32: negative size");
    if (nnz == 0) return LSK_OK;
    if (!col32 || !order) return lsk::fail(LSK_ERR_ARG, "lsk_coo_block_order_u32: null pointer");
    if (block_cols < 1 || block_cols > 0xffffffffLL || x_extent > 0xffffffffLL)
        return lsk::fail(LSK_ERR_ARG, "lsk_coo_block_order_u32: block of %lld columns, window of %lld",
                         (long long)block_cols, (long long)x_extent);
    if (nnz > 0x7fffffffLL)
        return lsk::decline(LSK_ERR_UNSUPPORTED, "lsk_coo_block_order_u32: more than 2^31 - 1 entries");
    hipStream_t s = lsk::as_stream(stream);
    const uint32_t last_block = static_cast<uint32_t>(lsk::ceil_div(std::max<int64_t>(x_extent, 1), block_cols));
    int bits = 1;
    while ((1u << bits) <= last_block && bits < 32) ++bits;
    uint32_t *keys = nullptr, *keys_out = nullptr, *pos = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    int status = LSK_OK;
    auto release = [&] {
        (void)hipFree(keys);
        (void)hipFree(keys_out);
        (void)hipFree(pos);
        (void)hipFree(temp);
    };
#define LSK_TRY(expr)                                                                               \
    do {                                                                                            \
        const hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess && status == LSK_OK) status = lsk::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)
    const size_t bytes = static_cast<size_t>(nnz) * sizeof(uint32_t);
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&keys), bytes));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&keys_out), bytes));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&pos), bytes));
    if (status == LSK_OK) {
        int blocks = grid_for(nnz);
        if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
        block_keys_kernel<<<blocks, kBlock, 0, s>>>(nnz, col32, static_cast<uint32_t>(block_cols), last_block, keys, pos);
        LSK_TRY(hipGetLastError());
        // radix sort is stable: inside a block the entries keep their order (rows ascending)
        LSK_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_out, pos, order,
                                                   static_cast<int>(nnz), 0, bits, s));
        LSK_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 1));
        if (status == LSK_OK)
            LSK_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_out, pos, order,
                                                       static_cast<int>(nnz), 0, bits, s));
        LSK_TRY(hipStreamSynchronize(s)); // set-up time: the scratch is freed before returning
    }
#undef LSK_TRY
    release();
    return status;
}

int lsk_gather_u32(int64_t n, const uint32_t *src, const uint32_t *order, uint32_t *dst, lsk_stream_t stream) {
    return launch_gather<uint32_t>("lsk_gather_u32", n, src, order, dst, stream);
}
int lsk_gather_f32(int64_t n, const float *src, const uint32_t *order, float *dst, lsk_stream_t stream) {
    return launch_gather<float>("lsk_gather_f32", n, src, order, dst, stream);
}
int lsk_gather_f64(int64_t n, const double *src, const uint32_t *order, double *dst, lsk_stream_t stream) {
    return launch_gather<double>("lsk_gather_f64", n, src, order, dst, stream);
}

int lsk_csr_entry_rows_u32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, uint32_t *row32,
                           lsk_stream_t stream) {
    if (rows < 0 || nnz < 0) return lsk::fail(LSK_ERR_ARG, "lsk_csr_entry_rows_u32: negative size");
    if (rows > 0xfffffffeLL)
        return lsk::fail(LSK_ERR_UNSUPPORTED, "lsk_csr_entry_rows_u32: %lld rows do not fit 32 bits", (long long)rows);
    if (nnz == 0) return LSK_OK;
    if (!row32 || (rows > 0 && !rects)) return lsk::fail(LSK_ERR_ARG, "lsk_csr_entry_rows_u32: null pointer");
    hipStream_t s = lsk::as_stream(stream);
    int blocks = grid_for(nnz);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    mark_all_kernel<<<blocks, kBlock, 0, s>>>(nnz, row32);
    LSK_LAUNCH_CHECK();
    if (rows == 0) return LSK_OK;
    blocks = grid_for(rows);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    entry_rows_kernel<<<blocks, kBlock, 0, s>>>(rows, nnz, reinterpret_cast<const long long *>(rects), k_base, row32);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

int lsk_sorted_key_offsets_u32(int64_t n, const uint32_t *keys, int64_t nkeys, uint32_t *offsets,
                               lsk_stream_t stream) {
    if (n < 0 || nkeys < 0) return lsk::fail(LSK_ERR_ARG, "lsk_sorted_key_offsets_u32: negative size");
    if (n > 0xffffffffLL || nkeys > 0xfffffffeLL)
        return lsk::fail(LSK_ERR_UNSUPPORTED, "lsk_sorted_key_offsets_u32: %lld entries, %lld keys do not fit 32 bits",
                         (long long)n, (long long)nkeys);
    if (!offsets || (n > 0 && !keys)) return lsk::fail(LSK_ERR_ARG, "lsk_sorted_key_offsets_u32: null pointer");
    int blocks = grid_for(nkeys + 1);
    if (blocks > 16 * lsk::kMaxGridBlocks) blocks = 16 * lsk::kMaxGridBlocks;
    sorted_key_offsets_kernel<<<blocks, kBlock, 0, lsk::as_stream(stream)>>>(n, keys, nkeys, offsets);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

int lsk_inclusive_sum_i64(int64_t n, const int64_t *in, int64_t *out, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "lsk_inclusive_sum_i64: negative length");
    if (n == 0) return LSK_OK;
    if (!in || !out) return lsk::fail(LSK_ERR_ARG, "lsk_inclusive_sum_i64: null pointer");
    if (n > 0x7fffffffLL) return lsk::fail(LSK_ERR_UNSUPPORTED, "lsk_inclusive_sum_i64: more than 2^31 - 1 values");
    hipStream_t s = lsk::as_stream(stream);
    void *temp = nullptr;
    size_t temp_bytes = 0;
    LSK_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, in, out, static_cast<int>(n), s));
    LSK_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 1));
    const hipError_t st = hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, in, out, static_cast<int>(n), s);
    const hipError_t sy = hipStreamSynchronize(s); // set-up time: the scratch is freed before returning
    (void)hipFree(temp);
    LSK_HIP(st);
    LSK_HIP(sy);
    return LSK_OK;
}

int lsk_rects_from_counts_i64(int64_t rows, const int64_t *counts, const int64_t *prefix_incl,
                              int64_t base, int64_t *rects, lsk_stream_t stream) {
    if (rows < 0) return lsk::fail(LSK_ERR_ARG, "lsk_rects_from_counts_i64: negative rows");
    if (rows == 0) return LSK_OK;
    if (!counts || !prefix_incl || !rects)
        return lsk::fail(LSK_ERR_ARG, "lsk_rects_from_counts_i64: null pointer");
    rects_from_counts_kernel<<<grid_for(rows), kBlock, 0, lsk::as_stream(stream)>>>(
        rows, reinterpret_cast<const long long *>(counts),
        reinterpret_cast<const long long *>(prefix_incl), base, reinterpret_cast<long long *>(rects));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

int lsk_index_bounds_i64(int64_t n, const int64_t *idx, int64_t *out_minmax, lsk_stream_t stream) {
    if (n < 0) return lsk::fail(LSK_ERR_ARG, "lsk_index_bounds_i64: negative length");
    if (!out_minmax) return lsk::fail(LSK_ERR_ARG, "lsk_index_bounds_i64: null output");
    hipStream_t s = lsk::as_stream(stream);
    bounds_init_kernel<<<1, 1, 0, s>>>(reinterpret_cast<long long *>(out_minmax));
    LSK_LAUNCH_CHECK();
    if (n == 0) return LSK_OK;
    if (!idx) return lsk::fail(LSK_ERR_ARG, "lsk_index_bounds_i64: null input");
    int blocks = grid_for(n);
    if (blocks > lsk::kMaxGridBlocks) blocks = lsk::kMaxGridBlocks;
    index_bounds_kernel<<<blocks, kBlock, 0, s>>>(n, reinterpret_cast<const long long *>(idx),
                                                  reinterpret_cast<long long *>(out_minmax));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // extern "C"
