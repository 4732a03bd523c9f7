// CSR SpMV for gfx950 on the reference's own layout: per row a 16-byte {lo, hi} rect of
// GLOBAL kernel indices, 64-bit global column ids, fp32/fp64 entries
// (src/CSRMatrixTasks.cu:14-156, src/CuSPARSEHelpers.hpp:9-101).  No rect->indptr rebasing
// pass, no descriptors, no workspace: one kernel launch per call.
//
// Two forms (option csr.variant):
//
//   wave tiles (default; csr_spmv_wave_kernel): every wave64 owns 1..256 consecutive rows and works
//   alone -- rects by 16-byte loads, the rows' contiguous run of entries by aligned 16-byte loads
//   of col and val (a whole tile in flight at once), x gathered through L2, products through a
//   wave-private LDS buffer, rows added up out of LDS, no workgroup barrier anywhere.  Optionally
//   fused with the dot product of the result and another vector.
//
//   workgroup tiles (csr_spmv_kernel): the four waves of a workgroup share R rows and one LDS
//   buffer, with barriers between the phases.  Kept for comparison.
//
// Rows whose rects are not back to back (gaps, overlaps, permuted) and single rows longer than
// the LDS buffer are walked one wave per row straight from HBM.  All paths skip entries whose
// column lies outside x_lo..x_hi, as the reference's CPU body does (src/CSRMatrixTasks.cpp:88).
//
// Compressed columns (lsk_csr_compress_cols_u32 + lsk_csr_spmv_c32_*): a set-up pass rewrites the
// 64-bit global column ids once as 32-bit offsets into the x window (0xffffffff = outside it);
// the wave-tile kernel then reads 4 instead of 8 bytes of index per entry.  Same tiles, same
// order of additions: results are bit-identical to the 64-bit form.
//
// HBM bytes per call: nnz*(8 + sizeof(T)) + rows*(16 + sizeof(T)) + the x window
// (compressed columns: nnz*(4 + sizeof(T)) + ...).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "lsk_common.hpp"

// The rows of a piece that are too long for one wave (lsk_csr_long_rows_create): each is cut into
// chunks of kLongChunk entries that one workgroup multiplies, and a last small kernel adds a row's
// chunk sums in chunk order.  Row indices count from the piece's first row, entries from its first entry.
struct lsk_csr_long_rows {
    int64_t rows = 0;            // rows of the piece the plan was made for
    int64_t nlong = 0;           // rows longer than `threshold`
    int64_t chunks = 0;          // chunks of all of them together
    int threshold = 0;
    uint32_t *row = nullptr;         // nlong: the long rows, ascending
    uint32_t *chunk_first = nullptr; // nlong + 1: first chunk of every long row
    uint32_t *chunk_row = nullptr;   // chunks: which long row (0 .. nlong) a chunk belongs to
    void *partial = nullptr;         // chunks sums, 8 bytes each (products through one plan are ordered on one stream)
};

namespace {

using lsk::kBlock;

template <typename T> struct Cfg;
// Products kept per workgroup: 30 KiB of LDS (+1 KiB of row offsets), so up to five workgroups
// (20 waves) fit the 160 KiB of a CU.
template <> struct Cfg<double> { static constexpr int CAP = 3840; static constexpr int U = 2; using V = double2; };
template <> struct Cfg<float> { static constexpr int CAP = 7680; static constexpr int U = 4; using V = float4; };

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct Rect { long long lo, hi; };

// Native 16-byte vectors (the nontemporal builtin does not take HIP's struct vector types).
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef long long ll2_t __attribute__((ext_vector_type(2)));
// The same, promised only 8-byte alignment: col pairs sit at the val units' entries, whose
// 16-byte phase may differ from the col array's.
typedef ll2_t ll2_a8_t __attribute__((aligned(8)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
// val words of the compressed form sit at the col quads' entries: natural alignment only.
typedef d2_t d2_a8_t __attribute__((aligned(8)));
typedef f4_t f4_a4_t __attribute__((aligned(4)));

// Column field: 64-bit global ids checked against x_lo..x_hi, or 32-bit offsets into the window
// (x_lo = 0, x_hi = extent - 1 < 0xffffffff; the marker 0xffffffff fails the check).
template <typename CT> struct ColOps;
template <> struct ColOps<int64_t> {
    static __device__ __forceinline__ bool in(int64_t c, int64_t lo, int64_t hi) { return c >= lo && c <= hi; }
    static __device__ __forceinline__ int64_t off(int64_t c, int64_t lo) { return c - lo; }
};
template <> struct ColOps<uint32_t> {
    static __device__ __forceinline__ bool in(uint32_t c, int64_t, int64_t hi) { return c <= static_cast<uint32_t>(hi); }
    static __device__ __forceinline__ uint32_t off(uint32_t c, int64_t) { return c; }
};

// Row field: with 64-bit columns the reference's {lo, hi} rects of global entry indices; with the
// compressed form rows + 1 32-bit offsets of each row's first entry from the start of the piece's
// col / val arrays (validated when they were made: ascending, inside the piece).
template <typename CT> struct RowField;
template <> struct RowField<int64_t> {
    using type = Rect;
    static __device__ __forceinline__ void bounds(const Rect *p, int64_t r, int64_t k_base, int64_t nnz,
                                                  int64_t &lo, int64_t &hi) {
        const Rect rc = p[r]; // clipped to the piece so that a malformed rect cannot fault
        lo = rc.lo > k_base ? rc.lo : k_base;
        hi = rc.hi < k_base + nnz - 1 ? rc.hi : k_base + nnz - 1;
    }
};
template <> struct RowField<uint32_t> {
    using type = uint32_t;
    static __device__ __forceinline__ void bounds(const uint32_t *p, int64_t r, int64_t, int64_t,
                                                  int64_t &lo, int64_t &hi) {
        lo = p[r];
        hi = static_cast<int64_t>(p[r + 1]) - 1;
    }
};

template <typename T> struct Native;
template <> struct Native<double> { using V = d2_t; };
template <> struct Native<float> { using V = f4_t; };

template <bool NT, typename VT>
__device__ __forceinline__ VT vload(const VT *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT> __device__ __forceinline__ ll2_t load_col_pair(const int64_t *p) {
    const ll2_a8_t *q = reinterpret_cast<const ll2_a8_t *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <bool NT> __device__ __forceinline__ d2_t load_val_pair_a8(const double *p) {
    const d2_a8_t *q = reinterpret_cast<const d2_a8_t *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}
template <bool NT> __device__ __forceinline__ f4_t load_val_quad_a4(const float *p) {
    const f4_a4_t *q = reinterpret_cast<const f4_a4_t *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <typename T> __device__ __forceinline__ T vget(const double2 &v, int i) { return i == 0 ? v.x : v.y; }
template <typename T> __device__ __forceinline__ T vget(const float4 &v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// Walk rows [r_begin, r_end) one wave per row, entries straight from HBM.  Correct for any
// rects (gaps, overlaps, empty, long).
template <typename T>
__device__ void rows_by_wave(int r_begin, int r_end, const Rect *__restrict__ rects,
                             int64_t k_base, int64_t nnz, const int64_t *__restrict__ col,
                             const T *__restrict__ val, const T *__restrict__ x, int64_t x_lo,
                             int64_t x_hi, T *__restrict__ y, int beta) {
    const int lane = threadIdx.x % lsk::kWave, wave = threadIdx.x / lsk::kWave;
    for (int r = r_begin + wave; r < r_end; r += kBlock / lsk::kWave) {
        const Rect rc = rects[r];
        T acc = T(0);
        // Rects are clipped to the piece so that a malformed rect cannot fault.
        const int64_t lo = rc.lo > k_base ? rc.lo : k_base;
        const int64_t hi = rc.hi < k_base + nnz - 1 ? rc.hi : k_base + nnz - 1;
        for (int64_t k = lo + lane; k <= hi; k += lsk::kWave) {
            const int64_t c = col[k - k_base];
            if (c >= x_lo && c <= x_hi) acc = fma_t(val[k - k_base], x[c - x_lo], acc);
        }
        acc = lsk::wave_sum(acc);
        if (lane == 0) y[r] = beta ? y[r] + acc : acc;
    }
}

// LOG_R: log2 of rows per workgroup (R = 1 << LOG_R <= 256); G = 256 / R lanes add one row.
template <typename T, int LOG_R>
__global__ __launch_bounds__(kBlock) void csr_spmv_kernel(
    int64_t rows, const Rect *__restrict__ rects, int64_t k_base, int64_t nnz,
    const int64_t *__restrict__ col, const T *__restrict__ val, const T *__restrict__ x,
    int64_t x_lo, int64_t x_hi, T *__restrict__ y, int beta, int remap) {
    constexpr int R = 1 << LOG_R;
    constexpr int G = kBlock / R;
    constexpr int CAP = Cfg<T>::CAP;
    constexpr int U = Cfg<T>::U; // entries per 16-byte val access
    using V = typename Cfg<T>::V;

    __shared__ T s_prod[CAP];
    __shared__ int s_off[R + 1];
    __shared__ long long s_k0;

    const unsigned tile = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t r0 = int64_t(tile) * R;
    const int nr = static_cast<int>(rows - r0 < R ? rows - r0 : R);
    const int tid = threadIdx.x;

    // ---- 1. rects -> row offsets relative to the first entry of the tile --------------
    Rect rc{0, -1};
    if (tid < nr) rc = rects[r0 + tid];
    if (tid == 0) s_k0 = rc.lo;
    __syncthreads();
    const long long k0 = s_k0;
    const long long rel_lo = rc.lo - k0, rel_end = rc.hi + 1 - k0;
    bool ok = true;
    if (tid < nr) {
        ok = rel_lo >= 0 && rel_end >= rel_lo && rel_end <= CAP && k0 >= k_base &&
             rc.hi - k_base < nnz;
        s_off[tid] = static_cast<int>(rel_lo);
        if (tid == nr - 1) s_off[nr] = static_cast<int>(rel_end);
    }
    __syncthreads();
    if (tid < nr) ok = ok && (static_cast<long long>(s_off[tid + 1]) == rel_end);
    const bool stream = __syncthreads_and(ok);

    if (!stream) {
        rows_by_wave<T>(0, nr, rects + r0, k_base, nnz, col, val, x, x_lo, x_hi, y + r0, beta);
        return;
    }

    // ---- 2. stream the tile's entries; products into LDS ------------------------------
    const int nnzb = s_off[nr];
    const int64_t *colb = col + (k0 - k_base);
    const T *valb = val + (k0 - k_base);
    // Entry e of the tile sits at colb[e] / valb[e].  Lane accesses are 16-byte aligned:
    // a unit is U consecutive entries starting at a multiple of U minus `shift`, where
    // `shift` makes the val address 16-byte aligned; col is read as U/2 aligned pairs when
    // its parity agrees, else entry by entry.  Entries of a unit that fall outside
    // [0, nnzb) are loaded (the aligned 16-byte word holding an entry of this tile never
    // leaves the page that entry is on) and discarded.
    const int shift = static_cast<int>((reinterpret_cast<uintptr_t>(valb) / sizeof(T)) % U);
    const bool col_pairs =
        ((reinterpret_cast<uintptr_t>(colb) / 8 + (U - shift)) % 2) == 0; // pair-aligned col
    constexpr int UNROLL = 4;
    if (nnzb > 0)
    for (int base = tid * U - shift; base < nnzb; base += kBlock * U * UNROLL) {
        V vv[UNROLL];
        long long cc[UNROLL][U];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int b = base + u * kBlock * U;
            if (b < nnzb) {
                vv[u] = *reinterpret_cast<const V *>(valb + b);
                if (col_pairs) {
#pragma unroll
                    for (int p = 0; p < U; p += 2) {
                        const int e = b + p;
                        longlong2 c2{-1, -1};
                        if (e + 1 >= 0 && e < nnzb) // the pair holds an entry of this tile
                            c2 = *reinterpret_cast<const longlong2 *>(colb + e);
                        cc[u][p] = c2.x;
                        cc[u][p + 1] = c2.y;
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < U; ++p) {
                        const int e = b + p;
                        cc[u][p] = (e >= 0 && e < nnzb) ? colb[e] : -1;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int b = base + u * kBlock * U;
            if (b < nnzb) {
                T xv[U];
#pragma unroll
                for (int p = 0; p < U; ++p) {
                    const int e = b + p;
                    const long long c = cc[u][p];
                    const bool live = e >= 0 && e < nnzb && c >= x_lo && c <= x_hi;
                    xv[p] = live ? x[c - x_lo] : T(0);
                }
#pragma unroll
                for (int p = 0; p < U; ++p) {
                    const int e = b + p;
                    if (e >= 0 && e < nnzb) {
                        const long long c = cc[u][p];
                        const bool live = c >= x_lo && c <= x_hi;
                        s_prod[e] = live ? vget<T>(vv[u], p) * xv[p] : T(0);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- 3. G lanes per row fold the products; lane 0 of the group stores y ------------
    const int row = tid / G, g = tid % G;
    T acc = T(0);
    if (row < nr) {
        const int e1 = s_off[row + 1];
        for (int e = s_off[row] + g; e < e1; e += G) acc += s_prod[e];
    }
    if constexpr (G > 1) {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, lsk::kWave);
    }
    if (row < nr && g == 0) {
        T *yp = y + r0 + row;
        *yp = beta ? *yp + acc : acc;
    }
}

// One row of the tile per wave, entries straight from HBM (wave-level form of rows_by_wave).
// With w != nullptr lane 0 also adds y_new[r] * w[r] into *dot_acc.
template <typename T, typename CT>
__device__ void wave_rows_direct(int nr, const typename RowField<CT>::type *__restrict__ rects, int64_t k_base,
                                 int64_t nnz, const CT *__restrict__ col,
                                 const T *__restrict__ val, const T *__restrict__ x, int64_t x_lo,
                                 int64_t x_hi, T *__restrict__ y, int beta,
                                 const T *__restrict__ w = nullptr, T *dot_acc = nullptr, int skip_len = 0) {
    const int lane = threadIdx.x % lsk::kWave;
    for (int r = 0; r < nr; ++r) {
        T acc = T(0);
        int64_t lo, hi;
        RowField<CT>::bounds(rects, r, k_base, nnz, lo, hi);
        if (skip_len > 0 && hi - lo + 1 > skip_len) continue; // the split product's own kernels take this row
        for (int64_t k = lo + lane; k <= hi; k += lsk::kWave) {
            const CT c = col[k - k_base];
            if (ColOps<CT>::in(c, x_lo, x_hi)) acc = fma_t(val[k - k_base], x[ColOps<CT>::off(c, x_lo)], acc);
        }
        acc = lsk::wave_sum(acc);
        if (lane == 0) {
            const T out = beta ? y[r] + acc : acc;
            y[r] = out;
            if (w) *dot_acc += out * w[r];
        }
    }
}

// Orders this wave's LDS writes before its later LDS reads (other lanes' data) without a
// workgroup barrier: LDS operations of one wave execute in issue order, so only the
// compiler must be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One chunk of the streaming phase: K units per lane (unit u = ubase + k*64 + lane).  All
// 2K 16-byte loads of val and col are issued before the first use, then all x gathers, then
// the products go to LDS.  Lanes whose unit lies past the end re-read the last unit (clamped
// index, same cache line) and drop the result, so the chunk has no branches around loads.
template <typename T, int K, bool NT>
__device__ __forceinline__ void stream_chunk(int ubase, int lane, int nfull,
                                             const T *__restrict__ vbody,
                                             const int64_t *__restrict__ cbody,
                                             const T *__restrict__ x, int64_t x_lo, int64_t x_hi,
                                             T *__restrict__ sp_body) {
    constexpr int U = Cfg<T>::U;
    using V = typename Native<T>::V;
    V vv[K];
    long long cc[K][U];
    int uu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int u = ubase + k * lsk::kWave + lane;
        uu[k] = u;
        const int uc = u < nfull ? u : nfull - 1;
        vv[k] = vload<NT>(reinterpret_cast<const V *>(vbody) + uc);
#pragma unroll
        for (int p = 0; p < U; p += 2) {
            const ll2_t c2 = load_col_pair<NT>(cbody + uc * U + p);
            cc[k][p] = c2.x;
            cc[k][p + 1] = c2.y;
        }
    }
    T xv[K][U];
    bool in[K][U];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int p = 0; p < U; ++p) {
            const long long c = cc[k][p];
            in[k][p] = c >= x_lo && c <= x_hi;   // the CPU body's input-domain guard
            xv[k][p] = x[in[k][p] ? c - x_lo : 0];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (uu[k] < nfull) {
#pragma unroll
            for (int p = 0; p < U; ++p)
                sp_body[uu[k] * U + p] = in[k][p] ? vv[k][p] * xv[k][p] : T(0);
        }
    }
}

// The same for 32-bit columns: the unit's U column offsets come as one 8- or 16-byte load
// (4-byte aligned: the val word fixes the unit's phase).
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
typedef u2_t u2_a4_t __attribute__((aligned(4)));
typedef u4_t u4_a4_t __attribute__((aligned(4)));
template <bool NT> __device__ __forceinline__ u2_t load_col32_unit(const uint32_t *p, u2_t) {
    const u2_a4_t *q = reinterpret_cast<const u2_a4_t *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}
template <bool NT> __device__ __forceinline__ u4_t load_col32_unit(const uint32_t *p, u4_t) {
    const u4_a4_t *q = reinterpret_cast<const u4_a4_t *>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <typename T, int K, bool NT>
__device__ __forceinline__ void stream_chunk32(int ubase, int lane, int nfull,
                                               const T *__restrict__ vbody,
                                               const uint32_t *__restrict__ cbody,
                                               const T *__restrict__ x, uint32_t x_last,
                                               T *__restrict__ sp_body) {
    constexpr int U = Cfg<T>::U;
    using V = typename Native<T>::V;
    using CV = unsigned int __attribute__((ext_vector_type(U)));
    V vv[K];
    CV cc[K];
    int uu[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int u = ubase + k * lsk::kWave + lane;
        uu[k] = u;
        const int uc = u < nfull ? u : nfull - 1;
        vv[k] = vload<NT>(reinterpret_cast<const V *>(vbody) + uc);
        cc[k] = load_col32_unit<NT>(cbody + uc * U, CV{});
    }
    T xv[K][U];
    bool in[K][U];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int p = 0; p < U; ++p) {
            const uint32_t c = cc[k][p];
            in[k][p] = c <= x_last;
            xv[k][p] = x[in[k][p] ? c : 0u];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (uu[k] < nfull) {
#pragma unroll
            for (int p = 0; p < U; ++p)
                sp_body[uu[k] * U + p] = in[k][p] ? vv[k][p] * xv[k][p] : T(0);
        }
    }
}

// The streaming phase of one batch of consecutive rows: n entries starting at valb / colb go as
// products into sp[0, n).  Units are U consecutive entries whose val word is 16-byte aligned;
// the entries before the first and after the last whole unit are handled one per lane.
template <typename T, typename CT, bool NT, int KMAX>
__device__ __forceinline__ void stream_batch(int lane, int n, const T *__restrict__ valb,
                                             const CT *__restrict__ colb,
                                             const T *__restrict__ x, int64_t x_lo, int64_t x_hi,
                                             T *__restrict__ sp) {
    constexpr bool C32 = sizeof(CT) == 4;
    constexpr int U = Cfg<T>::U;
    const int shift = static_cast<int>((reinterpret_cast<uintptr_t>(valb) / sizeof(T)) % U);
    const int head0 = (U - shift) % U;
    const int head = head0 < n ? head0 : n;
    const int nfull = (n - head) / U;
    const int tail0 = head + nfull * U; // first entry after the whole units
    const T *vbody = valb + head;
    const CT *cbody = colb + head;
    // Up to KMAX units per lane in one go.  KMAX = 8 holds a whole tile of 32 rows x 27 entries
    // (7 units): all its col/val loads are in flight together, then all its gathers; two rounds
    // of 4 units cost 13 % at that size.  The instance with KMAX = 4 needs 68 instead of 103
    // vector registers (5 instead of 4 waves per SIMD); the launcher picks by mean tile size.
    for (int ubase = 0; ubase < nfull; ubase += lsk::kWave * KMAX) {
        const int left = nfull - ubase; // wave-uniform
#define LSK_CHUNK(K)                                                                              \
    do {                                                                                          \
        if constexpr (C32)                                                                        \
            stream_chunk32<T, K, NT>(ubase, lane, nfull, vbody,                                   \
                                     reinterpret_cast<const uint32_t *>(cbody), x,                \
                                     static_cast<uint32_t>(x_hi), sp + head);                     \
        else                                                                                      \
            stream_chunk<T, K, NT>(ubase, lane, nfull, vbody,                                     \
                                   reinterpret_cast<const int64_t *>(cbody), x, x_lo, x_hi,       \
                                   sp + head);                                                    \
    } while (0)
        if constexpr (KMAX == 8) {
            if (left > 7 * lsk::kWave) { LSK_CHUNK(8); continue; }
            if (left > 6 * lsk::kWave) { LSK_CHUNK(7); continue; }
            if (left > 5 * lsk::kWave) { LSK_CHUNK(6); continue; }
            if (left > 4 * lsk::kWave) { LSK_CHUNK(5); continue; }
        }
        if (left > 3 * lsk::kWave) LSK_CHUNK(4);
        else if (left > 2 * lsk::kWave) LSK_CHUNK(3);
        else if (left > lsk::kWave) LSK_CHUNK(2);
        else LSK_CHUNK(1);
#undef LSK_CHUNK
    }
    const int n_edge = head + (n - tail0);
    if (lane < n_edge) {
        const int e = lane < head ? lane : tail0 + (lane - head);
        const CT c = colb[e];
        const bool in = ColOps<CT>::in(c, x_lo, x_hi);
        sp[e] = in ? valb[e] * x[in ? ColOps<CT>::off(c, x_lo) : 0] : T(0);
    }
}

// Sum of sp[e0, e1) taken in steps of `stride` from e0 + g, four LDS reads in flight; reads
// past the end are clamped to a valid slot and dropped.
template <typename T, int CAPW>
__device__ __forceinline__ T fold_range(const T *__restrict__ sp, int e0, int e1, int g, int stride) {
    T acc = T(0);
    for (int e = e0 + g; e < e1; e += 4 * stride) {
        T a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ek = e + k * stride;
            a[k] = sp[ek < CAPW ? ek : CAPW - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += (e + k * stride < e1) ? a[k] : T(0);
    }
    return acc;
}

// A row of n entries that is longer than the wave's LDS buffer: streamed through the buffer a
// bufferful at a time with the same wide loads as a batch of rows (all of a bufferful's col / val
// units in flight, then its gathers), every lane keeping the sum of its own share of the products;
// returns this lane's share.  Always with 4 units per lane in flight, whatever the tile path uses: the
// rare path must stay below the tile path's register peak (an out-of-line call costs far more).
template <typename T, typename CT, bool NT, int CAPW>
__device__ __forceinline__ T wave_long_row(int lane, int n, const T *__restrict__ valb,
                                                     const CT *__restrict__ colb, const T *__restrict__ x,
                                                     int64_t x_lo, int64_t x_hi, T *__restrict__ sp) {
    T acc = T(0);
    for (int s0 = 0; s0 < n; s0 += CAPW) {
        const int len = n - s0 < CAPW ? n - s0 : CAPW;
        stream_batch<T, CT, NT, 4>(lane, len, valb + s0, colb + s0, x, x_lo, x_hi, sp);
        wave_lds_fence();
        acc += fold_range<T, CAPW>(sp, 0, len, lane, lsk::kWave);
        wave_lds_fence(); // the next bufferful overwrites the buffer
    }
    return acc;
}

// Wave-tile CSR SpMV: every wave owns ROWS = 1 << LOG_ROWS consecutive rows (1..256) and never
// meets the other waves of its workgroup (no s_barrier), so a wave that waits on HBM never holds
// up three others.  Per tile:
//   1. lanes load the rects, ROWS/64 per lane when ROWS > 64 (row j*64 + lane, coalesced).  If the
//      rows do not tile one contiguous run of entries inside the piece (wave vote), they are
//      walked one by one straight from HBM;
//   2. otherwise the rows are taken in batches of as many consecutive rows as fit the wave's
//      LDS buffer -- the whole tile at once in the normal case; a row that alone exceeds the
//      buffer is walked straight from HBM.  A batch's run is read in chunks of up to KMAX aligned
//      16-byte units per lane (stream_batch / stream_chunk);
//   3. the rows' products are added out of LDS: 64/ROWS lanes per row when ROWS < 64, else every
//      lane folds its own ROWS/64 rows; y is stored one lane per row.
//
// DOT: the kernel also leaves, per wave, the sum over its rows of y_new[r] * w[r] in
// partials[tile * 4 + wave] (0 for waves past the last row): the dot product a Krylov step
// takes right after the product (CG's p.Ap) without a second pass over y and w.
template <typename T, typename CT, int LOG_ROWS, bool NT, int KMAX, bool DOT>
#ifndef LSK_MIN_WAVES
#define LSK_MIN_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, LSK_MIN_WAVES) void csr_spmv_wave_kernel( // at least 4 waves per SIMD: <= 128 VGPRs
    int64_t rows, const typename RowField<CT>::type *__restrict__ rects, int64_t k_base, int64_t nnz,
    const CT *__restrict__ col, const T *__restrict__ val, const T *__restrict__ x,
    int64_t x_lo, int64_t x_hi, T *__restrict__ y, int beta, int remap,
    const T *__restrict__ w, T *__restrict__ partials, int skip_len) {
    constexpr int ROWS = 1 << LOG_ROWS;
    constexpr int RPL = ROWS > lsk::kWave ? ROWS / lsk::kWave : 1; // rects per lane
    constexpr int GW = ROWS < lsk::kWave ? lsk::kWave / ROWS : 1;  // lanes per row in the fold
    constexpr int WAVES = kBlock / lsk::kWave;
    constexpr int CAPW = Cfg<T>::CAP / WAVES;

    __shared__ T s_prod[WAVES][CAPW];

    // the wave number as a scalar: everything derived from it (first row, LDS base, partial
    // slot) then lives in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / lsk::kWave);
    const int lane = threadIdx.x % lsk::kWave;
    const unsigned tile = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t r0 = (int64_t(tile) * WAVES + wave) * ROWS;
    // indexed by tile, not by workgroup: the order of the partial sums does not follow the placement
    T *const my_partial = DOT ? partials + (int64_t(tile) * WAVES + wave) : nullptr;
    if (r0 >= rows) { // wave-uniform; nothing below synchronises across waves
        if (DOT && lane == 0) *my_partial = T(0);
        return;
    }
    const int nr = static_cast<int>(rows - r0 < ROWS ? rows - r0 : ROWS);
    T *sp = s_prod[wave];
    T dot_acc = T(0); // this lane's share of sum_r y_new[r] * w[r]
    const T *wt = DOT ? w + r0 : nullptr;
    // w of the rows this lane will store, fetched now so that its latency hides behind the tile
    T wv[RPL];
    if constexpr (DOT) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int row = RPL == 1 ? lane / GW : j * lsk::kWave + lane;
            wv[j] = wt[row < nr ? row : nr - 1];
        }
    }

    // ---- 1. rects (or row offsets) --------------------------------------------------------
    long long k0;
    int rel_lo[RPL], rel_end[RPL]; // this lane's rows relative to k0 (32-bit once validated)
    bool ok = true;
    if constexpr (sizeof(CT) == 4) {
        // offsets were validated when they were made; only a tile too long for 32-bit
        // arithmetic below (rows of 2^30 entries) is sent the slow way
        const uint32_t *off = rects + r0;
        uint32_t lo32[RPL], end32[RPL];
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int row = j * lsk::kWave + lane;
            lo32[j] = off[row < nr ? row : nr];
            end32[j] = off[row < nr ? row + 1 : nr];
        }
        const uint32_t first = __builtin_amdgcn_readfirstlane(lo32[0]);
        k0 = first;
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            ok = ok && (end32[j] - first <= 0x3fffffffu);
            rel_lo[j] = static_cast<int>(lo32[j] - first);
            rel_end[j] = static_cast<int>(end32[j] - first);
        }
    } else {
    Rect rc[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const int row = j * lsk::kWave + lane;
        rc[j] = Rect{0, -1};
        if (row < nr) rc[j] = rects[r0 + row];
    }
    k0 = lsk::wave_first(rc[0].lo);
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const long long lo64 = rc[j].lo - k0, end64 = rc[j].hi + 1 - k0;
        const int row = j * lsk::kWave + lane;
        if (row < nr)
            ok = ok && lo64 >= 0 && end64 >= lo64 && end64 <= 0x3fffffff && k0 >= k_base && rc[j].hi - k_base < nnz;
        rel_lo[j] = static_cast<int>(lo64);
        rel_end[j] = static_cast<int>(end64);
    }
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const int row = j * lsk::kWave + lane;
        // the next row's first entry: the neighbouring lane's, or lane 0's next rect
        int next_lo = __shfl_down(rel_lo[j], 1, lsk::kWave);
        if constexpr (RPL > 1) {
            const int wrap = __builtin_amdgcn_readfirstlane(rel_lo[j + 1 < RPL ? j + 1 : j]);
            if (lane == lsk::kWave - 1) next_lo = wrap;
        }
        if (row < nr - 1) ok = ok && (next_lo == rel_end[j]);
    }
    }
    if (!__all(ok)) {
        wave_rows_direct<T, CT>(nr, rects + r0, k_base, nnz, col, val, x, x_lo, x_hi, y + r0, beta, wt, &dot_acc,
                                skip_len);
        if (DOT && lane == 0) *my_partial = dot_acc;
        return;
    }
    const CT *colb = col + (k0 - k_base);
    const T *valb = val + (k0 - k_base);

    // value of a per-row quantity at tile row q (q wave-uniform), as a wave-uniform value
    auto at_row = [&](const int (&v)[RPL], int q) {
        int out = 0;
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int got = __builtin_amdgcn_readlane(v[j], q % lsk::kWave);
            if (j == q / lsk::kWave) out = got;
        }
        return out;
    };

    // ---- 2./3. batches of consecutive rows that fit the buffer ---------------------------
    for (int ra = 0; ra < nr;) {
        const int base = at_row(rel_lo, ra);
        int m = 0; // rows ra .. ra+m-1 fit (the condition is monotone in the row)
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int row = j * lsk::kWave + lane;
            const bool fits = row >= ra && row < nr && rel_end[j] - base <= CAPW;
            m += __popcll(__ballot(fits));
        }
        if (m == 0) {
            // this row alone is longer than the buffer: streamed through it (wave_long_row)
            const int n_row = at_row(rel_end, ra) - base;
            if (skip_len > 0 && n_row > skip_len) { // left to the split product's own kernels (lsk_csr_spmv_c32_split_*)
                ra += 1;
                continue;
            }
            T acc = wave_long_row<T, CT, NT, CAPW>(lane, n_row, valb + base, colb + base, x, x_lo, x_hi, sp);
            acc = lsk::wave_sum(acc);
            if (lane == 0) {
                T *yp = y + r0 + ra;
                const T out = beta ? *yp + acc : acc;
                *yp = out;
                if constexpr (DOT) dot_acc += out * wt[ra];
            }
            ra += 1;
            continue;
        }
        const int n = at_row(rel_end, ra + m - 1) - base;
        stream_batch<T, CT, NT, KMAX>(lane, n, valb + base, colb + base, x, x_lo, x_hi, sp);
        wave_lds_fence();
        if constexpr (RPL == 1) {
            const int row = lane / GW, g = lane % GW;
            const int e0 = __shfl(rel_lo[0], row, lsk::kWave) - base;
            const int e1 = __shfl(rel_end[0], row, lsk::kWave) - base;
            const bool mine = row >= ra && row < ra + m;
            T acc = mine ? fold_range<T, CAPW>(sp, e0, e1, g, GW) : T(0);
            if constexpr (GW > 1) {
#pragma unroll
                for (int off = GW / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, lsk::kWave);
            }
            if (mine && g == 0) {
                T *yp = y + r0 + row;
                const T out = beta ? *yp + acc : acc;
                *yp = out;
                if constexpr (DOT) dot_acc += out * wv[0];
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int row = j * lsk::kWave + lane;
                if (row >= ra && row < ra + m) {
                    const T acc = fold_range<T, CAPW>(sp, rel_lo[j] - base, rel_end[j] - base, 0, 1);
                    T *yp = y + r0 + row;
                    const T out = beta ? *yp + acc : acc;
                    *yp = out;
                    if constexpr (DOT) dot_acc += out * wv[j];
                }
            }
        }
        ra += m;
        if (ra < nr) wave_lds_fence(); // the next batch overwrites the buffer
    }
    if constexpr (DOT) {
        dot_acc = lsk::wave_sum(dot_acc);
        if (lane == 0) *my_partial = dot_acc;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void zero_kernel(int64_t n, T *__restrict__ y) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) y[i] = T(0);
}

// Rows per wave of the wave-tile form (log2): as many as keep the tile's products inside the
// wave's quarter of the LDS buffer at the piece's mean row length, a power of two in 1..256.
// (A tile that turns out larger is taken in several batches, so the estimate only has to be
// typical.)
template <typename T>
int pick_log_rows_per_wave(int64_t rows, int64_t nnz) {
    const double mean = rows > 0 ? double(nnz) / double(rows) : 0.0;
    const double want_w = mean > 0.0 ? 0.95 * (Cfg<T>::CAP / 4) / mean : 256.0;
    int log_rw = 8;
    while (log_rw > 0 && double(1 << log_rw) > want_w) --log_rw;
    const int forced = lsk::options().csr_log_rw;
    if (forced >= 0 && forced <= 8) log_rw = forced; // (a tile larger than the buffer is taken in several batches)
    return log_rw;
}

template <typename T>
__global__ void store_zero_kernel(T *out) { *out = T(0); }

// w != nullptr selects the fused form: besides y, partials[0 .. *count) receive per-wave sums
// of y_new[r] * w[r] (w[0] pairs with the piece's first row).
template <typename T, typename CT = int64_t>
int launch_csr_spmv(const char *name, int64_t rows, int64_t nnz, const void *rects,
                    int64_t k_base, const CT *col, const T *val, const T *x, int64_t x_lo,
                    int64_t x_hi, T *y, int beta, lsk_stream_t stream, const T *w = nullptr,
                    T *partials = nullptr, int64_t capacity = 0, int64_t *count = nullptr, int skip_len = 0) {
    const bool fused = (w != nullptr);
    if (fused) {
        if (!partials || !count) return lsk::fail(LSK_ERR_ARG, "%s: null partials or count", name);
        *count = 0;
    }
    if (rows < 0 || nnz < 0)
        return lsk::fail(LSK_ERR_ARG, "%s: negative size (rows %lld, nnz %lld)", name,
                         (long long)rows, (long long)nnz);
    if (beta != 0 && beta != 1) return lsk::fail(LSK_ERR_ARG, "%s: beta %d not in {0,1}", name, beta);
    if (rows == 0) return LSK_OK; // "Break out if there are no rows to process." (.cu:75-76)
    if (!rects || !y) return lsk::fail(LSK_ERR_ARG, "%s: null rects or output", name);
    if (nnz > 0 && (!col || !val)) return lsk::fail(LSK_ERR_ARG, "%s: null kernel arrays", name);
    if (x_hi >= x_lo && !x) return lsk::fail(LSK_ERR_ARG, "%s: null input vector", name);
    constexpr bool C32 = sizeof(CT) == 4;
    if (C32 ? (reinterpret_cast<uintptr_t>(rects) & 3u) != 0 : !lsk::aligned16(rects))
        return lsk::fail(LSK_ERR_ALIGN, "%s: %s", name,
                         C32 ? "row offsets must be 4-byte aligned" : "rects must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(col) & (sizeof(CT) - 1)) || (reinterpret_cast<uintptr_t>(val) & (sizeof(T) - 1)))
        return lsk::fail(LSK_ERR_ALIGN, "%s: col/val must be naturally aligned", name);

    if (x_hi < x_lo) { // an empty input domain admits no entry: y <- beta * y
        if (fused && beta != 0)
            return lsk::fail(LSK_ERR_UNSUPPORTED, "%s: empty input window with beta = 1", name);
        if (beta == 0) {
            zero_kernel<T><<<static_cast<unsigned>(std::min<int64_t>(lsk::ceil_div(rows, kBlock),
                                                                  lsk::kMaxGridBlocks)),
                             kBlock, 0, lsk::as_stream(stream)>>>(rows, y);
            LSK_LAUNCH_CHECK();
        }
        if (fused) {
            if (capacity < 1) return lsk::fail(LSK_ERR_ARG, "%s: no room for partial sums", name);
            store_zero_kernel<T><<<1, 1, 0, lsk::as_stream(stream)>>>(partials);
            LSK_LAUNCH_CHECK();
            *count = 1;
        }
        return LSK_OK;
    }
    const double mean = rows > 0 ? double(nnz) / double(rows) : 0.0;
    const typename RowField<CT>::type *rc = static_cast<const typename RowField<CT>::type *>(rects);
    hipStream_t s = lsk::as_stream(stream);
    const lsk::Options &tune = lsk::options();
    if (tune.csr_variant == 1 || fused || C32) {
        const int log_rw = pick_log_rows_per_wave<T>(rows, nnz);
        const int64_t blocks_w = lsk::ceil_div(rows, int64_t(4) << log_rw);
        if (blocks_w > 0x7fffffffLL) return lsk::fail(LSK_ERR_UNSUPPORTED, "%s: too many rows", name);
        if (fused && capacity < blocks_w * 4)
            return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %lld needed", name,
                             (long long)capacity, (long long)(blocks_w * 4));
        // Units per lane a mean tile needs: more than 4 -> the 8-unit instance.
        constexpr int UNIT = Cfg<T>::U;
        constexpr int KWIDE = 8;
        const double units = 1.05 * mean * double(1 << log_rw) / double(lsk::kWave * UNIT);
        const bool wide = tune.csr_max_units == 0 ? units > 4.0 : tune.csr_max_units == 8;
#define LSK_CSRW_LAUNCH(L, NT, KMAX)                                                              \
    csr_spmv_wave_kernel<T, CT, L, NT, KMAX, false><<<unsigned(blocks_w), kBlock, 0, s>>>(        \
        rows, rc, k_base, nnz, col, val, x, x_lo, x_hi, y, beta, lsk::xcd_remap_option(), nullptr, nullptr, skip_len)
#define LSK_CSRW_LAUNCH_DOT(L, KMAX)                                                              \
    csr_spmv_wave_kernel<T, CT, L, false, KMAX, true><<<unsigned(blocks_w), kBlock, 0, s>>>(      \
        rows, rc, k_base, nnz, col, val, x, x_lo, x_hi, y, beta, lsk::xcd_remap_option(), w, partials, 0)
#define LSK_CSRW_CASE(L)                                                                          \
    case L:                                                                                       \
        if (fused) {                                                                              \
            if (wide) LSK_CSRW_LAUNCH_DOT(L, KWIDE);                                              \
            else LSK_CSRW_LAUNCH_DOT(L, 4);                                                       \
        } else if constexpr (C32) {                                                               \
            if (wide) LSK_CSRW_LAUNCH(L, false, KWIDE);                                           \
            else LSK_CSRW_LAUNCH(L, false, 4);                                                    \
        } else if (tune.csr_nontemporal) {                                                        \
            if (wide) LSK_CSRW_LAUNCH(L, true, 8);                                                \
            else LSK_CSRW_LAUNCH(L, true, 4);                                                     \
        } else {                                                                                  \
            if (wide) LSK_CSRW_LAUNCH(L, false, 8);                                               \
            else LSK_CSRW_LAUNCH(L, false, 4);                                                    \
        }                                                                                         \
        break;
        switch (log_rw) {
            LSK_CSRW_CASE(0)
            LSK_CSRW_CASE(1)
            LSK_CSRW_CASE(2)
            LSK_CSRW_CASE(3)
            LSK_CSRW_CASE(4)
            LSK_CSRW_CASE(5)
            LSK_CSRW_CASE(6)
            LSK_CSRW_CASE(7)
            LSK_CSRW_CASE(8)
        }
#undef LSK_CSRW_LAUNCH
#undef LSK_CSRW_LAUNCH_DOT
#undef LSK_CSRW_CASE
        LSK_LAUNCH_CHECK();
        if (fused) *count = blocks_w * 4;
        return LSK_OK;
    }

    if constexpr (!C32) {
    // Rows per workgroup: as many as keep the tile's products inside the LDS buffer at the
    // piece's mean row length, a power of two in 4..256.
    const double want = mean > 0.0 ? 0.9 * Cfg<T>::CAP / mean : 256.0;
    int log_r = 8;
    while (log_r > 2 && double(1 << log_r) > want) --log_r;

    const int64_t blocks64 = lsk::ceil_div(rows, int64_t(1) << log_r);
    if (blocks64 > 0x7fffffffLL) return lsk::fail(LSK_ERR_UNSUPPORTED, "%s: too many rows", name);
    const unsigned blocks = static_cast<unsigned>(blocks64);
#define LSK_CSR_CASE(L)                                                                       \
    case L:                                                                                   \
        csr_spmv_kernel<T, L><<<blocks, kBlock, 0, s>>>(rows, rc, k_base, nnz, col, val, x,   \
                                                        x_lo, x_hi, y, beta, lsk::xcd_remap_option()); \
        break;
    switch (log_r) {
        LSK_CSR_CASE(2)
        LSK_CSR_CASE(3)
        LSK_CSR_CASE(4)
        LSK_CSR_CASE(5)
        LSK_CSR_CASE(6)
        LSK_CSR_CASE(7)
        LSK_CSR_CASE(8)
    }
#undef LSK_CSR_CASE
    LSK_LAUNCH_CHECK();
    }
    return LSK_OK;
}

// col32[k] <- col[k] - x_lo if x_lo <= col[k] <= x_hi, else 0xffffffff.
__global__ __launch_bounds__(kBlock) void compress_cols_kernel(int64_t nnz, const int64_t *__restrict__ col,
                                                               int64_t x_lo, int64_t x_hi,
                                                               uint32_t *__restrict__ col32) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t k = int64_t(blockIdx.x) * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int64_t c = col[k];
        col32[k] = (c >= x_lo && c <= x_hi) ? static_cast<uint32_t>(c - x_lo) : 0xffffffffu;
    }
}

// indptr conversion, bit-exact with convertGlobalRowptrToLocalIndPtr
// (src/CuSPARSEHelpers.hpp:20-30).
__global__ __launch_bounds__(kBlock) void rects_to_indptr_kernel(int64_t rows,
                                                                 const Rect *__restrict__ rects,
                                                                 long long *__restrict__ indptr) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const long long first = rects[0].lo;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < rows; i += stride) {
        indptr[i] = rects[i].lo - first;
        if (i == 0) indptr[rows] = rects[rows - 1].hi + 1 - first;
    }
}

// rowoff[r] <- rects[r].lo - k_base, rowoff[rows] <- rects[rows-1].hi + 1 - k_base; *bad is set if
// the rects are not back to back, ascending and inside the piece [k_base, k_base + nnz).
__global__ __launch_bounds__(kBlock) void compress_rows_kernel(int64_t rows, int64_t nnz,
                                                               const Rect *__restrict__ rects, int64_t k_base,
                                                               uint32_t *__restrict__ rowoff, int *__restrict__ bad) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    bool wrong = false;
    for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride) {
        const Rect rc = rects[r];
        const long long lo = rc.lo - k_base, end = rc.hi + 1 - k_base;
        wrong = wrong || lo < 0 || end < lo || end > nnz;
        if (r + 1 < rows) wrong = wrong || rects[r + 1].lo != rc.hi + 1;
        rowoff[r] = static_cast<uint32_t>(lo);
        if (r + 1 == rows) rowoff[rows] = static_cast<uint32_t>(end);
    }
    if (wrong) *bad = 1; // every writer stores the same value
}

__global__ void clear_flag_kernel(int *flag) { *flag = 0; }

// Times the candidates of the tile -> XCD placement on this piece and reports the fastest.
template <typename T, typename CT = int64_t>
int tune_csr_spmv(const char *name, int64_t rows, int64_t nnz, const void *rects, int64_t k_base,
                  const CT *col, const T *val, const T *x, int64_t x_lo, int64_t x_hi, T *y,
                  lsk_stream_t stream, int *remap_out, float *ms_out) {
    static const int candidates[] = {0, 16, 64, 256, 1024, 4096};
    constexpr int kWarm = 2, kTimed = 5;
    LSK_NOT_WHILE_RECORDING(name, stream);
    int &placement = lsk::xcd_remap_option();
    const int saved = placement;
    hipStream_t s = lsk::as_stream(stream);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    LSK_HIP(hipEventCreate(&t0));
    if (const hipError_t e = hipEventCreate(&t1); e != hipSuccess) {
        (void)hipEventDestroy(t0);
        return lsk::hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
    }
    int best = saved, status = LSK_OK;
    float best_ms = -1.0f;
    for (int c : candidates) {
        placement = c;
        for (int i = 0; i < kWarm && status == LSK_OK; ++i)
            status = launch_csr_spmv<T, CT>(name, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, 0, stream);
        if (status != LSK_OK) break;
        if (hipEventRecord(t0, s) != hipSuccess) { status = lsk::fail(LSK_ERR_HIP, "%s: event record", name); break; }
        for (int i = 0; i < kTimed && status == LSK_OK; ++i)
            status = launch_csr_spmv<T, CT>(name, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, 0, stream);
        if (status != LSK_OK) break;
        float ms = 0.0f;
        if (hipEventRecord(t1, s) != hipSuccess || hipEventSynchronize(t1) != hipSuccess ||
            hipEventElapsedTime(&ms, t0, t1) != hipSuccess) {
            status = lsk::fail(LSK_ERR_HIP, "%s: event timing", name);
            break;
        }
        ms /= kTimed;
        if (std::getenv("LSK_TUNE_VERBOSE"))
            std::fprintf(stderr, "[lsk] %s: xcd_remap %d -> %.3f ms per launch\n", name, c, ms);
        if (best_ms < 0.0f || ms < best_ms) {
            best_ms = ms;
            best = c;
        }
    }
    placement = saved;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (status != LSK_OK) return status;
    if (remap_out) *remap_out = best;
    if (ms_out) *ms_out = best_ms;
    return LSK_OK;
}


// ---------------------------------------------------------------- rows too long for one wave
//
// The wave-tile kernel gives every wave a fixed number of rows; a row of 10^5 or 10^6 entries in a
// matrix whose mean row has 30 keeps one wave busy for milliseconds after all others have finished.
// A plan made once per piece lists those rows; the split product leaves them out of the tile kernel,
// multiplies each in chunks of kLongChunk entries, one workgroup per chunk, and adds a row's chunk sums
// in chunk order: the result does not depend on how the hardware schedules anything.
constexpr int kLongThreshold = 8192;    // entries: the default
constexpr int kLongThresholdMin = 2048; // not below any instance's wave buffer (CAP / 4 <= 1920)
constexpr int kLongChunk = 8192;

__global__ __launch_bounds__(kBlock) void long_count_kernel(int64_t rows, const uint32_t *__restrict__ off,
                                                            uint32_t threshold, unsigned int *__restrict__ count) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    unsigned int mine = 0;
    for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride)
        mine += (off[r + 1] - off[r] > threshold) ? 1u : 0u;
    if (mine) atomicAdd(count, mine);
}

__global__ __launch_bounds__(kBlock) void long_list_kernel(int64_t rows, const uint32_t *__restrict__ off,
                                                           uint32_t threshold, unsigned int *__restrict__ fill,
                                                           uint32_t *__restrict__ row_out) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x; r < rows; r += stride)
        if (off[r + 1] - off[r] > threshold) row_out[atomicAdd(fill, 1u)] = static_cast<uint32_t>(r);
}

// nchunks[q] = chunks of long row q (slot nlong: 0, so that an exclusive sum ends with the total)
__global__ __launch_bounds__(kBlock) void long_chunks_kernel(int64_t nlong, const uint32_t *__restrict__ row,
                                                             const uint32_t *__restrict__ off,
                                                             uint32_t *__restrict__ nchunks) {
    const int64_t q = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (q > nlong) return;
    const uint32_t len = q < nlong ? off[row[q] + 1] - off[row[q]] : 0u;
    nchunks[q] = (len + kLongChunk - 1) / kLongChunk;
}

__global__ __launch_bounds__(kBlock) void long_chunk_rows_kernel(int64_t nlong, const uint32_t *__restrict__ chunk_first,
                                                                 uint32_t *__restrict__ chunk_row) {
    const int64_t q = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (q >= nlong) return;
    for (uint32_t w = chunk_first[q]; w < chunk_first[q + 1]; ++w) chunk_row[w] = static_cast<uint32_t>(q);
}

// One chunk of one long row per workgroup: lane t adds the products of the chunk's entries t, t + 256, ...
// in that order (eight of them in flight), then the lanes of a wave and the four waves are added in a fixed
// order.  off0: the piece's row offsets (of plan row 0).
template <typename T>
__global__ __launch_bounds__(kBlock) void long_chunk_kernel(
    const uint32_t *__restrict__ lrow, const uint32_t *__restrict__ chunk_first, const uint32_t *__restrict__ chunk_row,
    const uint32_t *__restrict__ off0, const uint32_t *__restrict__ col, const T *__restrict__ val,
    const T *__restrict__ x, uint32_t x_last, long long row_begin, long long row_end, T *__restrict__ partial) {
    __shared__ T s_part[kBlock / lsk::kWave];
    const uint32_t w = blockIdx.x, q = chunk_row[w], r = lrow[q];
    if (static_cast<long long>(r) < row_begin || static_cast<long long>(r) >= row_end) return; // uniform
    const uint32_t lo = off0[r], hi = off0[r + 1];
    const uint32_t start = lo + (w - chunk_first[q]) * static_cast<uint32_t>(kLongChunk);
    const uint32_t end = hi - start > static_cast<uint32_t>(kLongChunk) ? start + kLongChunk : hi;
    constexpr int K = 8;
    T acc = T(0);
    for (uint32_t k0 = start + threadIdx.x; k0 < end; k0 += kBlock * K) {
        T v[K];
        uint32_t c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t k = k0 + j * kBlock;
            const uint32_t kc = k < end ? k : end - 1; // past the end: re-read the last entry, dropped below
            v[j] = __builtin_nontemporal_load(val + kc);
            c[j] = __builtin_nontemporal_load(col + kc);
        }
        T xv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) xv[j] = x[c[j] <= x_last ? c[j] : 0u];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const T prod = c[j] <= x_last ? v[j] * xv[j] : T(0); // the CPU body's input-domain guard
            acc += (k0 + j * kBlock < end) ? prod : T(0);
        }
    }
    acc = lsk::wave_sum(acc);
    if (threadIdx.x % lsk::kWave == 0) s_part[threadIdx.x / lsk::kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T sum = s_part[0];
#pragma unroll
        for (int i = 1; i < kBlock / lsk::kWave; ++i) sum += s_part[i];
        partial[w] = sum;
    }
}

// One wave per long row: its chunk sums in chunk order (lane l takes chunks l, l + 64, ...), then y.
template <typename T>
__global__ __launch_bounds__(kBlock) void long_fold_kernel(int64_t nlong, const uint32_t *__restrict__ lrow,
                                                           const uint32_t *__restrict__ chunk_first,
                                                           const T *__restrict__ partial, long long row_begin,
                                                           long long row_end, T *__restrict__ y, int beta) {
    const int64_t q = int64_t(blockIdx.x) * (kBlock / lsk::kWave) + threadIdx.x / lsk::kWave;
    if (q >= nlong) return;
    const long long r = lrow[q];
    if (r < row_begin || r >= row_end) return;
    const int lane = threadIdx.x % lsk::kWave;
    T acc = T(0);
    for (uint32_t j = chunk_first[q] + lane; j < chunk_first[q + 1]; j += lsk::kWave) acc += partial[j];
    acc = lsk::wave_sum(acc);
    if (lane == 0) {
        T *yp = y + (r - row_begin);
        *yp = beta ? *yp + acc : acc;
    }
}

void release_long_rows(lsk_csr_long_rows *m) {
    if (!m) return;
    (void)hipFree(m->row);
    (void)hipFree(m->chunk_first);
    (void)hipFree(m->chunk_row);
    (void)hipFree(m->partial);
    delete m;
}

int create_long_rows(int64_t rows, const uint32_t *rowoff, int64_t threshold, lsk_stream_t stream,
                     lsk_csr_long_rows **out) {
    const char *name = "lsk_csr_long_rows_create";
    if (!out) return lsk::fail(LSK_ERR_ARG, "%s: null handle pointer", name);
    *out = nullptr;
    if (rows < 0 || threshold < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative size", name);
    if (threshold == 0) threshold = kLongThreshold;
    if (threshold < kLongThresholdMin || threshold > 0x3fffffff)
        return lsk::fail(LSK_ERR_ARG, "%s: threshold %lld (at least %d entries)", name, (long long)threshold, kLongThresholdMin);
    if (rows == 0) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: no rows", name);
    if (!rowoff) return lsk::fail(LSK_ERR_ARG, "%s: null row offsets", name);
    if (rows > 0xfffffffeLL) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: %lld rows", name, (long long)rows);
    hipStream_t s = lsk::as_stream(stream);
    // a counter of this call's own (set-up time): several host threads may make plans on one device at once
    unsigned int *counter = nullptr;
    LSK_HIP(hipMalloc(reinterpret_cast<void **>(&counter), sizeof(unsigned int)));
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(lsk::ceil_div(rows, kBlock), lsk::kMaxGridBlocks));
    unsigned int h_count = 0;
    hipError_t e = hipMemcpyAsync(counter, &h_count, sizeof(h_count), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        long_count_kernel<<<grid, kBlock, 0, s>>>(rows, rowoff, static_cast<uint32_t>(threshold), counter);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_count, counter, sizeof(h_count), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess || h_count == 0) {
        (void)hipFree(counter);
        if (e != hipSuccess) return lsk::hip_fail(e, "counting the long rows", __FILE__, __LINE__);
        return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: no row longer than %lld entries", name, (long long)threshold);
    }

    lsk_csr_long_rows *m = new lsk_csr_long_rows;
    m->rows = rows;
    m->nlong = h_count;
    m->threshold = static_cast<int>(threshold);
    uint32_t *unsorted = nullptr, *nchunks = nullptr;
    void *temp = nullptr;
    int status = LSK_OK;
#define LSK_TRY(expr)                                                                                    \
    do {                                                                                                 \
        if (status == LSK_OK) {                                                                          \
            const hipError_t e_ = (expr);                                                                \
            if (e_ != hipSuccess) status = lsk::hip_fail(e_, #expr, __FILE__, __LINE__);                 \
        }                                                                                                \
    } while (0)
    const size_t nl = static_cast<size_t>(m->nlong);
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&unsorted), nl * sizeof(uint32_t)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->row), nl * sizeof(uint32_t)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&nchunks), (nl + 1) * sizeof(uint32_t)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->chunk_first), (nl + 1) * sizeof(uint32_t)));
    unsigned int zero = 0;
    LSK_TRY(hipMemcpyAsync(counter, &zero, sizeof(zero), hipMemcpyHostToDevice, s));
    if (status == LSK_OK) {
        long_list_kernel<<<grid, kBlock, 0, s>>>(rows, rowoff, static_cast<uint32_t>(threshold), counter, unsorted);
        LSK_TRY(hipGetLastError());
    }
    size_t sort_bytes = 0, scan_bytes = 0;
    LSK_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, unsorted, m->row, static_cast<int>(nl), 0, 32, s));
    LSK_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, nchunks, m->chunk_first, static_cast<int>(nl + 1), s));
    size_t temp_bytes = std::max(sort_bytes, scan_bytes);
    LSK_TRY(hipMalloc(&temp, std::max<size_t>(temp_bytes, 1)));
    size_t tb = temp_bytes;
    LSK_TRY(hipcub::DeviceRadixSort::SortKeys(temp, tb, unsorted, m->row, static_cast<int>(nl), 0, 32, s));
    const unsigned lgrid = static_cast<unsigned>(lsk::ceil_div(static_cast<int64_t>(nl) + 1, kBlock));
    if (status == LSK_OK) {
        long_chunks_kernel<<<lgrid, kBlock, 0, s>>>(m->nlong, m->row, rowoff, nchunks);
        LSK_TRY(hipGetLastError());
    }
    tb = temp_bytes;
    LSK_TRY(hipcub::DeviceScan::ExclusiveSum(temp, tb, nchunks, m->chunk_first, static_cast<int>(nl + 1), s));
    uint32_t h_chunks = 0;
    LSK_TRY(hipMemcpyAsync(&h_chunks, m->chunk_first + nl, sizeof(h_chunks), hipMemcpyDeviceToHost, s));
    LSK_TRY(hipStreamSynchronize(s));
    if (status == LSK_OK) {
        m->chunks = h_chunks;
        LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->chunk_row), std::max<size_t>(h_chunks, 1) * sizeof(uint32_t)));
        LSK_TRY(hipMalloc(&m->partial, std::max<size_t>(h_chunks, 1) * 8));
    }
    if (status == LSK_OK) {
        long_chunk_rows_kernel<<<lgrid, kBlock, 0, s>>>(m->nlong, m->chunk_first, m->chunk_row);
        LSK_TRY(hipGetLastError());
    }
    LSK_TRY(hipStreamSynchronize(s)); // set-up time: the scratch is freed before returning
#undef LSK_TRY
    (void)hipFree(unsorted);
    (void)hipFree(nchunks);
    (void)hipFree(temp);
    (void)hipFree(counter);
    if (status != LSK_OK) {
        release_long_rows(m);
        return status;
    }
    *out = m;
    return LSK_OK;
}

// The product of rows row_begin .. row_begin + rows - 1 of the plan's piece; rowoff and y are those of
// row_begin, col32 / val / x the piece's.
template <typename T>
int spmv_c32_split(const char *name, const lsk_csr_long_rows *plan, int64_t row_begin, int64_t rows, int64_t nnz,
                   const uint32_t *rowoff, const uint32_t *col32, const T *val, const T *x, int64_t x_extent, T *y,
                   int beta, lsk_stream_t stream) {
    if (x_extent < 0 || x_extent > 0xffffffffLL) return lsk::fail(LSK_ERR_ARG, "%s: window of %lld columns", name, (long long)x_extent);
    if (nnz > 0xffffffffLL) return lsk::fail(LSK_ERR_ARG, "%s: %lld entries", name, (long long)nnz);
    if (!plan)
        return launch_csr_spmv<T, uint32_t>(name, rows, nnz, rowoff, 0, col32, val, x, 0, x_extent - 1, y, beta, stream);
    if (row_begin < 0 || rows < 0 || row_begin + rows > plan->rows)
        return lsk::fail(LSK_ERR_ARG, "%s: rows %lld .. %lld of a plan made for %lld rows", name, (long long)row_begin,
                         (long long)(row_begin + rows - 1), (long long)plan->rows);
    const int rc = launch_csr_spmv<T, uint32_t>(name, rows, nnz, rowoff, 0, col32, val, x, 0, x_extent - 1, y, beta,
                                                stream, nullptr, nullptr, 0, nullptr, plan->threshold);
    if (rc != LSK_OK || rows == 0 || x_extent == 0 || plan->chunks == 0) return rc; // (an empty window admits no entry)
    hipStream_t s = lsk::as_stream(stream);
    long_chunk_kernel<T><<<static_cast<unsigned>(plan->chunks), kBlock, 0, s>>>(
        plan->row, plan->chunk_first, plan->chunk_row, rowoff - row_begin, col32, val, x,
        static_cast<uint32_t>(x_extent - 1), row_begin, row_begin + rows, static_cast<T *>(plan->partial));
    LSK_LAUNCH_CHECK();
    long_fold_kernel<T><<<static_cast<unsigned>(lsk::ceil_div(plan->nlong, kBlock / lsk::kWave)), kBlock, 0, s>>>(
        plan->nlong, plan->row, plan->chunk_first, static_cast<const T *>(plan->partial), row_begin, row_begin + rows, y,
        beta);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_csr_long_rows_create(int64_t rows, const uint32_t *rowoff32, int64_t threshold, lsk_stream_t stream,
                             lsk_csr_long_rows_t **out) {
    return create_long_rows(rows, rowoff32, threshold, stream, out);
}

int lsk_csr_long_rows_info(const lsk_csr_long_rows_t *plan, int64_t *out4) {
    if (!plan || !out4) return lsk::fail(LSK_ERR_ARG, "lsk_csr_long_rows_info: null pointer");
    out4[0] = plan->nlong;
    out4[1] = plan->chunks;
    out4[2] = plan->threshold;
    out4[3] = kLongChunk;
    return LSK_OK;
}

int lsk_csr_long_rows_destroy(lsk_csr_long_rows_t *plan) {
    release_long_rows(plan);
    return LSK_OK;
}

int lsk_csr_spmv_c32_split_f32(const lsk_csr_long_rows_t *plan, int64_t row_begin, int64_t rows, int64_t nnz,
                               const uint32_t *rowoff32, const uint32_t *col32, const float *val, const float *x,
                               int64_t x_extent, float *y, int beta, lsk_stream_t stream) {
    return spmv_c32_split<float>("lsk_csr_spmv_c32_split_f32", plan, row_begin, rows, nnz, rowoff32, col32, val, x,
                                 x_extent, y, beta, stream);
}
int lsk_csr_spmv_c32_split_f64(const lsk_csr_long_rows_t *plan, int64_t row_begin, int64_t rows, int64_t nnz,
                               const uint32_t *rowoff32, const uint32_t *col32, const double *val, const double *x,
                               int64_t x_extent, double *y, int beta, lsk_stream_t stream) {
    return spmv_c32_split<double>("lsk_csr_spmv_c32_split_f64", plan, row_begin, rows, nnz, rowoff32, col32, val, x,
                                  x_extent, y, beta, stream);
}

int64_t lsk_csr_spmv_dot_partials(int64_t rows, int64_t nnz, int elem_size) {
    if (rows <= 0) return 1;
    const int log_rw = elem_size == 4 ? pick_log_rows_per_wave<float>(rows, nnz)
                                      : pick_log_rows_per_wave<double>(rows, nnz);
    return std::max<int64_t>(lsk::ceil_div(rows, int64_t(4) << log_rw) * 4, 1);
}

int lsk_csr_spmv_dot_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                         const int64_t *col, const float *val, const float *x, int64_t x_lo,
                         int64_t x_hi, float *y, int beta, const float *w, float *partials,
                         int64_t capacity, int64_t *count, lsk_stream_t stream) {
    if (!w) return lsk::fail(LSK_ERR_ARG, "lsk_csr_spmv_dot_f32: null w");
    return launch_csr_spmv<float>("lsk_csr_spmv_dot_f32", rows, nnz, rects, k_base, col, val, x, x_lo, x_hi,
                                  y, beta, stream, w, partials, capacity, count);
}

int lsk_csr_spmv_dot_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                         const int64_t *col, const double *val, const double *x, int64_t x_lo,
                         int64_t x_hi, double *y, int beta, const double *w, double *partials,
                         int64_t capacity, int64_t *count, lsk_stream_t stream) {
    if (!w) return lsk::fail(LSK_ERR_ARG, "lsk_csr_spmv_dot_f64: null w");
    return launch_csr_spmv<double>("lsk_csr_spmv_dot_f64", rows, nnz, rects, k_base, col, val, x, x_lo, x_hi,
                                   y, beta, stream, w, partials, capacity, count);
}

int lsk_csr_spmv_tune_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, const float *val, const float *x, int64_t x_lo,
                          int64_t x_hi, float *y, lsk_stream_t stream, int *remap_out, float *ms_out) {
    return tune_csr_spmv<float>("lsk_csr_spmv_tune_f32", rows, nnz, rects, k_base, col, val, x, x_lo, x_hi,
                                y, stream, remap_out, ms_out);
}

int lsk_csr_spmv_tune_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                          const int64_t *col, const double *val, const double *x, int64_t x_lo,
                          int64_t x_hi, double *y, lsk_stream_t stream, int *remap_out, float *ms_out) {
    return tune_csr_spmv<double>("lsk_csr_spmv_tune_f64", rows, nnz, rects, k_base, col, val, x, x_lo, x_hi,
                                 y, stream, remap_out, ms_out);
}


int lsk_csr_spmv_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                     const int64_t *col, const float *val, const float *x, int64_t x_lo,
                     int64_t x_hi, float *y, int beta, lsk_stream_t stream) {
    return launch_csr_spmv<float>("lsk_csr_spmv_f32", rows, nnz, rects, k_base, col, val, x, x_lo,
                                  x_hi, y, beta, stream);
}

int lsk_csr_spmv_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                     const int64_t *col, const double *val, const double *x, int64_t x_lo,
                     int64_t x_hi, double *y, int beta, lsk_stream_t stream) {
    return launch_csr_spmv<double>("lsk_csr_spmv_f64", rows, nnz, rects, k_base, col, val, x, x_lo,
                                   x_hi, y, beta, stream);
}

int lsk_csr_compress_cols_u32(int64_t nnz, const int64_t *col, int64_t x_lo, int64_t x_hi,
                               uint32_t *col32, lsk_stream_t stream) {
    if (nnz < 0) return lsk::fail(LSK_ERR_ARG, "lsk_csr_compress_cols_u32: negative nnz");
    if (nnz == 0) return LSK_OK;
    if (!col || !col32) return lsk::fail(LSK_ERR_ARG, "lsk_csr_compress_cols_u32: null pointer");
    if (x_hi >= x_lo && x_hi - x_lo >= 0xffffffffLL)
        return lsk::decline(LSK_ERR_UNSUPPORTED,
                            "lsk_csr_compress_cols_u32: a window of %lld columns does not fit 32-bit offsets",
                            (long long)(x_hi - x_lo + 1));
    int64_t blocks = lsk::ceil_div(nnz, kBlock * 4);
    if (blocks > lsk::kMaxGridBlocks) blocks = lsk::kMaxGridBlocks;
    compress_cols_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(nnz, col, x_lo,
                                                                                               x_hi, col32);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

int lsk_csr_compress_rows_u32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                              uint32_t *rowoff32, lsk_stream_t stream) {
    if (rows < 0 || nnz < 0) return lsk::fail(LSK_ERR_ARG, "lsk_csr_compress_rows_u32: negative size");
    if (!rowoff32) return lsk::fail(LSK_ERR_ARG, "lsk_csr_compress_rows_u32: null output");
    if (nnz > 0xffffffffLL)
        return lsk::decline(LSK_ERR_UNSUPPORTED, "lsk_csr_compress_rows_u32: %lld entries do not fit 32-bit offsets",
                         (long long)nnz);
    hipStream_t s = lsk::as_stream(stream);
    if (rows == 0) {
        const uint32_t zero = 0;
        LSK_HIP(hipMemcpyAsync(rowoff32, &zero, sizeof(zero), hipMemcpyHostToDevice, s));
        LSK_HIP(hipStreamSynchronize(s));
        return LSK_OK;
    }
    if (!rects) return lsk::fail(LSK_ERR_ARG, "lsk_csr_compress_rows_u32: null rects");
    if (!lsk::aligned16(rects))
        return lsk::fail(LSK_ERR_ALIGN, "lsk_csr_compress_rows_u32: rects must be 16-byte aligned");
    // a flag of this call's own (set-up time): several host threads may compress pieces on one device at once
    int *flag = nullptr;
    LSK_HIP(hipMalloc(reinterpret_cast<void **>(&flag), sizeof(int)));
    clear_flag_kernel<<<1, 1, 0, s>>>(flag);
    int64_t blocks = lsk::ceil_div(rows, kBlock);
    if (blocks > lsk::kMaxGridBlocks) blocks = lsk::kMaxGridBlocks;
    compress_rows_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, s>>>(
        rows, nnz, reinterpret_cast<const Rect *>(rects), k_base, rowoff32, flag);
    int bad = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, sizeof(bad), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(flag);
    if (e != hipSuccess) return lsk::hip_fail(e, "lsk_csr_compress_rows_u32", __FILE__, __LINE__);
    if (bad)
        return lsk::decline(LSK_ERR_UNSUPPORTED,
                            "lsk_csr_compress_rows_u32: rects are not back to back inside the piece");
    return LSK_OK;
}

#define LSK_C32_WINDOW(NAME)                                                                          \
    if (x_extent < 0 || x_extent > 0xffffffffLL)                                                      \
        return lsk::fail(LSK_ERR_ARG, NAME ": window of %lld columns", (long long)x_extent);          \
    if (nnz > 0xffffffffLL) return lsk::fail(LSK_ERR_ARG, NAME ": %lld entries", (long long)nnz)
#define LSK_C32(SUF, T)                                                                               \
    int lsk_csr_spmv_c32_##SUF(int64_t rows, int64_t nnz, const uint32_t *rowoff32,                   \
                               const uint32_t *col32, const T *val, const T *x, int64_t x_extent, T *y, \
                               int beta, lsk_stream_t stream) {                                       \
        LSK_C32_WINDOW("lsk_csr_spmv_c32_" #SUF);                                                     \
        return launch_csr_spmv<T, uint32_t>("lsk_csr_spmv_c32_" #SUF, rows, nnz, rowoff32, 0, col32, val, x, \
                                            0, x_extent - 1, y, beta, stream);                        \
    }                                                                                                 \
    int lsk_csr_spmv_c32_dot_##SUF(int64_t rows, int64_t nnz, const uint32_t *rowoff32,               \
                                   const uint32_t *col32, const T *val, const T *x, int64_t x_extent, \
                                   T *y, int beta, const T *w, T *partials, int64_t capacity,         \
                                   int64_t *count, lsk_stream_t stream) {                             \
        if (!w) return lsk::fail(LSK_ERR_ARG, "lsk_csr_spmv_c32_dot_" #SUF ": null w");               \
        LSK_C32_WINDOW("lsk_csr_spmv_c32_dot_" #SUF);                                                 \
        return launch_csr_spmv<T, uint32_t>("lsk_csr_spmv_c32_dot_" #SUF, rows, nnz, rowoff32, 0, col32, val, \
                                            x, 0, x_extent - 1, y, beta, stream, w, partials, capacity, count); \
    }                                                                                                 \
    int lsk_csr_spmv_c32_tune_##SUF(int64_t rows, int64_t nnz, const uint32_t *rowoff32,              \
                                    const uint32_t *col32, const T *val, const T *x, int64_t x_extent, \
                                    T *y, lsk_stream_t stream, int *remap_out, float *ms_out) {       \
        LSK_C32_WINDOW("lsk_csr_spmv_c32_tune_" #SUF);                                                \
        return tune_csr_spmv<T, uint32_t>("lsk_csr_spmv_c32_tune_" #SUF, rows, nnz, rowoff32, 0, col32, val, \
                                          x, 0, x_extent - 1, y, stream, remap_out, ms_out);          \
    }
LSK_C32(f32, float)
LSK_C32(f64, double)
#undef LSK_C32_WINDOW
#undef LSK_C32

int lsk_rects_to_indptr_i64(int64_t rows, const int64_t *rects, int64_t *indptr,
                            lsk_stream_t stream) {
    if (rows < 0) return lsk::fail(LSK_ERR_ARG, "lsk_rects_to_indptr_i64: negative rows");
    if (rows == 0) return LSK_OK;
    if (!rects || !indptr) return lsk::fail(LSK_ERR_ARG, "lsk_rects_to_indptr_i64: null pointer");
    int64_t blocks = lsk::ceil_div(rows, kBlock);
    if (blocks > lsk::kMaxGridBlocks) blocks = lsk::kMaxGridBlocks;
    rects_to_indptr_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(
        rows, reinterpret_cast<const Rect *>(rects), reinterpret_cast<long long *>(indptr));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // extern "C"
