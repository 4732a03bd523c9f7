// The binned COO product, for matrices whose columns have no locality (the x gathers of the scan
// kernel then pull a 128-byte line through L2 for every 8 bytes they use, and that, not the entry
// stream, bounds it).  What COOMatvecTask computes (src/COOMatrixTasks.cpp:58-76,
// src/COOMatrixTasks.cu:12-43: y[row] += val * x[col] for every entry inside the bounds), in two
// streaming passes over a set-up-time copy of the entries:
//
//   pass 1 (binned_multiply_kernel): the copy is ordered by (block of columns, tile of rows).  A
//     workgroup stages its block of x in LDS (128 KiB), streams a contiguous run of the block's
//     entries -- value and a 16-bit column offset -- and writes the products val * x[col] to a
//     workspace in the same order.  Nothing is gathered from memory.
//   pass 2 (binned_reduce_kernel): a workgroup owns a tile of rows of y in LDS and reads the tile's
//     segments of the workspace (one contiguous run per block of columns) with the entries' 16-bit
//     row offsets, adding with LDS atomics; the tile is written once.  No global atomics.
//
// 2 (s + 1) + s bytes per entry cross HBM instead of 8 + s, all of them as contiguous streams.
// Like the reference's atomicAdd kernel the sums depend on the order in which the hardware serves
// the additions (here: LDS atomics of different waves), so results agree with the CPU loop to
// rounding, not bit for bit.
#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "lsk_common.hpp"

struct lsk_coo_binned {
    int elem_bytes = 0;
    int device = 0;
    int64_t nnz = 0;         // entries handed over
    int64_t live = 0;        // ... of which inside both windows
    int64_t padded = 0;      // entries of the copy (every segment padded to a multiple of kPad)
    int64_t rows = 0;        // y window
    int64_t x_extent = 0;    // x window
    int block_cols = 0;      // columns per block (what pass 1 stages of x)
    int tile_rows = 0;       // rows per tile (what pass 2 holds of y)
    int64_t nblocks = 0, ntiles = 0;
    int64_t units = 0;       // workgroups of pass 1
    int64_t segments = 0;    // non-empty (block, tile) segments
    void *val = nullptr;     // padded values, (block, tile) order
    void *prod = nullptr;    // workspace: the products, same order
    uint16_t *lc = nullptr;  // column offset inside the block (block_cols for a pad: a slot that holds 0)
    uint16_t *lr = nullptr;  // row offset inside the tile (tile_rows for a pad: a slot nobody reads)
    uint32_t *src = nullptr; // position of the entry in the caller's arrays (0xffffffff for a pad)
    uint32_t *unit = nullptr;    // 4 per unit: block, first entry, end, stride of its runs
    uint32_t *segptr = nullptr;  // ntiles + 1
    uint32_t *segs = nullptr;    // 2 per segment: first entry, length
    std::vector<uint32_t> unit_first; // host: first unit of every block of columns (nblocks + 1; units are in block order)
};

namespace {

using lsk::kWave;

constexpr int kBinBlock = 1024;              // threads per workgroup of the two product kernels
constexpr int kBinWaves = kBinBlock / kWave; // 16
constexpr int kPad = 8;                      // segments are padded to multiples of 8 entries
#ifndef LSK_BIN_UNIT_LOG
#define LSK_BIN_UNIT_LOG 15
#endif
#ifndef LSK_BIN_U1
#define LSK_BIN_U1 4
#endif
#ifndef LSK_BIN_U2
#define LSK_BIN_U2 4
#endif
constexpr int kU1 = LSK_BIN_U1;              // 16-byte value loads per lane in flight, pass 1
constexpr int kU2 = LSK_BIN_U2;              // ... pass 2
constexpr int64_t kMaxBins = int64_t(1) << 24;
constexpr int64_t kUnitEntries = int64_t(1) << LSK_BIN_UNIT_LOG; // entries one workgroup of pass 1 streams
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kSetupBlock = 256;

typedef double double2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));

template <typename T> struct Lane;
template <> struct Lane<double> {
    static constexpr int E = 2; // entries per 16 bytes
    using V = double2v;
    using I = ushort2v;
};
template <> struct Lane<float> {
    static constexpr int E = 4;
    using V = float4v;
    using I = ushort4v;
};

int grid_for(int64_t n) {
    int64_t b = lsk::ceil_div(std::max<int64_t>(n, 1), kSetupBlock);
    return static_cast<int>(std::min<int64_t>(b, 16 * lsk::kMaxGridBlocks));
}

// ---------------------------------------------------------------- set-up kernels (integer work)

// Bin of every entry: block of columns x tile of rows, or `nbins` for an entry outside a window.
__global__ __launch_bounds__(kSetupBlock) void bin_keys_kernel(
    int64_t nnz, const uint32_t *__restrict__ row32, const uint32_t *__restrict__ col32, uint32_t y_first,
    uint32_t y_last, uint32_t x_extent, uint32_t block_cols, uint32_t tile_rows, uint32_t ntiles, uint32_t nbins,
    uint32_t *__restrict__ keys, uint32_t *__restrict__ pos) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t k = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; k < nnz; k += stride) {
        const uint32_t r = row32[k], c = col32[k];
        const bool live = r != kNone && c != kNone && r >= y_first && r <= y_last && c < x_extent;
        keys[k] = live ? (c / block_cols) * ntiles + (r - y_first) / tile_rows : nbins;
        pos[k] = static_cast<uint32_t>(k);
    }
}

// first[b] = number of sorted keys below b, b = 0 .. nbins (first[nbins] = live entries).
__global__ __launch_bounds__(kSetupBlock) void bin_first_kernel(int64_t n, const uint32_t *__restrict__ keys,
                                                                int64_t nbins, uint32_t *__restrict__ first) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t b = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; b <= nbins; b += stride) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (static_cast<int64_t>(keys[mid]) < b) lo = mid + 1;
            else hi = mid;
        }
        first[b] = static_cast<uint32_t>(lo);
    }
}

// Padded length of every bin (slot nbins: 0, so that an exclusive sum ends with the total).
__global__ __launch_bounds__(kSetupBlock) void bin_padded_kernel(int64_t nbins, const uint32_t *__restrict__ first,
                                                                 long long *__restrict__ padded) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t b = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; b <= nbins; b += stride) {
        const long long n = b < nbins ? static_cast<long long>(first[b + 1]) - first[b] : 0;
        padded[b] = (n + kPad - 1) / kPad * kPad;
    }
}

// start[c] = where block c's entries start in the copy, c = 0 .. nblocks.
__global__ __launch_bounds__(kSetupBlock) void block_starts_kernel(int64_t nblocks, int64_t ntiles,
                                                                   const long long *__restrict__ where,
                                                                   long long *__restrict__ start) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock; // (grid_for caps the grid)
    for (int64_t c = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; c <= nblocks; c += stride)
        start[c] = where[c * ntiles];
}

template <typename T>
__global__ __launch_bounds__(kSetupBlock) void fill_pads_kernel(int64_t padded, T *__restrict__ val,
                                                                uint16_t *__restrict__ lc, uint16_t *__restrict__ lr,
                                                                uint32_t *__restrict__ src, uint16_t block_cols,
                                                                uint16_t tile_rows) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t p = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; p < padded; p += stride) {
        val[p] = T(0);
        lc[p] = block_cols;
        lr[p] = tile_rows;
        src[p] = kNone;
    }
}

// Entry i of the sorted order goes to where[bin] + (i - first[bin]).
template <typename T>
__global__ __launch_bounds__(kSetupBlock) void place_entries_kernel(
    int64_t live, const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ first, const long long *__restrict__ where, const uint32_t *__restrict__ row32,
    const uint32_t *__restrict__ col32, const T *__restrict__ val_in, uint32_t y_first, uint32_t block_cols,
    uint32_t tile_rows, uint32_t ntiles, T *__restrict__ val, uint16_t *__restrict__ lc, uint16_t *__restrict__ lr,
    uint32_t *__restrict__ src) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t i = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; i < live; i += stride) {
        const uint32_t bin = keys_sorted[i], o = order[i];
        const long long p = where[bin] + (i - first[bin]);
        const uint32_t c = bin / ntiles, r = bin % ntiles;
        val[p] = val_in[o];
        lc[p] = static_cast<uint16_t>(col32[o] - c * block_cols);
        lr[p] = static_cast<uint16_t>(row32[o] - y_first - r * tile_rows);
        src[p] = o;
    }
}

template <typename T>
__global__ __launch_bounds__(kSetupBlock) void refill_values_kernel(int64_t padded, const uint32_t *__restrict__ src,
                                                                    const T *__restrict__ val_in,
                                                                    T *__restrict__ val) {
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t p = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; p < padded; p += stride) {
        const uint32_t o = src[p];
        val[p] = o == kNone ? T(0) : val_in[o];
    }
}

// Tile-major view of the bins: flag[r * nblocks + c] = bin (c, r) is not empty (slot ntiles * nblocks: 0).
__global__ __launch_bounds__(kSetupBlock) void segment_flags_kernel(int64_t nblocks, int64_t ntiles,
                                                                    const long long *__restrict__ padded,
                                                                    int *__restrict__ flag) {
    const int64_t n = nblocks * ntiles;
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t j = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; j <= n; j += stride) {
        if (j == n) { flag[j] = 0; continue; }
        const int64_t r = j / nblocks, c = j % nblocks;
        flag[j] = padded[c * ntiles + r] > 0 ? 1 : 0;
    }
}

__global__ __launch_bounds__(kSetupBlock) void segment_lists_kernel(int64_t nblocks, int64_t ntiles,
                                                                    const long long *__restrict__ padded,
                                                                    const long long *__restrict__ where,
                                                                    const int *__restrict__ slot,
                                                                    uint32_t *__restrict__ segptr,
                                                                    uint32_t *__restrict__ segs) {
    const int64_t n = nblocks * ntiles;
    const int64_t stride = int64_t(gridDim.x) * kSetupBlock;
    for (int64_t j = int64_t(blockIdx.x) * kSetupBlock + threadIdx.x; j <= n; j += stride) {
        if (j % nblocks == 0) segptr[j / nblocks] = static_cast<uint32_t>(slot[j]); // also j == n: the total
        if (j == n) continue;
        const int64_t r = j / nblocks, c = j % nblocks;
        const long long len = padded[c * ntiles + r];
        if (len > 0) {
            segs[2 * int64_t(slot[j])] = static_cast<uint32_t>(where[c * ntiles + r]);
            segs[2 * int64_t(slot[j]) + 1] = static_cast<uint32_t>(len);
        }
    }
}

// ---------------------------------------------------------------- the two passes of the product

#ifndef LSK_BIN_NT1
#define LSK_BIN_NT1 1
#endif
#ifndef LSK_BIN_NTS
#define LSK_BIN_NTS 1
#endif
#ifndef LSK_BIN_NT2
#define LSK_BIN_NT2 1
#endif
template <typename V, bool NT, typename P>
__device__ __forceinline__ V load_stream(const P *p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
    else return *reinterpret_cast<const V *>(p);
}

// Pass 1.  Workgroup u streams entries of column block unit[4u]: runs of one workgroup sweep
// (kBinBlock lanes x kU1 loads) from unit[4u+1], unit[4u+3] apart, up to unit[4u+2].
template <typename T>
__global__ __launch_bounds__(kBinBlock) void binned_multiply_kernel(
    const uint32_t *__restrict__ unit, const T *__restrict__ val, const uint16_t *__restrict__ lc,
    const T *__restrict__ x, long long x_extent, int block_cols, T *__restrict__ prod, int x_aligned) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::V;
    using I = typename Lane<T>::I;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *xs = reinterpret_cast<T *>(lds_raw);

    const uint32_t block = unit[4 * blockIdx.x], begin = unit[4 * blockIdx.x + 1], end = unit[4 * blockIdx.x + 2];
    const uint32_t stride = unit[4 * blockIdx.x + 3]; // from one run of the workgroup to its next
    const long long col0 = static_cast<long long>(block) * block_cols;
    const int ncols = static_cast<int>(min(static_cast<long long>(block_cols), x_extent - col0));
    const T *xb = x + col0;
    if (x_aligned) { // x + col0 is 16-byte aligned: block_cols is a multiple of 8
        for (int i = threadIdx.x * E; i < ncols; i += kBinBlock * E) {
            if (i + E <= ncols) {
                *reinterpret_cast<V *>(xs + i) = *reinterpret_cast<const V *>(xb + i);
            } else {
                for (int e = 0; i + e < ncols; ++e) xs[i + e] = xb[i + e];
            }
        }
    } else {
        for (int i = threadIdx.x; i < ncols; i += kBinBlock) xs[i] = xb[i];
    }
    if (threadIdx.x < kPad) xs[block_cols + threadIdx.x] = T(0); // what a pad entry multiplies
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    constexpr uint32_t kStep = kWave * E;          // entries one load instruction of a wave covers
    constexpr uint32_t kWaveRun = kStep * kU1;      // ... the kU1 loads a wave has in flight
    // begin and end are multiples of kPad, so a lane's E entries are inside or outside together
    for (uint32_t base = begin + wave * kWaveRun; base < end; base += stride) {
        V v[kU1];
        I c[kU1];
#pragma unroll
        for (int u = 0; u < kU1; ++u) {
            const uint32_t p = base + u * kStep + lane * E;
            const uint32_t q = p < end ? p : end - E; // past the end: re-read the last entries, not stored
            v[u] = load_stream<V, LSK_BIN_NT1>(val + q);
            c[u] = load_stream<I, LSK_BIN_NT1>(lc + q);
        }
#pragma unroll
        for (int u = 0; u < kU1; ++u) {
            const uint32_t p = base + u * kStep + lane * E;
            V out;
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = v[u][e] * xs[c[u][e]];
            if (p < end) {
                if constexpr (LSK_BIN_NTS) __builtin_nontemporal_store(out, reinterpret_cast<V *>(prod + p));
                else *reinterpret_cast<V *>(prod + p) = out;
            }
        }
    }
}

// Pass 2.  Workgroup r owns rows r * tile_rows ... of y.  The tile is held in fp64 whatever T is:
// for fp32 the sums are rounded once at the end, and the LDS adds fp64 faster than fp32 (measured:
// 0.9 against 2.5 ms for the same 5e8 additions).
template <typename T>
__global__ __launch_bounds__(kBinBlock) void binned_reduce_kernel(
    const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ segs, const T *__restrict__ prod,
    const uint16_t *__restrict__ lr, T *__restrict__ y, long long rows, int tile_rows, int beta, int y_aligned) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::V;
    using I = typename Lane<T>::I;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *ys = reinterpret_cast<double *>(lds_raw);

    const long long row0 = static_cast<long long>(blockIdx.x) * tile_rows;
    const int nrows = static_cast<int>(min(static_cast<long long>(tile_rows), rows - row0));
    T *yb = y + row0;
    if (beta != 0) {
        if (y_aligned) {
            for (int i = threadIdx.x * E; i < nrows; i += kBinBlock * E) {
                if (i + E <= nrows) {
                    const V v = *reinterpret_cast<const V *>(yb + i);
#pragma unroll
                    for (int e = 0; e < E; ++e) ys[i + e] = static_cast<double>(v[e]);
                } else {
                    for (int e = 0; i + e < nrows; ++e) ys[i + e] = static_cast<double>(yb[i + e]);
                }
            }
        } else {
            for (int i = threadIdx.x; i < nrows; i += kBinBlock) ys[i] = static_cast<double>(yb[i]);
        }
    } else {
        for (int i = threadIdx.x; i < nrows; i += kBinBlock) ys[i] = 0.0;
    }
    if (threadIdx.x < kPad) ys[tile_rows + threadIdx.x] = 0.0; // where the pads' products go
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    constexpr uint32_t kStep = kWave * E;
    constexpr uint32_t kWaveRun = kStep * kU2;
    const uint32_t s_begin = segptr[blockIdx.x], s_end = segptr[blockIdx.x + 1];
    for (uint32_t s = s_begin + wave; s < s_end; s += kBinWaves) {
        const uint32_t first = __builtin_amdgcn_readfirstlane(segs[2 * size_t(s)]);
        const uint32_t len = __builtin_amdgcn_readfirstlane(segs[2 * size_t(s) + 1]);
        const uint32_t end = first + len;
        for (uint32_t base = first; base < end; base += kWaveRun) {
            V v[kU2];
            I r[kU2];
#pragma unroll
            for (int u = 0; u < kU2; ++u) {
                const uint32_t p = base + u * kStep + lane * E;
                const uint32_t q = p < end ? p : end - E;
                v[u] = load_stream<V, LSK_BIN_NT2>(prod + q);
                r[u] = load_stream<I, LSK_BIN_NT2>(lr + q);
            }
#pragma unroll
            for (int u = 0; u < kU2; ++u) {
                const uint32_t p = base + u * kStep + lane * E;
                if (p < end) {
#pragma unroll
                    for (int e = 0; e < E; ++e) unsafeAtomicAdd(&ys[r[u][e]], static_cast<double>(v[u][e]));
                }
            }
        }
    }
    __syncthreads();
    if (y_aligned) {
        for (int i = threadIdx.x * E; i < nrows; i += kBinBlock * E) {
            if (i + E <= nrows) {
                V v;
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = static_cast<T>(ys[i + e]);
                *reinterpret_cast<V *>(yb + i) = v;
            } else {
                for (int e = 0; i + e < nrows; ++e) yb[i + e] = static_cast<T>(ys[i + e]);
            }
        }
    } else {
        for (int i = threadIdx.x; i < nrows; i += kBinBlock) yb[i] = static_cast<T>(ys[i]);
    }
}

// ---------------------------------------------------------------- host side

void release(lsk_coo_binned *m) {
    if (!m) return;
    (void)hipFree(m->val);
    (void)hipFree(m->prod);
    (void)hipFree(m->lc);
    (void)hipFree(m->lr);
    (void)hipFree(m->src);
    (void)hipFree(m->unit);
    (void)hipFree(m->segptr);
    (void)hipFree(m->segs);
    delete m;
}

template <typename T>
size_t lds_bytes(int elems) { return (static_cast<size_t>(elems) + kPad) * sizeof(T); }

// Both product kernels may ask for the whole LDS of a CU (see lsk::allow_large_lds).
template <typename T>
int allow_lds(hipStream_t s) {
    static std::atomic<uint64_t> multiply_done{0}, reduce_done{0};
    const int rc = lsk::allow_large_lds(reinterpret_cast<const void *>(binned_multiply_kernel<T>), multiply_done, s);
    if (rc != LSK_OK) return rc;
    return lsk::allow_large_lds(reinterpret_cast<const void *>(binned_reduce_kernel<T>), reduce_done, s);
}

template <typename T>
int create(const char *name, int64_t nnz, const uint32_t *row32, const uint32_t *col32, const T *val,
           int64_t y_first, int64_t y_last, int64_t x_extent, int block_cols, int tile_rows, lsk_stream_t stream,
           lsk_coo_binned **out) {
    if (!out) return lsk::fail(LSK_ERR_ARG, "%s: null handle pointer", name);
    *out = nullptr;
    if (nnz < 0 || x_extent < 0 || y_first < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative size", name);
    const int64_t rows = y_last - y_first + 1;
    if (nnz == 0 || rows <= 0 || x_extent == 0)
        return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: nothing to multiply", name);
    if (!row32 || !col32 || !val) return lsk::fail(LSK_ERR_ARG, "%s: null kernel arrays", name);
    if (nnz > 0x7fffffffLL) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: more than 2^31 - 1 entries", name);
    if (x_extent > 0xfffffffeLL || y_last > 0xfffffffeLL)
        return lsk::fail(LSK_ERR_ARG, "%s: window does not fit 32-bit offsets", name);
    // x blocks are held as T, y tiles as fp64; the CU has 160 KiB; 16-bit offsets
    constexpr int kDefaultCols = sizeof(T) == 8 ? 16384 : 32768; // 128 KiB
    constexpr int kLimitCols = sizeof(T) == 8 ? 20472 : 40952;
    constexpr int kLimitRows = 20472;
    if (block_cols == 0) block_cols = kDefaultCols;
    if (tile_rows == 0) {
        // the largest tiles that fit, then as many more as make the tiles a whole number of rounds
        // over the 256 CUs (one workgroup of pass 2 per CU at a time, all about equally long)
        int64_t tiles = lsk::ceil_div(rows, kLimitRows);
        if (tiles > 256) tiles = lsk::ceil_div(tiles, 256) * 256;
        tile_rows = static_cast<int>(lsk::ceil_div(lsk::ceil_div(rows, tiles), kPad) * kPad);
    }
    if (block_cols < kPad || block_cols > kLimitCols || block_cols % kPad || tile_rows < kPad ||
        tile_rows > kLimitRows || tile_rows % kPad)
        return lsk::fail(LSK_ERR_ARG, "%s: blocks of %d columns (up to %d), tiles of %d rows (up to %d), multiples of %d",
                         name, block_cols, kLimitCols, tile_rows, kLimitRows, kPad);
    const int64_t nblocks = lsk::ceil_div(x_extent, block_cols), ntiles = lsk::ceil_div(rows, tile_rows);
    const int64_t nbins = nblocks * ntiles;
    if (nbins > kMaxBins)
        return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: %lld x %lld bins", name, (long long)nblocks, (long long)ntiles);

    hipStream_t s = lsk::as_stream(stream);
    lsk_coo_binned *m = new lsk_coo_binned;
    m->elem_bytes = sizeof(T);
    m->nnz = nnz;
    m->rows = rows;
    m->x_extent = x_extent;
    m->block_cols = block_cols;
    m->tile_rows = tile_rows;
    m->nblocks = nblocks;
    m->ntiles = ntiles;
    (void)hipGetDevice(&m->device);

    uint32_t *keys = nullptr, *keys_sorted = nullptr, *pos = nullptr, *order = nullptr, *first = nullptr;
    long long *padded = nullptr, *where = nullptr, *starts = nullptr;
    int *flag = nullptr, *slot = nullptr;
    void *temp = nullptr;
    int status = LSK_OK;
#define LSK_TRY(expr)                                                                                    \
    do {                                                                                                 \
        if (status == LSK_OK) {                                                                          \
            const hipError_t e_ = (expr);                                                                \
            if (e_ != hipSuccess) status = lsk::hip_fail(e_, #expr, __FILE__, __LINE__);                 \
        }                                                                                                \
    } while (0)
    auto alloc = [&](auto **p, size_t count) {
        LSK_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(**p)));
    };
    const size_t n = static_cast<size_t>(nnz);
    alloc(&keys, n);
    alloc(&keys_sorted, n);
    alloc(&pos, n);
    alloc(&order, n);
    alloc(&first, nbins + 2);
    alloc(&padded, nbins + 1);
    alloc(&where, nbins + 1);
    alloc(&starts, nblocks + 1);
    alloc(&flag, nbins + 1);
    alloc(&slot, nbins + 1);

    std::vector<long long> h_starts(nblocks + 1, 0);
    uint32_t h_live = 0;
    int h_segments = 0;
    if (status == LSK_OK) {
        bin_keys_kernel<<<grid_for(nnz), kSetupBlock, 0, s>>>(
            nnz, row32, col32, static_cast<uint32_t>(y_first), static_cast<uint32_t>(y_last),
            static_cast<uint32_t>(x_extent), static_cast<uint32_t>(block_cols), static_cast<uint32_t>(tile_rows),
            static_cast<uint32_t>(ntiles), static_cast<uint32_t>(nbins), keys, pos);
        LSK_TRY(hipGetLastError());
        int bits = 1;
        while ((int64_t(1) << bits) <= nbins && bits < 32) ++bits;
        size_t temp_bytes = 0, scan_bytes = 0, scan2_bytes = 0;
        // the radix sort is stable: inside a bin the entries keep the caller's order
        LSK_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_sorted, pos, order,
                                                   static_cast<int>(nnz), 0, bits, s));
        LSK_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, padded, where, static_cast<int>(nbins + 1), s));
        LSK_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan2_bytes, flag, slot, static_cast<int>(nbins + 1), s));
        temp_bytes = std::max(temp_bytes, std::max(scan_bytes, scan2_bytes));
        LSK_TRY(hipMalloc(&temp, std::max<size_t>(temp_bytes, 1)));
        size_t tb = temp_bytes;
        LSK_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb, keys, keys_sorted, pos, order, static_cast<int>(nnz), 0,
                                                   bits, s));
        if (status == LSK_OK) {
            bin_first_kernel<<<grid_for(nbins + 1), kSetupBlock, 0, s>>>(nnz, keys_sorted, nbins, first);
            LSK_TRY(hipGetLastError());
            bin_padded_kernel<<<grid_for(nbins + 1), kSetupBlock, 0, s>>>(nbins, first, padded);
            LSK_TRY(hipGetLastError());
        }
        tb = temp_bytes;
        LSK_TRY(hipcub::DeviceScan::ExclusiveSum(temp, tb, padded, where, static_cast<int>(nbins + 1), s));
        if (status == LSK_OK) {
            block_starts_kernel<<<grid_for(nblocks + 1), kSetupBlock, 0, s>>>(nblocks, ntiles, where, starts);
            LSK_TRY(hipGetLastError());
            segment_flags_kernel<<<grid_for(nbins + 1), kSetupBlock, 0, s>>>(nblocks, ntiles, padded, flag);
            LSK_TRY(hipGetLastError());
        }
        tb = temp_bytes;
        LSK_TRY(hipcub::DeviceScan::ExclusiveSum(temp, tb, flag, slot, static_cast<int>(nbins + 1), s));
        LSK_TRY(hipMemcpyAsync(h_starts.data(), starts, (nblocks + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipMemcpyAsync(&h_live, first + nbins, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipMemcpyAsync(&h_segments, slot + nbins, sizeof(int), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
    }
    if (status == LSK_OK) {
        m->live = h_live;
        m->padded = h_starts[nblocks];
        m->segments = h_segments;
        if (m->padded > 0xfffffff0LL)
            status = lsk::decline(LSK_ERR_UNSUPPORTED, "%s: %lld padded entries do not fit 32-bit positions", name,
                                  (long long)m->padded);
        else if (m->live == 0)
            status = lsk::decline(LSK_ERR_UNSUPPORTED, "%s: no entry inside the windows", name);
    }
    if (status == LSK_OK) {
        // the workgroups of pass 1: a block's entries in equal runs of at most kUnitEntries
        std::vector<uint32_t> h_unit;
        m->unit_first.assign(static_cast<size_t>(nblocks) + 1, 0);
        for (int64_t c = 0; c < nblocks; ++c) {
            m->unit_first[c] = static_cast<uint32_t>(h_unit.size() / 4);
            const long long n = h_starts[c + 1] - h_starts[c];
            if (n <= 0) continue;
            const long long parts = lsk::ceil_div(n, kUnitEntries);
            constexpr long long kGroupRun = static_cast<long long>(kWave) * Lane<T>::E * kU1 * kBinWaves;
            const long long size = lsk::ceil_div(lsk::ceil_div(n, parts), kPad) * kPad;
            for (long long b = h_starts[c]; b < h_starts[c + 1]; b += size) {
                h_unit.push_back(static_cast<uint32_t>(c));
                h_unit.push_back(static_cast<uint32_t>(b));
                h_unit.push_back(static_cast<uint32_t>(std::min<long long>(b + size, h_starts[c + 1])));
                h_unit.push_back(static_cast<uint32_t>(kGroupRun));
            }
        }
        m->units = static_cast<int64_t>(h_unit.size() / 4);
        m->unit_first[nblocks] = static_cast<uint32_t>(m->units);
        const size_t p = static_cast<size_t>(m->padded);
        LSK_TRY(hipMalloc(&m->val, p * sizeof(T)));
        LSK_TRY(hipMalloc(&m->prod, p * sizeof(T)));
        alloc(&m->lc, p);
        alloc(&m->lr, p);
        alloc(&m->src, p);
        alloc(&m->unit, h_unit.size());
        alloc(&m->segptr, static_cast<size_t>(ntiles) + 1);
        alloc(&m->segs, 2 * static_cast<size_t>(m->segments));
        LSK_TRY(hipMemcpyAsync(m->unit, h_unit.data(), h_unit.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (status == LSK_OK) {
            fill_pads_kernel<T><<<grid_for(m->padded), kSetupBlock, 0, s>>>(
                m->padded, static_cast<T *>(m->val), m->lc, m->lr, m->src, static_cast<uint16_t>(block_cols),
                static_cast<uint16_t>(tile_rows));
            LSK_TRY(hipGetLastError());
            place_entries_kernel<T><<<grid_for(m->live), kSetupBlock, 0, s>>>(
                m->live, keys_sorted, order, first, where, row32, col32, val, static_cast<uint32_t>(y_first),
                static_cast<uint32_t>(block_cols), static_cast<uint32_t>(tile_rows), static_cast<uint32_t>(ntiles),
                static_cast<T *>(m->val), m->lc, m->lr, m->src);
            LSK_TRY(hipGetLastError());
            segment_lists_kernel<<<grid_for(nbins + 1), kSetupBlock, 0, s>>>(nblocks, ntiles, padded, where, slot,
                                                                            m->segptr, m->segs);
            LSK_TRY(hipGetLastError());
        }
        if (status == LSK_OK) status = allow_lds<T>(s);
        LSK_TRY(hipStreamSynchronize(s)); // set-up time: the scratch is freed before returning
    }
#undef LSK_TRY
    (void)hipFree(keys);
    (void)hipFree(keys_sorted);
    (void)hipFree(pos);
    (void)hipFree(order);
    (void)hipFree(first);
    (void)hipFree(padded);
    (void)hipFree(where);
    (void)hipFree(starts);
    (void)hipFree(flag);
    (void)hipFree(slot);
    (void)hipFree(temp);
    if (status != LSK_OK) {
        release(m);
        return status;
    }
    *out = m;
    return LSK_OK;
}

// phase < 0: the whole product.  phase 0: pass 1 of the blocks of columns that lie wholly inside
// col_lo .. col_hi (window-relative), which reads x only there; phase 1: pass 1 of all other blocks, then
// pass 2.  Phases 0 and 1 in that order are the whole product (pass 1 writes the same products either way).
template <typename T>
int spmv(const char *name, const lsk_coo_binned *m, const T *x, T *y, int beta, lsk_stream_t stream, int phase = -1,
         int64_t col_lo = 0, int64_t col_hi = -1) {
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null handle", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T)))
        return lsk::fail(LSK_ERR_ARG, "%s: the copy holds %d-byte values", name, m->elem_bytes);
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    if (beta != 0 && beta != 1) return lsk::fail(LSK_ERR_ARG, "%s: beta must be 0 or 1", name);
    if (phase > 1) return lsk::fail(LSK_ERR_ARG, "%s: phase %d (0 or 1)", name, phase);
    hipStream_t s = lsk::as_stream(stream);
    if (const int rc = allow_lds<T>(s); rc != LSK_OK) return rc; // (a handle made on another device)
    // units [u0, u1): those of the blocks inside the interval (none when phase < 0)
    int64_t u0 = 0, u1 = 0;
    if (phase >= 0 && col_hi >= col_lo) {
        const int64_t b0 = std::min<int64_t>(lsk::ceil_div(std::max<int64_t>(col_lo, 0), m->block_cols), m->nblocks);
        // a block ends at the window's end at the latest: the last one counts as inside when col_hi reaches that end
        const int64_t b1 = col_hi >= m->x_extent - 1 ? m->nblocks : std::max<int64_t>((col_hi + 1) / m->block_cols, 0);
        if (b1 > b0) {
            u0 = m->unit_first[b0];
            u1 = m->unit_first[std::min<int64_t>(b1, m->nblocks)];
        }
    }
    auto multiply = [&](int64_t first, int64_t end) -> int {
        if (end <= first) return LSK_OK;
        binned_multiply_kernel<T><<<static_cast<unsigned>(end - first), kBinBlock, lds_bytes<T>(m->block_cols), s>>>(
            m->unit + 4 * first, static_cast<const T *>(m->val), m->lc, x, m->x_extent, m->block_cols,
            static_cast<T *>(m->prod), lsk::aligned16(x) ? 1 : 0);
        LSK_LAUNCH_CHECK();
        return LSK_OK;
    };
    if (phase == 0) return multiply(u0, u1);
    if (const int rc = multiply(0, u0); rc != LSK_OK) return rc;
    if (const int rc = multiply(u1, m->units); rc != LSK_OK) return rc;
    binned_reduce_kernel<T><<<static_cast<unsigned>(m->ntiles), kBinBlock, lds_bytes<double>(m->tile_rows), s>>>(
        m->segptr, m->segs, static_cast<const T *>(m->prod), m->lr, y, m->rows, m->tile_rows, beta,
        lsk::aligned16(y) ? 1 : 0);
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

template <typename T>
int update_values(const char *name, lsk_coo_binned *m, int64_t nnz, const T *val, lsk_stream_t stream) {
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null handle", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T)))
        return lsk::fail(LSK_ERR_ARG, "%s: the copy holds %d-byte values", name, m->elem_bytes);
    if (nnz != m->nnz) return lsk::fail(LSK_ERR_ARG, "%s: %lld entries, the copy was made from %lld", name,
                                        (long long)nnz, (long long)m->nnz);
    if (!val) return lsk::fail(LSK_ERR_ARG, "%s: null values", name);
    refill_values_kernel<T><<<grid_for(m->padded), kSetupBlock, 0, lsk::as_stream(stream)>>>(
        m->padded, m->src, val, static_cast<T *>(m->val));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_coo_binned_create_f32(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const float *val,
                              int64_t y_first, int64_t y_last, int64_t x_extent, int block_cols, int tile_rows,
                              lsk_stream_t stream, lsk_coo_binned_t **out) {
    return create<float>("lsk_coo_binned_create_f32", nnz, row32, col32, val, y_first, y_last, x_extent, block_cols,
                         tile_rows, stream, out);
}
int lsk_coo_binned_create_f64(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const double *val,
                              int64_t y_first, int64_t y_last, int64_t x_extent, int block_cols, int tile_rows,
                              lsk_stream_t stream, lsk_coo_binned_t **out) {
    return create<double>("lsk_coo_binned_create_f64", nnz, row32, col32, val, y_first, y_last, x_extent, block_cols,
                          tile_rows, stream, out);
}

int lsk_coo_binned_spmv_f32(const lsk_coo_binned_t *m, const float *x, float *y, int beta, lsk_stream_t stream) {
    return spmv<float>("lsk_coo_binned_spmv_f32", m, x, y, beta, stream);
}
int lsk_coo_binned_spmv_f64(const lsk_coo_binned_t *m, const double *x, double *y, int beta, lsk_stream_t stream) {
    return spmv<double>("lsk_coo_binned_spmv_f64", m, x, y, beta, stream);
}

int lsk_coo_binned_spmv_phase_f32(const lsk_coo_binned_t *m, const float *x, float *y, int beta, int phase,
                                  int64_t col_lo, int64_t col_hi, lsk_stream_t stream) {
    if (phase < 0) return lsk::fail(LSK_ERR_ARG, "lsk_coo_binned_spmv_phase_f32: phase %d (0 or 1)", phase);
    return spmv<float>("lsk_coo_binned_spmv_phase_f32", m, x, y, beta, stream, phase, col_lo, col_hi);
}
int lsk_coo_binned_spmv_phase_f64(const lsk_coo_binned_t *m, const double *x, double *y, int beta, int phase,
                                  int64_t col_lo, int64_t col_hi, lsk_stream_t stream) {
    if (phase < 0) return lsk::fail(LSK_ERR_ARG, "lsk_coo_binned_spmv_phase_f64: phase %d (0 or 1)", phase);
    return spmv<double>("lsk_coo_binned_spmv_phase_f64", m, x, y, beta, stream, phase, col_lo, col_hi);
}

int lsk_coo_binned_update_values_f32(lsk_coo_binned_t *m, int64_t nnz, const float *val, lsk_stream_t stream) {
    return update_values<float>("lsk_coo_binned_update_values_f32", m, nnz, val, stream);
}
int lsk_coo_binned_update_values_f64(lsk_coo_binned_t *m, int64_t nnz, const double *val, lsk_stream_t stream) {
    return update_values<double>("lsk_coo_binned_update_values_f64", m, nnz, val, stream);
}

int lsk_coo_binned_info(const lsk_coo_binned_t *m, int64_t *out8) {
    if (!m || !out8) return lsk::fail(LSK_ERR_ARG, "lsk_coo_binned_info: null pointer");
    out8[0] = m->live;
    out8[1] = m->padded;
    out8[2] = m->block_cols;
    out8[3] = m->tile_rows;
    out8[4] = m->nblocks;
    out8[5] = m->ntiles;
    out8[6] = m->units;
    out8[7] = m->segments;
    return LSK_OK;
}

int lsk_coo_binned_destroy(lsk_coo_binned_t *m) {
    release(m);
    return LSK_OK;
}

} // extern "C"
