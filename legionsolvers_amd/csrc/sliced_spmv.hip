// CSR product on the "sliced pattern" structure: a set-up-time re-layout for matrices whose rows
// repeat a few column patterns (every linearised stencil: col = row + one of a few distance lists).
//
// At set-up each row gets the id of its pattern (the list of col - row over its entries, in entry
// order), the patterns go into a small table, and the entry values are copied into slices of 64
// consecutive rows stored entry-major (value j of lane l of a slice at base + 64 j + l, short rows
// padded with zeros).  The product then reads NO column array and no row offsets: one wave takes
// one slice, lane l owns row l, and for j = 0, 1, ... every lane loads its j-th value (one
// contiguous 64-element run per wave) and x[row + distance_j] (for a slice whose rows share one
// pattern also one contiguous run, the distance being wave-uniform).  Each row is summed by one
// lane in ascending entry order with separate multiply and add: bit for bit the loop of the
// reference's CPU task (src/CSRMatrixTasks.cpp:73-91 visits the entries in ascending k and adds
// val * x into the row's sum).  What the reference does on the GPU instead is hand the arrays to
// cuSPARSE on every call (src/CSRMatrixTasks.cu:117-150).
//
// A slice whose rows differ (the ends of a grid line) still runs on one distance list when every
// row's list is an ordered part of the slice's longest one: the row's 32-bit mask says which
// entries it has, the others hold a zero that is never used.
//
// Bytes per launch: padded entries * sizeof(T) + rows * 2 sizeof(T) (x once, y once) + 16 per
// slice + 4 per row of a masked slice (+ 1 per row of a slice that fits no common list).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <new>

#include <hipcub/hipcub.hpp>

#include "lsk_common.hpp"

struct lsk_csr_sliced {
    int elem_bytes = 0;
    int64_t rows = 0, nnz = 0, k_base = 0, row_first = 0, x_lo = 0, x_hi = -1;
    int64_t slices = 0, padded_entries = 0, mixed_slices = 0, masked_slices = 0;
    int patterns = 0;
    void *val = nullptr;           // padded_entries values, slice by slice, entry-major inside a slice
    void *meta = nullptr;          // per slice {int64 base; int32 maxlen; int32 pattern (| kSliceMasked) or -1}
    unsigned char *row_pattern = nullptr; // per row
    unsigned *row_mask = nullptr;  // per row: which entries of its slice's list the row has (masked slices)
    int *table = nullptr;          // kMaxPatterns lengths, then kMaxPatterns x kMaxLength distances
};

namespace {

using lsk::kBlock;
using lsk::kWave;
constexpr int kMaxPatterns = 64;
constexpr int kMaxLength = 32;
constexpr int kTableInts = kMaxPatterns + kMaxPatterns * kMaxLength;

constexpr int kSliceMasked = 0x100;
struct SliceMeta {
    long long base;
    int maxlen;
    // pattern id: every row of the slice has exactly that distance list; id | kSliceMasked: every row
    // has an ordered part of that list (its row_mask says which); -1: neither
    int pattern;
};
static_assert(sizeof(SliceMeta) == 16, "slice record is 16 bytes");

// Flags the set-up kernels raise (any non-zero: the matrix does not get this structure).
enum { kFlagTooManyPatterns = 0, kFlagRowTooLong = 1, kFlagMismatch = 2, kFlagOutsideWindow = 3, kFlagBadRect = 4, kFlags = 8 };

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 32);
}

__global__ void sliced_init_kernel(unsigned long long *keys, int *flags, long long *mixed, int *table) {
    for (int i = threadIdx.x; i < kTableInts; i += blockDim.x) table[i] = 0;
    if (threadIdx.x < kMaxPatterns) keys[threadIdx.x] = 0;
    if (threadIdx.x < kFlags) flags[threadIdx.x] = 0;
    if (threadIdx.x < 2) mixed[threadIdx.x] = 0;
}

// Pass 1: hash of every row's (length, distances) list; first come, first served into the pattern
// table (keys[]: 0 = free), the inserting row recorded as the pattern's representative.  A hash
// only proposes: pass 3 compares every entry with the table.
__global__ __launch_bounds__(kBlock) void sliced_assign_kernel(
    int64_t rows, int64_t nnz, const long long *__restrict__ rects, long long k_base,
    const long long *__restrict__ col, long long row_first, unsigned long long *keys, long long *rep,
    unsigned char *__restrict__ row_pattern, int *flags) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= rows) return;
    const long long lo = rects[2 * r] - k_base, hi = rects[2 * r + 1] - k_base;
    const long long len = hi >= lo ? hi - lo + 1 : 0; // hi < lo: an empty rect, whatever its bounds
    if (len > 0 && (lo < 0 || hi >= nnz)) {
        flags[kFlagBadRect] = 1;
        return;
    }
    if (len > kMaxLength) {
        flags[kFlagRowTooLong] = 1;
        return;
    }
    unsigned long long h = mix64(0x2545f4914f6cdd1dull, static_cast<unsigned long long>(len));
    for (long long k = lo; k < lo + len; ++k) h = mix64(h, static_cast<unsigned long long>(col[k] - (row_first + r)));
    if (h == 0) h = 1;
    for (int s = 0; s < kMaxPatterns; ++s) {
        unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            cur = atomicCAS(&keys[s], 0ull, h);
            if (cur == 0) {
                rep[s] = r;
                cur = h;
            }
        }
        if (cur == h) {
            row_pattern[r] = static_cast<unsigned char>(s);
            return;
        }
    }
    flags[kFlagTooManyPatterns] = 1;
}

// Pass 2: the table, from the representatives.  table[s] = length, table[kMaxPatterns + 32 s + j] =
// distance of entry j.
__global__ void sliced_table_kernel(const long long *__restrict__ rects, long long k_base,
                                    const long long *__restrict__ col, long long row_first,
                                    const unsigned long long *__restrict__ keys, const long long *__restrict__ rep,
                                    int *__restrict__ table, int *flags) {
    const int s = threadIdx.x;
    if (s >= kMaxPatterns) return;
    if (keys[s] == 0) {
        table[s] = -1;
        return;
    }
    const long long r = rep[s];
    const long long lo = rects[2 * r] - k_base, hi = rects[2 * r + 1] - k_base;
    table[s] = hi >= lo ? static_cast<int>(hi - lo + 1) : 0;
    for (long long k = lo; k <= hi; ++k) {
        const long long d = col[k] - (row_first + r);
        if (d > INT_MAX || d < INT_MIN) flags[kFlagOutsideWindow] = 1;
        table[kMaxPatterns + kMaxLength * s + static_cast<int>(k - lo)] = static_cast<int>(d);
    }
}

// Pass 3: every entry against the table (so that a hash collision can never produce a wrong
// column) and against the window of x; per slice, the longest row and whether one pattern covers it.
__global__ __launch_bounds__(kBlock) void sliced_verify_kernel(
    int64_t rows, const long long *__restrict__ rects, long long k_base, const long long *__restrict__ col,
    long long row_first, long long x_lo, long long x_hi, const unsigned char *__restrict__ row_pattern,
    const int *__restrict__ table, unsigned *__restrict__ row_mask, long long *__restrict__ slice_entries,
    SliceMeta *__restrict__ meta, int *flags) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool live = r < rows;
    int len = 0, pat = -1;
    if (live) {
        pat = row_pattern[r];
        const long long lo = rects[2 * r] - k_base;
        len = max(static_cast<int>(rects[2 * r + 1] - k_base - lo + 1), 0);
        bool ok = table[pat] == len;
        for (int j = 0; ok && j < len; ++j) {
            const long long c = col[lo + j];
            ok = c == row_first + r + table[kMaxPatterns + kMaxLength * pat + j];
            if (c < x_lo || c > x_hi) flags[kFlagOutsideWindow] = 1;
        }
        if (!ok) flags[kFlagMismatch] = 1;
    }
    int maxlen = len;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, kWave));
    // the first live lane's pattern, compared by all live lanes
    const int first = __builtin_amdgcn_readfirstlane(pat); // lane 0 of a slice is live whenever the slice exists
    const bool uniform = __all(!live || pat == first);
    int slice_pattern = uniform ? first : -1;
    unsigned mask = live ? 0xffffffffu : 0u;
    if (!uniform) {
        // The slice's longest list (lowest lane among equals) as the common list: does every row's own
        // list occur in it, in order?  Then the row's mask marks its entries' positions there.
        const unsigned long long holders = __ballot(live && len == maxlen);
        const int common = __shfl(pat, __ffsll(static_cast<long long>(holders)) - 1, kWave);
        bool fits = true;
        mask = 0u;
        if (live) {
            const int *mine = table + kMaxPatterns + kMaxLength * pat;
            const int *list = table + kMaxPatterns + kMaxLength * common;
            int pos = 0;
            for (int j = 0; fits && j < len; ++j) {
                while (pos < maxlen && list[pos] != mine[j]) ++pos;
                fits = pos < maxlen;
                if (fits) mask |= 1u << pos++;
            }
        }
        if (__all(fits)) slice_pattern = common | kSliceMasked;
    }
    if (live) row_mask[r] = mask;
    if (threadIdx.x % kWave == 0 && live) {
        const int64_t slice = r / kWave;
        slice_entries[slice] = static_cast<long long>(maxlen) * kWave;
        meta[slice].maxlen = maxlen;
        meta[slice].pattern = slice_pattern;
    }
}

__global__ __launch_bounds__(kBlock) void sliced_bases_kernel(int64_t slices, const long long *__restrict__ entries,
                                                              const long long *__restrict__ inclusive,
                                                              SliceMeta *__restrict__ meta, long long *mixed) {
    const int64_t s = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (s >= slices) return;
    meta[s].base = inclusive[s] - entries[s];
    if (meta[s].pattern < 0) atomicAdd(reinterpret_cast<unsigned long long *>(mixed), 1ull);
    else if (meta[s].pattern & kSliceMasked) atomicAdd(reinterpret_cast<unsigned long long *>(mixed) + 1, 1ull);
}

// Pass 4: the values into their slices -- entry j of a row at position j, or, in a masked slice, at
// the position of its distance in the slice's list -- zeros everywhere else.
template <typename T>
__global__ __launch_bounds__(kBlock) void sliced_fill_kernel(int64_t rows, const long long *__restrict__ rects,
                                                             long long k_base, const T *__restrict__ val,
                                                             const SliceMeta *__restrict__ meta,
                                                             const unsigned *__restrict__ row_mask, T *__restrict__ out) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const int64_t slice = r / kWave;
    if (slice * kWave >= rows) return;
    const SliceMeta m = meta[slice];
    long long lo = 0;
    int len = 0;
    unsigned mask = 0u;
    if (r < rows) {
        lo = rects[2 * r] - k_base;
        len = max(static_cast<int>(rects[2 * r + 1] - k_base - lo + 1), 0);
        mask = (m.pattern >= 0 && (m.pattern & kSliceMasked)) ? row_mask[r] : (len >= 32 ? 0xffffffffu : (1u << len) - 1u);
    }
    T *dst = out + m.base + (r % kWave);
    for (int j = 0; j < m.maxlen; ++j) {
        const bool has = (mask >> j) & 1u;
        dst[int64_t(j) * kWave] = has ? val[lo + __popc(mask & ((1u << j) - 1u))] : T(0);
    }
}

// Entries j0 .. j0 + B - 1 of a slice whose lanes all follow the distance list `dist` (wave-uniform,
// so the x loads of a wave are one contiguous run): B loads of values (streamed past the caches:
// nothing reads them twice) and of x in flight per lane, then the sums in entry order.  `mask` bit j
// says whether the lane's row has entry j; a padding zero is never multiplied with an x value.
template <typename T, int B>
__device__ __forceinline__ T sliced_batch(T acc, const T *vp, const T *xp, const int *dist, int j0, int maxlen,
                                          unsigned mask) {
    T v[B], xv[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
        if (j0 + i < maxlen) { // wave-uniform
            v[i] = __builtin_nontemporal_load(vp + int64_t(j0 + i) * kWave);
            xv[i] = (mask >> (j0 + i)) & 1u ? xp[dist[j0 + i]] : T(0);
        }
    }
#pragma unroll
    for (int i = 0; i < B; ++i)
        if (j0 + i < maxlen && ((mask >> (j0 + i)) & 1u)) acc = acc + v[i] * xv[i];
    return acc;
}

// The product.  One wave per slice, lane l = row l of the slice.  DOT: workgroup g (of the logical
// order, whatever its placement) also leaves partials[g] = the sum over its rows of y_new[r] * w[r],
// added up in a fixed order.
template <typename T, int B, bool DOT>
__global__ __launch_bounds__(kBlock) void sliced_spmv_kernel(
    int64_t slice_begin, int64_t slice_count, int64_t row_begin, int64_t row_end, const SliceMeta *__restrict__ meta,
    const T *__restrict__ val, const unsigned char *__restrict__ row_pattern, const unsigned *__restrict__ row_mask,
    const int *__restrict__ table, const T *__restrict__ x, long long x_shift, T *__restrict__ y, int beta, int remap,
    const T *__restrict__ w, T *__restrict__ partials) {
    __shared__ T wave_share[kBlock / kWave];
    const unsigned wg = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t local = int64_t(wg) * (kBlock / kWave) + threadIdx.x / kWave;
    if (local >= slice_count) { // (only in the last workgroup)
        if constexpr (DOT) {
            wave_share[threadIdx.x / kWave] = T(0);
            __syncthreads();
        }
        return;
    }
    const int64_t slice = slice_begin + local;
    const int lane = threadIdx.x % kWave;
    const int64_t r = slice * kWave + lane;
    const bool live = r >= row_begin && r < row_end;
    const SliceMeta m = meta[slice];
    const long long base = lsk::wave_first(m.base);
    const int maxlen = __builtin_amdgcn_readfirstlane(m.maxlen);
    const int pattern = __builtin_amdgcn_readfirstlane(m.pattern);
    const T *vp = val + base + lane;
    const T *xp = x + (r + x_shift); // x[column] = xp[distance]
    T acc = (beta && live) ? y[r] : T(0);
    if (pattern >= 0) {
        // One distance list for the whole slice: every row has all of it (kSliceMasked clear) or an
        // ordered part of it, given by the row's mask.
        const int *dist = table + kMaxPatterns + kMaxLength * (pattern & 0xff); // wave-uniform address
        const unsigned mask = !live ? 0u : (pattern & kSliceMasked) ? row_mask[r] : 0xffffffffu;
        for (int j = 0; j < maxlen; j += B) acc = sliced_batch<T, B>(acc, vp, xp, dist, j, maxlen, mask);
    } else {
        // Rows whose patterns fit no common list: every lane looks its own distances up, eight at a
        // time (the table's rows are 128-byte aligned), and adds only the entries its row has.
        const int pat = live ? row_pattern[r] : 0;
        const int len = live ? table[pat] : 0;
        const int4 *dist4 = reinterpret_cast<const int4 *>(table + kMaxPatterns + kMaxLength * pat);
        for (int j = 0; j < maxlen; j += 8) {
            const int4 da = dist4[j / 4], db = dist4[j / 4 + 1];
            const int d[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
            T v[8], xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = j + i < maxlen ? __builtin_nontemporal_load(vp + int64_t(j + i) * kWave) : T(0);
                xv[i] = j + i < len ? xp[d[i]] : T(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (j + i < len) acc = acc + v[i] * xv[i];
        }
    }
    if (live) y[r] = acc;
    if constexpr (DOT) {
        const T share = lsk::wave_sum(live ? acc * w[r] : T(0));
        if (lane == 0) wave_share[threadIdx.x / kWave] = share;
        __syncthreads();
        if (threadIdx.x == 0) {
            T total = wave_share[0];
#pragma unroll
            for (int i = 1; i < kBlock / kWave; ++i) total = total + wave_share[i];
            partials[wg] = total;
        }
    }
}

template <typename T>
int create(const char *name, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
           const T *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream, lsk_csr_sliced **out) {
    if (!out) return lsk::fail(LSK_ERR_ARG, "%s: null output", name);
    *out = nullptr;
    if (rows < 0 || nnz < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative size", name);
    if (rows == 0 || nnz == 0 || x_hi < x_lo) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: empty piece", name);
    if (!rects || !col || !val) return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    if (rows > (int64_t(1) << 31) * kBlock - kBlock) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: too many rows", name);
    hipStream_t s = lsk::as_stream(stream);
    const int64_t slices = lsk::ceil_div(rows, kWave);
    const unsigned row_blocks = static_cast<unsigned>(lsk::ceil_div(rows, kBlock));

    lsk_csr_sliced *m = new (std::nothrow) lsk_csr_sliced;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: out of host memory", name);
    m->elem_bytes = sizeof(T);
    m->rows = rows;
    m->nnz = nnz;
    m->k_base = k_base;
    m->row_first = row_first;
    m->x_lo = x_lo;
    m->x_hi = x_hi;
    m->slices = slices;
    unsigned long long *keys = nullptr;
    long long *rep = nullptr, *slice_entries = nullptr, *inclusive = nullptr, *mixed = nullptr;
    int *flags = nullptr;
    void *temp = nullptr;
    int status = LSK_OK;
    int host_flags[kFlags] = {0};
    auto release_scratch = [&] {
        (void)hipFree(keys);
        (void)hipFree(rep);
        (void)hipFree(slice_entries);
        (void)hipFree(inclusive);
        (void)hipFree(mixed);
        (void)hipFree(flags);
        (void)hipFree(temp);
    };
#define LSK_TRY(expr)                                                                               \
    do {                                                                                            \
        const hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess && status == LSK_OK) status = lsk::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&keys), kMaxPatterns * sizeof(unsigned long long)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&rep), kMaxPatterns * sizeof(long long)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&flags), kFlags * sizeof(int)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&mixed), 2 * sizeof(long long)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&slice_entries), static_cast<size_t>(slices) * sizeof(long long)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&inclusive), static_cast<size_t>(slices) * sizeof(long long)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->row_pattern), static_cast<size_t>(rows)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->row_mask), static_cast<size_t>(rows) * sizeof(unsigned)));
    LSK_TRY(hipMalloc(reinterpret_cast<void **>(&m->table), kTableInts * sizeof(int)));
    LSK_TRY(hipMalloc(&m->meta, static_cast<size_t>(slices) * sizeof(SliceMeta)));
    bool declined = false;
    if (status == LSK_OK) {
        sliced_init_kernel<<<1, kBlock, 0, s>>>(keys, flags, mixed, m->table);
        LSK_TRY(hipGetLastError());
        const long long *rc = reinterpret_cast<const long long *>(rects);
        const long long *cl = reinterpret_cast<const long long *>(col);
        sliced_assign_kernel<<<row_blocks, kBlock, 0, s>>>(rows, nnz, rc, k_base, cl, row_first, keys, rep,
                                                           m->row_pattern, flags);
        LSK_TRY(hipGetLastError());
        LSK_TRY(hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        declined = host_flags[kFlagTooManyPatterns] || host_flags[kFlagRowTooLong] || host_flags[kFlagBadRect];
    }
    if (status == LSK_OK && !declined) {
        const long long *rc = reinterpret_cast<const long long *>(rects);
        const long long *cl = reinterpret_cast<const long long *>(col);
        sliced_table_kernel<<<1, kMaxPatterns, 0, s>>>(rc, k_base, cl, row_first, keys, rep, m->table, flags);
        LSK_TRY(hipGetLastError());
        sliced_verify_kernel<<<row_blocks, kBlock, 0, s>>>(rows, rc, k_base, cl, row_first, x_lo, x_hi, m->row_pattern,
                                                           m->table, m->row_mask, slice_entries,
                                                           static_cast<SliceMeta *>(m->meta), flags);
        LSK_TRY(hipGetLastError());
        LSK_TRY(hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        declined = host_flags[kFlagMismatch] || host_flags[kFlagOutsideWindow];
    }
    if (status == LSK_OK && !declined) {
        size_t temp_bytes = 0;
        LSK_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, slice_entries, inclusive, static_cast<int>(slices), s));
        LSK_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 1));
        if (status == LSK_OK)
            LSK_TRY(hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, slice_entries, inclusive, static_cast<int>(slices), s));
        const unsigned slice_blocks = static_cast<unsigned>(lsk::ceil_div(slices, kBlock));
        sliced_bases_kernel<<<slice_blocks, kBlock, 0, s>>>(slices, slice_entries, inclusive,
                                                            static_cast<SliceMeta *>(m->meta), mixed);
        LSK_TRY(hipGetLastError());
        long long total = 0, mixed_host[2] = {0, 0};
        LSK_TRY(hipMemcpyAsync(&total, inclusive + (slices - 1), sizeof(total), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipMemcpyAsync(mixed_host, mixed, sizeof(mixed_host), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        m->padded_entries = total;
        m->mixed_slices = mixed_host[0];
        m->masked_slices = mixed_host[1];
        int host_table[kMaxPatterns];
        LSK_TRY(hipMemcpy(host_table, m->table, sizeof(host_table), hipMemcpyDeviceToHost));
        for (int i = 0; i < kMaxPatterns; ++i) m->patterns += host_table[i] >= 0 ? 1 : 0;
        // not worth it when the padding would cost more than the column array saves
        if (status == LSK_OK && static_cast<double>(total) * sizeof(T) > static_cast<double>(nnz) * (sizeof(T) + 3.0))
            declined = true;
        if (status == LSK_OK && !declined) {
            LSK_TRY(hipMalloc(&m->val, static_cast<size_t>(std::max<long long>(total, 1)) * sizeof(T)));
            if (status == LSK_OK) {
                sliced_fill_kernel<T><<<static_cast<unsigned>(lsk::ceil_div(slices * kWave, kBlock)), kBlock, 0, s>>>(
                    rows, reinterpret_cast<const long long *>(rects), k_base, val,
                    static_cast<const SliceMeta *>(m->meta), m->row_mask, static_cast<T *>(m->val));
                LSK_TRY(hipGetLastError());
                LSK_TRY(hipStreamSynchronize(s));
            }
        }
    }
#undef LSK_TRY
    release_scratch();
    if (status != LSK_OK || declined) {
        (void)lsk_csr_sliced_destroy(m);
        if (status != LSK_OK) return status;
        return lsk::decline(LSK_ERR_UNSUPPORTED,
                            "%s: rows do not repeat a few column patterns (too many patterns %d, row too long %d, "
                            "rect outside the piece %d, mismatch %d, column outside the window %d, or too much padding)",
                            name, host_flags[kFlagTooManyPatterns], host_flags[kFlagRowTooLong], host_flags[kFlagBadRect],
                            host_flags[kFlagMismatch], host_flags[kFlagOutsideWindow]);
    }
    *out = m;
    return LSK_OK;
}

// The values again, from the caller's array into the slices (pass 4 alone): for a caller that has
// rewritten the entries of the matrix in place.  The structure (rects, columns) must be the one the
// copy was made from; that is the caller's promise, only the sizes can be checked here.
template <typename T>
int update_values(const char *name, lsk_csr_sliced *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                  const T *val, lsk_stream_t stream) {
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T))) return lsk::fail(LSK_ERR_ARG, "%s: structure of another entry type", name);
    if (rows != m->rows || nnz != m->nnz || k_base != m->k_base)
        return lsk::fail(LSK_ERR_ARG, "%s: %lld rows, %lld entries from %lld; the structure was made for %lld, %lld from %lld",
                         name, (long long)rows, (long long)nnz, (long long)k_base, (long long)m->rows, (long long)m->nnz,
                         (long long)m->k_base);
    if (!rects || !val) return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    sliced_fill_kernel<T><<<static_cast<unsigned>(lsk::ceil_div(m->slices * kWave, kBlock)), kBlock, 0, lsk::as_stream(stream)>>>(
        rows, reinterpret_cast<const long long *>(rects), k_base, val, static_cast<const SliceMeta *>(m->meta), m->row_mask,
        static_cast<T *>(m->val));
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

template <typename T>
int spmv(const char *name, const lsk_csr_sliced *m, int64_t row_begin, int64_t row_end, const T *x, T *y, int beta,
         const T *w, T *partials, int64_t capacity, int64_t *count, lsk_stream_t stream) {
    const bool dot = count != nullptr;
    if (dot) *count = 0;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T))) return lsk::fail(LSK_ERR_ARG, "%s: structure of another entry type", name);
    if (row_begin < 0 || row_end > m->rows || row_begin > row_end)
        return lsk::fail(LSK_ERR_ARG, "%s: rows %lld..%lld of %lld", name, (long long)row_begin, (long long)row_end,
                         (long long)m->rows);
    if (beta != 0 && beta != 1) return lsk::fail(LSK_ERR_ARG, "%s: beta must be 0 or 1", name);
    if (row_begin == row_end) return LSK_OK;
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    const int64_t slice_begin = row_begin / kWave;
    const int64_t slice_count = lsk::ceil_div(row_end, kWave) - slice_begin;
    const unsigned blocks = static_cast<unsigned>(lsk::ceil_div(slice_count, kBlock / kWave));
    if (dot) {
        if (!w || !partials) return lsk::fail(LSK_ERR_ARG, "%s: null w or partials", name);
        if (capacity < static_cast<int64_t>(blocks))
            return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %u needed", name, (long long)capacity, blocks);
    }
#define LSK_SLICED_LAUNCH(DOT)                                                                                  \
    sliced_spmv_kernel<T, 9, DOT><<<blocks, kBlock, 0, lsk::as_stream(stream)>>>(                                \
        slice_begin, slice_count, row_begin, row_end, static_cast<const SliceMeta *>(m->meta),                   \
        static_cast<const T *>(m->val), m->row_pattern, m->row_mask, m->table, x, m->row_first - m->x_lo, y, beta, \
        lsk::xcd_remap_option(), w, partials)
    if (dot) LSK_SLICED_LAUNCH(true);
    else LSK_SLICED_LAUNCH(false);
#undef LSK_SLICED_LAUNCH
    LSK_LAUNCH_CHECK();
    if (dot) *count = blocks;
    return LSK_OK;
}

// Times the candidates of the workgroup -> XCD placement on this structure and reports the fastest.
template <typename T>
int tune(const char *name, const lsk_csr_sliced *m, const T *x, T *y, lsk_stream_t stream, int *remap_out, float *ms_out) {
    static const int candidates[] = {0, 16, 64, 256, 1024, 4096};
    constexpr int kWarm = 1, kTimed = 3;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    int &placement = lsk::xcd_remap_option();
    const int saved = placement;
    hipStream_t s = lsk::as_stream(stream);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    LSK_HIP(hipEventCreate(&t0));
    LSK_HIP(hipEventCreate(&t1));
    int best = saved, status = LSK_OK;
    float best_ms = -1.0f;
    for (int c : candidates) {
        placement = c;
        for (int i = 0; i < kWarm && status == LSK_OK; ++i)
            status = spmv<T>(name, m, 0, m->rows, x, y, 0, nullptr, nullptr, 0, nullptr, stream);
        if (status != LSK_OK) break;
        if (hipEventRecord(t0, s) != hipSuccess) { status = lsk::fail(LSK_ERR_HIP, "%s: event record", name); break; }
        for (int i = 0; i < kTimed && status == LSK_OK; ++i)
            status = spmv<T>(name, m, 0, m->rows, x, y, 0, nullptr, nullptr, 0, nullptr, stream);
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

} // namespace

extern "C" {

int lsk_csr_sliced_tune_f32(const lsk_csr_sliced_t *m, const float *x, float *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms) {
    return tune<float>("lsk_csr_sliced_tune_f32", m, x, y, stream, xcd_remap, ms);
}
int lsk_csr_sliced_tune_f64(const lsk_csr_sliced_t *m, const double *x, double *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms) {
    return tune<double>("lsk_csr_sliced_tune_f64", m, x, y, stream, xcd_remap, ms);
}

int lsk_csr_sliced_create_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const float *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out) {
    return create<float>("lsk_csr_sliced_create_f32", rows, nnz, rects, k_base, col, val, row_first, x_lo, x_hi, stream, out);
}
int lsk_csr_sliced_create_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const double *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out) {
    return create<double>("lsk_csr_sliced_create_f64", rows, nnz, rects, k_base, col, val, row_first, x_lo, x_hi, stream, out);
}

int lsk_csr_sliced_update_values_f32(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                                     const float *val, lsk_stream_t stream) {
    return update_values<float>("lsk_csr_sliced_update_values_f32", m, rows, nnz, rects, k_base, val, stream);
}
int lsk_csr_sliced_update_values_f64(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                                     const double *val, lsk_stream_t stream) {
    return update_values<double>("lsk_csr_sliced_update_values_f64", m, rows, nnz, rects, k_base, val, stream);
}

int lsk_csr_sliced_spmv_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                            int beta, lsk_stream_t stream) {
    return spmv<float>("lsk_csr_sliced_spmv_f32", m, row_begin, row_end, x, y, beta, nullptr, nullptr, 0, nullptr, stream);
}
int lsk_csr_sliced_spmv_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                            int beta, lsk_stream_t stream) {
    return spmv<double>("lsk_csr_sliced_spmv_f64", m, row_begin, row_end, x, y, beta, nullptr, nullptr, 0, nullptr, stream);
}
int lsk_csr_sliced_spmv_dot_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                                int beta, const float *w, float *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream) {
    if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_spmv_dot_f32: null count");
    return spmv<float>("lsk_csr_sliced_spmv_dot_f32", m, row_begin, row_end, x, y, beta, w, partials, capacity, count, stream);
}
int lsk_csr_sliced_spmv_dot_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                                int beta, const double *w, double *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream) {
    if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_spmv_dot_f64: null count");
    return spmv<double>("lsk_csr_sliced_spmv_dot_f64", m, row_begin, row_end, x, y, beta, w, partials, capacity, count, stream);
}

int lsk_csr_sliced_info(const lsk_csr_sliced_t *m, int64_t *out) {
    if (!m || !out) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_info: null pointer");
    out[0] = m->padded_entries;
    out[1] = m->slices;
    out[2] = m->mixed_slices;
    out[3] = m->patterns;
    out[4] = m->masked_slices;
    return LSK_OK;
}

int lsk_csr_sliced_destroy(lsk_csr_sliced_t *m) {
    if (!m) return LSK_OK;
    (void)hipFree(m->val);
    (void)hipFree(m->meta);
    (void)hipFree(m->row_pattern);
    (void)hipFree(m->row_mask);
    (void)hipFree(m->table);
    delete m;
    return LSK_OK;
}

} // extern "C"
