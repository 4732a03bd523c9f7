// COO SpMV for gfx950: y[row[k]] += val[k] * x[col[k]] on the reference's layout (three
// parallel arrays: 64-bit global row ids, 64-bit global column ids, fp32/fp64 entries;
// src/COOMatrixTasks.cu:12-146), with the CPU body's domain guards
// (src/COOMatrixTasks.cpp:67-74).
//
// Two forms, both accumulating into y (beta = 1) and both correct for any entry order:
//
// Wave scan (default).  A wave takes 64*K consecutive entries, K steps of one entry per lane.
// All row/col/val loads of the K steps are issued first, then the K gathers of x.  Per step
// the products are summed per run of equal rows by a wave-wide segmented inclusive scan in
// registers (DPP row shifts and row broadcasts, no LDS, no barrier): the run boundaries are one
// ballot, and a lane adds its neighbour's partial only if that neighbour lies at or after the
// start of its own run.  The last lane of every run issues ONE floating-point atomic add; the
// run that is still open at lane 63 is carried into lane 0 of the next step, so row-sorted
// input costs one atomic per row plus one per wave boundary.
//
// Workgroup chunks (option coo.variant = 0).  A workgroup takes 2048 consecutive entries and
// parks (row, product) in LDS; runs of equal rows, cut every 32 entries, are summed by the
// lane that sees the run's head.
//
// HBM bytes per call: nnz*(16 + sizeof(T)) + the touched x and y windows.
#include <climits>

#include "lsk_common.hpp"

namespace {

using lsk::kBlock;
constexpr int kChunk = 2048; // entries per workgroup: 8 per lane
constexpr int kRunCut = 32;  // forced run boundary

template <typename T>
__global__ __launch_bounds__(kBlock) void coo_spmv_kernel(
    int64_t nnz, const int64_t *__restrict__ row, const int64_t *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ x, int64_t x_lo, int64_t x_hi,
    T *__restrict__ y, int64_t y_lo, int64_t y_hi) {
    __shared__ T s_prod[kChunk];
    __shared__ long long s_row[kChunk];

    const int64_t e0 = int64_t(blockIdx.x) * kChunk;
    const int cnt = static_cast<int>(nnz - e0 < kChunk ? nnz - e0 : kChunk);
    const int tid = threadIdx.x;
    constexpr int PER = kChunk / kBlock;

    long long rr[PER], cc[PER];
    T vv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + j * kBlock;
        if (e < cnt) {
            rr[j] = row[e0 + e];
            cc[j] = col[e0 + e];
            vv[j] = val[e0 + e];
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + j * kBlock;
        if (e < cnt) {
            const bool live = cc[j] >= x_lo && cc[j] <= x_hi && rr[j] >= y_lo && rr[j] <= y_hi;
            s_prod[e] = live ? vv[j] * x[cc[j] - x_lo] : T(0);
            // Entries outside the output domain get a row id no valid entry can have, so
            // that they never join a neighbour's run; their product is dropped below.
            s_row[e] = (rr[j] >= y_lo && rr[j] <= y_hi) ? rr[j] : LLONG_MIN;
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + j * kBlock;
        if (e >= cnt) continue;
        const long long r = s_row[e];
        const bool head = (e % kRunCut == 0) || (s_row[e - 1] != r);
        if (!head || r == LLONG_MIN) continue;
        T acc = s_prod[e];
        for (int f = e + 1; f < cnt && (f % kRunCut) != 0 && s_row[f] == r; ++f) acc += s_prod[f];
        unsafeAtomicAdd(&y[r - y_lo], acc);
    }
}

// ---- wave-scan form ---------------------------------------------------------------------

constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// Value of `v` in the lane the DPP control selects, 0 where the control selects no lane (past
// the start of a row of 16 for a shift; rows outside ROW_MASK for a broadcast).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_get(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

__device__ __forceinline__ long long wave_read_lane(long long v, int src) {
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), src);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), src);
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}
__device__ __forceinline__ uint32_t wave_read_lane(uint32_t v, int src) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src));
}
__device__ __forceinline__ double wave_read_lane(double v, int src) {
    return __longlong_as_double(wave_read_lane(__double_as_longlong(v), src));
}
__device__ __forceinline__ float wave_read_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Inclusive sum of s over the lanes start..lane of this wave, where `start` is the first lane
// of the run the lane belongs to.  Six rounds; a round adds the partial of a lower lane only
// if that lane is inside the run, in which case the partial it holds covers exactly the part
// of the run that is still missing.
template <typename T> __device__ __forceinline__ T run_scan(T s, int lane, int start) {
    T t;
    t = dpp_get<kDppRowShr1, 0xf>(s); s += (lane - 1 >= start) ? t : T(0);
    t = dpp_get<kDppRowShr2, 0xf>(s); s += (lane - 2 >= start) ? t : T(0);
    t = dpp_get<kDppRowShr4, 0xf>(s); s += (lane - 4 >= start) ? t : T(0);
    t = dpp_get<kDppRowShr8, 0xf>(s); s += (lane - 8 >= start) ? t : T(0);
    t = dpp_get<kDppRowBcast15, 0xa>(s); s += ((lane & ~15) - 1 >= start) ? t : T(0); // lanes 16-31, 48-63
    t = dpp_get<kDppRowBcast31, 0xc>(s); s += (31 >= start) ? t : T(0);               // lanes 32-63
    return s;
}

// Index field of the scan kernel: the reference's 64-bit global ids, or 32-bit offsets into the x
// and y windows made once at set-up (0xffffffff = outside the window, dropped like an id outside
// x_lo..x_hi / y_lo..y_hi is dropped).
template <typename IT> struct CooIndex;
template <> struct CooIndex<int64_t> {
    using type = long long;
    static constexpr long long kNone = LLONG_MIN; // a row id no live entry has
};
template <> struct CooIndex<uint32_t> {
    using type = uint32_t;
    static constexpr uint32_t kNone = 0xffffffffu;
};

template <typename T, typename IT, int K>
__global__ __launch_bounds__(kBlock) void coo_spmv_scan_kernel(
    int64_t nnz, const IT *__restrict__ row, const IT *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ x, int64_t x_lo, int64_t x_hi,
    T *__restrict__ y, int64_t y_lo, int64_t y_hi) {
    using I = typename CooIndex<IT>::type;
    constexpr I kNone = CooIndex<IT>::kNone;
    const int lane = threadIdx.x & (lsk::kWave - 1);
    const int64_t wave = int64_t(blockIdx.x) * (kBlock / lsk::kWave) + (threadIdx.x / lsk::kWave);
    const int64_t base = wave * (lsk::kWave * K);
    if (base >= nnz) return; // the whole wave leaves together
    const int64_t last = nnz - 1;
    // window bounds in the index type (32-bit form: x_lo = 0 and the bounds fit by construction)
    const I xl = static_cast<I>(x_lo), xh = static_cast<I>(x_hi), yl = static_cast<I>(y_lo), yh = static_cast<I>(y_hi);

    I rr[K], cc[K];
    T vv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { // entries past the end re-read the last one and are dropped below
        const int64_t e = base + j * lsk::kWave + lane;
        const int64_t ec = e < last ? e : last;
        // streamed past the caches (nothing reads an entry twice), so that x and y keep L2 to themselves
        rr[j] = __builtin_nontemporal_load(row + ec);
        cc[j] = __builtin_nontemporal_load(col + ec);
        vv[j] = __builtin_nontemporal_load(val + ec);
    }
    T pp[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int64_t e = base + j * lsk::kWave + lane;
        const bool live = e <= last && cc[j] >= xl && cc[j] <= xh && rr[j] >= yl && rr[j] <= yh;
        const T xv = x[live ? cc[j] - xl : 0];
        pp[j] = live ? vv[j] * xv : T(0);
        // Dropped entries get a row id no live entry has: they form runs of their own, whose
        // sums are never written.
        rr[j] = live ? rr[j] : kNone;
    }

    I carry_row = kNone; // run still open at the end of the previous step
    T carry = T(0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (base + j * lsk::kWave > last) break; // wave-uniform: nothing but dropped entries left
        const I r = rr[j];
        T s = pp[j];
        const I r0 = wave_read_lane(r, 0);
        if (lane == 0) {
            if (r0 == carry_row) s += carry;
            else if (carry_row != kNone) unsafeAtomicAdd(&y[carry_row - yl], carry);
        }
        const I rp = __shfl_up(r, 1);
        const bool head = lane == 0 || rp != r;
        const unsigned long long heads = __ballot(head);
        const int start = 63 - __clzll(heads & (~0ull >> (63 - lane))); // bit 0 is always set
        s = run_scan(s, lane, start);
        const bool tail = lane < 63 && ((heads >> (lane + 1)) & 1ull);
        if (tail && r != kNone) unsafeAtomicAdd(&y[r - yl], s);
        carry_row = wave_read_lane(r, 63);
        carry = wave_read_lane(s, 63);
    }
    if (lane == 0 && carry_row != kNone) unsafeAtomicAdd(&y[carry_row - yl], carry);
}

template <typename T, typename IT = int64_t>
int launch_coo_spmv(const char *name, int64_t nnz, const IT *row, const IT *col,
                    const T *val, const T *x, int64_t x_lo, int64_t x_hi, T *y, int64_t y_lo,
                    int64_t y_hi, lsk_stream_t stream) {
    if (nnz < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative nnz %lld", name, (long long)nnz);
    if (nnz == 0) return LSK_OK; // "If there are no coordinates to process, break out." (.cu:72-73)
    if (!row || !col || !val) return lsk::fail(LSK_ERR_ARG, "%s: null kernel arrays", name);
    if (x_hi < x_lo || y_hi < y_lo) return LSK_OK; // an empty domain admits no entry
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    const int64_t blocks = lsk::ceil_div(nnz, kChunk); // both forms: 2048 entries per workgroup
    if (blocks > 0x7fffffffLL) return lsk::fail(LSK_ERR_UNSUPPORTED, "%s: too many entries", name);
    if constexpr (sizeof(IT) == 4) {
        constexpr int K = kChunk / kBlock;
        coo_spmv_scan_kernel<T, IT, K><<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(
            nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi);
    } else if (lsk::options().coo_variant == 1) {
        constexpr int K = kChunk / kBlock;
        coo_spmv_scan_kernel<T, IT, K><<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(
            nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi);
    } else {
        coo_spmv_kernel<T><<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(
            nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi);
    }
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_coo_spmv_f32(int64_t nnz, const int64_t *row, const int64_t *col, const float *val,
                     const float *x, int64_t x_lo, int64_t x_hi, float *y, int64_t y_lo,
                     int64_t y_hi, lsk_stream_t stream) {
    return launch_coo_spmv<float>("lsk_coo_spmv_f32", nnz, row, col, val, x, x_lo, x_hi, y, y_lo,
                                  y_hi, stream);
}

int lsk_coo_spmv_f64(int64_t nnz, const int64_t *row, const int64_t *col, const double *val,
                     const double *x, int64_t x_lo, int64_t x_hi, double *y, int64_t y_lo,
                     int64_t y_hi, lsk_stream_t stream) {
    return launch_coo_spmv<double>("lsk_coo_spmv_f64", nnz, row, col, val, x, x_lo, x_hi, y, y_lo,
                                   y_hi, stream);
}

#define LSK_COO_C32(SUF, T)                                                                        \
    int lsk_coo_spmv_c32_##SUF(int64_t nnz, const uint32_t *row32, const uint32_t *col32, const T *val, \
                               const T *x, int64_t x_extent, T *y, int64_t y_first, int64_t y_last,   \
                               lsk_stream_t stream) {                                                 \
        if (x_extent < 0 || x_extent > 0xffffffffLL || y_first < 0 || y_last >= 0xffffffffLL)         \
            return lsk::fail(LSK_ERR_ARG, "lsk_coo_spmv_c32_" #SUF ": window does not fit 32-bit offsets"); \
        return launch_coo_spmv<T, uint32_t>("lsk_coo_spmv_c32_" #SUF, nnz, row32, col32, val, x, 0,   \
                                            x_extent - 1, y, y_first, y_last, stream);                \
    }
LSK_COO_C32(f32, float)
LSK_COO_C32(f64, double)
#undef LSK_COO_C32

} // extern "C"
