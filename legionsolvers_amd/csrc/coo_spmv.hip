// COO SpMV for gfx950: y[row[k]] += val[k] * x[col[k]] on the reference's layout (three
// parallel arrays: 64-bit global row ids, 64-bit global column ids, fp32/fp64 entries;
// src/COOMatrixTasks.cu:12-146), with the CPU body's domain guards
// (src/COOMatrixTasks.cpp:67-74).
//
// A workgroup takes a chunk of 2048 consecutive entries: lanes load row/col/val coalesced,
// gather x through L2 and park (row, product) in LDS.  Runs of equal rows inside the chunk,
// cut every 32 entries so that no lane walks a long run, are summed on chip by the lane that
// sees the run's head, which then issues ONE floating-point atomic add to y.  Row-sorted
// input therefore costs one atomic per (row, chunk) pair; unsorted input degrades to one
// atomic per entry and stays correct.  y accumulates (beta = 1).
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

template <typename T>
int launch_coo_spmv(const char *name, int64_t nnz, const int64_t *row, const int64_t *col,
                    const T *val, const T *x, int64_t x_lo, int64_t x_hi, T *y, int64_t y_lo,
                    int64_t y_hi, lsk_stream_t stream) {
    if (nnz < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative nnz %lld", name, (long long)nnz);
    if (nnz == 0) return LSK_OK; // "If there are no coordinates to process, break out." (.cu:72-73)
    if (!row || !col || !val) return lsk::fail(LSK_ERR_ARG, "%s: null kernel arrays", name);
    if (x_hi < x_lo || y_hi < y_lo) return LSK_OK; // an empty domain admits no entry
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    const int64_t blocks = lsk::ceil_div(nnz, kChunk);
    if (blocks > 0x7fffffffLL) return lsk::fail(LSK_ERR_UNSUPPORTED, "%s: too many entries", name);
    coo_spmv_kernel<T><<<static_cast<unsigned>(blocks), kBlock, 0, lsk::as_stream(stream)>>>(
        nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi);
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

} // extern "C"
