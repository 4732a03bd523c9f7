// Experiment bench for the sliced-pattern product: candidate kernel forms on a synthetic all-interior
// 27-point matrix (every row has the full distance list; x carries a halo of H elements on both
// sides so that no row needs a mask).  Not part of liblsk.so; built and driven by tools/sliced_exp.py.
// Every form sums a row's entries in ascending entry order with separate multiply and add, so the
// forms' results are bit-identical and one of them (form 0 = the shipped kernel's inner loop) checks
// the others.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kLen = 27;
typedef double d2 __attribute__((ext_vector_type(2)));  // a 16-byte vector the nontemporal builtins take

__constant__ int c_dist[32];

__device__ __forceinline__ double hash_unit(unsigned long long k) {
    k += 0x9e3779b97f4a7c15ull;
    k = (k ^ (k >> 30)) * 0xbf58476d1ce4e5b9ull;
    k = (k ^ (k >> 27)) * 0x94d049bb133111ebull;
    k ^= k >> 31;
    return double(k >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

// layout 0: slices of 64 rows, entry-major (value j of lane l at slice * 27 * 64 + j * 64 + l)
// layout 1: slices of 128 rows, lane l holds rows 2l and 2l + 1 side by side
//           (value j of row 2l + h at slice * 27 * 128 + j * 128 + 2 l + h)
__global__ void init_val_kernel(int layout, int64_t rows, double *val) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * kLen) return;
    int64_t row, j;
    if (layout == 0) {
        const int64_t slice = i / (kLen * 64), in = i % (kLen * 64);
        j = in / 64;
        row = slice * 64 + in % 64;
    } else {
        const int64_t slice = i / (kLen * 128), in = i % (kLen * 128);
        j = in / 128;
        row = slice * 128 + in % 128;
    }
    val[i] = hash_unit(static_cast<unsigned long long>(row) * 32 + j);
}

__global__ void init_x_kernel(int64_t n, double *x) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) x[i] = hash_unit(0x5555555555ull + static_cast<unsigned long long>(i));
}

// ---- form 0: one wave per slice of 64 rows, B loads of each kind in flight -----------------------
// FLAT: the row length is a compile-time constant, so the compiler may (and does) put all 27 + 27
// loads in flight at once; otherwise it is a kernel argument, as in the shipped kernel, and the loop
// runs batch by batch.
template <int B, bool NOX, bool NT_STORE, int S, bool FLAT = false, bool NO_STORE = false>
__global__ __launch_bounds__(kBlock) void form0_kernel(int64_t slices, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        int len_arg) {
    const int len = FLAT ? kLen : len_arg;
    const int lane = threadIdx.x % kWave;
    const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const int64_t slice = wave * S + s;
        if (slice >= slices) return;
        const int64_t r = slice * kWave + lane;
        const double *vp = val + slice * (kLen * kWave) + lane;
        const double *xp = x + r;
        double acc = 0.0;
        for (int j0 = 0; j0 < len; j0 += B) {
            double v[B], xv[B];
#pragma unroll
            for (int i = 0; i < B; ++i)
                if (j0 + i < len) {
                    v[i] = __builtin_nontemporal_load(vp + (j0 + i) * kWave);
                    xv[i] = NOX ? 1.0 : xp[c_dist[j0 + i]];
                }
#pragma unroll
            for (int i = 0; i < B; ++i)
                if (j0 + i < len) acc = acc + v[i] * xv[i];
        }
        if (NO_STORE) {
            if (acc == 1.2345e300) y[r] = acc; // never true: keeps the arithmetic alive
        } else if (NT_STORE) __builtin_nontemporal_store(acc, y + r);
        else y[r] = acc;
    }
}

// ---- form 1: slices of 128 rows, a lane owns rows 2l, 2l + 1; 16-byte value and x loads ----------
template <int B, bool NT_STORE, bool NOX = false>
__global__ __launch_bounds__(kBlock) void form1_kernel(int64_t slices128, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x % kWave;
    const int64_t slice = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
    if (slice >= slices128) return;
    const int64_t r = slice * 128 + 2 * lane;
    const double *vp = val + slice * (kLen * 128) + 2 * lane;
    const double *xp = x + r;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < kLen; j0 += B) {
        d2 v[B];
        double xa[B], xb[B];
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (j0 + i < kLen) {
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vp + (j0 + i) * 128));
                // two neighbouring x values; 16-byte aligned only for even distances
                if (NOX) {
                    xa[i] = 1.0;
                    xb[i] = 1.0;
                } else {
                    const double *p = xp + c_dist[j0 + i];
                    d2 t;
                    __builtin_memcpy(&t, p, 16);
                    xa[i] = t.x;
                    xb[i] = t.y;
                }
            }
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (j0 + i < kLen) {
                acc0 = acc0 + v[i].x * xa[i];
                acc1 = acc1 + v[i].y * xb[i];
            }
    }
    d2 out{acc0, acc1};
    if (NT_STORE) __builtin_nontemporal_store(out, reinterpret_cast<d2 *>(y + r));
    else *reinterpret_cast<d2 *>(y + r) = out;
}

// ---- form 2: x staged through LDS.  A workgroup takes 256 consecutive rows (4 slices); the distance
// list falls into runs of consecutive distances (27-point: 9 runs of 3); each run's window of
// 256 + run length - 1 elements of x is loaded once, coalesced, into LDS, and the lanes read their
// x values from there. --------------------------------------------------------------------------------
constexpr int kRuns = 9, kRunLen = 3;
template <bool NT_STORE>
__global__ __launch_bounds__(kBlock) void form2_kernel(int64_t slices, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y) {
    __shared__ double xs[kRuns][kBlock + 8];
    const int lane = threadIdx.x % kWave, w = threadIdx.x / kWave;
    const int64_t row0 = int64_t(blockIdx.x) * kBlock;
    const int64_t slice = int64_t(blockIdx.x) * (kBlock / kWave) + w;
    const double *vp = val + slice * (kLen * kWave) + lane;
    // values first: they are what comes from HBM
    double v[kLen];
    if (slice < slices) {
#pragma unroll
        for (int j = 0; j < kLen; ++j) v[j] = __builtin_nontemporal_load(vp + j * kWave);
    }
#pragma unroll
    for (int q = 0; q < kRuns; ++q) {
        const double *src = x + row0 + c_dist[q * kRunLen];
        xs[q][threadIdx.x] = src[threadIdx.x];
        if (threadIdx.x < kRunLen - 1) xs[q][kBlock + threadIdx.x] = src[kBlock + threadIdx.x];
    }
    __syncthreads();
    if (slice >= slices) return;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kLen; ++j) acc = acc + v[j] * xs[j / kRunLen][threadIdx.x + j % kRunLen];
    if (NT_STORE) __builtin_nontemporal_store(acc, y + row0 + threadIdx.x);
    else y[row0 + threadIdx.x] = acc;
}

// ---- form 3: both: slices of 128 rows with 16-byte value loads (layout 1), x through LDS.  A
// workgroup takes 512 consecutive rows; each run's window of 512 + 2 elements is staged once. -------
template <bool NT_STORE, int kBlock = 256>
__global__ __launch_bounds__(kBlock) void form3_kernel(int64_t slices128, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y) {
    __shared__ double xs[kRuns][2 * kBlock + 8];
    const int lane = threadIdx.x % kWave, w = threadIdx.x / kWave;
    const int64_t row0 = int64_t(blockIdx.x) * (2 * kBlock);
    const int64_t slice = int64_t(blockIdx.x) * (kBlock / kWave) + w;
    const double *vp = val + slice * (kLen * 128) + 2 * lane;
    d2 v[kLen];
    if (slice < slices128) {
#pragma unroll
        for (int j = 0; j < kLen; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vp + j * 128));
    }
#pragma unroll
    for (int q = 0; q < kRuns; ++q) {
        const double *src = x + row0 + c_dist[q * kRunLen];
        xs[q][threadIdx.x] = src[threadIdx.x];
        xs[q][kBlock + threadIdx.x] = src[kBlock + threadIdx.x];
        if (threadIdx.x < kRunLen - 1) xs[q][2 * kBlock + threadIdx.x] = src[2 * kBlock + threadIdx.x];
    }
    __syncthreads();
    if (slice >= slices128) return;
    const int t2 = 2 * threadIdx.x;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int j = 0; j < kLen; ++j) {
        acc0 = acc0 + v[j].x * xs[j / kRunLen][t2 + j % kRunLen];
        acc1 = acc1 + v[j].y * xs[j / kRunLen][t2 + 1 + j % kRunLen];
    }
    d2 out{acc0, acc1};
    if (NT_STORE) __builtin_nontemporal_store(out, reinterpret_cast<d2 *>(y + row0 + t2));
    else *reinterpret_cast<d2 *>(y + row0 + t2) = out;
}

// ---- streams: how fast the value array alone can be read ------------------------------------------
template <typename E, bool NT>
__global__ __launch_bounds__(kBlock) void stream_kernel(int64_t n_e, const E *__restrict__ p, double *__restrict__ out) {
    // same shape as form 0: a wave reads 27 consecutive 64-element runs, then the next wave's
    const int lane = threadIdx.x % kWave;
    const int64_t wave = int64_t(blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
    const int64_t base = wave * (kLen * kWave) + lane;
    if (base + (kLen - 1) * kWave >= n_e) return;
    double acc = 0.0;
    E v[kLen];
#pragma unroll
    for (int j = 0; j < kLen; ++j) v[j] = NT ? __builtin_nontemporal_load(p + base + j * kWave) : p[base + j * kWave];
#pragma unroll
    for (int j = 0; j < kLen; ++j) {
        if constexpr (sizeof(E) == 8) acc += v[j];
        else acc += v[j].x + v[j].y;
    }
    if (acc == 1.2345e300) out[wave] = acc; // never true: keeps the loads alive
}

} // namespace

extern "C" {

int exp_set_dist(const int *dist, int n) {
    return hipMemcpyToSymbol(HIP_SYMBOL(c_dist), dist, n * sizeof(int)) == hipSuccess ? 0 : -1;
}

int exp_init(int layout, int64_t rows, double *val, int64_t nx, double *x, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = rows * kLen;
    init_val_kernel<<<unsigned((n + 255) / 256), 256, 0, s>>>(layout, rows, val);
    init_x_kernel<<<unsigned((nx + 255) / 256), 256, 0, s>>>(nx, x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// form: 0 shipped loop (B = 9), 1 without the x loads, 2..4 = 2 / 4 / 8 slices per wave, 5 = B 14,
// 6 = nontemporal y store, 14 = row pairs without x, 21 = LDS + nontemporal store, 22 / 23 = row pairs
// + LDS (plain / nontemporal store), 40 = 8 + nontemporal store, 41 = 8 without the store, 42 = 7 +
// nontemporal store, 43 = 7 without the store, 7 = all 27 + 27 loads in flight (compile-time row length), 8 = 7 without x, 10 = row pairs with 16-byte loads (layout 1), 11 = the same with a
// nontemporal y store, 12 = row pairs B = 14, 20 = x through LDS, 30..33 = streams of the value array
// (8 B plain, 8 B nontemporal, 16 B plain, 16 B nontemporal)
int exp_launch(int form, int64_t rows, const double *val, const double *x_shifted, double *y, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t slices = rows / 64;
    const unsigned blocks = unsigned((slices + 3) / 4);
    switch (form) {
    case 0: form0_kernel<9, false, false, 1><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 1: form0_kernel<9, true, false, 1><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 2: form0_kernel<9, false, false, 2><<<(blocks + 1) / 2, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 3: form0_kernel<9, false, false, 4><<<(blocks + 3) / 4, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 4: form0_kernel<9, false, false, 8><<<(blocks + 7) / 8, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 5: form0_kernel<14, false, false, 1><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 6: form0_kernel<9, false, true, 1><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 7: form0_kernel<27, false, false, 1, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 8: form0_kernel<27, true, false, 1, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 10: form1_kernel<9, false><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 11: form1_kernel<9, true><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 12: form1_kernel<14, false><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 13: form1_kernel<5, false><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 14: form1_kernel<9, true, true><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 20: form2_kernel<false><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y); break;
    case 21: form2_kernel<true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y); break;
    case 22: form3_kernel<false><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 23: form3_kernel<true><<<unsigned((slices / 2 + 3) / 4), kBlock, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 24: form3_kernel<true, 512><<<unsigned((slices / 2 + 7) / 8), 512, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 25: form3_kernel<true, 128><<<unsigned((slices / 2 + 1) / 2), 128, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 26: form3_kernel<true, 1024><<<unsigned((slices / 2 + 15) / 16), 1024, 0, s>>>(slices / 2, val, x_shifted, y); break;
    case 40: form0_kernel<27, true, true, 1, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 41: form0_kernel<27, true, false, 1, true, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 42: form0_kernel<27, false, true, 1, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 43: form0_kernel<27, false, false, 1, true, true><<<blocks, kBlock, 0, s>>>(slices, val, x_shifted, y, kLen); break;
    case 30: stream_kernel<double, false><<<blocks, kBlock, 0, s>>>(rows * kLen, val, y); break;
    case 31: stream_kernel<double, true><<<blocks, kBlock, 0, s>>>(rows * kLen, val, y); break;
    case 32: stream_kernel<d2, false><<<(blocks + 1) / 2, kBlock, 0, s>>>(rows * kLen / 2, reinterpret_cast<const d2 *>(val), y); break;
    case 33: stream_kernel<d2, true><<<(blocks + 1) / 2, kBlock, 0, s>>>(rows * kLen / 2, reinterpret_cast<const d2 *>(val), y); break;
    default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // extern "C"
