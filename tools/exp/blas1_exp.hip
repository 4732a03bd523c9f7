// Experiment bench for the BLAS-1 streams: plain against nontemporal loads and stores, in the
// shape of csrc/blas1.hip's kernels (16 bytes per lane per access, four accesses in flight per
// lane, a workgroup sweeping contiguous 16 KiB chunks, grid capped at 2048).  Not part of
// liblsk.so; built and driven by tools/blas1_exp.py.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int kBlock = 256;
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(d2 *p, d2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// op 0: y = a x + y (2 reads, 1 write); op 1: dot (2 reads); op 2: z = a p + z, p = b p + r (3 reads,
// 2 writes); op 3: copy (1 read, 1 write); op 4: dot(v, v) (1 read)
template <int OP, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void stream_kernel(int64_t ne, double a, double b, const d2 *x, d2 *y, d2 *z,
                                                        double *out) {
    const int64_t gstride = int64_t(gridDim.x) * kBlock;
    const int64_t outer = 4 * gstride;
    int64_t i = int64_t(blockIdx.x) * (4 * kBlock) + threadIdx.x;
    double acc = 0.0;
    for (; i + 3 * kBlock < ne; i += outer) {
        d2 xv[4], yv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = ld<NTL>(x + i + u * kBlock);
            if (OP != 3 && OP != 4) yv[u] = ld<NTL>(y + i + u * kBlock);
            if (OP == 2) zv[u] = ld<NTL>(z + i + u * kBlock);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) {
                d2 r{__builtin_fma(a, xv[u].x, yv[u].x), __builtin_fma(a, xv[u].y, yv[u].y)};
                st<NTS>(y + i + u * kBlock, r);
            } else if (OP == 1) {
                acc = __builtin_fma(xv[u].x, yv[u].x, acc);
                acc = __builtin_fma(xv[u].y, yv[u].y, acc);
            } else if (OP == 2) { // x = r, y = p, z = z
                d2 zo{__builtin_fma(a, yv[u].x, zv[u].x), __builtin_fma(a, yv[u].y, zv[u].y)};
                d2 po{__builtin_fma(b, yv[u].x, xv[u].x), __builtin_fma(b, yv[u].y, xv[u].y)};
                st<NTS>(z + i + u * kBlock, zo);
                st<NTS>(y + i + u * kBlock, po);
            } else if (OP == 3) {
                st<NTS>(y + i + u * kBlock, xv[u]);
            } else {
                acc = __builtin_fma(xv[u].x, xv[u].x, acc);
                acc = __builtin_fma(xv[u].y, xv[u].y, acc);
            }
        }
    }
    if (OP == 1 || OP == 4) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = acc;
    }
}

} // namespace

extern "C" int blas1_exp_launch(int op, int ntl, int nts, int64_t n, const double *x, double *y, double *z,
                                double *out /* 8192 doubles */, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t ne = n / 2;
    int64_t blocks = (ne + 4 * kBlock - 1) / (4 * kBlock);
    if (blocks > 2048) blocks = 2048;
    if (n % (8 * kBlock) != 0) return -2; // whole chunks only: the bench has no tail code
    const d2 *xe = reinterpret_cast<const d2 *>(x);
    d2 *ye = reinterpret_cast<d2 *>(y), *ze = reinterpret_cast<d2 *>(z);
#define GO(OP, L, S) stream_kernel<OP, L, S><<<unsigned(blocks), kBlock, 0, s>>>(ne, 0.5, 0.25, xe, ye, ze, out)
#define PICK(OP)                                   \
    do {                                           \
        if (ntl && nts) GO(OP, true, true);        \
        else if (ntl) GO(OP, true, false);         \
        else if (nts) GO(OP, false, true);         \
        else GO(OP, false, false);                 \
    } while (0)
    switch (op) {
    case 0: PICK(0); break;
    case 1: PICK(1); break;
    case 2: PICK(2); break;
    case 3: PICK(3); break;
    case 4: PICK(4); break;
    default: return -3;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
