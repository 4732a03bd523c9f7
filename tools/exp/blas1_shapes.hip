// Experiment bench: launch shapes of the BLAS-1 streams (all with nontemporal loads and stores, the
// library's choice for vectors of 256 MiB and more): accesses in flight per lane (U), workgroups in the
// grid, grid-stride sweeps against one contiguous slab per workgroup, workgroup size.  Not part of
// liblsk.so; built and driven by tools/blas1_shapes.py.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 ld(const d2 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st(d2 *p, d2 v) { __builtin_nontemporal_store(v, p); }

// op 0: y = a x + y; op 2: z = a p + z, p = b p + r (x = r, y = p); op 3: copy y = x;
// op 5: y = a x + y and the squared norm of the new y (per-workgroup partial)
template <int OP, int U, int BLOCK, bool SLAB>
__global__ __launch_bounds__(BLOCK) void shape_kernel(int64_t ne, double a, double b, const d2 *x, d2 *y, d2 *z,
                                                      double *out) {
    // SLAB: workgroup g owns elements [g * per, (g + 1) * per); else chunks of U * BLOCK dealt round robin
    const int64_t chunk = int64_t(U) * BLOCK;
    const int64_t nchunks = ne / chunk;
    int64_t c, c_end, c_step;
    if (SLAB) {
        const int64_t per = (nchunks + gridDim.x - 1) / gridDim.x;
        c = int64_t(blockIdx.x) * per;
        c_end = c + per < nchunks ? c + per : nchunks;
        c_step = 1;
    } else {
        c = blockIdx.x;
        c_end = nchunks;
        c_step = gridDim.x;
    }
    double acc = 0.0;
    for (; c < c_end; c += c_step) {
        const int64_t i = c * chunk + threadIdx.x;
        d2 xv[U], yv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = ld(x + i + u * BLOCK);
            if (OP != 3) yv[u] = ld(y + i + u * BLOCK);
            if (OP == 2) zv[u] = ld(z + i + u * BLOCK);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (OP == 0 || OP == 5) {
                d2 r{__builtin_fma(a, xv[u].x, yv[u].x), __builtin_fma(a, xv[u].y, yv[u].y)};
                st(y + i + u * BLOCK, r);
                if (OP == 5) acc = __builtin_fma(r.y, r.y, __builtin_fma(r.x, r.x, acc));
            } else if (OP == 2) {
                d2 zo{__builtin_fma(a, yv[u].x, zv[u].x), __builtin_fma(a, yv[u].y, zv[u].y)};
                d2 po{__builtin_fma(b, yv[u].x, xv[u].x), __builtin_fma(b, yv[u].y, xv[u].y)};
                st(z + i + u * BLOCK, zo);
                st(y + i + u * BLOCK, po);
            } else {
                st(y + i + u * BLOCK, xv[u]);
            }
        }
    }
    if (OP == 5) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (threadIdx.x % 64 == 0) atomicAdd(out + (blockIdx.x % 4096), acc); // (timing only)
    }
}

template <int OP, int U, int BLOCK>
int go(bool slab, int64_t blocks, int64_t ne, const d2 *x, d2 *y, d2 *z, double *out, hipStream_t s) {
    if (slab) shape_kernel<OP, U, BLOCK, true><<<unsigned(blocks), BLOCK, 0, s>>>(ne, 0.5, 0.25, x, y, z, out);
    else shape_kernel<OP, U, BLOCK, false><<<unsigned(blocks), BLOCK, 0, s>>>(ne, 0.5, 0.25, x, y, z, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int OP>
int pick(int u, int block, bool slab, int64_t blocks, int64_t ne, const d2 *x, d2 *y, d2 *z, double *out, hipStream_t s) {
#define CASE(UU, BB) if (u == UU && block == BB) return go<OP, UU, BB>(slab, blocks, ne, x, y, z, out, s)
    CASE(2, 256); CASE(4, 256); CASE(8, 256); CASE(4, 512); CASE(2, 512); CASE(4, 1024); CASE(2, 1024); CASE(1, 1024);
    CASE(4, 128); CASE(8, 128); CASE(4, 64); CASE(8, 64); CASE(16, 64);
#undef CASE
    return -3;
}

} // namespace

extern "C" int blas1_shapes_launch(int op, int u, int block, int slab, int64_t blocks, int64_t n, const double *x,
                                   double *y, double *z, double *out /* 4096 doubles */, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t ne = n / 2;
    if (ne % (int64_t(u) * block) != 0) return -2; // whole chunks only
    const d2 *xe = reinterpret_cast<const d2 *>(x);
    d2 *ye = reinterpret_cast<d2 *>(y), *ze = reinterpret_cast<d2 *>(z);
    switch (op) {
    case 0: return pick<0>(u, block, slab != 0, blocks, ne, xe, ye, ze, out, s);
    case 2: return pick<2>(u, block, slab != 0, blocks, ne, xe, ye, ze, out, s);
    case 3: return pick<3>(u, block, slab != 0, blocks, ne, xe, ye, ze, out, s);
    case 5: return pick<5>(u, block, slab != 0, blocks, ne, xe, ye, ze, out, s);
    default: return -3;
    }
}
