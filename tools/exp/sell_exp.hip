// Experiment: a CSR product on a copy in slices of 64 rows with the rows' own 32-bit columns (SELL-64): lane =
// row, 16-byte loads of 4 columns and of 2 values per lane, every row summed by its lane in entry order.
// Slice s holds width4[s] "quads" (4 entries per row); quad q of lane l:
//   cols at col[((base[s] + q) * 64 + l) * 4 .. + 4)
//   vals at val[(((base[s] + q) * 2 + h) * 64 + l) * 2 .. + 2), h = 0, 1
// padding: column 0xffffffff (skipped), value 0.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long wave_first(long long v) {
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
    const int hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}

__global__ __launch_bounds__(256) void sell_fill_kernel(int64_t rows, int64_t padded_rows, const uint32_t *__restrict__ rowoff,
                                                        const uint32_t *__restrict__ col32, const double *__restrict__ val,
                                                        const long long *__restrict__ base, const int *__restrict__ width4,
                                                        uint32_t *__restrict__ col_out, double *__restrict__ val_out) {
    const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= padded_rows) return;
    const int64_t s = r / 64;
    const int lane = static_cast<int>(r % 64);
    uint32_t lo = 0, len = 0;
    if (r < rows) {
        lo = rowoff[r];
        len = rowoff[r + 1] - lo;
    }
    const long long b = base[s];
    const int w = width4[s];
    for (int j = 0; j < 4 * w; ++j) {
        const bool has = static_cast<uint32_t>(j) < len;
        const int q = j / 4, e = j % 4;
        col_out[((b + q) * 64 + lane) * 4 + e] = has ? col32[lo + j] : 0xffffffffu;
        val_out[(((b + q) * 2 + e / 2) * 64 + lane) * 2 + e % 2] = has ? val[lo + j] : 0.0;
    }
}

template <int CAPQ, int WAVES, bool NT>
__global__ __launch_bounds__(256, WAVES) void sell_spmv_kernel(int64_t rows, const long long *__restrict__ base,
                                                                const int *__restrict__ width4,
                                                                const u4_t *__restrict__ col, const d2_t *__restrict__ val,
                                                                const double *__restrict__ x, uint32_t x_last,
                                                                double *__restrict__ y, int beta) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
    const int64_t s = int64_t(blockIdx.x) * 4 + wave;
    if (s * 64 >= rows) return;
    const int64_t r = s * 64 + lane;
    const long long b = wave_first(base[s]);
    const int wq = __builtin_amdgcn_readfirstlane(width4[s]);
    double acc = (beta && r < rows) ? y[r] : 0.0;
    for (int q0 = 0; q0 < wq; q0 += CAPQ) {
        u4_t c[CAPQ];
        d2_t v[CAPQ][2];
#pragma unroll
        for (int q = 0; q < CAPQ; ++q) {
            if (q0 + q < wq) {
                const long long u = b + q0 + q;
                if constexpr (NT) {
                    c[q] = __builtin_nontemporal_load(col + u * 64 + lane);
                    v[q][0] = __builtin_nontemporal_load(val + (u * 2) * 64 + lane);
                    v[q][1] = __builtin_nontemporal_load(val + (u * 2 + 1) * 64 + lane);
                } else {
                    c[q] = col[u * 64 + lane];
                    v[q][0] = val[(u * 2) * 64 + lane];
                    v[q][1] = val[(u * 2 + 1) * 64 + lane];
                }
            } else {
                c[q] = u4_t{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                v[q][0] = d2_t{0.0, 0.0};
                v[q][1] = d2_t{0.0, 0.0};
            }
        }
        double xv[CAPQ][4];
#pragma unroll
        for (int q = 0; q < CAPQ; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t cc = c[q][e];
                xv[q][e] = x[cc <= x_last ? cc : 0u];
            }
        }
#pragma unroll
        for (int q = 0; q < CAPQ; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double sum = acc + v[q][e / 2][e % 2] * xv[q][e];
                acc = c[q][e] <= x_last ? sum : acc;
            }
        }
    }
    if (r < rows) y[r] = acc;
}

} // namespace

extern "C" {

int sell_exp_fill(int64_t rows, int64_t padded_rows, const uint32_t *rowoff, const uint32_t *col32, const double *val,
                  const long long *base, const int *width4, uint32_t *col_out, double *val_out, hipStream_t s) {
    sell_fill_kernel<<<dim3(static_cast<unsigned>((padded_rows + 255) / 256)), dim3(256), 0, s>>>(
        rows, padded_rows, rowoff, col32, val, base, width4, col_out, val_out);
    return static_cast<int>(hipGetLastError());
}

int sell_exp_spmv(int form, int64_t rows, const long long *base, const int *width4, const void *col, const void *val,
                  const double *x, uint32_t x_last, double *y, int beta, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>((rows + 255) / 256)), block(256);
    const u4_t *c = static_cast<const u4_t *>(col);
    const d2_t *v = static_cast<const d2_t *>(val);
#define LAUNCH(CAPQ, WAVES, NT) sell_spmv_kernel<CAPQ, WAVES, NT><<<grid, block, 0, s>>>(rows, base, width4, c, v, x, x_last, y, beta)
    switch (form) {
    case 0: LAUNCH(7, 3, true); break;
    case 1: LAUNCH(7, 2, true); break;
    case 2: LAUNCH(4, 4, true); break;
    case 3: LAUNCH(2, 6, true); break;
    case 4: LAUNCH(7, 3, false); break;
    case 5: LAUNCH(8, 3, true); break;
    case 6: LAUNCH(1, 8, true); break;
    default: return -1;
    }
#undef LAUNCH
    return static_cast<int>(hipGetLastError());
}
}
