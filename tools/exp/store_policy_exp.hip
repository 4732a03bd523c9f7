// Experiment: the rate of a pure write stream (fill of 1 GiB) and of a copy by the cache-policy bits of the 16-byte
// store (gfx950: sc0, sc1, nt), one 16 KiB chunk per workgroup as the library's element-wise kernels on large vectors.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/store_policy_exp.hip -o tools/exp/store_policy_exp && tools/exp/store_policy_exp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int POLICY> __device__ __forceinline__ void store16(f4 *p, f4 v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// one chunk of 16 KiB per workgroup: 256 lanes x 4 accesses of 16 bytes
template <int POLICY, bool COPY>
__global__ __launch_bounds__(256) void stream_kernel(f4 *__restrict__ dst, const f4 *__restrict__ src, f4 value) {
    const size_t base = size_t(blockIdx.x) * 1024 + threadIdx.x;
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = COPY ? __builtin_nontemporal_load(src + base + k * 256) : value;
#pragma unroll
    for (int k = 0; k < 4; ++k) store16<POLICY>(dst + base + k * 256, v[k]);
}

#define CHECK(e)                                                                   \
    do {                                                                           \
        hipError_t s_ = (e);                                                       \
        if (s_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(s_));           \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

template <int POLICY, bool COPY> float run(f4 *dst, const f4 *src, size_t bytes) {
    const unsigned blocks = static_cast<unsigned>(bytes / 16384);
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    const f4 value = {1.0f, 2.0f, 3.0f, 4.0f};
    for (int i = 0; i < 3; ++i) stream_kernel<POLICY, COPY><<<blocks, 256>>>(dst, src, value);
    CHECK(hipEventRecord(t0));
    for (int i = 0; i < 20; ++i) stream_kernel<POLICY, COPY><<<blocks, 256>>>(dst, src, value);
    CHECK(hipEventRecord(t1));
    CHECK(hipEventSynchronize(t1));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    return ms / 20;
}

int main() {
    const size_t bytes = size_t(1) << 30;
    f4 *dst = nullptr, *src = nullptr;
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 0, bytes));
    const char *names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    float fill[8], copy[8];
    for (int round = 0; round < 2; ++round) {
        fill[0] = run<0, false>(dst, src, bytes); copy[0] = run<0, true>(dst, src, bytes);
        fill[1] = run<1, false>(dst, src, bytes); copy[1] = run<1, true>(dst, src, bytes);
        fill[2] = run<2, false>(dst, src, bytes); copy[2] = run<2, true>(dst, src, bytes);
        fill[3] = run<3, false>(dst, src, bytes); copy[3] = run<3, true>(dst, src, bytes);
        fill[4] = run<4, false>(dst, src, bytes); copy[4] = run<4, true>(dst, src, bytes);
        fill[5] = run<5, false>(dst, src, bytes); copy[5] = run<5, true>(dst, src, bytes);
        fill[6] = run<6, false>(dst, src, bytes); copy[6] = run<6, true>(dst, src, bytes);
        fill[7] = run<7, false>(dst, src, bytes); copy[7] = run<7, true>(dst, src, bytes);
        for (int p = 0; p < 8; ++p)
            std::printf("round %d  store bits %-11s  fill %.4f ms = %5.0f GB/s   copy %.4f ms = %5.0f GB/s (read + write)\n", round,
                        names[p], fill[p], bytes / fill[p] / 1e6, copy[p], 2.0 * bytes / copy[p] / 1e6);
    }
    CHECK(hipFree(dst));
    CHECK(hipFree(src));
    return 0;
}
