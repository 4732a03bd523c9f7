// Shared helpers for the HIP translation units behind include/lsk.h.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "lsk.h"

namespace lsk {

// Records `code` and a printf-style message for lsk_last_error(); prints and abort()s when
// lsk_set_abort_on_error(1) is in force.  Returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// The same for an answer the caller is expected to handle (a set-up probe that says "not for this
// input"): records the message and returns the code, but never aborts.
int decline(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

int hip_fail(hipError_t status, const char *expr, const char *file, int line);

#define LSK_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t lsk_status_ = (expr);                                               \
        if (lsk_status_ != hipSuccess)                                                 \
            return ::lsk::hip_fail(lsk_status_, #expr, __FILE__, __LINE__);            \
    } while (false)

// After a kernel launch: surfaces launch-configuration errors without synchronising.
// After every launch: the launch's own status and, with option "debug.sync" (LSK_DEBUG_SYNC=1), the whole
// device waited for and its pending asynchronous error reported at THIS call site -- what the reference's
// debug builds do at task exit (CHECK_CUDA_STREAM, src/CUDAUtilities.cpp:110-122).
#define LSK_LAUNCH_CHECK()                                                                           \
    do {                                                                                             \
        LSK_HIP(hipGetLastError());                                                                  \
        if (lsk::debug_sync()) {                                                                     \
            LSK_HIP(hipDeviceSynchronize());                                                         \
            LSK_HIP(hipGetLastError());                                                              \
        }                                                                                            \
    } while (false)

bool debug_sync(); // option "debug.sync"

constexpr int kBlock = 256;          // threads per workgroup: 4 wave64
constexpr int kWave = 64;            // CDNA wavefront width
constexpr int kNumXcd = 8;           // MI355X accelerator complex dies
constexpr int kMaxGridBlocks = 2048; // 256 CUs x 8 workgroups for grid-stride kernels

inline hipStream_t as_stream(lsk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kMaxLdsBytes = 160 << 10; // LDS of one gfx950 CU, and the most one workgroup can ask for

// A launch with more than 64 KiB of dynamic LDS needs the kernel's cap raised first.  The cap belongs
// to the kernel (per device), not to a launch or to a handle: it is raised to the whole CU's LDS, once
// per kernel instance and device (`done`: one bit per device, a static of the call site), so that no
// later launch with another size lowers it under a live one, and a product launched again makes no
// runtime call besides the launch itself.  The one call that does raise it must not be recorded into
// a graph: while `s` is capturing it is refused with a message that says so.
inline int allow_large_lds(const void *kernel, std::atomic<uint64_t> &done, hipStream_t s) {
    int dev = 0;
    LSK_HIP(hipGetDevice(&dev));
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return LSK_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) (void)hipGetLastError();
    if (capturing != hipStreamCaptureStatusNone)
        return fail(LSK_ERR_UNSUPPORTED, "the first launch of this kernel on device %d raises its LDS limit, which cannot be "
                                         "captured: launch the product once before capturing it", dev);
    hipFuncAttributes attr;
    LSK_HIP(hipFuncGetAttributes(&attr, kernel)); // (what the kernel declares statically counts against the CU's LDS too)
    LSK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kMaxLdsBytes - static_cast<int>(attr.sharedSizeBytes)));
    done.fetch_or(bit, std::memory_order_release);
    return LSK_OK;
}

// Whether `s` is recording a trace (a query that fails itself means a spoiled recording: still one).
inline bool stream_recording(hipStream_t s) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &status) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return status != hipStreamCaptureStatusNone;
}

// For the calls that wait for the device (timing, results read on the host): refused while the stream
// records, BEFORE anything is enqueued, so that the recording stays whole.
#define LSK_NOT_WHILE_RECORDING(name, stream)                                                                         \
    do {                                                                                                              \
        if (::lsk::stream_recording(::lsk::as_stream(stream)))                                                        \
            return ::lsk::fail(LSK_ERR_UNSUPPORTED, "%s: the stream is recording a trace and this call waits for "    \
                                                    "the device; nothing was enqueued", name);                       \
    } while (false)

// Per-device library state: the partial-sum scratch of the reductions.
struct DeviceState {
    void *scratch = nullptr; // kScratchBytes of HBM
    bool ready = false;
};
constexpr size_t kScratchBytes = size_t(kMaxGridBlocks) * 16; // one 16-byte slot per block

// State of the current device, created on first use (lsk_init does it eagerly).
int device_state(DeviceState **out);

// Looks up, for the current device, the address of the static device array behind
// lsk_dot_noalloc_f64 (defined next to the array in blas1.hip; lsk_init calls it).
int noalloc_scratch_init();

// Library options (lsk_set_option); initial values may come from the environment variables
// LSK_CSR_VARIANT / LSK_CSR_NT / LSK_XCD_REMAP.
struct Options {
    int csr_variant = 1;     // 0: workgroup tiles, 1: wave tiles
    int csr_nontemporal = 0; // stream col/val with nontemporal loads
    int csr_log_rw = -1;     // log2 rows per wave for the wave-tile form; -1 = from the mean row length
    int csr_max_units = 0;   // 16-byte units per lane in flight in the wave-tile form: 0 = by tile size, 4 or 8
    int blas1_form = 1;      // 0: grid-strided accesses, 1: contiguous chunk per workgroup
    int blas1_nontemporal = -1; // BLAS-1 streams past the caches: 1, 0, -1 = for vectors of 128 MiB and more
    int coo_variant = 1;     // 0: workgroup chunks through LDS, 1: wave-wide segmented scan
    int comm_skip_single_rank = 1; // one-rank communicators: answer without calling RCCL
    // Sliced-pattern product: 1 wave-staged (every wave stages the windows of x of its own rows and
    // meets the block's other waves only to write y), 0 block-staged (one barrier before the sums).
    // Measured, fused dot shares included: 4.90 against 5.13 ms on the 27-point 512^3 fp64 matrix, 3 %
    // on the 7-point one, equal in fp32.
    int sliced_form = 1;
    int sliced_dma = 1;       // the wave-staged form sends the windows of x from memory straight into LDS
    int sliced_runs = 1;      // ... and reads lists of runs of three run by run (0: entry by entry)
    int sliced_uniform = 0;   // ... and takes the values as scalars where every list position has one value (no value loads)
    // Two right-hand sides: 1 wave-staged, one vector after the other through one LDS area; 0 both
    // vectors' windows in LDS at once (block-staged); -1: 1 where the wave-staged form serves.
    int sliced_pair_form = -1;
    int debug_sync = 0;       // every launch followed by a device synchronisation and an error check
};
Options &options();

// Option "xcd_remap" (tiles per XCD chunk; 0 off, < 0 one contiguous eighth per XCD) is kept per
// HOST THREAD: a caller sets it right before its launch and the tuners flip it while they time
// candidates, neither of which may leak into launches made from other threads (Legion runs one
// thread per GPU processor).  A thread's first value is LSK_XCD_REMAP from the environment, or 0.
int &xcd_remap_option();

// Workgroup -> logical tile remap for XCD-local reuse of the input-vector window.  Workgroups
// are dealt round-robin over the 8 XCDs (block b shares an XCD with b+8, b+16, ...).  With
// chunk c > 0, every group of 8*c consecutive block ids is re-dealt so that each XCD's c blocks
// of the group take c CONSECUTIVE tiles; c < 0 gives each XCD one contiguous eighth of the
// grid; c == 0 is the identity.  Bijective for any grid size (a trailing partial group keeps
// the identity).  Placement is a speed hint only: results never depend on it.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks, int chunk) {
    if (chunk == 0) return bid;
    if (chunk < 0) {
        const unsigned xcd = bid % kNumXcd;
        const unsigned q = nblocks / kNumXcd;
        const unsigned r = nblocks % kNumXcd;
        const unsigned first = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        return first + bid / kNumXcd;
    }
    const unsigned group = kNumXcd * unsigned(chunk);
    const unsigned g = bid / group;
    if ((g + 1) * group > nblocks) return bid; // trailing partial group
    const unsigned in = bid % group;
    return g * group + (in % kNumXcd) * unsigned(chunk) + in / kNumXcd;
}

// Sum over the 64 lanes of a wave; every lane returns the total.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Lane 0's value, as a wave-uniform (scalar-register) value.
__device__ __forceinline__ long long wave_first(long long v) {
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
    const int hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}

} // namespace lsk
