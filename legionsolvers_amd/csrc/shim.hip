// Runtime shim: error reporting, per-device state, memory/stream wrappers, get_alpha and the
// task-id arithmetic.  Replaces src/CUDAUtilities.{hpp,cpp} and src/CudaLibs.{hpp,cu} of
// the reference; there are no library handles to keep any more, only a reduction scratch.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "lsk_common.hpp"

namespace {

thread_local char g_error[512] = "";
int g_abort_on_error = 0;

constexpr int kMaxDevices = 64;
lsk::DeviceState g_devices[kMaxDevices];
std::mutex g_device_mutex;

} // namespace

namespace lsk {

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    if (g_abort_on_error) {
        std::fprintf(stderr, "[lsk] %s\n", g_error);
        std::abort();
    }
    return code;
}

int decline(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t status, const char *expr, const char *file, int line) {
    // The failure is reported here, once: what HIP keeps as the thread's last error is taken away, or the launch
    // check of some later, successful call would report it a second time (an allocation that did not fit, which
    // a caller may well survive by going on without the copy it wanted).  A fault of the device is not
    // forgotten this way: every later call fails with it again.
    (void)hipGetLastError();
    // Same fields the reference prints in check_cuda (src/CUDAUtilities.cpp:13-27).
    return fail(LSK_ERR_HIP, "HIP failure with status code %d (%s: %s) in %s at %s:%d",
                static_cast<int>(status), hipGetErrorName(status), hipGetErrorString(status),
                expr, file, line);
}

Options &options() {
    static Options opts = [] {
        Options o;
        if (const char *e = std::getenv("LSK_CSR_VARIANT")) o.csr_variant = std::atoi(e);
        if (const char *e = std::getenv("LSK_CSR_NT")) o.csr_nontemporal = std::atoi(e);
        if (const char *e = std::getenv("LSK_CSR_MAX_UNITS")) {
            const int v = std::atoi(e);
            if (v == 0 || v == 4 || v == 8) o.csr_max_units = v;
        }
        if (const char *e = std::getenv("LSK_COO_VARIANT")) o.coo_variant = std::atoi(e) != 0;
        if (const char *e = std::getenv("LSK_DEBUG_SYNC")) o.debug_sync = std::atoi(e) != 0;
        return o;
    }();
    return opts;
}

bool debug_sync() { return options().debug_sync != 0; }

int &xcd_remap_option() {
    static const int initial = [] {
        const char *e = std::getenv("LSK_XCD_REMAP");
        return e ? std::atoi(e) : 0;
    }();
    thread_local int value = initial;
    return value;
}

int device_state(DeviceState **out) {
    int dev = -1;
    LSK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices)
        return fail(LSK_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_device_mutex);
    DeviceState &st = g_devices[dev];
    if (!st.ready) {
        LSK_HIP(hipMalloc(&st.scratch, kScratchBytes));
        st.ready = true;
    }
    *out = &st;
    return LSK_OK;
}

} // namespace lsk

extern "C" {

const char *lsk_version(void) { return "lsk 0.1 gfx950"; }

const char *lsk_last_error(void) { return g_error; }

void lsk_set_abort_on_error(int enable) { g_abort_on_error = enable; }

int lsk_set_option(const char *key, int value) {
    if (!key) return lsk::fail(LSK_ERR_ARG, "lsk_set_option: null key");
    lsk::Options &o = lsk::options();
    if (!std::strcmp(key, "csr.variant")) {
        if (value != 0 && value != 1) return lsk::fail(LSK_ERR_ARG, "csr.variant must be 0 or 1");
        o.csr_variant = value;
    } else if (!std::strcmp(key, "csr.nontemporal")) {
        o.csr_nontemporal = value != 0;
    } else if (!std::strcmp(key, "xcd_remap")) {
        lsk::xcd_remap_option() = value;
    } else if (!std::strcmp(key, "csr.log_rows_per_wave")) {
        if (value < -1 || value > 8) return lsk::fail(LSK_ERR_ARG, "csr.log_rows_per_wave must be -1..8");
        o.csr_log_rw = value;
    } else if (!std::strcmp(key, "csr.max_units")) {
        if (value != 0 && value != 4 && value != 8) return lsk::fail(LSK_ERR_ARG, "csr.max_units must be 0, 4 or 8");
        o.csr_max_units = value;
    } else if (!std::strcmp(key, "blas1.form")) {
        o.blas1_form = value != 0;
    } else if (!std::strcmp(key, "blas1.nontemporal")) {
        if (value < -1 || value > 1) return lsk::fail(LSK_ERR_ARG, "blas1.nontemporal must be -1, 0 or 1");
        o.blas1_nontemporal = value;
    } else if (!std::strcmp(key, "coo.variant")) {
        if (value != 0 && value != 1) return lsk::fail(LSK_ERR_ARG, "coo.variant must be 0 or 1");
        o.coo_variant = value;
    } else if (!std::strcmp(key, "comm.skip_single_rank")) {
        o.comm_skip_single_rank = value != 0;
    } else if (!std::strcmp(key, "sliced.form")) {
        if (value != 0 && value != 1) return lsk::fail(LSK_ERR_ARG, "sliced.form must be 0 or 1");
        o.sliced_form = value;
    } else if (!std::strcmp(key, "sliced.dma")) {
        o.sliced_dma = value != 0;
    } else if (!std::strcmp(key, "sliced.runs")) {
        o.sliced_runs = value != 0;
    } else if (!std::strcmp(key, "sliced.uniform")) {
        o.sliced_uniform = value != 0;
    } else if (!std::strcmp(key, "debug.sync")) {
        o.debug_sync = value != 0;
    } else if (!std::strcmp(key, "sliced.pair_form")) {
        if (value < -1 || value > 1) return lsk::fail(LSK_ERR_ARG, "sliced.pair_form must be -1, 0 or 1");
        o.sliced_pair_form = value;
    } else {
        return lsk::fail(LSK_ERR_ARG, "lsk_set_option: unknown key '%s'", key);
    }
    return LSK_OK;
}

int lsk_get_option(const char *key, int *value) {
    if (!key || !value) return lsk::fail(LSK_ERR_ARG, "lsk_get_option: null argument");
    const lsk::Options &o = lsk::options();
    if (!std::strcmp(key, "csr.variant")) *value = o.csr_variant;
    else if (!std::strcmp(key, "csr.nontemporal")) *value = o.csr_nontemporal;
    else if (!std::strcmp(key, "xcd_remap")) *value = lsk::xcd_remap_option();
    else if (!std::strcmp(key, "csr.log_rows_per_wave")) *value = o.csr_log_rw;
    else if (!std::strcmp(key, "csr.max_units")) *value = o.csr_max_units;
    else if (!std::strcmp(key, "blas1.form")) *value = o.blas1_form;
    else if (!std::strcmp(key, "blas1.nontemporal")) *value = o.blas1_nontemporal;
    else if (!std::strcmp(key, "coo.variant")) *value = o.coo_variant;
    else if (!std::strcmp(key, "comm.skip_single_rank")) *value = o.comm_skip_single_rank;
    else if (!std::strcmp(key, "sliced.form")) *value = o.sliced_form;
    else if (!std::strcmp(key, "sliced.dma")) *value = o.sliced_dma;
    else if (!std::strcmp(key, "sliced.runs")) *value = o.sliced_runs;
    else if (!std::strcmp(key, "sliced.uniform")) *value = o.sliced_uniform;
    else if (!std::strcmp(key, "sliced.pair_form")) *value = o.sliced_pair_form;
    else if (!std::strcmp(key, "debug.sync")) *value = o.debug_sync;
    else return lsk::fail(LSK_ERR_ARG, "lsk_get_option: unknown key '%s'", key);
    return LSK_OK;
}

int lsk_device_count(int *count) {
    if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_device_count: null output");
    int n = 0;
    hipError_t st = hipGetDeviceCount(&n);
    if (st == hipErrorNoDevice) {
        (void)hipGetLastError();
        *count = 0;
        return LSK_OK;
    }
    LSK_HIP(st);
    *count = n;
    return LSK_OK;
}

int lsk_init(int device) {
    int n = 0;
    int rc = lsk_device_count(&n);
    if (rc != LSK_OK) return rc;
    if (n == 0) return lsk::fail(LSK_ERR_NO_DEVICE, "lsk_init: no HIP device visible");
    if (device < 0 || device >= n)
        return lsk::fail(LSK_ERR_ARG, "lsk_init: device %d outside 0..%d", device, n - 1);
    LSK_HIP(hipSetDevice(device));
    lsk::DeviceState *st = nullptr;
    rc = lsk::device_state(&st);
    if (rc != LSK_OK) return rc;
    return lsk::noalloc_scratch_init();
}

int lsk_malloc(void **ptr, size_t bytes) {
    if (!ptr) return lsk::fail(LSK_ERR_ARG, "lsk_malloc: null output");
    *ptr = nullptr;
    if (bytes == 0) return LSK_OK;
    LSK_HIP(hipMalloc(ptr, bytes));
    return LSK_OK;
}

int lsk_free(void *ptr) {
    if (!ptr) return LSK_OK;
    LSK_HIP(hipFree(ptr));
    return LSK_OK;
}

int lsk_mem_info(size_t *free_bytes, size_t *total_bytes) {
    if (!free_bytes || !total_bytes) return lsk::fail(LSK_ERR_ARG, "lsk_mem_info: null output");
    LSK_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return LSK_OK;
}

int lsk_stream_create(lsk_stream_t *stream) {
    if (!stream) return lsk::fail(LSK_ERR_ARG, "lsk_stream_create: null output");
    hipStream_t s = nullptr;
    // Default priority on purpose: mixed-priority streams are not used anywhere.
    LSK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = reinterpret_cast<lsk_stream_t>(s);
    return LSK_OK;
}

int lsk_stream_destroy(lsk_stream_t stream) {
    if (!stream) return LSK_OK;
    LSK_HIP(hipStreamDestroy(lsk::as_stream(stream)));
    return LSK_OK;
}

int lsk_stream_synchronize(lsk_stream_t stream) {
    LSK_NOT_WHILE_RECORDING("lsk_stream_synchronize", stream);
    LSK_HIP(hipStreamSynchronize(lsk::as_stream(stream)));
    return LSK_OK;
}

int lsk_stream_check(lsk_stream_t stream) {
    LSK_NOT_WHILE_RECORDING("lsk_stream_check", stream);
    LSK_HIP(hipStreamSynchronize(lsk::as_stream(stream)));
    LSK_HIP(hipPeekAtLastError());
    return LSK_OK;
}

// ---- traces: a recorded sequence of leaf calls as a HIP graph ---------------------------

} // extern "C"

struct lsk_trace {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t nodes = 0;
    mutable int64_t replays = 0;
};

extern "C" {

int lsk_stream_begin_capture(lsk_stream_t stream) {
    if (!stream) return lsk::fail(LSK_ERR_ARG, "lsk_stream_begin_capture: the default stream cannot record");
    if (lsk::debug_sync())
        return lsk::fail(LSK_ERR_UNSUPPORTED, "lsk_stream_begin_capture: option debug.sync is on (every launch waits for "
                                              "the device, which a recording cannot hold)");
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    LSK_HIP(hipStreamIsCapturing(lsk::as_stream(stream), &status));
    if (status != hipStreamCaptureStatusNone)
        return lsk::fail(LSK_ERR_ARG, "lsk_stream_begin_capture: the stream is recording already");
    // thread-local mode: only this thread's unsafe calls (allocation, synchronisation) spoil the recording
    LSK_HIP(hipStreamBeginCapture(lsk::as_stream(stream), hipStreamCaptureModeThreadLocal));
    return LSK_OK;
}

int lsk_stream_is_recording(lsk_stream_t stream, int *recording) {
    if (!recording) return lsk::fail(LSK_ERR_ARG, "lsk_stream_is_recording: null output");
    *recording = (stream && lsk::stream_recording(lsk::as_stream(stream))) ? 1 : 0;
    return LSK_OK;
}

int lsk_stream_end_capture(lsk_stream_t stream, lsk_trace_t **trace) {
    if (!trace) return lsk::fail(LSK_ERR_ARG, "lsk_stream_end_capture: null output");
    *trace = nullptr;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    // (asked before HIP is: its refusal to end what never began would stay behind as the thread's last
    // error; a query that fails itself means a spoiled recording, which must still be ended)
    if (hipStreamIsCapturing(lsk::as_stream(stream), &status) == hipSuccess && status == hipStreamCaptureStatusNone)
        return lsk::fail(LSK_ERR_ARG, "lsk_stream_end_capture: the stream is not recording");
    hipGraph_t graph = nullptr;
    const hipError_t ended = hipStreamEndCapture(lsk::as_stream(stream), &graph);
    if (ended != hipSuccess || !graph) {
        for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {} // the spoiled recording's error must not meet a later launch check
        if (graph) (void)hipGraphDestroy(graph);
        return lsk::hip_fail(ended != hipSuccess ? ended : hipErrorStreamCaptureInvalidated,
                             "hipStreamEndCapture (a call inside the recording failed: no trace, and the stream may stay unusable)",
                             __FILE__, __LINE__);
    }
    lsk_trace *t = new lsk_trace;
    t->graph = graph;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e == hipSuccess) e = hipGraphInstantiate(&t->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        delete t;
        return lsk::hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
    }
    t->nodes = static_cast<int64_t>(n);
    *trace = t;
    return LSK_OK;
}

int lsk_trace_replay(const lsk_trace_t *trace, lsk_stream_t stream) {
    if (!trace || !trace->exec) return lsk::fail(LSK_ERR_ARG, "lsk_trace_replay: null trace");
    LSK_HIP(hipGraphLaunch(trace->exec, lsk::as_stream(stream)));
    trace->replays += 1;
    return LSK_OK;
}

int lsk_trace_info(const lsk_trace_t *trace, int64_t *out2) {
    if (!trace || !out2) return lsk::fail(LSK_ERR_ARG, "lsk_trace_info: null pointer");
    out2[0] = trace->nodes;
    out2[1] = trace->replays;
    return LSK_OK;
}

int lsk_trace_destroy(lsk_trace_t *trace) {
    if (!trace) return LSK_OK;
    hipError_t e = hipSuccess;
    if (trace->exec) e = hipGraphExecDestroy(trace->exec);
    if (trace->graph) {
        const hipError_t e2 = hipGraphDestroy(trace->graph);
        if (e == hipSuccess) e = e2;
    }
    delete trace;
    if (e != hipSuccess) return lsk::hip_fail(e, "lsk_trace_destroy", __FILE__, __LINE__);
    return LSK_OK;
}

int lsk_memcpy_h2d(void *dst, const void *src, size_t bytes, lsk_stream_t stream) {
    if (bytes == 0) return LSK_OK;
    if (!dst || !src) return lsk::fail(LSK_ERR_ARG, "lsk_memcpy_h2d: null pointer");
    LSK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, lsk::as_stream(stream)));
    return LSK_OK;
}

int lsk_memcpy_d2h(void *dst, const void *src, size_t bytes, lsk_stream_t stream) {
    if (bytes == 0) return LSK_OK;
    if (!dst || !src) return lsk::fail(LSK_ERR_ARG, "lsk_memcpy_d2h: null pointer");
    LSK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, lsk::as_stream(stream)));
    return LSK_OK;
}

int lsk_memcpy_d2d(void *dst, const void *src, size_t bytes, lsk_stream_t stream) {
    if (bytes == 0) return LSK_OK;
    if (!dst || !src) return lsk::fail(LSK_ERR_ARG, "lsk_memcpy_d2d: null pointer");
    LSK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, lsk::as_stream(stream)));
    return LSK_OK;
}

// ---- get_alpha (src/LegionUtilities.cpp:72-97), host side -----------------------------

#define LSK_GET_ALPHA(SUFFIX, T)                                                       \
    int lsk_get_alpha_##SUFFIX(int nf, const T *f, T *alpha) {                         \
        if (!alpha) return lsk::fail(LSK_ERR_ARG, "lsk_get_alpha: null output");       \
        if (nf < 0 || nf > 4)                                                          \
            return lsk::fail(LSK_ERR_ARG, "lsk_get_alpha: %d futures (0..4 allowed)", nf); \
        if (nf > 0 && !f) return lsk::fail(LSK_ERR_ARG, "lsk_get_alpha: null futures"); \
        switch (nf) {                                                                  \
        case 0: *alpha = static_cast<T>(1); break;                                     \
        case 1: *alpha = f[0]; break;                                                  \
        case 2: *alpha = f[0] / f[1]; break;                                           \
        case 3: *alpha = (f[0] * f[1]) / f[2]; break;                                  \
        default: *alpha = (f[0] * f[1]) / (f[2] * f[3]); break;                        \
        }                                                                              \
        return LSK_OK;                                                                 \
    }
LSK_GET_ALPHA(f32, float)
LSK_GET_ALPHA(f64, double)
#undef LSK_GET_ALPHA

// ---- task ids (src/TaskBaseClasses.hpp) ----------------------------------------------

namespace {
constexpr int64_t kTaskIdOrigin = 500000; // LEGION_SOLVERS_TASK_ID_ORIGIN, LibraryOptions.hpp:25
constexpr int64_t kNumMetaTaskIds = 1;    // LOAD_CUDA_LIBS_META_TASK_ID only, TaskIDs.hpp:9-14
constexpr int64_t kNumEntryTypes = 3;     // TypeList<float, double, void>
constexpr int64_t kNumIndexTypes = 4;     // TypeList<int, unsigned, long long, void>
constexpr int64_t kMaxDim = 3;            // LEGION_SOLVERS_MAX_DIM
constexpr int64_t kIdx1 = kNumIndexTypes, kIdx2 = kIdx1 * kNumIndexTypes,
                  kIdx3 = kIdx2 * kNumIndexTypes;
constexpr int64_t kDim1 = kMaxDim, kDim2 = kDim1 * kMaxDim, kDim3 = kDim2 * kMaxDim;
constexpr int64_t kBlockSize = kNumEntryTypes * kDim3 * kIdx3; // 5184
inline int64_t block_origin(int block_id) {
    return kTaskIdOrigin + kNumMetaTaskIds + kBlockSize * block_id;
}
} // namespace

int64_t lsk_task_id_t(int block_id, int entry_index) { return block_origin(block_id) + entry_index; }

int64_t lsk_task_id_di(int block_id, int dim, int index_type_index) {
    return block_origin(block_id) + kMaxDim * index_type_index + (dim - 1);
}

int64_t lsk_task_id_tdi(int block_id, int entry_index, int dim, int index_type_index) {
    return block_origin(block_id) + kNumEntryTypes * kMaxDim * index_type_index +
           kNumEntryTypes * (dim - 1) + entry_index;
}

int64_t lsk_task_id_tdddiii(int block_id, int entry_index, int n1, int n2, int n3, int i1,
                            int i2, int i3) {
    return block_origin(block_id) + kIdx3 * kNumEntryTypes * kDim2 * (n1 - 1) +
           kIdx3 * kNumEntryTypes * kDim1 * (n2 - 1) + kIdx3 * kNumEntryTypes * 1 * (n3 - 1) +
           kIdx2 * kNumEntryTypes * i1 + kIdx1 * kNumEntryTypes * i2 +
           1 * kNumEntryTypes * i3 + entry_index;
}

} // extern "C"
