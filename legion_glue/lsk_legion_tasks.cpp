// Legion-side binding of the lsk C ABI: the GPU (TOC_PROC) bodies of LegionSolvers' six leaf
// tasks, for a Legion built with HIP (Legion_USE_HIP).  It replaces src/LinearAlgebraTasks.cu,
// src/CSRMatrixTasks.cu, src/COOMatrixTasks.cu, src/CUDAUtilities.{hpp,cpp}, src/CudaLibs.{hpp,cu}
// and src/CuSPARSEHelpers.hpp of the reference; nothing above the task bodies changes.
//
// NOT COMPILED IN THIS REPOSITORY: it needs <legion.h> and the reference's own headers
// (TaskBaseClasses.hpp, LegionUtilities.hpp, ...), neither of which is available where liblsk.so
// is built.  It is the stub a LegionSolvers maintainer adds, together with three widened
// preprocessor guards (src/TaskBaseClasses.hpp:211,304, src/LegionUtilities.hpp:68,
// src/Initialize.cpp:14,484: `LEGION_USE_CUDA` -> `LEGION_USE_CUDA || LEGION_USE_HIP`) and the
// `Legion_USE_HIP` branch of legion_glue/LegionUseHIP.cmake, which adds this file and links liblsk.so
// in place of cudart/cublas/cusparse (CMakeLists.txt:35-60).  The member names keep the
// reference's `cuda_task_body`, because that is the name TaskTDI::preregister and
// HasCUDAVariantMixin look for (src/TaskBaseClasses.hpp:211-223, src/MetaprogrammingUtilities.hpp:
// 123-145).
//
// What it takes from Legion / Realm beyond what the reference's own .cu files use.  These are ASSUMED,
// by analogy with the CUDA module the reference calls (realm/cuda/cuda_module.h, used at
// src/CUDAUtilities.cpp:7,70-74), and have to be checked against the Legion that is installed:
//   #include <realm/hip/hip_module.h>
//   Realm::Runtime::get_runtime().get_module<Realm::Hip::HipModule>("hip")
//   Realm::Hip::HipModule::get_task_hip_stream()          (CUDA: get_task_cuda_stream)
//   Realm::Hip::HipModule::set_task_ctxsync_required(bool) (CUDA: the same name)
//   hipGetDevice() inside a TOC_PROC task naming the processor's GPU (Realm makes it current, as the
//     CUDA module does for cudaGetDevice)
// Everything else (accessors, get_bounds, task->futures, preregister_task) is what the reference's
// CPU and CUDA bodies already use.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include <hip/hip_runtime_api.h>
#include <legion.h>
#include <realm/hip/hip_module.h>

#include "COOMatrixTasks.hpp"
#include "CSRMatrixTasks.hpp"
#include "CudaLibs.hpp"   // LoadCUDALibsTask (its includes of cuda_runtime / cublas / cusparse go behind the widened guard)
#include "LegionUtilities.hpp"
#include "LinearAlgebraTasks.hpp"

#include "lsk.h"
#include "lsk_legion.h"

namespace LegionSolvers {

namespace {

// The task's stream: Realm's per-task HIP stream, with the context synchroniser switched off
// because every kernel goes onto that one stream (what CUDALibraryContext::get_cuda_stream
// does with the CUDA module, src/CUDAUtilities.cpp:66-75).
lsk_stream_t task_stream() {
    Realm::Hip::HipModule *module =
        Realm::Runtime::get_runtime().get_module<Realm::Hip::HipModule>("hip");
    assert(module);
    module->set_task_ctxsync_required(false);
    return reinterpret_cast<lsk_stream_t>(module->get_task_hip_stream());
}

// The reference prints and asserts on any library failure (src/CUDAUtilities.cpp:13-63).
void check(int status) {
    if (status != LSK_OK) {
        std::fprintf(stderr, "[LegionSolvers] %s\n", lsk_last_error());
        assert(false);
    }
}

template <typename T> struct Abi;
template <> struct Abi<float> {
    static constexpr auto scal = lsk_scal_f32;
    static constexpr auto axpy = lsk_axpy_f32;
    static constexpr auto xpay = lsk_xpay_f32;
    static constexpr auto xpay_nd = lsk_xpay_nd_f32;
    static constexpr auto dot_host = lsk_dot_host_f32;
    static constexpr auto csr_spmv = lsk_csr_spmv_f32;
    static constexpr auto csr_spmv_c32 = lsk_csr_spmv_c32_f32;
    static constexpr auto csr_sliced_create = lsk_csr_sliced_create_f32;
    static constexpr auto csr_sliced_update_values = lsk_csr_sliced_update_values_f32;
    static constexpr auto csr_sliced_spmv = lsk_csr_sliced_spmv_f32;
    static constexpr auto coo_spmv = lsk_coo_spmv_f32;
};
template <> struct Abi<double> {
    static constexpr auto scal = lsk_scal_f64;
    static constexpr auto axpy = lsk_axpy_f64;
    static constexpr auto xpay = lsk_xpay_f64;
    static constexpr auto xpay_nd = lsk_xpay_nd_f64;
    static constexpr auto dot_host = lsk_dot_host_f64;
    static constexpr auto csr_spmv = lsk_csr_spmv_f64;
    static constexpr auto csr_spmv_c32 = lsk_csr_spmv_c32_f64;
    static constexpr auto csr_sliced_create = lsk_csr_sliced_create_f64;
    static constexpr auto csr_sliced_update_values = lsk_csr_sliced_update_values_f64;
    static constexpr auto csr_sliced_spmv = lsk_csr_sliced_spmv_f64;
    static constexpr auto coo_spmv = lsk_coo_spmv_f64;
};

// Set-up-time 32-bit copy of a CSR piece's structure, made on the first product of a piece and
// reused by every later one: the once-only form of what the reference redoes per call
// (src/CuSPARSEHelpers.hpp:9-31,71-80).  Entries are keyed by GPU processor, like CUDALibraryContext
// (src/CUDAUtilities.cpp:125-130); the tables themselves are shared by the processors' threads, so
// every look-up and insertion holds table_mutex (a std::map may rebalance under an insertion).  The
// copy holds STRUCTURE only (row offsets, columns): the entry values are read from the region on
// every call, as the reference does (src/CSRMatrixTasks.cu:88-118).
std::mutex table_mutex;
struct Structure32 {
    const void *rects = nullptr;
    int64_t rows = 0, nnz = 0, x_lo = 0, x_hi = -1;
    uint32_t *rowoff = nullptr, *col = nullptr; // null after a refusal: stay with the 64-bit form
};
std::map<std::pair<unsigned long long, const void *>, Structure32> structure_table;

const Structure32 &structure32(const int64_t *rects, int64_t rows, int64_t k_base, const int64_t *col,
                               int64_t nnz, int64_t x_lo, int64_t x_hi, lsk_stream_t stream) {
    const auto key = std::make_pair(Legion::Processor::get_executing_processor().id,
                                    static_cast<const void *>(col));
    std::lock_guard<std::mutex> lock(table_mutex);
    Structure32 &s = structure_table[key];
    if (s.rects == rects && s.rows == rows && s.nnz == nnz && s.x_lo == x_lo && s.x_hi == x_hi) return s;
    if (s.rowoff) check(lsk_free(s.rowoff));
    if (s.col) check(lsk_free(s.col));
    s = Structure32{rects, rows, nnz, x_lo, x_hi, nullptr, nullptr};
    void *rowoff = nullptr, *col32 = nullptr;
    // (the copies are optional memory: without room for them in the framebuffer the product reads the region's own arrays)
    const int ra = lsk_malloc(&rowoff, static_cast<size_t>(rows + 1) * sizeof(uint32_t));
    const int rb = ra == LSK_OK ? lsk_malloc(&col32, static_cast<size_t>(nnz) * sizeof(uint32_t)) : ra;
    if (rb != LSK_OK) {
        const bool no_room = std::strstr(lsk_last_error(), "OutOfMemory") != nullptr;
        if (rowoff) check(lsk_free(rowoff));
        if (!no_room) check(rb);
        return s;
    }
    const int a = lsk_csr_compress_rows_u32(rows, nnz, rects, k_base, static_cast<uint32_t *>(rowoff), stream);
    const int b = a == LSK_OK
                      ? lsk_csr_compress_cols_u32(nnz, col, x_lo, x_hi, static_cast<uint32_t *>(col32), stream)
                      : a;
    if (a == LSK_OK && b == LSK_OK) {
        s.rowoff = static_cast<uint32_t *>(rowoff);
        s.col = static_cast<uint32_t *>(col32);
    } else {
        if (a != LSK_ERR_UNSUPPORTED) check(a);
        if (b != LSK_ERR_UNSUPPORTED) check(b);
        check(lsk_free(rowoff));
        check(lsk_free(col32));
    }
    return s;
}

// The sliced-pattern copy of a CSR piece (values in blocks of 512 / 1024 rows, no column array; made
// for matrices whose rows repeat a few column patterns, i.e. every stencil; 1.9 x the rate of the
// 32-bit structure on the 27-point matrix).  It holds a COPY of the entry values, which the reference's
// task does not: OPT-IN, for applications that promise to say when they rewrite a matrix's entries.
//   LSK_SLICED=1 in the environment switches it on; without it every product reads live values
//   through the 32-bit structure above.
//   lsk_legion_matrix_values_changed(col_ptr) is the hook: call it (from any thread) after rewriting
//   the entries of the matrix whose column field instance starts at col_ptr; the next product of every
//   GPU processor then refills its copy from the region (lsk_csr_sliced_update_values_*: one pass over
//   the values) before it multiplies.
struct Sliced {
    const void *rects = nullptr, *val = nullptr;
    int64_t rows = 0, nnz = 0, x_lo = 0, x_hi = -1;
    bool tried = false, stale = false;
    lsk_csr_sliced_t *handle = nullptr; // null after a refusal
};
std::map<std::pair<unsigned long long, const void *>, Sliced> sliced_table;

template <typename T>
const lsk_csr_sliced_t *sliced_structure(const int64_t *rects, int64_t rows, int64_t k_base, const int64_t *col,
                                         const T *val, int64_t nnz, int64_t row_first, int64_t x_lo, int64_t x_hi,
                                         lsk_stream_t stream) {
    static const bool on = std::getenv("LSK_SLICED") && std::getenv("LSK_SLICED")[0] == '1';
    if (!on) return nullptr;
    const auto key = std::make_pair(Legion::Processor::get_executing_processor().id, static_cast<const void *>(col));
    std::lock_guard<std::mutex> lock(table_mutex);
    Sliced &s = sliced_table[key];
    if (s.tried && s.rects == rects && s.val == val && s.rows == rows && s.nnz == nnz && s.x_lo == x_lo && s.x_hi == x_hi) {
        if (s.handle && s.stale) {
            check(Abi<T>::csr_sliced_update_values(s.handle, rows, nnz, rects, k_base, val, stream));
            s.stale = false;
        }
        return s.handle;
    }
    if (s.handle) check(lsk_csr_sliced_destroy(s.handle));
    s = Sliced{rects, val, rows, nnz, x_lo, x_hi, true, false, nullptr};
    const int rc = Abi<T>::csr_sliced_create(rows, nnz, rects, k_base, col, val, row_first, x_lo, x_hi, stream, &s.handle);
    // (declined, or no room for the copy in the framebuffer: the product goes on with the next form)
    if (rc != LSK_OK && rc != LSK_ERR_UNSUPPORTED && std::strstr(lsk_last_error(), "OutOfMemory") == nullptr) check(rc);
    return s.handle;
}

} // namespace

// See the note above sliced_table; declared in legion_glue/lsk_legion.h.
void lsk_legion_matrix_values_changed(const void *col_ptr) {
    std::lock_guard<std::mutex> lock(table_mutex);
    for (auto &entry : sliced_table)
        if (entry.first.second == col_ptr) entry.second.stale = true;
}

namespace {

// One vector argument of a BLAS-1 task: base pointer of the dense piece and its volume.
template <typename T, int DIM, typename COORD_T, typename ACC>
struct Piece {
    Legion::Domain domain;
    T *ptr = nullptr;
    int64_t volume = 0;
    Piece(Legion::Context ctx, Legion::Runtime *rt, const Legion::Task *task,
          const std::vector<Legion::PhysicalRegion> &regions, unsigned index) {
        const auto &req = task->regions[index];
        assert(req.privilege_fields.size() == 1);
        ACC acc(regions[index], *req.privilege_fields.begin());
        domain = rt->get_index_space_domain(ctx, req.region.get_index_space());
        assert(domain.dense());
        volume = static_cast<int64_t>(domain.get_volume());
        if (volume > 0) ptr = const_cast<T *>(acc.ptr(domain.lo()));
    }
};

} // namespace

template <typename T, int DIM, typename COORD_T>
void ScalTask<T, DIM, COORD_T>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    Piece<T, DIM, COORD_T, AffineReaderWriter<T, DIM, COORD_T>> x(ctx, rt, task, regions, 0);
    const T alpha = get_alpha<T>(task->futures);
    check(Abi<T>::scal(x.volume, alpha, x.ptr, task_stream()));
}

template <typename T, int DIM, typename COORD_T>
void AxpyTask<T, DIM, COORD_T>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    Piece<T, DIM, COORD_T, AffineReaderWriter<T, DIM, COORD_T>> y(ctx, rt, task, regions, 0);
    Piece<T, DIM, COORD_T, AffineReader<T, DIM, COORD_T>> x(ctx, rt, task, regions, 1);
    assert(y.domain == x.domain);
    const T alpha = get_alpha<T>(task->futures);
    check(Abi<T>::axpy(y.volume, alpha, x.ptr, y.ptr, task_stream()));
}

template <typename T, int DIM, typename COORD_T>
void XpayTask<T, DIM, COORD_T>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    Piece<T, DIM, COORD_T, AffineReaderWriter<T, DIM, COORD_T>> y(ctx, rt, task, regions, 0);
    Piece<T, DIM, COORD_T, AffineReader<T, DIM, COORD_T>> x(ctx, rt, task, regions, 1);
    assert(y.domain == x.domain);
    const T alpha = get_alpha<T>(task->futures);
    // Exact (compact) instances are contiguous; a caller that maps padded N-D instances
    // passes extents and accessor strides to lsk_xpay_nd_* instead.
    check(Abi<T>::xpay(y.volume, alpha, x.ptr, y.ptr, task_stream()));
}

template <typename T, int DIM, typename COORD_T>
T DotTask<T, DIM, COORD_T>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    Piece<T, DIM, COORD_T, AffineReader<T, DIM, COORD_T>> v(ctx, rt, task, regions, 0);
    Piece<T, DIM, COORD_T, AffineReader<T, DIM, COORD_T>> w(ctx, rt, task, regions, 1);
    assert(v.domain == w.domain);
    T result = static_cast<T>(0);
    check(Abi<T>::dot_host(v.volume, v.ptr, w.ptr, &result, task_stream())); // syncs the stream
    return result;
}

LEGION_SOLVERS_KDR_TEMPLATE
void CSRMatvecTask<LEGION_SOLVERS_KDR_TEMPLATE_ARGS>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    static_assert(KERNEL_DIM == 1 && DOMAIN_DIM == 1 && RANGE_DIM == 1);
    static_assert(sizeof(Legion::Rect<1, KERNEL_COORD_T>) == 16 && sizeof(DOMAIN_COORD_T) == 8);
    assert(regions.size() == 4 && task->arglen == sizeof(Args));
    const Args args = *reinterpret_cast<const Args *>(task->args);
    const auto out_bounds = regions[0].template get_bounds<1, RANGE_COORD_T>().bounds;
    const auto ker_bounds = regions[1].template get_bounds<1, KERNEL_COORD_T>().bounds;
    const auto in_bounds = regions[3].template get_bounds<1, DOMAIN_COORD_T>().bounds;
    if (out_bounds.empty()) return;
    const AffineReaderWriter<ENTRY_T, 1, RANGE_COORD_T> y(regions[0], *task->regions[0].privilege_fields.begin());
    const AffineReader<ENTRY_T, 1, KERNEL_COORD_T> val(regions[1], args.fid_entry);
    const AffineReader<Legion::Point<1, DOMAIN_COORD_T>, 1, KERNEL_COORD_T> col(regions[1], args.fid_col);
    const AffineReader<Legion::Rect<1, KERNEL_COORD_T>, 1, RANGE_COORD_T> rowptr(
        regions[2], *task->regions[2].privilege_fields.begin());
    const AffineReader<ENTRY_T, 1, DOMAIN_COORD_T> x(regions[3], *task->regions[3].privilege_fields.begin());
    const bool no_x = in_bounds.empty(), no_k = ker_bounds.empty();
    if (!no_x && !no_k) {
        if (const lsk_csr_sliced_t *sl = sliced_structure<ENTRY_T>(
                reinterpret_cast<const int64_t *>(rowptr.ptr(out_bounds.lo)), static_cast<int64_t>(out_bounds.volume()),
                ker_bounds.lo[0], reinterpret_cast<const int64_t *>(col.ptr(ker_bounds.lo)), val.ptr(ker_bounds.lo),
                static_cast<int64_t>(ker_bounds.volume()), out_bounds.lo[0], in_bounds.lo[0], in_bounds.hi[0],
                task_stream())) {
            check(Abi<ENTRY_T>::csr_sliced_spmv(sl, 0, static_cast<int64_t>(out_bounds.volume()), x.ptr(in_bounds.lo),
                                                const_cast<ENTRY_T *>(y.ptr(out_bounds.lo)), /*beta=*/0, task_stream()));
            return;
        }
        const Structure32 &s32 = structure32(
            reinterpret_cast<const int64_t *>(rowptr.ptr(out_bounds.lo)), static_cast<int64_t>(out_bounds.volume()),
            ker_bounds.lo[0], reinterpret_cast<const int64_t *>(col.ptr(ker_bounds.lo)),
            static_cast<int64_t>(ker_bounds.volume()), in_bounds.lo[0], in_bounds.hi[0], task_stream());
        if (s32.rowoff) {
            check(Abi<ENTRY_T>::csr_spmv_c32(s32.rows, s32.nnz, s32.rowoff, s32.col, val.ptr(ker_bounds.lo),
                                             x.ptr(in_bounds.lo), in_bounds.hi[0] - in_bounds.lo[0] + 1,
                                             const_cast<ENTRY_T *>(y.ptr(out_bounds.lo)), /*beta=*/0,
                                             task_stream()));
            return;
        }
    }
    // beta = 0: what the reference's cuSPARSE call does (src/CSRMatrixTasks.cu:117-118).
    check(Abi<ENTRY_T>::csr_spmv(
        static_cast<int64_t>(out_bounds.volume()), no_k ? 0 : static_cast<int64_t>(ker_bounds.volume()),
        reinterpret_cast<const int64_t *>(rowptr.ptr(out_bounds.lo)), no_k ? 0 : ker_bounds.lo[0],
        no_k ? nullptr : reinterpret_cast<const int64_t *>(col.ptr(ker_bounds.lo)),
        no_k ? nullptr : val.ptr(ker_bounds.lo), no_x ? nullptr : x.ptr(in_bounds.lo),
        no_x ? 0 : in_bounds.lo[0], no_x ? -1 : in_bounds.hi[0],
        const_cast<ENTRY_T *>(y.ptr(out_bounds.lo)), /*beta=*/0, task_stream()));
}

LEGION_SOLVERS_KDR_TEMPLATE
void COOMatvecTask<LEGION_SOLVERS_KDR_TEMPLATE_ARGS>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) {
    static_assert(KERNEL_DIM == 1 && DOMAIN_DIM == 1 && RANGE_DIM == 1);
    assert(regions.size() == 3 && task->arglen == sizeof(Args));
    const Args args = *reinterpret_cast<const Args *>(task->args);
    const auto out_bounds = regions[0].template get_bounds<1, RANGE_COORD_T>().bounds;
    const auto ker_bounds = regions[1].template get_bounds<1, KERNEL_COORD_T>().bounds;
    const auto in_bounds = regions[2].template get_bounds<1, DOMAIN_COORD_T>().bounds;
    if (ker_bounds.empty() || out_bounds.empty() || in_bounds.empty()) return;
    // Reduction-privileged output, accumulated into (src/COOMatrixTasks.cu:101-108).
    const AffineSumAccessor<ENTRY_T, 1, RANGE_COORD_T> y(regions[0], *task->regions[0].privilege_fields.begin(),
                                                         LEGION_REDOP_SUM<ENTRY_T>);
    const AffineReader<ENTRY_T, 1, KERNEL_COORD_T> val(regions[1], args.fid_entry);
    const AffineReader<Legion::Point<1, RANGE_COORD_T>, 1, KERNEL_COORD_T> row(regions[1], args.fid_row);
    const AffineReader<Legion::Point<1, DOMAIN_COORD_T>, 1, KERNEL_COORD_T> col(regions[1], args.fid_col);
    const AffineReader<ENTRY_T, 1, DOMAIN_COORD_T> x(regions[2], *task->regions[2].privilege_fields.begin());
    check(Abi<ENTRY_T>::coo_spmv(
        static_cast<int64_t>(ker_bounds.volume()), reinterpret_cast<const int64_t *>(row.ptr(ker_bounds.lo)),
        reinterpret_cast<const int64_t *>(col.ptr(ker_bounds.lo)), val.ptr(ker_bounds.lo),
        x.ptr(in_bounds.lo), in_bounds.lo[0], in_bounds.hi[0],
        reinterpret_cast<ENTRY_T *>(y.ptr(out_bounds.lo)), out_bounds.lo[0], out_bounds.hi[0], task_stream()));
}

// The transposed products are registered (src/Initialize.cpp:464-479) and never launched; their
// bodies stay what the reference's are (src/CSRMatrixTasks.cu:159-164, src/COOMatrixTasks.cu:149-154).
// (liblsk has transposed products -- a set-up-time transposition onto lsk_csr_spmv_c32_* -- for the day a
// solver launches these tasks; a task body has no place to keep that set-up between calls: it belongs to the
// matrix object, which would make the transposed structure when it is built and hand it to the task.)
LEGION_SOLVERS_KDR_TEMPLATE
void CSRRmatvecTask<LEGION_SOLVERS_KDR_TEMPLATE_ARGS>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) { assert(false); }

LEGION_SOLVERS_KDR_TEMPLATE
void COORmatvecTask<LEGION_SOLVERS_KDR_TEMPLATE_ARGS>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS) { assert(false); }

// LoadCUDALibsTask (src/CudaLibs.cu:11-20 warms the stream and the cuBLAS / cuSPARSE handles of every
// GPU): there are no library handles left; what is worth doing before the first timed task is the
// library's per-device set-up (its reduction scratch, the address of its static scratch array).  The
// registration and loadCUDALibs() stay the reference's (src/CudaLibs.cu:23-66); only this body changes.
/* static */
LoadCUDALibsTask::return_type LoadCUDALibsTask::task_body(const Legion::Task *, const std::vector<Legion::PhysicalRegion> &,
                                                          Legion::Context, Legion::Runtime *) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) assert(false);
    check(lsk_init(device));
    (void)task_stream();
}

// Explicit instantiations: the same set the reference's .cu files instantiate, under the same switches
// (src/LinearAlgebraTasks.cu:241-366: entry types x index types x dimensions 1..LEGION_SOLVERS_MAX_DIM;
// src/CSRMatrixTasks.cu:167-172 and src/COOMatrixTasks.cu:157-162: <T, 1, 1, 1, long long x 3>).
#define LSK_BLAS1_BODIES(T, DIM, I)                                                    \
    template void ScalTask<T, DIM, I>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);       \
    template void AxpyTask<T, DIM, I>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);       \
    template void XpayTask<T, DIM, I>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);       \
    template T DotTask<T, DIM, I>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);
#if LEGION_SOLVERS_MAX_DIM >= 3
#define LSK_BLAS1_DIMS(T, I) LSK_BLAS1_BODIES(T, 1, I) LSK_BLAS1_BODIES(T, 2, I) LSK_BLAS1_BODIES(T, 3, I)
#elif LEGION_SOLVERS_MAX_DIM >= 2
#define LSK_BLAS1_DIMS(T, I) LSK_BLAS1_BODIES(T, 1, I) LSK_BLAS1_BODIES(T, 2, I)
#else
#define LSK_BLAS1_DIMS(T, I) LSK_BLAS1_BODIES(T, 1, I)
#endif
#ifdef LEGION_SOLVERS_USE_S32_INDICES
#define LSK_BLAS1_S32(T) LSK_BLAS1_DIMS(T, int)
#else
#define LSK_BLAS1_S32(T)
#endif
#ifdef LEGION_SOLVERS_USE_U32_INDICES
#define LSK_BLAS1_U32(T) LSK_BLAS1_DIMS(T, unsigned)
#else
#define LSK_BLAS1_U32(T)
#endif
#ifdef LEGION_SOLVERS_USE_S64_INDICES
#define LSK_BLAS1_S64(T) LSK_BLAS1_DIMS(T, long long)
#else
#define LSK_BLAS1_S64(T)
#endif
#define LSK_MATRIX_BODIES(T)                                                                                              \
    template void CSRMatvecTask<T, 1, 1, 1, long long, long long, long long>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);  \
    template void CSRRmatvecTask<T, 1, 1, 1, long long, long long, long long>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS); \
    template void COOMatvecTask<T, 1, 1, 1, long long, long long, long long>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);  \
    template void COORmatvecTask<T, 1, 1, 1, long long, long long, long long>::cuda_task_body(LEGION_SOLVERS_TASK_ARGS);
#ifdef LEGION_SOLVERS_USE_F32
LSK_BLAS1_S32(float) LSK_BLAS1_U32(float) LSK_BLAS1_S64(float)
#endif
#ifdef LEGION_SOLVERS_USE_F64
LSK_BLAS1_S32(double) LSK_BLAS1_U32(double) LSK_BLAS1_S64(double)
#endif
// (the reference instantiates the matrix tasks for both entry types whatever LEGION_SOLVERS_USE_F32 / F64 say)
LSK_MATRIX_BODIES(float)
LSK_MATRIX_BODIES(double)

} // namespace LegionSolvers
