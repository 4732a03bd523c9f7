// C++ host layer above the lsk C ABI: the reference's caller classes with their own names and call
// shapes -- Scalar (src/Scalar.hpp), PartitionedVector (src/PartitionedVector.hpp), CSRMatrix /
// COOMatrix (src/CSRMatrix.hpp, src/COOMatrix.hpp), SquarePlanner (src/SquarePlanner.hpp),
// CGSolver / BiCGStabSolver / GMRESSolver (src/CGSolver.hpp, src/BiCGStabSolver.hpp,
// src/GMRESSolver.hpp) -- for ONE process driving ONE GPU, without Legion.  A vector of n elements is
// split into `pieces` equal row blocks (the colours of Legion's create_equal_partition); every
// operation launches one leaf per piece through include/lsk.h, as the reference's index launches
// do.  Scalars are device-resident futures, so a solver step enqueues without a host round trip.
//
// Header-only; needs only lsk.h and liblsk.so.  No arithmetic happens here.
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lsk.h"

namespace LegionSolversHost {

inline void check(int status) {
    if (status != LSK_OK) throw std::runtime_error(std::string("lsk: ") + lsk_last_error());
}

// Whether a failed call failed for lack of device memory: the set-up-time copies are optional, a matrix
// that fills the GPU is multiplied on the arrays it came in.
inline bool no_room(const std::exception &e) { return std::strstr(e.what(), "OutOfMemory") != nullptr; }

// Device memory a program can work without: the set-up-time copies of the matrices.  Their owners register a
// function that frees some of it and says whether it did; an allocation that finds no room calls them, one at
// a time, and tries again (DeviceArray).  Nothing is given up while a recorded trace may still refer to it.
struct OptionalMemory {
    std::vector<std::pair<const void *, std::function<bool()>>> owners;
    int recordings = 0; // Trace objects alive
    static OptionalMemory &get() {
        static OptionalMemory registry;
        return registry;
    }
    int held_back = 0;  // allocations under way that must fail rather than cost a copy (SquarePlanner::allocate_workspace)
    bool release() {
        if (recordings > 0 || held_back > 0) return false;
        for (auto &owner : owners)
            if (owner.second()) return true;
        return false;
    }
    void forget(const void *who) {
        owners.erase(std::remove_if(owners.begin(), owners.end(), [who](const auto &o) { return o.first == who; }),
                     owners.end());
    }
};

// The typed entry points of the ABI, selected by the entry type.
template <typename T> struct Abi;
#define LSK_HOST_ABI(T, SUF)                                                                       \
    template <> struct Abi<T> {                                                                    \
        static constexpr auto fill = lsk_fill_##SUF;                                               \
        static constexpr auto copy = lsk_copy_##SUF;                                               \
        static constexpr auto scal_dev = lsk_scal_dev_##SUF;                                       \
        static constexpr auto axpy_dev = lsk_axpy_dev_##SUF;                                       \
        static constexpr auto xpay_dev = lsk_xpay_dev_##SUF;                                       \
        static constexpr auto dot = lsk_dot_##SUF;                                                 \
        static constexpr auto scalar_op = lsk_scalar_op_##SUF;                                     \
        static constexpr auto sum_scalars = lsk_sum_scalars_##SUF;                                 \
        static constexpr auto sum_partials = lsk_sum_partials_##SUF;                               \
        static constexpr auto axpy_sqnorm_dev = lsk_axpy_sqnorm_dev_##SUF;                         \
        static constexpr auto axpy_dot_dev = lsk_axpy_dot_dev_##SUF;                               \
        static constexpr auto axpy_xpay_dev = lsk_axpy_xpay_dev_##SUF;                             \
        static constexpr auto csr_spmv = lsk_csr_spmv_##SUF;                                       \
        static constexpr auto csr_spmv_c32 = lsk_csr_spmv_c32_##SUF;                               \
        static constexpr auto csr_spmv_c32_split = lsk_csr_spmv_c32_split_##SUF;                   \
        static constexpr auto csr_spmv_c32_tune = lsk_csr_spmv_c32_tune_##SUF;                     \
        static constexpr auto csr_spmv_tune = lsk_csr_spmv_tune_##SUF;                             \
        static constexpr auto coo_spmv = lsk_coo_spmv_##SUF;                                       \
        static constexpr auto coo_binned_create = lsk_coo_binned_create_##SUF;                     \
        static constexpr auto coo_binned_spmv = lsk_coo_binned_spmv_##SUF;                         \
        static constexpr auto coo_binned_update_values = lsk_coo_binned_update_values_##SUF;       \
        static constexpr auto multi_dot = lsk_multi_dot_##SUF;                                     \
        static constexpr auto multi_axpy = lsk_multi_axpy_##SUF;                                   \
        static constexpr auto gather = lsk_gather_##SUF;                                           \
        static constexpr auto csr_sliced_create = lsk_csr_sliced_create_##SUF;                     \
        static constexpr auto csr_sliced_spmv = lsk_csr_sliced_spmv_##SUF;                         \
        static constexpr auto csr_sliced_spmv_dot = lsk_csr_sliced_spmv_dot_##SUF;                 \
        static constexpr auto csr_sliced_spmv2 = lsk_csr_sliced_spmv2_##SUF;                       \
        static constexpr auto csr_sliced_spmv2_dot = lsk_csr_sliced_spmv2_dot_##SUF;               \
        static constexpr auto csr_sliced_tune = lsk_csr_sliced_tune_##SUF;                         \
        static constexpr auto csr_sliced_tune2 = lsk_csr_sliced_tune2_##SUF;                       \
        static constexpr auto csr_sliced_update_values = lsk_csr_sliced_update_values_##SUF;       \
        static constexpr auto stencil_fill_csr = lsk_stencil_fill_csr_##SUF;                       \
        static constexpr auto stencil_fill_coo = lsk_stencil_fill_coo_##SUF;                       \
    }
LSK_HOST_ABI(float, f32);
LSK_HOST_ABI(double, f64);
#undef LSK_HOST_ABI

// What the reference threads through every class as (Legion::Context, Legion::Runtime *): here the
// GPU, the stream every leaf is enqueued on, and the workspace of the dot products.
class Context {
public:
    explicit Context(int device = 0) {
        check(lsk_init(device));
        check(lsk_stream_create(&stream_));
        workspace_bytes_ = lsk_dot_workspace_bytes(0);
        check(lsk_malloc(&workspace_, workspace_bytes_));
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ~Context() {
        lsk_stream_synchronize(stream_);
        for (void *chunk : chunks_) lsk_free(chunk);
        lsk_free(workspace_);
        lsk_stream_destroy(stream_);
    }
    // 8-byte device slots for scalars, recycled without freeing device memory (a hipFree waits for
    // the device, once per scalar per solver step otherwise).  A recycled slot is rewritten by a
    // kernel enqueued after every reader of its old value: everything runs on the one stream.
    void *acquire_slot() const {
        if (free_slots_.empty()) add_slots();
        void *slot = free_slots_.back();
        free_slots_.pop_back();
        return slot;
    }
    // At least `count` free slots without another allocation (nothing may allocate while a trace records).
    void reserve_slots(size_t count) const {
        while (free_slots_.size() < count) add_slots();
    }
    void release_slot(void *slot) const { (parked_ ? *parked_ : free_slots_).push_back(slot); }
    // While a trace records, the scalars of the recorded steps must keep their slots for as long as the
    // trace may be replayed (a replay writes them again): slots released meanwhile go to `parked`
    // instead of back into circulation (Trace hands them back when it is destroyed).
    void park_released_slots(std::vector<void *> *parked) const { parked_ = parked; }
    void return_slots(const std::vector<void *> &slots) const { free_slots_.insert(free_slots_.end(), slots.begin(), slots.end()); }
    lsk_stream_t stream() const { return stream_; }
    void *workspace() const { return workspace_; }
    size_t workspace_bytes() const { return workspace_bytes_; }
    void synchronize() const { check(lsk_stream_synchronize(stream_)); }

private:
    void add_slots() const {
        constexpr size_t kChunk = 4096;
        void *chunk = nullptr;
        check(lsk_malloc(&chunk, kChunk * 8));
        chunks_.push_back(chunk);
        for (size_t i = kChunk; i-- > 0;) free_slots_.push_back(static_cast<char *>(chunk) + 8 * i);
    }
    lsk_stream_t stream_ = nullptr;
    void *workspace_ = nullptr;
    size_t workspace_bytes_ = 0;
    mutable std::vector<void *> chunks_, free_slots_;
    mutable std::vector<void *> *parked_ = nullptr;
};

// Device memory that frees itself.
template <typename E> class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(int64_t n) : n_(n) {
        void *p = nullptr;
        for (;;) {
            const int status = lsk_malloc(&p, static_cast<size_t>(std::max<int64_t>(n, 1)) * sizeof(E));
            if (status == LSK_OK) break;
            // no room: the matrices' set-up-time copies go before an allocation fails (OptionalMemory)
            const bool no_room_for_it = std::strstr(lsk_last_error(), "OutOfMemory") != nullptr;
            const std::string message = std::string("lsk: ") + lsk_last_error();
            if (!no_room_for_it || !OptionalMemory::get().release()) throw std::runtime_error(message);
        }
        p_ = static_cast<E *>(p);
    }
    DeviceArray(DeviceArray &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceArray &operator=(DeviceArray &&o) noexcept {
        if (this != &o) {
            lsk_free(p_);
            p_ = o.p_;
            n_ = o.n_;
            o.p_ = nullptr;
            o.n_ = 0;
        }
        return *this;
    }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    ~DeviceArray() { lsk_free(p_); }
    E *get() const { return p_; }
    int64_t size() const { return n_; }
    std::vector<E> to_host(const Context &ctx, int64_t first = 0, int64_t count = -1) const {
        if (count < 0) count = n_ - first;
        std::vector<E> out(static_cast<size_t>(count));
        check(lsk_memcpy_d2h(out.data(), p_ + first, static_cast<size_t>(count) * sizeof(E), ctx.stream()));
        ctx.synchronize();
        return out;
    }
    void from_host(const Context &ctx, const std::vector<E> &src) {
        check(lsk_memcpy_h2d(p_, src.data(), src.size() * sizeof(E), ctx.stream()));
        ctx.synchronize();
    }

private:
    E *p_ = nullptr;
    int64_t n_ = 0;
};

// Scalar<T> (src/Scalar.hpp): a future of one T, here one T in device memory written by a
// stream-ordered kernel; the operators are the scalar tasks of src/UtilityTasks.cpp:34-99.
template <typename T> class Scalar {
public:
    Scalar() = default;
    Scalar(const Context &ctx, T value) : Scalar(uninitialised(ctx)) {
        check(Abi<T>::fill(1, value, writable(), ctx.stream()));
    }
    // A future nobody has written yet.  The Context must outlive every Scalar made from it.
    static Scalar uninitialised(const Context &ctx) {
        static_assert(sizeof(T) <= 8, "scalar slots are 8 bytes");
        Scalar s;
        s.ctx_ = &ctx;
        const Context *owner = &ctx;
        s.v_ = std::shared_ptr<T>(static_cast<T *>(ctx.acquire_slot()), [owner](T *p) { owner->release_slot(p); });
        return s;
    }
    const T *get_future() const { return v_.get(); }
    T *writable() const { return v_.get(); }
    T get() const { // blocks, like Future::get_result
        T out;
        check(lsk_memcpy_d2h(&out, v_.get(), sizeof(T), ctx_->stream()));
        ctx_->synchronize();
        return out;
    }
    Scalar operator-() const { return op(LSK_SOP_NEG, nullptr); }
    Scalar operator+(const Scalar &r) const { return op(LSK_SOP_ADD, &r); }
    Scalar operator-(const Scalar &r) const { return op(LSK_SOP_SUB, &r); }
    Scalar operator*(const Scalar &r) const { return op(LSK_SOP_MUL, &r); }
    Scalar operator/(const Scalar &r) const { return op(LSK_SOP_DIV, &r); }
    Scalar sqrt() const { return op(LSK_SOP_SQRT, nullptr); }
    Scalar rsqrt() const { return op(LSK_SOP_RSQRT, nullptr); }

private:
    Scalar op(int code, const Scalar *r) const {
        Scalar out = uninitialised(*ctx_);
        check(Abi<T>::scalar_op(code, get_future(), r ? r->get_future() : nullptr, out.writable(), ctx_->stream()));
        return out;
    }
    const Context *ctx_ = nullptr;
    std::shared_ptr<T> v_;
};

// Bounds of `pieces` consecutive blocks of 0..n-1 as Legion's create_equal_partition colours
// them: sizes differ by at most one, the larger blocks come last.
struct EqualPartition {
    int64_t n = 0;
    std::vector<std::pair<int64_t, int64_t>> piece; // inclusive; hi < lo when empty
    EqualPartition() = default;
    EqualPartition(int64_t size, int pieces) : n(size) {
        if (pieces < 1) throw std::invalid_argument("a partition needs at least one colour");
        int64_t px = 0;
        for (int i = 0; i < pieces; ++i) {
            const int64_t nx = (n - px) / (pieces - i);
            piece.emplace_back(px, px + nx - 1);
            px += nx;
        }
    }
    int colors() const { return static_cast<int>(piece.size()); }
    int64_t volume(int c) const { return piece[c].second - piece[c].first + 1; }
    bool operator==(const EqualPartition &o) const { return n == o.n && piece == o.piece; }
};

// PartitionedVector<T> (src/PartitionedVector.hpp): one dense buffer, coloured by the partition.
//
// Held-back updates.  axpy(alpha, x, /*defer=*/true) does not launch at once: the two vectors
// remember it, and the next thing that happens decides how it runs.  x.xpay(beta, r) next (CG:
// x_sol += alpha p, then p <- r + beta p): both in one pass over x.  A dot product with the updated
// vector next (CG: r -= alpha q, then r.r; GMRES: w -= h v_k, then v_{k+1}.w): the update leaves
// that dot's partial sums behind.  Anything else that reaches either vector's storage (data(),
// piece(), to_host(), any operation) runs the update first, on its own.  Values are the same
// either way; pointers from data() / piece() are to be fetched again after an operation rather than
// kept, as a Legion accessor is valid for one mapping only.
template <typename T> class PartitionedVector {
public:
    PartitionedVector(const Context &ctx, std::string name, const EqualPartition &partition)
        : ctx_(&ctx), name_(std::move(name)), part_(partition), data_(partition.n), slots_(partition.colors()),
          partial_stride_(partial_slots(partition)), partials_(static_cast<int64_t>(partition.colors()) * partial_stride_) {}
    const Context &context() const { return *ctx_; }
    const EqualPartition &partition() const { return part_; }
    const std::string &name() const { return name_; }
    T *data() const { settle(); return data_.get(); }
    void settle_now() const { settle(); } // run a held-back update on its own (SquarePlanner::settle)
    T *piece(int c) const { settle(); return raw_piece(c); }
    std::vector<T> to_host() const { settle(); return data_.to_host(*ctx_); }

    void constant_fill(T value) { each([&](int c) { check(Abi<T>::fill(part_.volume(c), value, piece(c), ctx_->stream())); }); }
    void zero_fill() { constant_fill(T(0)); }
    PartitionedVector &operator=(const PartitionedVector &src) { // CopyTask per piece
        same_partition(src);
        each([&](int c) { check(Abi<T>::copy(part_.volume(c), src.piece(c), piece(c), ctx_->stream())); });
        return *this;
    }
    PartitionedVector(const PartitionedVector &) = delete;
    ~PartitionedVector() {
        if (deferred_) detach(); // never leave the partner pointing at a dead vector
    }

    // alpha = get_alpha(futures) (src/LegionUtilities.cpp:72-97): f0, f0/f1, f0*f1/f2, f0*f1/(f2*f3)
    void scal(const std::vector<Scalar<T>> &alpha) {
        const T *f[4];
        const int nf = futures(alpha, f);
        each([&](int c) { check(Abi<T>::scal_dev(part_.volume(c), nf, f[0], f[1], f[2], f[3], piece(c), ctx_->stream())); });
    }
    void axpy(const std::vector<Scalar<T>> &alpha, PartitionedVector &x, bool defer = false) {
        same_partition(x);
        if (alpha.size() > 4) throw std::invalid_argument("alpha takes at most 4 futures");
        if (defer && &x != this) {
            settle();
            x.settle();
            deferred_ = x.deferred_ = std::make_shared<Deferred>(Deferred{this, &x, alpha});
            return;
        }
        const T *f[4];
        const int nf = futures(alpha, f);
        each([&](int c) {
            check(Abi<T>::axpy_dev(part_.volume(c), nf, f[0], f[1], f[2], f[3], x.piece(c), piece(c), ctx_->stream()));
        });
    }
    void xpay(const std::vector<Scalar<T>> &alpha, const PartitionedVector &x) {
        same_partition(x);
        const T *f[4];
        const int nf = futures(alpha, f);
        if (deferred_ && deferred_->p == this && &x != deferred_->z && &x != this && !x.deferred_) {
            // z <- fma(a, this, z) was held back: one pass over this vector serves both updates
            const std::shared_ptr<Deferred> d = detach();
            const T *fa[4];
            const int nfa = futures(d->scalars, fa);
            each([&](int c) {
                check(Abi<T>::axpy_xpay_dev(part_.volume(c), nfa, fa[0], fa[1], fa[2], fa[3], nf, f[0], f[1], f[2], f[3],
                                            x.raw_piece(c), raw_piece(c), d->z->raw_piece(c), ctx_->stream()));
            });
            return;
        }
        each([&](int c) {
            check(Abi<T>::xpay_dev(part_.volume(c), nf, f[0], f[1], f[2], f[3], x.piece(c), piece(c), ctx_->stream()));
        });
    }
    // One dot leaf per piece, then the pieces' results added in colour order: Legion's REDOP_SUM
    // over the future map (src/PartitionedVector.cpp:353-357).
    Scalar<T> dot(const PartitionedVector &w) const {
        same_partition(w);
        Scalar<T> out = Scalar<T>::uninitialised(*ctx_);
        // the pieces' results live in a buffer of this vector; everything is ordered on one stream,
        // so the next dot may reuse it
        std::shared_ptr<Deferred> d = (deferred_ && deferred_->z == this) ? deferred_ : w.deferred_;
        if (d && (d->z == this || d->z == &w)) {
            PartitionedVector *z = d->z;
            const PartitionedVector *other = z == this ? &w : this;
            if (other == z || !other->deferred_ || other->deferred_ == d) {
                // an update of one operand was held back: it runs now and leaves this dot's partial sums
                z->detach();
                const T *f[4];
                const int nf = futures(d->scalars, f);
                for (int c = 0; c < part_.colors(); ++c) {
                    const int64_t kPartials = z->partial_stride_;
                    T *buf = z->partials_.get() + static_cast<int64_t>(c) * kPartials;
                    int count = 0;
                    if (part_.volume(c) > 0) {
                        if (other == z)
                            check(Abi<T>::axpy_sqnorm_dev(part_.volume(c), nf, f[0], f[1], f[2], f[3], d->p->raw_piece(c),
                                                          z->raw_piece(c), buf, kPartials, &count, ctx_->stream()));
                        else
                            check(Abi<T>::axpy_dot_dev(part_.volume(c), nf, f[0], f[1], f[2], f[3], d->p->raw_piece(c),
                                                       z->raw_piece(c), other->raw_piece(c), buf, kPartials, &count,
                                                       ctx_->stream()));
                    }
                    check(Abi<T>::sum_partials(count, buf, slots_.get() + c, nullptr, 0, ctx_->stream()));
                }
                check(Abi<T>::sum_scalars(part_.colors(), slots_.get(), out.writable(), ctx_->stream()));
                return out;
            }
        }
        for (int c = 0; c < part_.colors(); ++c)
            check(Abi<T>::dot(part_.volume(c), piece(c), w.piece(c), slots_.get() + c, ctx_->workspace(),
                              ctx_->workspace_bytes(), ctx_->stream()));
        check(Abi<T>::sum_scalars(part_.colors(), slots_.get(), out.writable(), ctx_->stream()));
        return out;
    }

    // k dot products with this vector in one pass over it per 8 vectors: out[j] (device) <- v[j] . this,
    // each equal, bit for bit, to v[j]->dot(*this).  `rows` is scratch of k * colours values.
    void multi_dot_into(const std::vector<const PartitionedVector *> &v, T *out, T *rows, void *workspace,
                        size_t workspace_bytes) const {
        const int k = static_cast<int>(v.size());
        std::vector<const T *> ptrs(k);
        int live = 0;
        for (int c = 0; c < part_.colors(); ++c) {
            if (part_.volume(c) <= 0) continue;
            for (int j = 0; j < k; ++j) {
                same_partition(*v[j]);
                ptrs[j] = v[j]->piece(c);
            }
            check(Abi<T>::multi_dot(part_.volume(c), k, ptrs.data(), piece(c), live == 0 ? out : rows + static_cast<int64_t>(live) * k,
                                    workspace, workspace_bytes, ctx_->stream()));
            // the pieces' results are added in colour order, as dot() adds them
            if (live > 0)
                for (int j = 0; j < k; ++j)
                    check(Abi<T>::scalar_op(LSK_SOP_ADD, out + j, rows + static_cast<int64_t>(live) * k + j, out + j, ctx_->stream()));
            ++live;
        }
        if (live == 0) check(Abi<T>::fill(k, T(0), out, ctx_->stream()));
    }
    // this <- this -+ sum_j coef[j] x[j], element for element the chain of axpys in ascending j.
    void multi_axpy(const std::vector<const PartitionedVector *> &x, const T *coef_dev, bool negate) {
        const int k = static_cast<int>(x.size());
        std::vector<const T *> ptrs(k);
        each([&](int c) {
            for (int j = 0; j < k; ++j) {
                same_partition(*x[j]);
                if (x[j] == this) throw std::invalid_argument("a source vector is the destination");
                ptrs[j] = x[j]->piece(c);
            }
            check(Abi<T>::multi_axpy(part_.volume(c), k, ptrs.data(), coef_dev, negate ? 1 : 0, piece(c), nullptr, 0, nullptr,
                                     ctx_->stream()));
        });
    }

private:
    // Room for the partial sums a fused leaf leaves per piece: 2048 always suffice; a piece that no cache holds
    // is updated a few per cent faster with one workgroup (and one partial sum) per 16 KiB, up to LSK_MAX_PARTIALS.
    static int64_t partial_slots(const EqualPartition &partition) {
        int64_t largest = 0;
        for (int c = 0; c < partition.colors(); ++c) largest = std::max(largest, partition.volume(c));
        const int64_t chunks = largest * static_cast<int64_t>(sizeof(T)) / 16384 + 1;
        return std::min<int64_t>(std::max<int64_t>(chunks, 2048), LSK_MAX_PARTIALS);
    }
    struct Deferred {
        PartitionedVector *z, *p; // z <- fma(alpha, p, z), not launched yet
        std::vector<Scalar<T>> scalars;
    };
    std::shared_ptr<Deferred> detach() const {
        std::shared_ptr<Deferred> d = deferred_;
        d->z->deferred_.reset();
        d->p->deferred_.reset();
        return d;
    }
    void settle() const { // run an update that was being held back, before the storage is touched
        if (deferred_) {
            const std::shared_ptr<Deferred> d = detach();
            d->z->axpy(d->scalars, *d->p);
        }
    }
    T *raw_piece(int c) const { return data_.get() + part_.piece[c].first; }
    template <typename F> void each(F &&f) {
        for (int c = 0; c < part_.colors(); ++c)
            if (part_.volume(c) > 0) f(c);
    }
    void same_partition(const PartitionedVector &o) const {
        if (!(part_ == o.part_)) throw std::invalid_argument("vectors are partitioned differently");
    }
    static int futures(const std::vector<Scalar<T>> &alpha, const T *(&f)[4]) {
        if (alpha.size() > 4) throw std::invalid_argument("alpha takes at most 4 futures");
        for (int i = 0; i < 4; ++i) f[i] = i < static_cast<int>(alpha.size()) ? alpha[i].get_future() : nullptr;
        return static_cast<int>(alpha.size());
    }
    const Context *ctx_;
    std::string name_;
    EqualPartition part_;
    DeviceArray<T> data_;
    DeviceArray<T> slots_;    // one dot result per piece
    int64_t partial_stride_;  // room for partial sums per piece (partial_slots)
    DeviceArray<T> partials_; // partial sums of the fused leaves, partial_stride_ per piece
    mutable std::shared_ptr<Deferred> deferred_;
};

// AbstractMatrix (src/AbstractMatrix.hpp) as the planner uses it.
template <typename T> class AbstractMatrix {
public:
    virtual ~AbstractMatrix() = default;
    // dst <- beta * dst + A src, one leaf per colour of dst's partition
    virtual void matvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const = 0;
    // dst <- beta * dst + A^T src.  CSRRmatvecTask / COORmatvecTask are registered by the reference
    // (src/Initialize.cpp:464-479) with bodies that are assert(false) (src/CSRMatrixTasks.cpp:95-100,
    // src/COOMatrixTasks.cpp:78-83) and nothing launches them; here the product runs on the
    // transpose, built on the first call.
    virtual void rmatvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const = 0;
    virtual bool accumulates_only() const = 0;
    // Milliseconds of one product of the matrix's first LARGE piece (the rows a set-up-time timing is worth) from x
    // (the whole input vector) into y (that piece's rows), as the planner would launch it; 0: there is no such piece.
    virtual float time_large_product(const T *, T *) const { return 0.0f; }
    // Optional fused form: dst <- A src and, in `shares` (room for dot_shares() values), partial sums
    // that add up to dot(dst, w).  dot_shares() == 0: the matrix has no such form.
    virtual int64_t dot_shares() const { return 0; }
    virtual void matvec_dot(PartitionedVector<T> &, const PartitionedVector<T> &, const PartitionedVector<T> &, T *) const {
        throw std::logic_error("this matrix has no fused product + dot");
    }
    // Optional: dst_a <- A src_a and dst_b <- A src_b in one pass over the matrix (two systems on one
    // matrix), each bit for bit what matvec(dst, src, 0) gives; with w_a / w_b also the shares of
    // dot(dst_a, w_a) and dot(dst_b, w_b) as matvec_dot leaves them.  False: not served, nothing launched.
    virtual bool matvec_pair(PartitionedVector<T> &, PartitionedVector<T> &, const PartitionedVector<T> &,
                             const PartitionedVector<T> &, const PartitionedVector<T> *, const PartitionedVector<T> *, T *,
                             T *) const {
        return false;
    }
};

// The CSR form of the transpose of a matrix, in the library's 32-bit structure: one row per element
// of the domain, column ids = the matrix's rows.  Built by stable sorts of the entries by column, in
// runs of at most kChunk entries (a sort takes fewer than 2^31); every run becomes a structure of
// its own over the interval of the domain its columns span, and the runs' products accumulate in
// order.  Products are lsk_csr_spmv_c32_*: no atomics, the same bits on every call.
template <typename T> class Transposed {
public:
    static constexpr int64_t kChunk = int64_t(1) << 30;
    // col: the int64 global column of every entry; row32: its row (0xffffffff: entry to be dropped)
    Transposed(const Context &ctx, int64_t nnz, const int64_t *col, const uint32_t *row32, const T *val,
               int64_t domain, int64_t rows, int64_t chunk = kChunk)
        : ctx_(&ctx), domain_(domain), rows_(rows) {
        lsk_stream_t s = ctx.stream();
        for (int64_t k0 = 0; k0 < nnz && domain > 0; k0 += chunk) {
            const int64_t n = std::min(chunk, nnz - k0);
            DeviceArray<int64_t> bounds(2);
            check(lsk_index_bounds_i64(n, col + k0, bounds.get(), s));
            const auto b = bounds.to_host(ctx);
            Part p;
            p.first = std::max<int64_t>(b[0], 0);
            p.extent = std::min<int64_t>(b[1], domain - 1) - p.first + 1;
            if (p.extent <= 0) continue;
            DeviceArray<uint32_t> keys(n), order(n), sorted(n);
            check(lsk_csr_compress_cols_u32(n, col + k0, p.first, p.first + p.extent - 1, keys.get(), s));
            check(lsk_coo_block_order_u32(n, keys.get(), p.extent, 1, order.get(), s));
            check(lsk_gather_u32(n, keys.get(), order.get(), sorted.get(), s));
            p.offsets = DeviceArray<uint32_t>(p.extent + 1);
            check(lsk_sorted_key_offsets_u32(n, sorted.get(), p.extent, p.offsets.get(), s));
            p.nnz = p.offsets.to_host(ctx, p.extent, 1)[0]; // entries whose column lies inside the domain
            if (p.nnz <= 0) continue;
            p.col32 = DeviceArray<uint32_t>(n);
            p.val = DeviceArray<T>(n);
            check(lsk_gather_u32(n, row32 + k0, order.get(), p.col32.get(), s));
            check(Abi<T>::gather(n, val + k0, order.get(), p.val.get(), s));
            ctx.synchronize(); // the scratch arrays go out of scope
            parts_.push_back(std::move(p));
        }
    }
    int runs() const { return static_cast<int>(parts_.size()); }
    void apply(T *out, const T *src, int beta) const {
        if (domain_ <= 0) return;
        const bool whole = parts_.size() == 1 && parts_[0].extent == domain_;
        if (beta == 0 && !whole) check(Abi<T>::fill(domain_, T(0), out, ctx_->stream()));
        for (const Part &p : parts_)
            check(Abi<T>::csr_spmv_c32(p.extent, p.nnz, p.offsets.get(), p.col32.get(), p.val.get(), src, rows_,
                                       out + p.first, whole ? beta : 1, ctx_->stream()));
    }

private:
    struct Part {
        int64_t first = 0, extent = 0, nnz = 0;
        DeviceArray<uint32_t> offsets, col32;
        DeviceArray<T> val;
    };
    const Context *ctx_;
    int64_t domain_, rows_;
    std::vector<Part> parts_;
};

// CSRMatrix<T> (src/CSRMatrix.hpp) in the reference's layout: per row a {lo, hi} rect of global
// entry indices, int64 global columns, T entries.  The kernel piece of a row block is the image of
// its rects (src/CSRMatrix.cpp:89-109).  At construction the matrix gets the sliced-pattern copy where
// the library accepts it, else a 32-bit copy of the structure per piece (bit-identical products, fewer bytes).
template <typename T> class CSRMatrix : public AbstractMatrix<T> {
public:
    CSRMatrix(const Context &ctx, const EqualPartition &range, int64_t domain_size, DeviceArray<int64_t> rects,
              DeviceArray<int64_t> col, DeviceArray<T> val, bool compress = true, int redraws = 0, bool slice = true)
        : ctx_(&ctx), range_(range), domain_(domain_size), rects_(std::move(rects)), col_(std::move(col)),
          val_(std::move(val)) {
        for (int c = 0; c < range_.colors(); ++c) {
            Piece p;
            if (range_.volume(c) > 0) {
                const auto first = rects_.to_host(ctx, 2 * range_.piece[c].first, 2);
                const auto last = rects_.to_host(ctx, 2 * range_.piece[c].second, 2);
                p.k_lo = first[0];
                p.nnz = std::max<int64_t>(last[1] - first[0] + 1, 0);
            }
            pieces_.push_back(std::move(p));
        }
        OptionalMemory::get().owners.emplace_back(this, [this] { return release_optional_memory(); });
        // (the sliced-pattern copy first; a matrix that gets it gets no 32-bit copy of the structure as well: no
        // product would read it, and it is 4 bytes per entry that a larger problem can use)
        if (slice && col_.size() > 0 && domain_ > 0 && try_slice(redraws)) return;
        if (compress && col_.size() > 0 && domain_ > 0) try_compress();
        float best = tune();
        // Where the driver puts the arrays in HBM moves the product's time by several per cent, and
        // user code cannot steer it; so the arrays are copied into fresh allocations a few times
        // and the fastest set is kept (needs room for a second copy; stops quietly without).
        for (int i = 0; i < redraws && best > 0.0f && compressed(); ++i) {
            try {
                DeviceArray<uint32_t> col32(col32_.size());
                DeviceArray<T> val(val_.size());
                std::vector<DeviceArray<uint32_t>> offs;
                check(lsk_memcpy_d2d(col32.get(), col32_.get(), static_cast<size_t>(col32_.size()) * 4, ctx.stream()));
                check(lsk_memcpy_d2d(val.get(), val_.get(), static_cast<size_t>(val_.size()) * sizeof(T), ctx.stream()));
                for (auto &p : pieces_) {
                    offs.emplace_back(p.rowoff.size());
                    check(lsk_memcpy_d2d(offs.back().get(), p.rowoff.get(), static_cast<size_t>(p.rowoff.size()) * 4,
                                         ctx.stream()));
                }
                ctx.synchronize();
                std::vector<int> held;
                for (auto &p : pieces_) held.push_back(p.placement);
                std::swap(col32, col32_);
                std::swap(val, val_);
                for (size_t c = 0; c < pieces_.size(); ++c) std::swap(offs[c], pieces_[c].rowoff);
                const float ms = tune();
                if (ms > 0.0f && ms < best) {
                    best = ms; // the old set is freed on leaving the scope
                } else {
                    std::swap(col32, col32_);
                    std::swap(val, val_);
                    for (size_t c = 0; c < pieces_.size(); ++c) {
                        std::swap(offs[c], pieces_[c].rowoff);
                        pieces_[c].placement = held[c];
                    }
                }
            } catch (const std::runtime_error &) {
                break;
            }
        }
    }
    ~CSRMatrix() override { OptionalMemory::get().forget(this); }
    CSRMatrix(const CSRMatrix &) = delete;
    CSRMatrix &operator=(const CSRMatrix &) = delete;
    // Called when an allocation found no room: the sliced-pattern copies go first (a copy of all values) and
    // the 32-bit copy of the structure is made in their place where that fits (half the bytes); asked again,
    // that one goes as well and the products read the arrays as they were handed over.
    bool release_optional_memory() {
        if (sliced()) {
            std::fprintf(stderr, "[LegionSolversHost] an allocation found no room: sliced-pattern copy of the matrix given up\n");
            sliced_.clear();
            if (col_.size() > 0 && domain_ > 0) try_compress();
            return true;
        }
        if (compressed()) {
            std::fprintf(stderr, "[LegionSolversHost] an allocation found no room: 32-bit copy of the matrix structure given up\n");
            col32_ = DeviceArray<uint32_t>();
            for (auto &p : pieces_) {
                p.rowoff = DeviceArray<uint32_t>();
                p.long_rows.reset();
            }
            return true;
        }
        return false;
    }
    bool compressed() const { return col32_.size() > 0; }
    bool sliced() const { return !sliced_.empty(); }
    // Whether the first product of a large piece times the placements once more on the vectors it is given
    // (see place_on_these_vectors); a recording must not contain it: Trace runs its steps after a warm-up.
    void set_placement_on_vectors(bool on) { place_on_vectors_ = on; }
    bool accumulates_only() const override { return false; }
    void matvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const override {
        if (!(dst.partition() == range_) || src.partition().n != domain_)
            throw std::invalid_argument("matrix and vectors do not fit");
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            if (rows <= 0) continue;
            const Piece &p = pieces_[c];
            if (sliced() && sliced_[c].handle) {
                if (beta == 0) place_on_these_vectors(c, rows, src.data(), dst.piece(c));
                check(lsk_set_option("xcd_remap", sliced_[c].placement));
                check(Abi<T>::csr_sliced_spmv(sliced_[c].handle, 0, rows, src.data(), dst.piece(c), beta, ctx_->stream()));
                continue;
            }
            check(lsk_set_option("xcd_remap", p.placement));
            if (compressed()) // (p.long_rows: rows too long for one wave go to workgroups of their own; null = none)
                check(Abi<T>::csr_spmv_c32_split(p.long_rows.get(), 0, rows, p.nnz, p.rowoff.get(), col32_.get() + p.k_lo,
                                                 val_.get() + p.k_lo, src.data(), domain_, dst.piece(c), beta,
                                                 ctx_->stream()));
            else
                check(Abi<T>::csr_spmv(rows, p.nnz, rects_.get() + 2 * range_.piece[c].first, p.k_lo,
                                       col_.get() + p.k_lo, val_.get() + p.k_lo, src.data(), 0, domain_ - 1,
                                       dst.piece(c), beta, ctx_->stream()));
        }
    }
    float time_large_product(const T *x, T *y) const override {
        if (!sliced()) return 0.0f;
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            if (rows < kTuneMinRows || !sliced_[c].handle) continue;
            int64_t info[9] = {};
            check(lsk_csr_sliced_info(sliced_[c].handle, info));
            if (info[0] * static_cast<int64_t>(sizeof(T)) < kPlaceMinBytes) continue; // (lives in the caches)
            auto launch = [&] {
                check(lsk_set_option("xcd_remap", sliced_[c].placement));
                check(Abi<T>::csr_sliced_spmv(sliced_[c].handle, 0, rows, x, y, 0, ctx_->stream()));
            };
            launch();
            ctx_->synchronize();
            float ms = 0.0f;
            for (int reps = 3, round = 0;; ++round) { // (a short product: once more over enough launches for 6 ms)
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < reps; ++i) launch();
                ctx_->synchronize();
                ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / float(reps);
                if (ms * float(reps) >= 5.0f || reps >= 40 || round >= 1) break;
                reps = std::min(40, static_cast<int>(6.0f / std::max(ms, 0.15f)) + 1);
            }
            return ms;
        }
        return 0.0f;
    }
    // On the sliced-pattern copy a lane owns its rows, so the product can leave the shares of
    // dot(dst, w) for a wave reduction per block: one share per block of every piece.
    int64_t dot_shares() const override {
        if (!sliced()) return 0;
        int64_t total = 0;
        for (int c = 0; c < range_.colors(); ++c) {
            if (range_.volume(c) > 0 && !sliced_[c].handle) return 0; // a piece without entries: no fused form
            total += shares_of(c, range_.volume(c));
        }
        return total;
    }
    void matvec_dot(PartitionedVector<T> &dst, const PartitionedVector<T> &src, const PartitionedVector<T> &w,
                    T *shares) const override {
        if (!(dst.partition() == range_) || !(w.partition() == range_) || src.partition().n != domain_ || !dot_shares())
            throw std::invalid_argument("matrix and vectors do not fit");
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            if (rows <= 0) continue;
            int64_t count = 0;
            place_on_these_vectors(c, rows, src.data(), dst.piece(c));
            check(lsk_set_option("xcd_remap", sliced_[c].placement));
            check(Abi<T>::csr_sliced_spmv_dot(sliced_[c].handle, 0, rows, src.data(), dst.piece(c), 0, w.piece(c), shares,
                                              shares_of(c, rows), &count, ctx_->stream()));
            shares += count;
        }
    }
    bool matvec_pair(PartitionedVector<T> &dst_a, PartitionedVector<T> &dst_b, const PartitionedVector<T> &src_a,
                     const PartitionedVector<T> &src_b, const PartitionedVector<T> *w_a, const PartitionedVector<T> *w_b,
                     T *shares_a, T *shares_b) const override {
        if (!sliced()) return false;
        if (!(dst_a.partition() == range_) || !(dst_b.partition() == range_) || src_a.partition().n != domain_ ||
            src_b.partition().n != domain_)
            throw std::invalid_argument("matrix and vectors do not fit");
        for (int c = 0; c < range_.colors(); ++c) { // all pieces or none
            if (range_.volume(c) <= 0) continue;
            int64_t info[9] = {};
            if (!sliced_[c].handle) return false;
            check(lsk_csr_sliced_info(sliced_[c].handle, info));
            if (!info[7]) return false;
        }
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            if (rows <= 0) continue;
            const int untimed = sliced_[c].placement ? sliced_[c].placement : 64; // (only round robin is slow)
            const bool timed_later = sliced_[c].pair_placement < 0 && recording(); // (timing waits for the device)
            if (sliced_[c].pair_placement < 0 && !timed_later) {
                // two windows of x compete for each XCD's L2: the pair gets a placement of its own, timed once
                // on the vectors at hand (the destinations are about to be overwritten anyway)
                sliced_[c].pair_placement = untimed;
                if (rows >= kTuneMinRows) {
                    float ms = 0.0f;
                    check(Abi<T>::csr_sliced_tune2(sliced_[c].handle, src_a.data(), src_b.data(), dst_a.piece(c),
                                                   dst_b.piece(c), ctx_->stream(), &sliced_[c].pair_placement, &ms));
                }
            }
            check(lsk_set_option("xcd_remap", timed_later ? untimed : sliced_[c].pair_placement));
            int status;
            int64_t count = 0;
            if (w_a)
                status = Abi<T>::csr_sliced_spmv2_dot(sliced_[c].handle, 0, rows, src_a.data(), src_b.data(), dst_a.piece(c),
                                                      dst_b.piece(c), 0, w_a->piece(c), w_b->piece(c), shares_a, shares_b,
                                                      shares_of(c, rows), &count, ctx_->stream());
            else
                status = Abi<T>::csr_sliced_spmv2(sliced_[c].handle, 0, rows, src_a.data(), src_b.data(), dst_a.piece(c),
                                                  dst_b.piece(c), 0, ctx_->stream());
            check(status);
            if (w_a) {
                shares_a += count;
                shares_b += count;
            }
        }
        return true;
    }
    void rmatvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const override {
        if (!(src.partition() == range_) || dst.partition().n != domain_)
            throw std::invalid_argument("matrix and vectors do not fit");
        if (!transposed_) {
            DeviceArray<uint32_t> row32(col_.size());
            check(lsk_csr_entry_rows_u32(range_.n, col_.size(), rects_.get(), 0, row32.get(), ctx_->stream()));
            transposed_ = std::make_unique<Transposed<T>>(*ctx_, col_.size(), col_.get(), row32.get(), val_.get(),
                                                          domain_, range_.n, transpose_chunk);
        }
        check(lsk_set_option("xcd_remap", 0));
        transposed_->apply(dst.data(), src.data(), beta);
    }
    int transposed_runs() const { return transposed_ ? transposed_->runs() : 0; }
    // The entries, for a caller that rewrites them in place (same structure, new values) ...
    T *values() { return val_.get(); }
    // ... and then says so, before the next product: the set-up-time copies of the values (the
    // sliced-pattern copies, the transpose behind rmatvec) are brought up to date.  The reference's
    // task reads the region on every call (src/CSRMatrixTasks.cu:88-118).
    void values_changed() {
        for (int c = 0; c < static_cast<int>(sliced_.size()); ++c) {
            if (!sliced_[c].handle) continue;
            const Piece &p = pieces_[c];
            check(Abi<T>::csr_sliced_update_values(sliced_[c].handle, range_.volume(c), p.nnz,
                                                   rects_.get() + 2 * range_.piece[c].first, p.k_lo, val_.get() + p.k_lo,
                                                   ctx_->stream()));
        }
        transposed_.reset();
    }
    int64_t transpose_chunk = Transposed<T>::kChunk; // entries per run of the transposition (tests lower it)

private:
    struct Piece {
        int64_t k_lo = 0, nnz = 0;
        int placement = 0; // tile -> XCD placement the library measured fastest for this piece
        DeviceArray<uint32_t> rowoff;
        struct DestroyLongRows {
            void operator()(lsk_csr_long_rows_t *plan) const { (void)lsk_csr_long_rows_destroy(plan); }
        };
        std::unique_ptr<lsk_csr_long_rows_t, DestroyLongRows> long_rows; // lsk_csr_long_rows_create, where it applies
    };
    // Set-up-time choice of the tile placement, per piece of at least kTuneMinRows rows (the
    // library times a few candidates on the piece itself; a speed hint, results do not depend on it).
    static constexpr int64_t kTuneMinRows = int64_t(1) << 21;
    // a product that streams less than this from HBM says nothing about where its vectors should lie
    static constexpr int64_t kPlaceMinBytes = int64_t(1) << 30;
    float tune() { // returns the milliseconds of one product over the tuned pieces (0: nothing tuned)
        int64_t most = 0;
        float total = 0.0f;
        for (int c = 0; c < range_.colors(); ++c) most = std::max(most, range_.volume(c));
        if (most < kTuneMinRows || domain_ <= 0) return total;
        DeviceArray<T> x, y;
        try {
            x = DeviceArray<T>(domain_);
            y = DeviceArray<T>(most);
        } catch (const std::runtime_error &e) {
            if (!no_room(e)) throw;
            return total; // no room for the timing's two temporary vectors: no timing
        }
        check(Abi<T>::fill(domain_, T(0), x.get(), ctx_->stream()));
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            Piece &p = pieces_[c];
            if (rows < kTuneMinRows || p.nnz <= 0) continue;
            float ms = 0.0f;
            if (compressed())
                check(Abi<T>::csr_spmv_c32_tune(rows, p.nnz, p.rowoff.get(), col32_.get() + p.k_lo, val_.get() + p.k_lo,
                                                x.get(), domain_, y.get(), ctx_->stream(), &p.placement, &ms));
            else
                check(Abi<T>::csr_spmv_tune(rows, p.nnz, rects_.get() + 2 * range_.piece[c].first, p.k_lo,
                                            col_.get() + p.k_lo, val_.get() + p.k_lo, x.get(), 0, domain_ - 1, y.get(),
                                            ctx_->stream(), &p.placement, &ms));
            total += ms;
        }
        return total;
    }
    // The placement of a large piece's sliced product, timed once more on the first vectors it is really
    // multiplied with (every row of y is about to be overwritten): which placement is fastest depends on where
    // x and y lie in HBM as much as on the matrix (on the 27-point 512^3 matrix set-up's choice on temporary
    // vectors was 3-4 % off the best in three processes of five), and a solver multiplies the same few vectors
    // over and over.
    bool recording() const { // timing waits for the device, which a trace that is being recorded cannot hold
        int yes = 0;
        check(lsk_stream_is_recording(ctx_->stream(), &yes));
        return yes != 0;
    }
    void place_on_these_vectors(int c, int64_t rows, const T *x, T *y) const {
        const Sliced &s = sliced_[c];
        if (!place_on_vectors_ || s.placed_on_vectors || rows < kTuneMinRows || recording()) return;
        s.placed_on_vectors = true;
        float ms = 0.0f;
        check(Abi<T>::csr_sliced_tune(s.handle, x, y, ctx_->stream(), &s.placement, &ms));
    }
    // The sliced-pattern copy of a piece (lsk_csr_sliced_*): for matrices whose rows repeat a few
    // column patterns, the product then reads no column array at all and sums every row in entry
    // order, bit for bit the loop of src/CSRMatrixTasks.cpp:73-91.
    struct Sliced {
        lsk_csr_sliced_t *handle = nullptr;
        mutable int placement = 0;
        mutable bool placed_on_vectors = false; // place_on_these_vectors has run for this piece
        mutable int pair_placement = -1; // of the two-vector product; -1: not chosen yet (matvec_pair does)
        Sliced() = default;
        Sliced(Sliced &&o) noexcept : handle(o.handle), placement(o.placement), pair_placement(o.pair_placement) {
            o.handle = nullptr;
        }
        Sliced &operator=(Sliced &&o) noexcept {
            if (this != &o) {
                lsk_csr_sliced_destroy(handle);
                handle = o.handle;
                placement = o.placement;
                pair_placement = o.pair_placement;
                o.handle = nullptr;
            }
            return *this;
        }
        Sliced(const Sliced &) = delete;
        Sliced &operator=(const Sliced &) = delete;
        ~Sliced() { lsk_csr_sliced_destroy(handle); }
    };
    int64_t shares_of(int c, int64_t rows) const { return lsk_csr_sliced_dot_partials(sliced_[c].handle, 0, rows); }
    bool make_slice(int c, Sliced &out) const { // false: the library declined
        const Piece &p = pieces_[c];
        const int rc = Abi<T>::csr_sliced_create(range_.volume(c), p.nnz, rects_.get() + 2 * range_.piece[c].first, p.k_lo,
                                                 col_.get() + p.k_lo, val_.get() + p.k_lo, range_.piece[c].first, 0,
                                                 domain_ - 1, ctx_->stream(), &out.handle);
        if (rc == LSK_ERR_UNSUPPORTED) return false;
        check(rc);
        return true;
    }
    // All pieces or none.  Large pieces then get their placement by timing, and up to `redraws` more
    // copies are made, the fastest kept (where the driver puts the copy in HBM moves the product's
    // time by several per cent).
    bool try_slice(int redraws) {
        std::vector<Sliced> made(range_.colors());
        int64_t most = 0;
        for (int c = 0; c < range_.colors(); ++c) {
            if (range_.volume(c) <= 0 || pieces_[c].nnz <= 0) continue;
            try {
                if (!make_slice(c, made[c])) return false;
            } catch (const std::runtime_error &e) {
                if (!no_room(e)) throw;
                std::fprintf(stderr, "[LegionSolversHost] no room for the sliced-pattern copy of the matrix: left out\n");
                return false; // the copy does not fit beside the matrix: the products go on without it
            }
            most = std::max(most, range_.volume(c));
        }
        if (most == 0) return false;
        DeviceArray<T> x, y; // (the timing's two temporary vectors: without room for them, no timing)
        try {
            if (most >= kTuneMinRows) {
                x = DeviceArray<T>(domain_);
                y = DeviceArray<T>(most);
            }
        } catch (const std::runtime_error &e) {
            if (!no_room(e)) throw;
            x = DeviceArray<T>();
        }
        if (x.size() > 0 && y.size() > 0) {
            check(Abi<T>::fill(domain_, T(0), x.get(), ctx_->stream()));
            for (int c = 0; c < range_.colors(); ++c) {
                if (!made[c].handle || range_.volume(c) < kTuneMinRows) continue;
                float best = 0.0f;
                check(Abi<T>::csr_sliced_tune(made[c].handle, x.get(), y.get(), ctx_->stream(), &made[c].placement, &best));
                for (int i = 0; i < redraws; ++i) {
                    Sliced other;
                    float ms = 0.0f;
                    try {
                        if (!make_slice(c, other)) break;
                    } catch (const std::runtime_error &) {
                        break; // no room for a second copy
                    }
                    check(Abi<T>::csr_sliced_tune(other.handle, x.get(), y.get(), ctx_->stream(), &other.placement, &ms));
                    if (ms < best) {
                        best = ms;
                        made[c] = std::move(other);
                    }
                }
                // The driver scrubs the memory of the copies just freed in the background, and a
                // product that runs meanwhile loses a few per cent: leave once this one is back at
                // the time it was chosen for, three probes in a row (or after 10 s).
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(10);
                for (int good = 0; redraws > 0 && good < 3 && std::chrono::steady_clock::now() < deadline;) {
                    float ms = 0.0f;
                    int placement = 0;
                    check(Abi<T>::csr_sliced_tune(made[c].handle, x.get(), y.get(), ctx_->stream(), &placement, &ms));
                    good = ms <= 1.01f * best ? good + 1 : 0;
                }
            }
        }
        sliced_ = std::move(made);
        return true;
    }
    void try_compress() {
        try {
            compress_or_throw();
        } catch (const std::runtime_error &e) {
            if (!no_room(e)) throw;
            std::fprintf(stderr, "[LegionSolversHost] no room for the 32-bit copy of the matrix structure: left out\n");
            col32_ = DeviceArray<uint32_t>(); // no room for the copy: products on the arrays as handed over
            for (auto &p : pieces_) {
                p.rowoff = DeviceArray<uint32_t>();
                p.long_rows.reset();
            }
        }
    }
    void compress_or_throw() {
        DeviceArray<uint32_t> col32(col_.size());
        if (lsk_csr_compress_cols_u32(col_.size(), col_.get(), 0, domain_ - 1, col32.get(), ctx_->stream()) != LSK_OK)
            return;
        for (int c = 0; c < range_.colors(); ++c) {
            const int64_t rows = range_.volume(c);
            if (rows <= 0) continue;
            DeviceArray<uint32_t> off(rows + 1);
            const int rc = lsk_csr_compress_rows_u32(rows, pieces_[c].nnz, rects_.get() + 2 * range_.piece[c].first,
                                                     pieces_[c].k_lo, off.get(), ctx_->stream());
            if (rc == LSK_ERR_UNSUPPORTED) {
                for (auto &p : pieces_) {
                    p.rowoff = DeviceArray<uint32_t>();
                    p.long_rows.reset();
                }
                return;
            }
            check(rc);
            pieces_[c].rowoff = std::move(off);
            lsk_csr_long_rows_t *plan = nullptr;
            const int lrc = lsk_csr_long_rows_create(rows, pieces_[c].rowoff.get(), 0, ctx_->stream(), &plan);
            if (lrc != LSK_ERR_UNSUPPORTED) check(lrc); // unsupported: no row beyond a few thousand entries
            pieces_[c].long_rows.reset(plan);
        }
        col32_ = std::move(col32);
    }
    const Context *ctx_;
    EqualPartition range_;
    int64_t domain_;
    DeviceArray<int64_t> rects_, col_;
    DeviceArray<T> val_;
    DeviceArray<uint32_t> col32_;
    std::vector<Piece> pieces_;
    std::vector<Sliced> sliced_;
    bool place_on_vectors_ = true; // (see set_placement_on_vectors)
    mutable std::unique_ptr<Transposed<T>> transposed_;
};

// COOMatrix<T> (src/COOMatrix.hpp): int64 global row and column ids, T entries, row-sorted, so the
// kernel piece of a row block (the preimage of its rows, src/COOMatrix.cpp:77-95) is one run of
// entries, handed over as `piece_entries` (first entry, count) per colour.
template <typename T> class COOMatrix : public AbstractMatrix<T> {
public:
    COOMatrix(const Context &ctx, const EqualPartition &range, int64_t domain_size, DeviceArray<int64_t> row,
              DeviceArray<int64_t> col, DeviceArray<T> val, std::vector<std::pair<int64_t, int64_t>> piece_entries)
        : ctx_(&ctx), range_(range), domain_(domain_size), row_(std::move(row)), col_(std::move(col)),
          val_(std::move(val)), entries_(std::move(piece_entries)) {}
    ~COOMatrix() override {
        if (binned_) lsk_coo_binned_destroy(binned_);
    }
    COOMatrix(COOMatrix &&o) noexcept
        : ctx_(o.ctx_), range_(std::move(o.range_)), domain_(o.domain_), row_(std::move(o.row_)), col_(std::move(o.col_)),
          val_(std::move(o.val_)), entries_(std::move(o.entries_)), transposed_(std::move(o.transposed_)), binned_(o.binned_) {
        o.binned_ = nullptr;
    }
    // The set-up-time binned copy (lsk_coo_binned_*) over all rows: for matrices whose columns have no
    // locality.  Products then go through it and may overwrite (beta = 0).  False: the library declined.
    bool bin(int block_cols = 0, int tile_rows = 0) {
        DeviceArray<uint32_t> row32(row_.size()), col32(col_.size());
        check(lsk_csr_compress_cols_u32(row_.size(), row_.get(), 0, range_.n - 1, row32.get(), ctx_->stream()));
        check(lsk_csr_compress_cols_u32(col_.size(), col_.get(), 0, domain_ - 1, col32.get(), ctx_->stream()));
        lsk_coo_binned_t *handle = nullptr;
        const int status = Abi<T>::coo_binned_create(val_.size(), row32.get(), col32.get(), val_.get(), 0, range_.n - 1,
                                                     domain_, block_cols, tile_rows, ctx_->stream(), &handle);
        if (status == LSK_ERR_UNSUPPORTED) return false;
        check(status);
        if (binned_) lsk_coo_binned_destroy(binned_);
        binned_ = handle;
        return true;
    }
    // After the entries were rewritten in place (same rows and columns): refresh the copies.
    void values_changed() {
        if (binned_) check(Abi<T>::coo_binned_update_values(binned_, val_.size(), val_.get(), ctx_->stream()));
        transposed_.reset();
    }
    T *values() { return val_.get(); }
    bool accumulates_only() const override { return binned_ == nullptr; }
    void matvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const override {
        if (binned_) {
            check(Abi<T>::coo_binned_spmv(binned_, src.data(), dst.data(), beta, ctx_->stream()));
            return;
        }
        if (beta != 1) throw std::invalid_argument("COO matvec accumulates (beta = 1)");
        for (int c = 0; c < range_.colors(); ++c) {
            if (range_.volume(c) <= 0 || entries_[c].second <= 0) continue;
            const int64_t k = entries_[c].first;
            check(Abi<T>::coo_spmv(entries_[c].second, row_.get() + k, col_.get() + k, val_.get() + k, src.data(), 0,
                                   domain_ - 1, dst.piece(c), range_.piece[c].first, range_.piece[c].second,
                                   ctx_->stream()));
        }
    }
    void rmatvec(PartitionedVector<T> &dst, const PartitionedVector<T> &src, int beta) const override {
        if (!(src.partition() == range_) || dst.partition().n != domain_)
            throw std::invalid_argument("matrix and vectors do not fit");
        if (!transposed_) {
            DeviceArray<uint32_t> row32(row_.size());
            check(lsk_csr_compress_cols_u32(row_.size(), row_.get(), 0, range_.n - 1, row32.get(), ctx_->stream()));
            transposed_ = std::make_unique<Transposed<T>>(*ctx_, col_.size(), col_.get(), row32.get(), val_.get(),
                                                          domain_, range_.n);
        }
        check(lsk_set_option("xcd_remap", 0));
        transposed_->apply(dst.data(), src.data(), beta);
    }

private:
    const Context *ctx_;
    EqualPartition range_;
    int64_t domain_;
    DeviceArray<int64_t> row_, col_;
    DeviceArray<T> val_;
    std::vector<std::pair<int64_t, int64_t>> entries_;
    mutable std::unique_ptr<Transposed<T>> transposed_;
    lsk_coo_binned_t *binned_ = nullptr;
};

// SquarePlanner<T> (src/SquarePlanner.hpp): the solution / right-hand-side / workspace vectors of
// one or more spaces, addressed by index (0 = solution, 1 = right-hand side, 2.. = workspace),
// and the matrices between the spaces.  Every operation runs over all spaces; dot adds the spaces'
// results (:300-329); matvec is zero fill + one accumulating product per matrix (:340-357), the
// first CSR product into a space overwriting instead of the zero fill (same result).
template <typename T> class SquarePlanner {
public:
    static constexpr int SOL = 0, RHS = 1;
    explicit SquarePlanner(const Context &ctx) : ctx_(&ctx), vectors_(2) {}
    const Context &get_context() const { return *ctx_; }
    int add_sol_vector(PartitionedVector<T> &v) { return add(v, SOL); }
    int add_rhs_vector(PartitionedVector<T> &v) { return add(v, RHS); }
    int get_num_spaces() const {
        if (vectors_[SOL].size() != vectors_[RHS].size())
            throw std::logic_error("every space needs one solution and one right-hand-side vector");
        return static_cast<int>(vectors_[SOL].size());
    }
    void add_row_partitioned_matrix(const AbstractMatrix<T> &m, int domain_index, int range_index) {
        const int n = get_num_spaces();
        if (domain_index < 0 || domain_index >= n || range_index < 0 || range_index >= n)
            throw std::out_of_range("matrix refers to a space the planner does not have");
        matrices_.push_back({&m, domain_index, range_index});
    }
    void allocate_workspace(int count) {
        const int n = get_num_spaces();
        size_t need = 0;
        for (int s = 0; s < n; ++s) need += static_cast<size_t>(vectors_[SOL][s]->partition().n) * sizeof(T);
        std::vector<void *> hold = room_in_front_of_the_workspace(need * static_cast<size_t>(count));
        // While blocks are held in front of the workspace, a vector that finds no room must not cost a matrix its
        // set-up-time copy: the blocks go instead, and the vector is tried again.
        OptionalMemory &optional = OptionalMemory::get();
        auto let_go = [&hold, &optional]() {
            if (hold.empty()) return false;
            for (void *p : hold) (void)lsk_free(p);
            hold.clear();
            --optional.held_back;
            return true;
        };
        if (!hold.empty()) ++optional.held_back;
        try {
            for (int i = 0; i < count; ++i) {
                std::vector<PartitionedVector<T> *> row;
                for (int s = 0; s < n; ++s) {
                    for (;;) {
                        try {
                            owned_.push_back(std::make_unique<PartitionedVector<T>>(
                                *ctx_, "workspace_" + std::to_string(vectors_.size() - 2) + "_" + std::to_string(s),
                                vectors_[SOL][s]->partition()));
                            break;
                        } catch (const std::exception &e) {
                            if (!no_room(e) || !let_go()) throw;
                            workspace_blocks_ = 0; // (the rest of the workspace lies where it would have gone anyway)
                        }
                    }
                    row.push_back(owned_.back().get());
                }
                vectors_.push_back(std::move(row));
            }
        } catch (...) {
            let_go();
            throw;
        }
        let_go();
    }
    // Where the workspace went (see room_in_front_of_the_workspace): blocks of a third of the device held in front
    // of it while it was allocated, and the milliseconds of the product timed behind 0, 1 and 2 such blocks.
    int workspace_blocks_in_front() const { return workspace_blocks_; }
    const std::vector<float> &workspace_product_ms() const { return workspace_ms_; }
    PartitionedVector<T> &get_vector(int index, int space) const {
        pending_.reset(); // the caller may write through the reference
        return *vectors_.at(index).at(space);
    }
    void zero_fill(int dst) {
        pending_.reset();
        for (auto *v : vectors_.at(dst)) v->zero_fill();
    }
    void copy(int dst, int src) {
        pending_.reset();
        for (size_t s = 0; s < vectors_.at(dst).size(); ++s) *vectors_[dst][s] = *vectors_.at(src)[s];
    }
    void scal(int dst, const std::vector<Scalar<T>> &alpha) {
        pending_.reset();
        for (auto *v : vectors_.at(dst)) v->scal(alpha);
    }
    // An update by a workspace vector is held back (see PartitionedVector): into the solution it may
    // share a pass with that vector's own update, into a workspace vector it may serve the dot
    // product that follows.  A product between workspace vectors may leave the shares of the dot
    // product of its result with its input (CG's p.Ap), which a dot(dst, src) that comes next, before
    // any other operation of the planner, only adds up.  set_fusion(false) launches every leaf as it
    // is asked for.
    void set_fusion(bool on) { fuse_ = on; }
    // Brings every vector's storage up to date and forgets what a fused leaf left for a call that may
    // follow: after it, what the next operation launches depends on nothing an earlier one left
    // behind.  The two ends of a recorded trace (Trace below).
    void settle() {
        pending_.reset();
        for (auto &row : vectors_)
            for (auto *v : row) v->settle_now();
    }
    // Whether two systems on one matrix are multiplied in one pass over it (same results either way).
    void set_pairing(bool on) { pair_ = on; }
    long paired_products() const { return paired_products_; } // how many times two products shared a pass
    void axpy(int dst, const std::vector<Scalar<T>> &alpha, int src) {
        pending_.reset();
        const bool defer = fuse_ && src >= 2 && dst != RHS && dst != src;
        for (size_t s = 0; s < vectors_.at(dst).size(); ++s) vectors_[dst][s]->axpy(alpha, *vectors_.at(src)[s], defer);
    }
    void xpay(int dst, const std::vector<Scalar<T>> &alpha, int src) {
        pending_.reset();
        for (size_t s = 0; s < vectors_.at(dst).size(); ++s) vectors_[dst][s]->xpay(alpha, *vectors_.at(src)[s]);
    }
    Scalar<T> dot(int v, int w) const {
        const std::optional<Pending> pending = std::move(pending_);
        pending_.reset();
        const bool ready = pending && ((pending->dst == v && pending->src == w) || (pending->dst == w && pending->src == v));
        auto one = [&](size_t s) {
            if (ready && pending->shares[s] > 0) {
                Scalar<T> out = Scalar<T>::uninitialised(*ctx_);
                check(Abi<T>::sum_partials(pending->shares[s], share_buffers_[s].get(), out.writable(), nullptr, 0,
                                           ctx_->stream()));
                return out;
            }
            return vectors_.at(v)[s]->dot(*vectors_.at(w)[s]);
        };
        Scalar<T> acc = one(0);
        for (size_t s = 1; s < vectors_[v].size(); ++s) acc = acc + one(s);
        return acc;
    }
    // k dot products with one vector in one pass over it per 8 vectors: out[j] = dot(vs[j], w), bit for
    // bit.  The results are k device scalars in a buffer of the planner that the next multi_dot
    // reuses (everything is ordered on one stream); `coefficients()` is that buffer, for multi_axpy.
    // The reference's planner has no such method (its GMRES takes its inner products one by one,
    // src/GMRESSolver.hpp:92-100).
    const T *multi_dot(const std::vector<int> &vs, int w) {
        pending_.reset();
        const int k = static_cast<int>(vs.size());
        if (k < 1 || k > 64) throw std::invalid_argument("multi_dot takes 1..64 vectors");
        const size_t spaces = vectors_.at(w).size();
        int colours = 1;
        for (auto *v : vectors_[w]) colours = std::max(colours, v->partition().colors());
        if (multi_out_.size() < k) multi_out_ = DeviceArray<T>(64);
        if (multi_rows_.size() < static_cast<int64_t>(64) * (colours + 1)) multi_rows_ = DeviceArray<T>(64 * (colours + 1));
        const size_t bytes = lsk_multi_dot_workspace_bytes(64);
        if (multi_workspace_.size() < static_cast<int64_t>(bytes)) multi_workspace_ = DeviceArray<unsigned char>(bytes);
        for (size_t s = 0; s < spaces; ++s) {
            std::vector<const PartitionedVector<T> *> v(k);
            for (int j = 0; j < k; ++j) v[j] = vectors_.at(vs[j])[s];
            // space 0 writes the result, the others are added to it in order, as dot() adds the spaces
            T *out = s == 0 ? multi_out_.get() : multi_rows_.get();
            vectors_[w][s]->multi_dot_into(v, out, multi_rows_.get() + 64, multi_workspace_.get(), bytes);
            if (s > 0)
                for (int j = 0; j < k; ++j)
                    check(Abi<T>::scalar_op(LSK_SOP_ADD, multi_out_.get() + j, out + j, multi_out_.get() + j, ctx_->stream()));
        }
        return multi_out_.get();
    }
    // dst <- dst -+ sum_j coef[j] vs[j] (coef: k device scalars, e.g. what multi_dot returned).
    void multi_axpy(int dst, const T *coef_dev, const std::vector<int> &vs, bool negate) {
        pending_.reset();
        for (size_t s = 0; s < vectors_.at(dst).size(); ++s) {
            std::vector<const PartitionedVector<T> *> x(vs.size());
            for (size_t j = 0; j < vs.size(); ++j) x[j] = vectors_.at(vs[j])[s];
            vectors_[dst][s]->multi_axpy(x, coef_dev, negate);
        }
    }
    void matvec(int dst, int src) {
        pending_.reset();
        const size_t spaces = vectors_.at(dst).size();
        std::vector<bool> touched(spaces, false);
        std::vector<int> writers(spaces, 0);
        for (const auto &e : matrices_) ++writers[e.range];
        Pending pending{dst, src, std::vector<int64_t>(spaces, 0)};
        bool any = false;
        std::vector<bool> done(matrices_.size(), false);
        for (size_t i = 0; i < matrices_.size(); ++i) {
            const auto &e = matrices_[i];
            if (done[i]) continue;
            // Two systems on ONE matrix (the reference's BenchmarkStencil, test/BenchmarkStencil.cpp:181-199):
            // both products in one pass over the matrix values where each alone produces its range space.
            if (pair_ && writers[e.range] == 1 && e.domain == e.range && !touched[e.range]) {
                size_t mate = matrices_.size();
                for (size_t j = i + 1; j < matrices_.size() && mate == matrices_.size(); ++j) {
                    const auto &f = matrices_[j];
                    if (!done[j] && f.matrix == e.matrix && f.range != e.range && writers[f.range] == 1 &&
                        f.domain == f.range && !touched[f.range])
                        mate = j;
                }
                if (mate < matrices_.size()) {
                    const auto &f = matrices_[mate];
                    const int64_t shares = e.matrix->dot_shares();
                    const bool with_dot = fuse_ && dst >= 2 && src >= 2 && dst != src && shares > 0;
                    if (with_dot) {
                        if (share_buffers_.size() < spaces) share_buffers_.resize(spaces);
                        for (int r : {e.range, f.range})
                            if (share_buffers_[r].size() < shares) share_buffers_[r] = DeviceArray<T>(shares);
                    }
                    if (e.matrix->matvec_pair(*vectors_[dst][e.range], *vectors_[dst][f.range], *vectors_.at(src)[e.domain],
                                              *vectors_[src][f.domain], with_dot ? vectors_[src][e.range] : nullptr,
                                              with_dot ? vectors_[src][f.range] : nullptr,
                                              with_dot ? share_buffers_[e.range].get() : nullptr,
                                              with_dot ? share_buffers_[f.range].get() : nullptr)) {
                        done[mate] = true;
                        ++paired_products_;
                        touched[e.range] = touched[f.range] = true;
                        if (with_dot) {
                            pending.shares[e.range] = pending.shares[f.range] = shares;
                            any = true;
                        }
                        continue;
                    }
                }
            }
            PartitionedVector<T> &y = *vectors_[dst][e.range];
            const int64_t shares = e.matrix->dot_shares();
            if (fuse_ && dst >= 2 && src >= 2 && dst != src && writers[e.range] == 1 && e.domain == e.range && shares > 0) {
                if (share_buffers_.size() < spaces) share_buffers_.resize(spaces);
                if (share_buffers_[e.range].size() < shares) share_buffers_[e.range] = DeviceArray<T>(shares);
                e.matrix->matvec_dot(y, *vectors_.at(src)[e.domain], *vectors_[src][e.range], share_buffers_[e.range].get());
                pending.shares[e.range] = shares;
                any = true;
            } else {
                if (!touched[e.range] && e.matrix->accumulates_only()) y.zero_fill();
                e.matrix->matvec(y, *vectors_.at(src)[e.domain], touched[e.range] || e.matrix->accumulates_only() ? 1 : 0);
            }
            touched[e.range] = true;
        }
        for (size_t s = 0; s < touched.size(); ++s)
            if (!touched[s]) vectors_[dst][s]->zero_fill();
        if (any) pending_ = std::move(pending);
    }
    // dst <- A^T src over all matrices: `src` is read in every matrix's range space, `dst` written in
    // its domain space (the reference's planner has no such method).
    void rmatvec(int dst, int src) {
        pending_.reset();
        std::vector<bool> touched(vectors_.at(dst).size(), false);
        for (const auto &e : matrices_) {
            e.matrix->rmatvec(*vectors_[dst][e.domain], *vectors_.at(src)[e.range], touched[e.domain] ? 1 : 0);
            touched[e.domain] = true;
        }
        for (size_t s = 0; s < touched.size(); ++s)
            if (!touched[s]) vectors_[dst][s]->zero_fill();
    }

private:
    struct Entry {
        const AbstractMatrix<T> *matrix;
        int domain, range;
    };
    struct Pending { // shares of dot(vector dst, vector src) left by the last matvec, per space (0: none)
        int dst, src;
        std::vector<int64_t> shares;
    };
    int add(PartitionedVector<T> &v, int index) {
        vectors_[index].push_back(&v);
        return static_cast<int>(vectors_[index].size()) - 1;
    }
    const Context *ctx_;
    bool fuse_ = true;
    bool pair_ = true;
    long paired_products_ = 0;
    std::vector<std::vector<PartitionedVector<T> *>> vectors_;
    std::vector<std::unique_ptr<PartitionedVector<T>>> owned_;
    // Where in HBM the workspace vectors go.  The 288 GB of an MI355X answer in three parts of 96 GB (the three
    // ranks of its 12-high stacks): a read and a write stream are faster in two different parts (a copy: 6.2
    // against 5.7 TB/s), so the vectors a product writes are best kept out of the part the values of a large
    // sliced-pattern copy lie in (CG on the 27-point 512^3 matrix: 161 -> 170 it/s).  No interface says where an
    // allocation lands; memory is handed out from the bottom up, so a block of a third of the device allocated
    // first pushes what follows into the next part.  Nothing is assumed: one product of every such matrix is
    // timed into a vector here, into one behind such a block and into one behind two, and the workspace goes one
    // place deeper for as long as that is faster by 1.5 % (ties go to the nearer place).  Returns the allocations
    // to hold while the workspace is allocated (fewer when the workspace would not fit behind them).  Skipped
    // without a sliced copy of at least kPlaceMinBytes of values, with LSK_PLACE_VECTORS=0, or without room for
    // the blocks.
    std::vector<void *> room_in_front_of_the_workspace(size_t workspace_bytes) {
        std::vector<void *> hold;
        workspace_blocks_ = 0;
        workspace_ms_.clear();
        const char *off = std::getenv("LSK_PLACE_VECTORS");
        if (off && off[0] == '0') return hold;
        auto try_malloc = [](size_t bytes) -> void * {
            void *p = nullptr;
            return lsk_malloc(&p, bytes ? bytes : 1) == LSK_OK ? p : nullptr;
        };
        // one product of every matrix with a large sliced piece, into the same vector: their sum is what is compared
        int64_t n_dom = 0, n_rng = 0;
        for (const auto &e : matrices_) {
            n_dom = std::max<int64_t>(n_dom, vectors_[SOL][e.domain]->partition().n);
            n_rng = std::max<int64_t>(n_rng, vectors_[SOL][e.range]->partition().n);
        }
        size_t free_b = 0, total_b = 0;
        if (matrices_.empty() || lsk_mem_info(&free_b, &total_b) != LSK_OK) return hold;
        const size_t third = total_b / 3;
        // kept: input 0, output 0, block 1, input 1, output 1, block 2, input 2, output 2 -- at every place the product
        // runs from a first vector into a second one, as the solver's products will from the first workspace vector
        // into the second (what is freed here is what the workspace is then allocated from)
        std::vector<void *> kept;
        std::vector<float> times;
        try {
            for (int depth = 0; depth < 3; ++depth) {
                const size_t before = kept.size();
                if (depth) {
                    if (lsk_mem_info(&free_b, &total_b) != LSK_OK || free_b < third + (size_t(4) << 30)) break;
                    void *block = try_malloc(third);
                    if (!block) break;
                    kept.push_back(block);
                }
                void *x = try_malloc(static_cast<size_t>(n_dom) * sizeof(T));
                if (x) kept.push_back(x);
                void *y = x ? try_malloc(static_cast<size_t>(n_rng) * sizeof(T)) : nullptr;
                if (!y) { // a place without all of its vectors
                    while (kept.size() > before) {
                        (void)lsk_free(kept.back());
                        kept.pop_back();
                    }
                    break;
                }
                kept.push_back(y);
                check(Abi<T>::fill(n_dom, T(0), static_cast<T *>(x), ctx_->stream()));
                float ms = 0.0f;
                for (const auto &e : matrices_) ms += e.matrix->time_large_product(static_cast<const T *>(x), static_cast<T *>(y));
                if (ms <= 0.0f) break; // no matrix has a large sliced piece: nothing to place by
                times.push_back(ms);
            }
        } catch (...) {
            for (void *p : kept) (void)lsk_free(p);
            throw;
        }
        if (times.empty()) {
            ctx_->synchronize();
            for (void *p : kept) (void)lsk_free(p);
            return hold;
        }
        size_t best = 0;
        for (size_t d = 1; d < times.size(); ++d) // every step deeper must pay 1.5 %: ties go to the nearer place
            if (times[d] < 0.985f * times[best]) best = d;
        for (size_t k = 3 * best; k < kept.size(); ++k) (void)lsk_free(kept[k]); // the chosen place and all behind it
        kept.resize(3 * best);
        // the workspace itself must fit behind what is held: else the deepest block goes, and the next
        while (best > 0 && (lsk_mem_info(&free_b, &total_b) != LSK_OK || free_b < workspace_bytes + (size_t(2) << 30))) {
            for (int k = 0; k < 3; ++k) {
                (void)lsk_free(kept.back());
                kept.pop_back();
            }
            --best;
        }
        workspace_blocks_ = static_cast<int>(best);
        workspace_ms_ = times;
        return kept; // what lies in front of the chosen place stays until the workspace is there
    }
    int workspace_blocks_ = 0;
    std::vector<float> workspace_ms_;

    std::vector<Entry> matrices_;
    mutable std::optional<Pending> pending_;
    mutable std::vector<DeviceArray<T>> share_buffers_;
    DeviceArray<T> multi_out_, multi_rows_;
    DeviceArray<unsigned char> multi_workspace_;
};

// CGSolver<T> (src/CGSolver.hpp:13-57).
template <typename T> class CGSolver {
public:
    static constexpr int SOL = 0, RHS = 1, P = 2, Q = 3, R = 4;
    std::vector<Scalar<T>> residual_norm_squared;
    explicit CGSolver(SquarePlanner<T> &planner) : pl(planner), negative_one(planner.get_context(), T(-1)) {
        pl.allocate_workspace(3);
        pl.copy(P, RHS);
        pl.copy(R, RHS);
        residual_norm_squared.push_back(pl.dot(R, R));
    }
    void step() {
        pl.matvec(Q, P);
        const Scalar<T> p_norm = pl.dot(P, Q);
        const Scalar<T> r_norm2_old = residual_norm_squared.back();
        pl.axpy(SOL, {r_norm2_old, p_norm}, P);
        pl.axpy(R, {negative_one, r_norm2_old, p_norm}, Q);
        const Scalar<T> r_norm2_new = pl.dot(R, R);
        residual_norm_squared.push_back(r_norm2_new);
        pl.xpay(P, {r_norm2_new, r_norm2_old}, R);
    }

private:
    SquarePlanner<T> &pl;
    Scalar<T> negative_one;
};

// BiCGStabSolver<T> (src/BiCGStabSolver.hpp:13-84).
template <typename T> class BiCGStabSolver {
public:
    static constexpr int SOL = 0, RHS = 1, P = 2, R = 3, R_TILDE = 4, U = 5, V = 6;
    std::vector<Scalar<T>> rho, alpha, omega;
    explicit BiCGStabSolver(SquarePlanner<T> &planner)
        : pl(planner), negative_one(planner.get_context(), T(-1)), zero(planner.get_context(), T(0)),
          one(planner.get_context(), T(1)) {
        pl.allocate_workspace(5);
        pl.copy(R, RHS);
        pl.copy(R_TILDE, RHS);
        rho.push_back(one);
        alpha.push_back(zero);
        omega.push_back(one);
        pl.zero_fill(P);
        pl.zero_fill(V);
    }
    void step() {
        const Scalar<T> rho_new = pl.dot(R, R_TILDE);
        const Scalar<T> beta = (rho_new / rho.back()) * (alpha.back() / omega.back());
        rho.push_back(rho_new);
        pl.axpy(P, {-omega.back()}, V);
        pl.xpay(P, {beta}, R);
        pl.matvec(V, P);
        const Scalar<T> temp = pl.dot(R_TILDE, V);
        pl.axpy(R, {negative_one, rho.back(), temp}, V);
        alpha.push_back(rho.back() / temp);
        pl.matvec(U, R);
        const Scalar<T> r_anorm2 = pl.dot(R, U);
        const Scalar<T> u_norm2 = pl.dot(U, U);
        omega.push_back(r_anorm2 / u_norm2);
        pl.axpy(SOL, {alpha.back()}, P);
        pl.axpy(SOL, {omega.back()}, R);
        pl.axpy(R, {-omega.back()}, U);
    }

private:
    SquarePlanner<T> &pl;
    Scalar<T> negative_one, zero, one;
};

// GMRESSolver<T> (src/GMRESSolver.hpp:15-129).  As in the reference the least-squares solve is not
// there yet: the solution update takes DummyTask's coefficient 1 for every basis vector (:109-126).
template <typename T> class GMRESSolver {
public:
    static constexpr int SOL = 0, RHS = 1;
    // classical = false: the reference's modified Gram-Schmidt sweep (:92-100).  true: every inner
    // product of a step from the same vector (SquarePlanner::multi_dot / multi_axpy): about half the
    // vector traffic, different arithmetic -- an option, never the default.
    GMRESSolver(SquarePlanner<T> &planner, int restart_length, bool classical = false)
        : pl(planner), restart(restart_length), classical_(classical), negative_one(planner.get_context(), T(-1)),
          dummy_coefficient(planner.get_context(), T(1)) {
        // (one multi_dot call takes at most 64 vectors, and all inner products of a sweep share one buffer)
        if (classical && restart_length > 64)
            throw std::invalid_argument("GMRESSolver: the classical Gram-Schmidt option takes a restart length of at most 64");
        pl.allocate_workspace(restart + 1);
    }
    static int krylov_basis(int i) { return i + 2; }
    void step() {
        pl.matvec(krylov_basis(0), SOL);
        pl.xpay(krylov_basis(0), {negative_one}, RHS);
        pl.scal(krylov_basis(0), {pl.dot(krylov_basis(0), krylov_basis(0)).rsqrt()});
        for (int j = 0; j < restart; ++j) {
            pl.matvec(krylov_basis(j + 1), krylov_basis(j));
            if (classical_) {
                std::vector<int> basis(j + 1);
                for (int k = 0; k <= j; ++k) basis[k] = krylov_basis(k);
                const T *h = pl.multi_dot(basis, krylov_basis(j + 1));
                pl.multi_axpy(krylov_basis(j + 1), h, basis, /*negate=*/true);
            } else {
                for (int k = 0; k <= j; ++k) {
                    const Scalar<T> h = pl.dot(krylov_basis(k), krylov_basis(j + 1));
                    pl.axpy(krylov_basis(j + 1), {-h}, krylov_basis(k));
                }
            }
            const Scalar<T> d = pl.dot(krylov_basis(j + 1), krylov_basis(j + 1));
            if (j + 1 < restart) pl.scal(krylov_basis(j + 1), {d.rsqrt()});
        }
        for (int j = 0; j < restart; ++j) pl.axpy(SOL, {dummy_coefficient}, krylov_basis(j));
    }

private:
    SquarePlanner<T> &pl;
    int restart;
    bool classical_;
    Scalar<T> negative_one, dummy_coefficient;
};

// The scalars a step reads from the step before it (each solver's history vectors).
template <typename T> std::vector<std::vector<Scalar<T>> *> carried_scalars(CGSolver<T> &s) { return {&s.residual_norm_squared}; }
template <typename T> std::vector<std::vector<Scalar<T>> *> carried_scalars(BiCGStabSolver<T> &s) { return {&s.rho, &s.alpha, &s.omega}; }
template <typename T> std::vector<std::vector<Scalar<T>> *> carried_scalars(GMRESSolver<T> &) { return {}; }

// `steps` solver steps recorded once and replayed with one call each time: what rt->begin_trace /
// rt->end_trace around the steps are in the reference's benchmark (test/BenchmarkStencil.cpp:219-222),
// as a HIP graph (lsk_stream_begin_capture / lsk_trace_replay).  Recording executes nothing; every
// replay() is `steps` steps.  The recording starts and ends at SquarePlanner::settle(), and ends by
// copying the newest value of every scalar a step reads from the step before (carried_scalars) into
// the slot the first recorded step read it from, so a replay continues from the one before.  The
// solver's history vectors grow by `steps` entries when the trace is recorded; after a replay the last
// entry holds the newest value, and so does the entry the recording started from (the history of
// replayed steps is not kept).  Run the solver a few
// steps before recording (a kernel's first launch on a device is not recordable); the vectors, the
// matrix and the solver must stay as they were for as long as the trace is replayed.
template <typename T, typename Solver> class Trace {
public:
    Trace(const Context &ctx, SquarePlanner<T> &planner, Solver &solver, long steps = 1) : ctx_(&ctx) {
        OptionalMemory::get().recordings += 1; // (the recording refers to the matrices' copies: they stay)
        planner.settle();
        ctx.synchronize();
        const auto carried = carried_scalars<T>(solver);
        std::vector<Scalar<T>> before;
        for (auto *history : carried) before.push_back(history->back());
        ctx.reserve_slots(static_cast<size_t>(steps) * kSlotsPerStep); // (slots released while recording are not reused)
        ctx.park_released_slots(&parked_);
        check(lsk_stream_begin_capture(ctx.stream()));
        int status = LSK_OK;
        try {
            for (long i = 0; i < steps; ++i) solver.step();
            planner.settle();
            for (size_t i = 0; i < carried.size(); ++i)
                check(Abi<T>::copy(1, carried[i]->back().get_future(), before[i].writable(), ctx.stream()));
        } catch (...) {
            lsk_trace_t *spoiled = nullptr;
            (void)lsk_stream_end_capture(ctx.stream(), &spoiled);
            lsk_trace_destroy(spoiled);
            ctx.park_released_slots(nullptr);
            ctx.return_slots(parked_);
            throw;
        }
        status = lsk_stream_end_capture(ctx.stream(), &trace_);
        ctx.park_released_slots(nullptr);
        keep_ = std::move(before);
        check(status);
    }
    Trace(const Trace &) = delete;
    Trace &operator=(const Trace &) = delete;
    ~Trace() {
        OptionalMemory::get().recordings -= 1;
        lsk_stream_synchronize(ctx_->stream());
        lsk_trace_destroy(trace_);
        ctx_->return_slots(parked_);
    }
    void replay() { check(lsk_trace_replay(trace_, ctx_->stream())); }
    int64_t nodes() const {
        int64_t out[2] = {0, 0};
        check(lsk_trace_info(trace_, out));
        return out[0];
    }

private:
    static constexpr size_t kSlotsPerStep = 4096; // more scalars than a step of any of the solvers makes (GMRES(64): ~2300)
    const Context *ctx_;
    lsk_trace_t *trace_ = nullptr;
    std::vector<void *> parked_;
    std::vector<Scalar<T>> keep_; // the slots the first recorded step reads
};

} // namespace LegionSolversHost
