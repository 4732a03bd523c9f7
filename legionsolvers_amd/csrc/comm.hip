// RCCL layer: the dot all-reduce, the halo x-gather and the x all-gather, enqueued on the
// caller's stream.  The reference has no collective call sites (Legion moves the data:
// src/PartitionedVector.cpp:353-357, src/CSRMatrix.cpp:202-211); this is the mechanism a
// GPU task uses instead when one communicator spans the MI355Xs of a node.
#include <rccl/rccl.h>

#include <cstring>

#include "lsk_common.hpp"

static_assert(LSK_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct lsk_comm {
    ncclComm_t comm;
    int rank;
    int nranks;
};

namespace {

int nccl_fail(ncclResult_t status, const char *expr, const char *file, int line) {
    return lsk::fail(LSK_ERR_COMM, "RCCL failure with status code %d (%s) in %s at %s:%d",
                     static_cast<int>(status), ncclGetErrorString(status), expr, file, line);
}

#define LSK_NCCL(expr)                                                                 \
    do {                                                                               \
        ncclResult_t lsk_nccl_status_ = (expr);                                        \
        if (lsk_nccl_status_ != ncclSuccess)                                           \
            return nccl_fail(lsk_nccl_status_, #expr, __FILE__, __LINE__);             \
    } while (false)

template <typename T> struct NcclType;
template <> struct NcclType<float> { static constexpr ncclDataType_t value = ncclFloat32; };
template <> struct NcclType<double> { static constexpr ncclDataType_t value = ncclFloat64; };

template <typename T>
int allreduce_sum(lsk_comm_t c, T *buf, int count, lsk_stream_t stream) {
    if (!c) return lsk::fail(LSK_ERR_ARG, "lsk_allreduce_sum: null communicator");
    if (count < 0) return lsk::fail(LSK_ERR_ARG, "lsk_allreduce_sum: negative count");
    if (count == 0) return LSK_OK;
    if (!buf) return lsk::fail(LSK_ERR_ARG, "lsk_allreduce_sum: null buffer");
    // A sum over one rank changes nothing; option "comm.skip_single_rank" = 0 enqueues the RCCL
    // call all the same (what a one-GPU machine can exercise of this path).
    if (c->nranks == 1 && lsk::options().comm_skip_single_rank) return LSK_OK;
    LSK_NCCL(ncclAllReduce(buf, buf, size_t(count), NcclType<T>::value, ncclSum, c->comm,
                           lsk::as_stream(stream)));
    return LSK_OK;
}

template <typename T>
int halo_exchange(lsk_comm_t c, int nsend, const int *send_peer, const T *const *send_ptr,
                  const int64_t *send_count, int nrecv, const int *recv_peer, T *const *recv_ptr,
                  const int64_t *recv_count, lsk_stream_t stream) {
    if (!c) return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: null communicator");
    if (nsend < 0 || nrecv < 0) return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: negative count");
    if ((nsend && (!send_peer || !send_ptr || !send_count)) ||
        (nrecv && (!recv_peer || !recv_ptr || !recv_count)))
        return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: null list");
    // (a transfer to the calling rank itself is allowed: RCCL pairs a send and a receive of one rank
    // inside a group, in the order they were posted, and moves the data with a copy on the device)
    for (int i = 0; i < nsend; ++i)
        if (send_peer[i] < 0 || send_peer[i] >= c->nranks || send_count[i] < 0 ||
            (send_count[i] > 0 && !send_ptr[i]))
            return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: bad send entry %d", i);
    for (int i = 0; i < nrecv; ++i)
        if (recv_peer[i] < 0 || recv_peer[i] >= c->nranks || recv_count[i] < 0 ||
            (recv_count[i] > 0 && !recv_ptr[i]))
            return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: bad recv entry %d", i);
    hipStream_t s = lsk::as_stream(stream);
    if (c->nranks == 1 && lsk::options().comm_skip_single_rank) {
        // one rank: every transfer is to itself; the i-th send lands in the i-th receive
        if (nsend != nrecv) return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: %d sends to itself, %d receives", nsend, nrecv);
        for (int i = 0; i < nsend; ++i) {
            if (send_count[i] != recv_count[i])
                return lsk::fail(LSK_ERR_ARG, "lsk_halo_exchange: transfer %d to itself sends %lld and receives %lld elements",
                                 i, (long long)send_count[i], (long long)recv_count[i]);
            if (send_count[i] > 0 && send_ptr[i] != recv_ptr[i])
                LSK_HIP(hipMemcpyAsync(recv_ptr[i], send_ptr[i], size_t(send_count[i]) * sizeof(T),
                                       hipMemcpyDeviceToDevice, s));
        }
        return LSK_OK;
    }
    LSK_NCCL(ncclGroupStart());
    // From here on the group is open: a failing call is remembered, the group is closed whatever
    // happened (a group left open would swallow every later RCCL call of this thread), and the
    // first failure is what the caller gets.
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int i = 0; i < nsend && first == ncclSuccess; ++i)
        if (send_count[i] > 0) {
            first = ncclSend(send_ptr[i], size_t(send_count[i]), NcclType<T>::value, send_peer[i], c->comm, s);
            what = "ncclSend";
        }
    for (int i = 0; i < nrecv && first == ncclSuccess; ++i)
        if (recv_count[i] > 0) {
            first = ncclRecv(recv_ptr[i], size_t(recv_count[i]), NcclType<T>::value, recv_peer[i], c->comm, s);
            what = "ncclRecv";
        }
    const ncclResult_t closed = ncclGroupEnd();
    if (first != ncclSuccess) return nccl_fail(first, what, __FILE__, __LINE__);
    if (closed != ncclSuccess) return nccl_fail(closed, "ncclGroupEnd", __FILE__, __LINE__);
    return LSK_OK;
}

template <typename T>
int allgather(lsk_comm_t c, const T *send, T *recv, int64_t count, lsk_stream_t stream) {
    if (!c) return lsk::fail(LSK_ERR_ARG, "lsk_allgather: null communicator");
    if (count < 0) return lsk::fail(LSK_ERR_ARG, "lsk_allgather: negative count");
    if (count == 0) return LSK_OK;
    if (!send || !recv) return lsk::fail(LSK_ERR_ARG, "lsk_allgather: null buffer");
    if (c->nranks == 1 && lsk::options().comm_skip_single_rank) { // one rank: recv <- send
        if (send != recv)
            LSK_HIP(hipMemcpyAsync(recv, send, size_t(count) * sizeof(T), hipMemcpyDeviceToDevice,
                                   lsk::as_stream(stream)));
        return LSK_OK;
    }
    LSK_NCCL(ncclAllGather(send, recv, size_t(count), NcclType<T>::value, c->comm,
                           lsk::as_stream(stream)));
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_comm_unique_id(void *id_out) {
    if (!id_out) return lsk::fail(LSK_ERR_ARG, "lsk_comm_unique_id: null output");
    ncclUniqueId id;
    LSK_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return LSK_OK;
}

int lsk_comm_init_rank(lsk_comm_t *comm, int nranks, int rank, const void *id) {
    if (!comm || !id) return lsk::fail(LSK_ERR_ARG, "lsk_comm_init_rank: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return lsk::fail(LSK_ERR_ARG, "lsk_comm_init_rank: rank %d of %d", rank, nranks);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    LSK_NCCL(ncclCommInitRank(&c, nranks, uid, rank));
    *comm = new lsk_comm{c, rank, nranks};
    return LSK_OK;
}

int lsk_comm_destroy(lsk_comm_t comm) {
    if (!comm) return LSK_OK;
    ncclResult_t st = ncclCommDestroy(comm->comm);
    delete comm;
    if (st != ncclSuccess) return nccl_fail(st, "ncclCommDestroy", __FILE__, __LINE__);
    return LSK_OK;
}

int lsk_comm_abort(lsk_comm_t comm) {
    if (!comm) return LSK_OK;
    ncclResult_t st = ncclCommAbort(comm->comm);
    delete comm;
    if (st != ncclSuccess) return nccl_fail(st, "ncclCommAbort", __FILE__, __LINE__);
    return LSK_OK;
}

int lsk_comm_rank(lsk_comm_t comm, int *rank, int *nranks) {
    if (!comm) return lsk::fail(LSK_ERR_ARG, "lsk_comm_rank: null communicator");
    if (rank) *rank = comm->rank;
    if (nranks) *nranks = comm->nranks;
    return LSK_OK;
}

int lsk_allreduce_sum_f32(lsk_comm_t comm, float *buf, int count, lsk_stream_t stream) {
    return allreduce_sum<float>(comm, buf, count, stream);
}
int lsk_allreduce_sum_f64(lsk_comm_t comm, double *buf, int count, lsk_stream_t stream) {
    return allreduce_sum<double>(comm, buf, count, stream);
}

int lsk_halo_exchange_f32(lsk_comm_t comm, int nsend, const int *send_peer,
                          const float *const *send_ptr, const int64_t *send_count, int nrecv,
                          const int *recv_peer, float *const *recv_ptr, const int64_t *recv_count,
                          lsk_stream_t stream) {
    return halo_exchange<float>(comm, nsend, send_peer, send_ptr, send_count, nrecv, recv_peer,
                                recv_ptr, recv_count, stream);
}
int lsk_halo_exchange_f64(lsk_comm_t comm, int nsend, const int *send_peer,
                          const double *const *send_ptr, const int64_t *send_count, int nrecv,
                          const int *recv_peer, double *const *recv_ptr, const int64_t *recv_count,
                          lsk_stream_t stream) {
    return halo_exchange<double>(comm, nsend, send_peer, send_ptr, send_count, nrecv, recv_peer,
                                 recv_ptr, recv_count, stream);
}

int lsk_allgather_f32(lsk_comm_t comm, const float *send, float *recv, int64_t count,
                      lsk_stream_t stream) {
    return allgather<float>(comm, send, recv, count, stream);
}
int lsk_allgather_f64(lsk_comm_t comm, const double *send, double *recv, int64_t count,
                      lsk_stream_t stream) {
    return allgather<double>(comm, send, recv, count, stream);
}

} // extern "C"
