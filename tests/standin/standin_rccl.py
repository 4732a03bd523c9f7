"""Test stand-in for the RCCL layer of liblsk (lsk_comm_*, lsk_allreduce_sum_*, lsk_halo_exchange_*), for machines
with ONE GPU: RCCL refuses two ranks on one device, so everything ABOVE those six entry points -- comm.LskTransport
(bootstrap over the store, side stream and events, self-check), Communicator's collective decision, the planner's
overlapped halo exchange, bench.py's N > 1 legs and its transport fields -- could otherwise never run with more
than one rank here.  install() replaces exactly those entry points, in this process, by functions that move the
same bytes between the ranks over gloo through host memory, synchronously on the stream they were given.  Test
infrastructure only: enabled by LSK_TEST_STANDIN_RCCL=1 from tests/, never by the package."""
import ctypes
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from legionsolvers_amd import _lib

_state = {"rank": None, "world": None, "calls": {"allreduce": 0, "exchange": 0}}
_real_call = _lib.call
_NP = {"f32": np.float32, "f64": np.float64}


def _value(x):
    return x.value if hasattr(x, "value") else x


def _to_host(ptr, count, npdt, stream):
    host = np.empty(int(count), dtype=npdt)
    if count:
        _real_call("lsk_memcpy_d2h", host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(_value(ptr)), host.nbytes, stream)
    return host


def _to_device(ptr, host, stream):
    if host.size:
        _real_call("lsk_memcpy_h2d", ctypes.c_void_p(_value(ptr)), host.ctypes.data_as(ctypes.c_void_p), host.nbytes, stream)


def _unique_id(buf):
    tag = b"stand-in communicator id"
    ctypes.memmove(buf, tag + bytes(128 - len(tag)), 128)


def _init_rank(handle_ref, world, rank, uid):
    if os.environ.get("LSK_TEST_STANDIN_INIT_HANGS") == "1":   # a communicator that never comes up
        time.sleep(10 ** 6)
    assert bytes(uid.raw[:8]) == b"stand-in", "every rank must present the id rank 0 published"
    _state.update(rank=int(rank), world=int(world))
    assert dist.get_world_size() == _state["world"] and dist.get_rank() == _state["rank"]
    # (the handle stays null: nothing of the real library is ever given it)


def _comm_rank(handle, rank_ref, n_ref):
    rank_ref._obj.value, n_ref._obj.value = _state["rank"], _state["world"]


def _allreduce(suf, handle, ptr, count, stream):
    host = _to_host(ptr, count, _NP[suf], stream)
    _real_call("lsk_stream_synchronize", stream)
    t = torch.from_numpy(host)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    _to_device(ptr, host, stream)
    _real_call("lsk_stream_synchronize", stream)
    _state["calls"]["allreduce"] += 1


def _exchange(suf, handle, ns, speer, sptr, scount, nr, rpeer, rptr, rcount, stream):
    npdt = _NP[suf]
    out = [_to_host(sptr[i], scount[i], npdt, stream) for i in range(ns)]
    _real_call("lsk_stream_synchronize", stream)
    landing = [np.empty(int(rcount[i]), dtype=npdt) for i in range(nr)]
    me = _state["rank"]
    works = []
    # a transfer to the calling rank itself pairs the i-th such send with the i-th such receive (RCCL's rule)
    own_sends = [i for i in range(ns) if speer[i] == me]
    own_recvs = [i for i in range(nr) if rpeer[i] == me]
    for i, j in zip(own_sends, own_recvs):
        landing[j][:] = out[i]
    for i in range(nr):
        if rpeer[i] != me and landing[i].size:
            works.append(dist.irecv(torch.from_numpy(landing[i]), src=int(rpeer[i])))
    for i in range(ns):
        if speer[i] != me and out[i].size:
            works.append(dist.isend(torch.from_numpy(out[i]), dst=int(speer[i])))
    for w in works:
        w.wait()
    for i in range(nr):
        _to_device(rptr[i], landing[i], stream)
    _real_call("lsk_stream_synchronize", stream)
    _state["calls"]["exchange"] += 1


def _call(name, *args):
    if name == "lsk_comm_unique_id":
        return _unique_id(*args)
    if name == "lsk_comm_init_rank":
        return _init_rank(*args)
    if name == "lsk_comm_rank":
        return _comm_rank(*args)
    if name.startswith("lsk_allreduce_sum_"):
        return _allreduce(name[-3:], *args)
    if name.startswith("lsk_halo_exchange_"):
        return _exchange(name[-3:], *args)
    if name.startswith("lsk_allgather_") or name in ("lsk_comm_destroy", "lsk_comm_abort"):
        raise AssertionError(f"{name}: not expected under the stand-in")
    return _real_call(name, *args)


def install():
    _lib.call = _call


def calls():
    return dict(_state["calls"])
