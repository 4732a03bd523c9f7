"""Cross-rank data movement of the Krylov path: the scalar all-reduce behind `dot` and the
halo gather of `x` behind `matvec`.

In the reference both are implicit in Legion (a REDOP_SUM future map,
src/PartitionedVector.cpp:353-357; a READ_ONLY requirement on the ghost partition,
src/CSRMatrix.cpp:202-211).  Here one process drives one MI355X.  Two transports carry the
per-step traffic:

  "lsk"    the library's own RCCL layer (lsk_comm_*, lsk_allreduce_sum_*, lsk_halo_exchange_*
           of include/lsk.h) -- the calls a Legion GPU task would make.  The communicator is
           bootstrapped from the torch.distributed store (rank 0 makes the unique id, the
           others read it); all-reduces are enqueued on the current stream, halo exchanges on
           a second stream of the same priority, ordered against the current one with events,
           so the rows that read no ghost cell are multiplied while the halo travels.  The
           default when the ranks own one GPU each (process group backend "nccl").
  "torch"  torch.distributed itself: backend "nccl" is RCCL as well, "gloo" runs the same code
           on CPU tensors for tests.  LSK_TRANSPORT=torch selects it, and it is what a failed
           bootstrap of the first falls back to (every rank together) -- also one that is not up
           after LSK_COMM_TIMEOUT seconds (default 180), whose thread is then left behind.
           LSK_TRANSPORT=lsk insists on the first (a failure is raised); prefer-lsk tries it and
           falls back, which is also what happens by default where the process group is nccl.

Set-up-time traffic (host objects, barriers, timing brackets) always goes through
torch.distributed.  The plan arithmetic (who sends which interval to whom) is pure host code.

Rehearsal transport: when the group's backend is "gloo" and the tensors live on a GPU (several
ranks sharing one card on a one-GPU machine), every transfer is staged through host memory with
synchronous copies, because gloo is only ever handed CPU tensors here.  It exists to run the
multi-rank path on real kernels where RCCL cannot (RCCL refuses two ranks on one device); it is
not a data path for production.
"""
from __future__ import annotations

import ctypes
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

UNIQUE_ID_BYTES = 128   # LSK_UNIQUE_ID_BYTES of include/lsk.h


@dataclass(frozen=True)
class Transfer:
    """`count` elements starting at global index `start`, exchanged with rank `peer`."""
    peer: int
    start: int
    count: int


@dataclass
class HaloPlan:
    """For one (matrix, rank): the pieces of x to receive into the halo and to send away."""
    own: Tuple[int, int]      # interval this rank owns
    ghost: Tuple[int, int]    # interval its kernel pieces read (superset of what is needed)
    recvs: List[Transfer]
    sends: List[Transfer]

    @property
    def pad_lo(self) -> int:
        return max(self.own[0] - self.ghost[0], 0) if self.ghost[1] >= self.ghost[0] else 0

    @property
    def pad_hi(self) -> int:
        return max(self.ghost[1] - self.own[1], 0) if self.ghost[1] >= self.ghost[0] else 0


def _intersect(a: Tuple[int, int], b: Tuple[int, int]) -> Tuple[int, int]:
    return max(a[0], b[0]), min(a[1], b[1])


def build_halo_plan(rank: int, owns: Sequence[Tuple[int, int]],
                    ghosts: Sequence[Tuple[int, int]]) -> HaloPlan:
    """owns[q] / ghosts[q]: owned and ghost intervals of every rank q (inclusive; an empty
    interval has hi < lo).  Rank `rank` receives from q the part of its ghost interval that q
    owns, and sends to q the part of q's ghost interval that it owns itself."""
    me_own, me_ghost = owns[rank], ghosts[rank]
    recvs, sends = [], []
    for q in range(len(owns)):
        if q == rank:
            continue
        lo, hi = _intersect(me_ghost, owns[q])
        if hi >= lo:
            recvs.append(Transfer(q, lo, hi - lo + 1))
        lo, hi = _intersect(ghosts[q], me_own)
        if hi >= lo:
            sends.append(Transfer(q, lo, hi - lo + 1))
    return HaloPlan(me_own, me_ghost, recvs, sends)


def bootstrap_unique_id(rank: int, make_id, store, key: str) -> bytes:
    """The RCCL unique id every rank of a new communicator must present: rank 0 makes it
    (`make_id()` -> UNIQUE_ID_BYTES bytes) and publishes it under `key` in `store` (anything with
    the set / get of a torch.distributed store; get blocks until the key exists), the others
    read it."""
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != UNIQUE_ID_BYTES:
            raise ValueError(f"a unique id has {UNIQUE_ID_BYTES} bytes, got {len(uid)}")
        store.set(key, uid)
        return uid
    uid = bytes(store.get(key))
    if len(uid) != UNIQUE_ID_BYTES:
        raise RuntimeError(f"unique id under {key!r} has {len(uid)} bytes")
    return uid


class LskTransport:
    """One RCCL communicator made through the C ABI, and the streams and events that order its
    transfers against the kernels of the current stream."""

    _serial = 0   # communicators made by this process (part of the store key)

    def __init__(self, rank: int, world_size: int, device: torch.device, store, group_tag: str = ""):
        from . import _lib
        self._lib = _lib
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        LskTransport._serial += 1
        key = f"lsk_comm_unique_id/{group_tag}/{LskTransport._serial}"

        def make_id():
            buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
            _lib.call("lsk_comm_unique_id", buf)
            return buf.raw

        if world_size == 1:
            uid = make_id()
        else:
            uid = bootstrap_unique_id(rank, make_id, store, key)
        self.handle = ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            _lib.call("lsk_comm_init_rank", ctypes.byref(self.handle), world_size, rank,
                      ctypes.create_string_buffer(uid, UNIQUE_ID_BYTES))
            # the halo stream: default priority, like every stream of this package
            self.side = torch.cuda.Stream(self.device)
        r, n = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.call("lsk_comm_rank", self.handle, ctypes.byref(r), ctypes.byref(n))
        self.rccl_rank, self.rccl_ranks = r.value, n.value

    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        if not t.is_contiguous():
            raise ValueError("all_reduce_sum takes a contiguous tensor")
        self._lib.call(f"lsk_allreduce_sum_{self._lib.suffix(t.dtype)}", self.handle,
                       ctypes.c_void_p(t.data_ptr()), t.numel(), self._stream())

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        if recv.numel() != send.numel() * self.world_size or send.dtype != recv.dtype:
            raise ValueError("all_gather: recv holds world_size blocks of send's size")
        self._lib.call(f"lsk_allgather_{self._lib.suffix(send.dtype)}", self.handle,
                       ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), send.numel(), self._stream())

    def exchange_start(self, sends, recvs):
        """Posts the grouped transfers on the halo stream, after everything already enqueued on the
        current stream; returns the event that marks their completion."""
        tensors = [t for _, t in list(sends) + list(recvs)]
        dtype = tensors[0].dtype
        if any(t.dtype != dtype or not t.is_contiguous() for t in tensors):
            raise ValueError("exchange takes contiguous views of one entry type")
        ns, nr = len(sends), len(recvs)
        speer = (ctypes.c_int * max(ns, 1))(*[p for p, _ in sends])
        sptr = (ctypes.c_void_p * max(ns, 1))(*[t.data_ptr() for _, t in sends])
        scount = (ctypes.c_int64 * max(ns, 1))(*[t.numel() for _, t in sends])
        rpeer = (ctypes.c_int * max(nr, 1))(*[p for p, _ in recvs])
        rptr = (ctypes.c_void_p * max(nr, 1))(*[t.data_ptr() for _, t in recvs])
        rcount = (ctypes.c_int64 * max(nr, 1))(*[t.numel() for _, t in recvs])
        current = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(current)
        self.side.wait_event(ready)
        self._lib.call(f"lsk_halo_exchange_{self._lib.suffix(dtype)}", self.handle, ns, speer, sptr, scount,
                       nr, rpeer, rptr, rcount, ctypes.c_void_p(self.side.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.side)
        return done

    def exchange_finish(self, done) -> None:
        torch.cuda.current_stream(self.device).wait_event(done)

    def self_check(self) -> None:
        """One all-reduce and one ring exchange with known answers; raises when either is wrong."""
        n, r = self.world_size, self.rank
        t = torch.tensor([float(r + 1), 1.0], dtype=torch.float64, device=self.device)
        self.all_reduce_sum(t)
        want = [n * (n + 1) / 2.0, float(n)]
        if t.tolist() != want:
            raise RuntimeError(f"lsk transport self-check: all-reduce gave {t.tolist()}, expected {want}")
        if n > 1:
            out = torch.full((257,), float(r), dtype=torch.float64, device=self.device)
            inn = torch.full((257,), -1.0, dtype=torch.float64, device=self.device)
            self.exchange_finish(self.exchange_start([((r + 1) % n, out)], [((r - 1) % n, inn)]))
            got = inn.tolist()
            if got != [float((r - 1) % n)] * 257:
                raise RuntimeError(f"lsk transport self-check: ring exchange delivered {got[:3]}... from rank {(r - 1) % n}")

    def close(self, abort: bool = False) -> None:
        if self.handle:
            handle, self.handle = self.handle, ctypes.c_void_p(0)
            self._lib.lib().lsk_comm_abort(handle) if abort else self._lib.lib().lsk_comm_destroy(handle)


class Communicator:
    """A torch.distributed process group (or a single process) and, on top, the transport of the
    per-step traffic (module docstring).  `transport`: "lsk", "torch" or None = LSK_TRANSPORT from
    the environment, else "lsk" where the group's backend is nccl and "torch" elsewhere."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, transport: Optional[str] = None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self.backend = dist.get_backend(group) if self.world_size > 1 else None
        asked = transport if transport is not None else os.environ.get("LSK_TRANSPORT")
        if asked not in (None, "", "lsk", "torch", "prefer-lsk"):
            raise ValueError(f"transport {asked!r}: 'lsk', 'prefer-lsk' or 'torch'")
        self._asked = asked or None
        self.transport = "torch"      # until bind_device has made the RCCL communicator
        self.transport_note = None    # why the transport is not the one asked for / the default
        self.lsk: Optional[LskTransport] = None

    def bind_device(self, device: torch.device) -> None:
        """Called once the rank's GPU is known (Context does): makes the "lsk" transport where it is
        asked for or the default.  The decision is taken by all ranks together: when the bootstrap
        or its self-check fails on any of them, every rank falls back to "torch" -- unless "lsk" was
        asked for explicitly, then the failure is raised."""
        want = self._asked or ("prefer-lsk" if (self.world_size > 1 and self.backend is not None
                                                and "nccl" in str(self.backend)) else "torch")
        if want == "torch" or self.lsk is not None or torch.device(device).type != "cuda":
            return
        limit = float(os.environ.get("LSK_COMM_TIMEOUT", "180"))
        made = {}

        def make_and_check():
            # (in a thread of its own: a peer that never arrives leaves ncclCommInitRank, or the first
            # collective, waiting for good, and nothing can interrupt that call)
            try:
                torch.cuda.set_device(device)   # (a new thread starts on device 0: RCCL and the library go by the thread's device)
                store = None
                if self.world_size > 1:
                    from torch.distributed.distributed_c10d import _get_default_store
                    store = _get_default_store()
                # (a key every rank of the group forms alike: its members' global ranks)
                tag = "world" if self.group is None else "ranks" + "-".join(
                    str(r) for r in dist.get_process_group_ranks(self.group))
                made["transport"] = LskTransport(self.rank, self.world_size, device, store, tag)
                made["transport"].self_check()
                torch.cuda.synchronize(device)
            except Exception as exc:  # noqa: BLE001
                made["error"] = f"rank {self.rank}: {type(exc).__name__}: {exc}"

        worker = threading.Thread(target=make_and_check, name="lsk-comm-bootstrap", daemon=True)
        worker.start()
        worker.join(limit)
        if worker.is_alive():   # left behind: whatever it holds is not touched again
            error = (f"rank {self.rank}: the RCCL communicator was not up after {limit:.0f} s (LSK_COMM_TIMEOUT)")
            made = {}
        else:
            error = made.get("error")
            self.lsk = made.get("transport")
        errors = [e for e in self.all_gather_object(error) if e]
        if errors:
            if self.lsk is not None:
                self.lsk.close(abort=True)
                self.lsk = None
            if self._asked == "lsk":
                raise RuntimeError("the lsk transport was asked for and could not be made: " + "; ".join(errors))
            self.transport_note = "lsk transport failed, torch.distributed carries the traffic: " + "; ".join(errors)
            return
        self.transport = "lsk"

    @property
    def rccl_ranks(self) -> Optional[int]:
        """Ranks of the RCCL communicator the per-step traffic runs on, as RCCL reports it (None when
        torch.distributed carries it)."""
        return self.lsk.rccl_ranks if self.lsk is not None else None

    def _staged(self, t: torch.Tensor) -> bool:
        return self.backend == "gloo" and t.is_cuda

    # -- set-up time (host values) -----------------------------------------------------
    def all_gather_object(self, obj):
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)

    def max_over_ranks(self, value: float, device=None) -> float:
        """Maximum of a host value over the ranks (timing brackets)."""
        if self.world_size == 1:
            return float(value)
        on = device if (self.backend != "gloo" and device is not None) else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    # -- per step, on the tensors' device and stream -------------------------------------
    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """In-place sum over ranks (the dot all-reduce; 1..k scalars)."""
        if self.lsk is not None and t.is_cuda:
            self.lsk.all_reduce_sum(t)
            return
        if self.world_size == 1:
            return
        if self._staged(t):
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def exchange(self, sends: Sequence[Tuple[int, torch.Tensor]],
                 recvs: Sequence[Tuple[int, torch.Tensor]]) -> None:
        """Grouped point-to-point exchange: (peer, contiguous tensor view) pairs."""
        self.exchange_finish(self.exchange_start(sends, recvs))

    def exchange_start(self, sends: Sequence[Tuple[int, torch.Tensor]],
                       recvs: Sequence[Tuple[int, torch.Tensor]]):
        """Posts the exchange and returns a handle for exchange_finish.  With RCCL the transfers
        run on the communicator's own stream, ordered after everything already enqueued on the
        current stream; kernels enqueued on the current stream between start and finish overlap
        with them and must not touch the receive views.  Receives are posted in ascending peer
        order and sends likewise, so that both sides of every pair agree on the order of the
        operations between them."""
        if not sends and not recvs:
            return None
        recvs = sorted(recvs, key=lambda it: it[0])
        sends = sorted(sends, key=lambda it: it[0])
        if self.lsk is not None and all(t.is_cuda for _, t in list(recvs) + list(sends)):
            return self.lsk.exchange_start(sends, recvs), None, sends
        if self.world_size == 1:
            return None
        staged = any(self._staged(t) for _, t in list(recvs) + list(sends))
        landing = None
        if staged:  # .cpu() synchronises with the stream that produced the data
            landing = [(peer, t, torch.empty(t.shape, dtype=t.dtype)) for peer, t in recvs]
            recvs = [(peer, host) for peer, _, host in landing]
            sends = [(peer, t.cpu()) for peer, t in sends]
        ops = []
        for peer, t in recvs:
            ops.append(dist.P2POp(dist.irecv, t, self._global_rank(peer), group=self.group))
        for peer, t in sends:
            ops.append(dist.P2POp(dist.isend, t, self._global_rank(peer), group=self.group))
        return dist.batch_isend_irecv(ops), landing, sends

    def exchange_finish(self, handle) -> None:
        """Makes the current stream (and, for host transports, the host) wait for the exchange."""
        if handle is None:
            return
        works, landing, _keep_alive = handle
        if isinstance(works, torch.cuda.Event):
            self.lsk.exchange_finish(works)
            return
        for work in works:
            work.wait()
        if landing is not None:
            for _, t, host in landing:
                t.copy_(host)

    def close(self) -> None:
        """Destroys the RCCL communicator of the lsk transport (the caller has synchronised)."""
        if self.lsk is not None:
            self.lsk.close()
            self.lsk = None
            self.transport = "torch"

    def _global_rank(self, group_rank: int) -> int:
        if self.group is None:
            return group_rank
        return dist.get_global_rank(self.group, group_rank)


