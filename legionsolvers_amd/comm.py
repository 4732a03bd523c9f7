"""Cross-rank data movement of the Krylov path: the scalar all-reduce behind `dot` and the
halo gather of `x` behind `matvec`.

In the reference both are implicit in Legion (a REDOP_SUM future map,
src/PartitionedVector.cpp:353-357; a READ_ONLY requirement on the ghost partition,
src/CSRMatrix.cpp:202-211).  Here one process drives one MI355X and `torch.distributed`
carries them: backend "nccl" is RCCL over xGMI on the GPUs, "gloo" runs the same code on CPU
tensors for tests.  The plan arithmetic (who sends which interval to whom) is pure host code.

Rehearsal transport: when the group's backend is "gloo" and the tensors live on a GPU (several
ranks sharing one card on a one-GPU machine), every transfer is staged through host memory with
synchronous copies, because gloo is only ever handed CPU tensors here.  It exists to run the
multi-rank path on real kernels where RCCL cannot (RCCL refuses two ranks on one device); it is
not a data path for production.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


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


class Communicator:
    """Thin wrapper over a torch.distributed process group (or a single process)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self.backend = dist.get_backend(group) if self.world_size > 1 else None

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
        if self.world_size == 1 or (not sends and not recvs):
            return None
        recvs = sorted(recvs, key=lambda it: it[0])
        sends = sorted(sends, key=lambda it: it[0])
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
        for work in works:
            work.wait()
        if landing is not None:
            for _, t, host in landing:
                t.copy_(host)

    def _global_rank(self, group_rank: int) -> int:
        if self.group is None:
            return group_rank
        return dist.get_global_rank(self.group, group_rank)
