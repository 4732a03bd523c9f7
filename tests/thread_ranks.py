"""Ranks as threads of one process, all on the one GPU of the test machine: a communicator with the
interface of legionsolvers_amd.comm.Communicator whose transfers are device-to-device copies on the
shared stream.  RCCL refuses two ranks on one device and gloo only takes host tensors, so this is
how the N-rank path (row blocks dealt to ranks, halo plan, split products with both phases, fused dot
shares per phase, one all-reduce per dot) runs on real kernels with device-resident halos when
there is one card.  Test infrastructure only."""
from __future__ import annotations

import threading
from typing import Callable, List

import torch

BARRIER_SECONDS = 120.0


class _Shared:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=BARRIER_SECONDS)
        self.slots: List[object] = [None] * world
        self.mail = {}  # (src rank, dst rank) -> list of tensors, in posting order


class ThreadCommunicator:
    backend = "threads"

    def __init__(self, shared: _Shared, rank: int):
        self.shared, self.rank, self.world_size = shared, rank, shared.world
        self.group = None
        self.all_reduces = 0
        self.exchanges = 0

    def _sync(self):
        self.shared.barrier.wait()

    # -- set-up time ---------------------------------------------------------------------------
    def all_gather_object(self, obj):
        sh = self.shared
        sh.slots[self.rank] = obj
        self._sync()
        out = list(sh.slots)
        self._sync()
        return out

    def barrier(self):
        self._sync()

    def max_over_ranks(self, value: float, device=None) -> float:
        return max(self.all_gather_object(float(value)))

    # -- per step --------------------------------------------------------------------------------
    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """Every rank adds the ranks' tensors in rank order (all launched on the one shared stream,
        after every rank has enqueued what produces its tensor)."""
        if self.world_size == 1:
            return
        self.all_reduces += 1
        parts = self.all_gather_object(t)
        acc = parts[0].clone()
        for other in parts[1:]:
            acc += other
        self._sync()       # everyone has read everyone's tensor into its sum
        t.copy_(acc)

    def exchange(self, sends, recvs) -> None:
        self.exchange_finish(self.exchange_start(sends, recvs))

    def exchange_start(self, sends, recvs):
        """Same contract as Communicator.exchange_start: per pair of ranks the k-th send matches the
        k-th receive, both in ascending peer order.  Every rank of the group calls it."""
        sh = self.shared
        self.exchanges += 1
        for peer, t in sorted(sends, key=lambda it: it[0]):
            sh.mail.setdefault((self.rank, peer), []).append(t)
        self._sync()       # all posted (and all producers of the sent data enqueued)
        taken = {}
        for peer, t in sorted(recvs, key=lambda it: it[0]):
            k = taken.get(peer, 0)
            src = sh.mail[(peer, self.rank)][k]
            taken[peer] = k + 1
            assert src.shape == t.shape and src.dtype == t.dtype, (self.rank, peer, src.shape, t.shape)
            t.copy_(src)
        self._sync()       # all copies enqueued before any sender enqueues its next write
        if self.rank == 0:
            sh.mail.clear()
        self._sync()
        return True

    def exchange_finish(self, handle) -> None:
        return None


def run_ranks(world: int, body: Callable, device=None):
    """body(ctx) on `world` thread ranks; returns the list of results in rank order.  An exception in
    any rank breaks the barrier for all and is raised here."""
    from legionsolvers_amd import Context
    device = device or torch.device("cuda", 0)
    shared = _Shared(world)
    results, errors = [None] * world, [None] * world

    def main(rank):
        try:
            torch.cuda.set_device(device)
            ctx = Context(device, comm=ThreadCommunicator(shared, rank))
            results[rank] = body(ctx)
            ctx.synchronize()
        except BaseException as e:  # noqa: BLE001  (reported below)
            errors[rank] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=main, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    if first is None:
        first = next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results
