#!/usr/bin/env python3
"""The two transports of comm.Communicator on ONE rank of RCCL -- torch.distributed (backend nccl) and the
library's own layer (lsk_allreduce_sum_*, lsk_halo_exchange_* through comm.LskTransport, option
comm.skip_single_rank = 0 so that RCCL is really called): the dot's all-reduce and a halo exchange with
itself (send and receive in one group), ordered against kernels on the current stream, and what the two
calls cost the host and the stream.  (More ranks need more GPUs.)

    python tools/nccl_single_rank_probe.py [torch|lsk]      (default: both, one after the other)
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd.comm import Communicator  # noqa: E402


def probe(which):
    dev = torch.device("cuda", 0)
    if which == "lsk":
        from legionsolvers_amd import _lib
        _lib.call("lsk_set_option", b"comm.skip_single_rank", 0)
        comm = Communicator(transport="lsk")
        comm.bind_device(dev)
        assert comm.transport == "lsk" and comm.rccl_ranks == 1
    else:
        comm = Communicator(transport="torch")
        comm.world_size = 2          # (so that the calls are not short-circuited; the only peer is this rank)
        comm.backend = "nccl"
    print(f"== transport {which}")
    n = 262657                   # the halo of the 27-point 512^3 problem on 8 ranks, elements per side
    src = torch.arange(n, dtype=torch.float64, device=dev)
    dst = torch.zeros(n, dtype=torch.float64, device=dev)
    s = torch.ones(1, dtype=torch.float64, device=dev)
    for _ in range(3):           # (the first calls set the transports up)
        src.mul_(1.0)
        comm.exchange([(0, src)], [(0, dst)])
        comm.all_reduce_sum(s)
    torch.cuda.synchronize()
    ok = bool(torch.equal(src, dst)) and float(s.item()) == 1.0
    # a kernel, then the exchange of what it wrote, then a kernel on what arrived: stream order holds
    src.add_(1.0)
    h = comm.exchange_start([(0, src)], [(0, dst)])
    comm.exchange_finish(h)
    out = dst * 2.0
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(out, (torch.arange(n, dtype=torch.float64, device=dev) + 1.0) * 2.0))
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.all_reduce_sum(s)
    host_ar = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.all_reduce_sum(s)
    torch.cuda.synchronize()
    wall_ar = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.exchange_finish(comm.exchange_start([(0, src)], [(0, dst)]))
    host_ex = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.exchange_finish(comm.exchange_start([(0, src)], [(0, dst)]))
    torch.cuda.synchronize()
    wall_ex = (time.perf_counter() - t0) / reps
    print(f"results as expected: {ok}")
    print(f"all-reduce of one scalar: {host_ar * 1e6:.1f} us to enqueue, {wall_ar * 1e6:.1f} us per call back to back")
    print(f"exchange of {n} fp64 with itself: {host_ex * 1e6:.1f} us to enqueue, {wall_ex * 1e6:.1f} us per call back to back")
    comm.close()
    return ok


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ok = all([probe(w) for w in (sys.argv[1:] or ["torch", "lsk"])])
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
