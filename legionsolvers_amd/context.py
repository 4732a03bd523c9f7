"""Context: what the reference threads through every class as `(Legion::Context, Runtime*)`.

It names this process's GPU, the leaf backend that launches on it, and the communicator that
joins it to the other shards' processes.  One process drives one MI355X.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .comm import Communicator


class Context:
    def __init__(self, device: Optional[torch.device] = None, group=None, backend=None,
                 fuse: Optional[bool] = None, comm=None):
        """fuse: let the planner use the fused leaf forms (product + dot, axpy + squared norm)
        where a solver's next dot product can reuse them; default on, LSK_FUSE=0 turns it off.
        Results change only in the order of the additions inside the dot products.
        comm: an object with the interface of comm.Communicator in place of the torch.distributed
        one (ranks that are threads of one process, for instance)."""
        self.comm = comm if comm is not None else Communicator(group)
        self.fuse = (os.environ.get("LSK_FUSE", "1") != "0") if fuse is None else bool(fuse)
        if backend is None:
            from .backend import HipBackend
            if device is None:
                if not torch.cuda.is_available():
                    raise RuntimeError("no GPU visible: the Krylov path has no CPU fallback")
                device = torch.device("cuda", torch.cuda.current_device())
            backend = HipBackend(device)
        self.backend = backend
        self.device = backend.device
        bind = getattr(self.comm, "bind_device", None)
        if bind is not None:   # the RCCL communicator of the "lsk" transport needs the rank's GPU
            bind(self.device)

    @property
    def rank(self) -> int:
        return self.comm.rank

    @property
    def world_size(self) -> int:
        return self.comm.world_size

    def synchronize(self) -> None:
        self.backend.synchronize()
