#!/usr/bin/env python3
"""What an allocation does when the device is nearly full: torch holds all but 64 MiB of the free memory,
then the library's allocator and torch's are asked for 200 MB each."""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import _lib  # noqa: E402

torch.cuda.init()
free, total = torch.cuda.mem_get_info()
print("free, total", free, total, flush=True)
held = torch.empty(free - (64 << 20), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
print("free with the memory held", torch.cuda.mem_get_info()[0], flush=True)
p = ctypes.c_void_p(0)
rc = _lib.lib().lsk_malloc(ctypes.byref(p), ctypes.c_size_t(200 << 20))
print("lsk_malloc of 200 MiB:", rc, _lib.lib().lsk_last_error() if rc else "ok", "free now", torch.cuda.mem_get_info()[0], flush=True)
if rc == 0:
    _lib.lib().lsk_free(p)
try:
    t = torch.empty(200 << 20, dtype=torch.uint8, device="cuda")
    print("torch.empty of 200 MiB: ok, free now", torch.cuda.mem_get_info()[0], flush=True)
except torch.cuda.OutOfMemoryError as exc:
    print("torch.empty of 200 MiB:", str(exc)[:120], flush=True)
