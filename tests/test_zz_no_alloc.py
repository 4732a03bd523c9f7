"""lsk_dot_noalloc_f64: the allocation-free variant of lsk_dot_f64 (partial sums in a static device
array inside the library), called through the C ABI and compared bit for bit with lsk_dot_f64 on
the same inputs -- two calls in a row, the second on a smaller input, so that what the first left
in the shared array cannot leak into the second -- then the edge cases, and once under stream
capture, where a hipMalloc would fail the capture.

Collected last on purpose (the round's end runs `pytest -x`)."""
import ctypes

import numpy as np
import pytest
import torch

from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pair(n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    v = torch.randn(n, dtype=torch.float64, generator=g).cuda()
    w = torch.randn(n, dtype=torch.float64, generator=g).cuda()
    return v, w


def _reference(n, v, w, ws):
    out = torch.full((1,), -3.0, dtype=torch.float64, device="cuda")
    _lib.call("lsk_dot_f64", n, P(v), P(w), P(out), P(ws), ws.numel(), stream())
    return out


def test_noalloc_dot_matches_dot_two_calls_in_a_row():
    lib = _lib.lib()
    torch.cuda.set_device(0)
    ws = torch.empty(int(lib.lsk_dot_workspace_bytes(0)), dtype=torch.uint8, device="cuda")
    sizes = [(1 << 22) + 11, 100003, 1000, 1, 0, (1 << 20) + 1]  # large first, then smaller ones
    for k, n in enumerate(sizes):
        for off in (0, 1):  # off 1: vectors off 16-byte alignment (the scalar-access kernels)
            print(f"case noalloc_dot n={n} off={off}", flush=True)
            v, w = _pair(n + off, 100 + k)
            want = _reference(n, v[off:], w[off:], ws)
            got = torch.full((1,), 5.0, dtype=torch.float64, device="cuda")
            _lib.call("lsk_dot_noalloc_f64", n, P(v[off:]), P(w[off:]), P(got), stream())
            torch.cuda.synchronize()
            assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes(), (n, off, float(got), float(want))
            if n:
                exact = float(v[off:].cpu().numpy() @ w[off:].cpu().numpy())
                scale = float(np.abs(v[off:].cpu().numpy() * w[off:].cpu().numpy()).sum())
                assert abs(float(got) - exact) <= 1e-13 * scale
    # v.v takes the read-once kernels in both entry points
    v, _ = _pair(70001, 7)
    want = _reference(70001, v, v, ws)
    got = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.call("lsk_dot_noalloc_f64", 70001, P(v), P(v), P(got), stream())
    assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes()


def test_noalloc_dot_refuses_what_it_cannot_take():
    lib = _lib.lib()
    v, w = _pair(16, 3)
    out = torch.full((1,), 9.0, dtype=torch.float64, device="cuda")
    too_many = (1 << 34) + 1
    assert lib.lsk_dot_noalloc_f64(too_many, P(v), P(w), P(out), stream()) == _lib.LSK_ERR_ARG
    assert b"at most" in lib.lsk_last_error()
    assert lib.lsk_dot_noalloc_f64(-1, P(v), P(w), P(out), stream()) == _lib.LSK_ERR_ARG
    assert lib.lsk_dot_noalloc_f64(16, None, P(w), P(out), stream()) == _lib.LSK_ERR_ARG
    assert lib.lsk_dot_noalloc_f64(16, P(v), P(w), None, stream()) == _lib.LSK_ERR_ARG
    torch.cuda.synchronize()
    assert float(out) == 9.0  # a refused call has written nothing


def test_noalloc_dot_under_stream_capture():
    """Captured into a HIP graph (an allocation inside the call would fail the capture) and replayed
    on changing inputs, with eager work between the replays; every replay equals the eager result."""
    lib = _lib.lib()
    n = (1 << 21) + 5
    ws = torch.empty(int(lib.lsk_dot_workspace_bytes(0)), dtype=torch.uint8, device="cuda")
    v, w = _pair(n, 11)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    _lib.call("lsk_dot_noalloc_f64", n, P(v), P(w), P(out), stream())  # the address look-up happens here
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        rc = lib.lsk_dot_noalloc_f64(n, P(v), P(w), P(out), stream())
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    assert rc == 0, lib.lsk_last_error()
    for k in range(3):
        print(f"case noalloc_dot graph replay {k}", flush=True)
        fresh_v, fresh_w = _pair(n, 20 + k)
        v.copy_(fresh_v)
        w.copy_(fresh_w)
        out.fill_(-1.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = _reference(n, v, w, ws)
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == want.cpu().numpy().tobytes(), k
