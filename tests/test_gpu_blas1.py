"""GPU parity tests of the BLAS-1 leaf kernels, called through the C ABI and compared with
the CPU oracle.  Elementwise ops and scalar ops: bit-exact.  dot / nrm2: the summation order
differs from the sequential CPU loop; tolerance 1e-13 (fp64) / 2e-6 (fp32) relative to
sum |v_i w_i| against the exactly rounded sum, the sequential loop's own n*eps error bound
against the oracle, and bit-reproducible run to run."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (np.float32, torch.float32), "f64": (np.float64, torch.float64)}
SIZES = [0, 1, 2, 3, 63, 64, 255, 256, 257, 1023, 4099, 65536 + 5, (1 << 20) + 3]


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.fixture(scope="module", autouse=True)
def _init(gpu_ctx):
    return gpu_ctx


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("offset", [0, 1])
def test_elementwise_bit_exact(suf, offset):
    """scal / axpy / xpay, host alpha; `offset` shifts the vectors off 16-byte alignment."""
    npdt, _ = DT[suf]
    rng = np.random.default_rng(11)
    for n in SIZES:
        x = rng.standard_normal(n + offset).astype(npdt)
        y = rng.standard_normal(n + offset).astype(npdt)
        alpha = npdt(rng.standard_normal())
        for op in ("scal", "axpy", "xpay"):
            print(f"case {op}_{suf} n={n} offset={offset}", flush=True)
            want = y.copy()
            if op == "scal":
                oracle.scal(alpha, want[offset:])
            else:
                getattr(oracle, op)(alpha, np.ascontiguousarray(x[offset:]), want[offset:])
            xd, yd = dev(x), dev(y)
            if op == "scal":
                _lib.call(f"lsk_scal_{suf}", n, float(alpha), P(yd[offset:]), stream())
            else:
                _lib.call(f"lsk_{op}_{suf}", n, float(alpha), P(xd[offset:]), P(yd[offset:]), stream())
            got = yd.cpu().numpy()
            assert np.array_equal(bits(got), bits(want)), f"{op} n={n}"


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_elementwise_device_alpha_bit_exact(suf):
    """alpha = get_alpha(0..4 device futures), evaluated on the GPU in the reference's order."""
    npdt, _ = DT[suf]
    rng = np.random.default_rng(12)
    n = 10007
    for nf in range(5):
        f = (rng.standard_normal(4) + 2.0).astype(npdt)
        alpha = oracle.get_alpha(f[:nf], npdt)
        fd = dev(f)
        ptrs = [P(fd[i:i + 1]) if i < nf else ctypes.c_void_p(0) for i in range(4)]
        for op in ("scal", "axpy", "xpay"):
            print(f"case {op}_dev_{suf} nf={nf}", flush=True)
            x = rng.standard_normal(n).astype(npdt)
            y = rng.standard_normal(n).astype(npdt)
            want = y.copy()
            xd, yd = dev(x), dev(y)
            if op == "scal":
                oracle.scal(alpha, want)
                _lib.call(f"lsk_scal_dev_{suf}", n, nf, *ptrs, P(yd), stream())
            else:
                getattr(oracle, op)(alpha, x, want)
                _lib.call(f"lsk_{op}_dev_{suf}", n, nf, *ptrs, P(xd), P(yd), stream())
            assert np.array_equal(bits(yd.cpu().numpy()), bits(want)), f"{op} nf={nf}"
    assert _lib.lib().lsk_axpy_dev_f64(4, 5, *([ctypes.c_void_p(0)] * 4), P(fd), P(fd), stream()) == -2
    assert _lib.lib().lsk_axpy_dev_f64(4, 2, P(fd), *([ctypes.c_void_p(0)] * 3), P(fd), P(fd), stream()) == -2


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_fill_copy_and_aliasing(suf):
    npdt, tdt = DT[suf]
    for n in (0, 1, 5, 1000, 70001):
        for off in (0, 1):
            print(f"case fill/copy {suf} n={n} off={off}", flush=True)
            buf = torch.full((n + off + 2,), -7.0, dtype=tdt, device="cuda")
            _lib.call(f"lsk_fill_{suf}", n, 2.5, P(buf[off:]), stream())
            host = buf.cpu().numpy()
            assert np.all(host[off:off + n] == 2.5) and np.all(host[:off] == -7) and np.all(host[off + n:] == -7)
            src = dev(np.arange(n + off, dtype=npdt))
            dst = torch.zeros(n + 1, dtype=tdt, device="cuda")
            _lib.call(f"lsk_copy_{suf}", n, P(src[off:]), P(dst), stream())
            assert np.array_equal(dst.cpu().numpy()[:n], np.arange(off, n + off, dtype=npdt))
    # y == x is a legal call: y = fma(alpha, y, y)
    x = np.random.default_rng(1).standard_normal(4097).astype(npdt)
    want = x.copy()
    oracle.axpy(npdt(0.5), x.copy(), want)
    xd = dev(x)
    _lib.call(f"lsk_axpy_{suf}", x.size, 0.5, P(xd), P(xd), stream())
    assert np.array_equal(bits(xd.cpu().numpy()), bits(want))


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_dot_and_nrm2(suf):
    npdt, tdt = DT[suf]
    tol = 1e-13 if suf == "f64" else 2e-6
    rng = np.random.default_rng(13)
    ws_bytes = int(_lib.lib().lsk_dot_workspace_bytes(0))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    for n in SIZES + [(1 << 22) + 11]:
        for off in (0, 1):
            print(f"case dot_{suf} n={n} off={off}", flush=True)
            v = rng.standard_normal(n + off).astype(npdt)
            w = rng.standard_normal(n + off).astype(npdt)
            vd, wd = dev(v), dev(w)
            out = torch.full((3,), 99.0, dtype=tdt, device="cuda")
            _lib.call(f"lsk_dot_{suf}", n, P(vd[off:]), P(wd[off:]), P(out[0:1]), P(ws), ws_bytes, stream())
            _lib.call(f"lsk_dot_{suf}", n, P(vd[off:]), P(wd[off:]), P(out[1:2]), None, 0, stream())
            _lib.call(f"lsk_nrm2_{suf}", n, P(vd[off:]), P(out[2:3]), P(ws), ws_bytes, stream())
            got = out.cpu().numpy()
            v64, w64 = v[off:].astype(np.float64), w[off:].astype(np.float64)
            exact = float(v64 @ w64)
            scale = float(np.abs(v64 * w64).sum()) + 1e-300
            # against the exactly rounded value: the GPU's tree sum is accurate to a few ulps
            assert abs(float(got[0]) - exact) <= tol * scale, (n, got[0], exact)
            # against the oracle's left-to-right sum, whose own rounding error is bounded by
            # n*eps*sum|v w| (and is what separates the two for long fp32 vectors)
            want = float(oracle.dot(np.ascontiguousarray(v[off:]), np.ascontiguousarray(w[off:])))
            assert abs(float(got[0]) - want) <= (tol + n * float(np.finfo(npdt).eps)) * scale
            assert got[0].tobytes() == got[1].tobytes()          # reproducible, any workspace
            exact_nrm = float(np.sqrt(v64 @ v64))
            assert abs(float(got[2]) - exact_nrm) <= tol * (exact_nrm + 1e-300) * 4
            host = np.zeros(1, dtype=npdt)
            _lib.call(f"lsk_dot_host_{suf}", n, P(vd[off:]), P(wd[off:]), host.ctypes.data, stream())
            assert host[0].tobytes() == got[0].tobytes()
    # dot(v, v) takes the read-once path and equals dot of two equal copies
    v = rng.standard_normal(100003).astype(npdt)
    vd, vd2 = dev(v), dev(v)
    out = torch.zeros(2, dtype=tdt, device="cuda")
    _lib.call(f"lsk_dot_{suf}", v.size, P(vd), P(vd), P(out[0:1]), None, 0, stream())
    _lib.call(f"lsk_dot_{suf}", v.size, P(vd), P(vd2), P(out[1:2]), None, 0, stream())
    got = out.cpu().numpy()
    assert got[0].tobytes() == got[1].tobytes()
    # a caller-supplied workspace that is too small is refused, not overrun
    small = torch.empty(8, dtype=torch.uint8, device="cuda")
    big = dev(np.ones(1 << 20, dtype=npdt))
    rc = getattr(_lib.lib(), f"lsk_dot_{suf}")(1 << 20, P(big), P(big), P(out), P(small), 8, stream())
    assert rc == -2 and b"workspace" in _lib.lib().lsk_last_error()


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_scalar_ops_bit_exact(suf):
    """The scalar task bodies (src/UtilityTasks.cpp:34-99) on the device: IEEE division and
    square root, so bit-equal to the host."""
    npdt, tdt = DT[suf]
    rng = np.random.default_rng(14)
    names = ["neg", "add", "sub", "mul", "div", "sqrt", "rsqrt", "copy"]
    a = np.abs(rng.standard_normal(200)).astype(npdt) * npdt(1e3) + npdt(1e-3)
    b = (rng.standard_normal(200)).astype(npdt) + npdt(3)
    ad, bd = dev(a), dev(b)
    out = torch.zeros(200, dtype=tdt, device="cuda")
    for op, name in enumerate(names):
        print(f"case scalar_op {name} {suf}", flush=True)
        for i in range(200):
            _lib.call(f"lsk_scalar_op_{suf}", op, P(ad[i:i + 1]), P(bd[i:i + 1]), P(out[i:i + 1]), stream())
        got = out.cpu().numpy()
        want = np.array([oracle.scalar_op(name, a[i], b[i], npdt) for i in range(200)], dtype=npdt)
        assert np.array_equal(bits(got), bits(want)), name
    # sum_scalars folds left to right
    vals = rng.standard_normal(17).astype(npdt)
    acc = vals[0]
    for v in vals[1:]:
        acc = oracle.scalar_op("add", acc, v, npdt)
    res = torch.zeros(1, dtype=tdt, device="cuda")
    vals_d = dev(vals)
    _lib.call(f"lsk_sum_scalars_{suf}", 17, P(vals_d), P(res), stream())
    assert res.cpu().numpy()[0].tobytes() == np.asarray(acc, dtype=npdt).tobytes()


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_xpay_nd_bit_exact(suf):
    """N-D xpay over strided instances (what xpay_kernel does through Pitches + accessors),
    including Legion's default column-major (first dimension fastest) layout."""
    npdt, _ = DT[suf]
    rng = np.random.default_rng(15)
    cases = [
        ((100,), (1,), (1,)),
        ((10, 10), (1, 10), (1, 10)),            # column-major
        ((10, 10), (10, 1), (1, 10)),            # x row-major, y column-major
        ((4, 5, 5), (1, 4, 20), (1, 4, 20)),
        ((4, 5, 5), (30, 6, 1), (50, 7, 1)),     # padded rows
        ((0, 5, 5), (25, 5, 1), (25, 5, 1)),     # empty rect
    ]
    for extent, xs, ys in cases:
        print(f"case xpay_nd {suf} {extent} {xs} {ys}", flush=True)
        size = lambda st: 1 + sum((e - 1) * s for e, s in zip(extent, st)) if all(extent) else 1
        x = rng.standard_normal(size(xs)).astype(npdt)
        y = rng.standard_normal(size(ys)).astype(npdt)
        want = y.copy()
        oracle.xpay_nd(extent, npdt(1.25), x, xs, want, ys)
        xd, yd = dev(x), dev(y)
        ext = (ctypes.c_int64 * len(extent))(*extent)
        xsc = (ctypes.c_int64 * len(extent))(*xs)
        ysc = (ctypes.c_int64 * len(extent))(*ys)
        _lib.call(f"lsk_xpay_nd_{suf}", len(extent), ext, 1.25, P(xd), xsc, P(yd), ysc, stream())
        assert np.array_equal(bits(yd.cpu().numpy()), bits(want))


def test_shim_memory_and_streams():
    """lsk_malloc / lsk_free / streams / copies, with no torch memory involved."""
    lib = _lib.lib()
    n = ctypes.c_int(0)
    _lib.check(lib.lsk_device_count(ctypes.byref(n)))
    assert n.value >= 1
    s = ctypes.c_void_p()
    _lib.check(lib.lsk_stream_create(ctypes.byref(s)))
    ptr = ctypes.c_void_p()
    host = np.arange(1000, dtype=np.float64)
    _lib.check(lib.lsk_malloc(ctypes.byref(ptr), host.nbytes))
    assert ptr.value and ptr.value % 256 == 0
    _lib.check(lib.lsk_memcpy_h2d(ptr, host.ctypes.data, host.nbytes, s))
    _lib.check(lib.lsk_scal_f64(1000, 2.0, ptr, s))
    back = np.zeros_like(host)
    _lib.check(lib.lsk_memcpy_d2h(back.ctypes.data, ptr, host.nbytes, s))
    _lib.check(lib.lsk_stream_check(s))
    assert np.array_equal(back, host * 2)
    res = np.zeros(1)
    _lib.check(lib.lsk_dot_host_f64(1000, ptr, ptr, res.ctypes.data, s))
    assert res[0] == float(np.dot(host * 2, host * 2))
    # lsk_mem_info: what is free goes down by an allocation and comes back with its release
    free0, free1, total = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(lib.lsk_mem_info(ctypes.byref(free0), ctypes.byref(total)))
    assert 0 < free0.value <= total.value and total.value >= (16 << 30)
    big = ctypes.c_void_p()
    _lib.check(lib.lsk_malloc(ctypes.byref(big), 1 << 30))
    _lib.check(lib.lsk_mem_info(ctypes.byref(free1), ctypes.byref(total)))
    assert free0.value - free1.value >= (1 << 30) - (16 << 20)   # (the driver accounts in pages of its own)
    _lib.check(lib.lsk_free(big))
    _lib.check(lib.lsk_mem_info(ctypes.byref(free1), ctypes.byref(total)))
    assert free0.value - free1.value < (64 << 20)
    assert lib.lsk_mem_info(None, ctypes.byref(total)) == _lib.LSK_ERR_ARG
    _lib.check(lib.lsk_free(ptr))
    _lib.check(lib.lsk_stream_destroy(s))


def test_rccl_layer_single_rank():
    """The RCCL entry points with a one-rank communicator (all a 1-GPU box can run).  With option
    comm.skip_single_rank = 0 the collectives are really enqueued on the stream (ncclAllReduce,
    ncclAllGather over one rank), between kernels that produce and consume their operands, as a dot
    product's all-reduce sits between the fold kernel and the next axpy."""
    lib = _lib.lib()
    uid = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check(lib.lsk_comm_unique_id(uid))
    comm = ctypes.c_void_p()
    _lib.check(lib.lsk_comm_init_rank(ctypes.byref(comm), 1, 0, uid))
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(lib.lsk_comm_rank(comm, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)
    for skip in (1, 0):
        print(f"case rccl single rank, comm.skip_single_rank={skip}", flush=True)
        _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", skip))
        try:
            for suf, tdt in (("f64", torch.float64), ("f32", torch.float32)):
                v = torch.arange(1, 65, dtype=tdt, device="cuda")  # every partial sum exact in fp32
                buf = torch.zeros(2, dtype=tdt, device="cuda")
                # dot -> all-reduce -> scal by the reduced value, all stream-ordered
                _lib.call(f"lsk_dot_{suf}", 64, P(v), P(v), P(buf[0:1]), None, 0, stream())
                _lib.call(f"lsk_nrm2_{suf}", 64, P(v), P(buf[1:2]), None, 0, stream())
                _lib.check(getattr(lib, f"lsk_allreduce_sum_{suf}")(comm, P(buf), 2, stream()))
                y = torch.ones(8, dtype=tdt, device="cuda")
                _lib.call(f"lsk_scal_dev_{suf}", 8, 1, P(buf[0:1]), None, None, None, P(y), stream())
                out = torch.zeros(2, dtype=tdt, device="cuda")
                _lib.check(getattr(lib, f"lsk_allgather_{suf}")(comm, P(buf), P(out), 2, stream()))
                torch.cuda.synchronize()
                want = float(sum(i * i for i in range(1, 65)))
                assert buf[0].item() == want and abs(buf[1].item() - want ** 0.5) < 1e-4
                assert out.tolist() == buf.tolist()
                assert y.tolist() == [want] * 8
        finally:
            _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", 1))
    _lib.check(lib.lsk_halo_exchange_f64(comm, 0, None, None, None, 0, None, None, None, stream()))
    buf = dev(np.array([1.5, 2.5]))
    bad_peer = (ctypes.c_int * 1)(1)       # a rank the communicator does not have
    ptrs = (ctypes.c_void_p * 1)(buf.data_ptr())
    cnt = (ctypes.c_int64 * 1)(1)
    assert lib.lsk_halo_exchange_f64(comm, 1, bad_peer, ptrs, cnt, 0, None, None, None, stream()) == -2
    own_peer = (ctypes.c_int * 1)(0)       # a send to the rank itself that no receive takes
    assert lib.lsk_halo_exchange_f64(comm, 1, own_peer, ptrs, cnt, 0, None, None, None, stream()) == -2
    # after a refused exchange the communicator still works (no RCCL group was left open)
    _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", 0))
    try:
        _lib.check(lib.lsk_allreduce_sum_f64(comm, P(buf), 2, stream()))
    finally:
        _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", 1))
    torch.cuda.synchronize()
    assert buf.tolist() == [1.5, 2.5]
    _lib.check(lib.lsk_comm_destroy(comm))
    # a communicator given up without waiting (what a rank does whose peers never came): the next one works
    _lib.check(lib.lsk_comm_unique_id(uid))
    _lib.check(lib.lsk_comm_init_rank(ctypes.byref(comm), 1, 0, uid))
    _lib.check(lib.lsk_comm_abort(comm))
    _lib.check(lib.lsk_comm_abort(None))
    _lib.check(lib.lsk_comm_unique_id(uid))
    _lib.check(lib.lsk_comm_init_rank(ctypes.byref(comm), 1, 0, uid))
    _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", 0))
    try:
        _lib.check(lib.lsk_allreduce_sum_f64(comm, P(buf), 2, stream()))
    finally:
        _lib.check(lib.lsk_set_option(b"comm.skip_single_rank", 1))
    torch.cuda.synchronize()
    assert buf.tolist() == [1.5, 2.5]
    _lib.check(lib.lsk_comm_destroy(comm))


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("offset", [0, 1])
def test_axpy_sqnorm_and_sum_partials(suf, offset):
    """The fused axpy + squared norm: y bit-exact with the plain axpy (host and device alpha);
    the partial sums add up to dot(y, y) within the dot tolerance, reproducibly."""
    npdt, tdt = DT[suf]
    tol = 1e-13 if suf == "f64" else 2e-6
    rng = np.random.default_rng(17)
    for n in SIZES + [(1 << 22) + 11]:
        x = rng.standard_normal(n + offset).astype(npdt)
        y = rng.standard_normal(n + offset).astype(npdt)
        alpha = npdt(rng.standard_normal())
        print(f"case axpy_sqnorm_{suf} n={n} offset={offset}", flush=True)
        want = y.copy()
        oracle.axpy(alpha, np.ascontiguousarray(x[offset:]), want[offset:])
        exact = float(np.sum(want[offset:].astype(np.float64) ** 2))
        results = []
        for form in ("host", "dev", "dev"):
            xd, yd = dev(x), dev(y)
            partials = torch.full((2048,), 7.0, dtype=tdt, device="cuda")
            count = ctypes.c_int(-1)
            if form == "host":
                _lib.call(f"lsk_axpy_sqnorm_{suf}", n, float(alpha), P(xd[offset:]), P(yd[offset:]), P(partials),
                          2048, ctypes.byref(count), stream())
            else:
                # alpha = (f0 * f1) / f2 on the device, chosen to reproduce alpha exactly
                f = dev(np.array([alpha, 1.0, 1.0], dtype=npdt))
                _lib.call(f"lsk_axpy_sqnorm_dev_{suf}", n, 3, P(f[0:1]), P(f[1:2]), P(f[2:3]), None,
                          P(xd[offset:]), P(yd[offset:]), P(partials), 2048, ctypes.byref(count), stream())
            assert np.array_equal(bits(yd.cpu().numpy()), bits(want)), (n, form)
            assert (count.value == 0) == (n == 0) and count.value <= 2048
            out = torch.full((1,), -1.0, dtype=tdt, device="cuda")
            _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(out), None, 0, stream())
            assert abs(float(out) - exact) <= tol * max(exact, 1e-300) + 1e-300, (n, form)
            results.append(float(out))
        assert results[1] == results[2] == results[0]  # bit-reproducible
    # too little room for the partial sums is an error, not a silent truncation
    xd, yd = dev(np.ones(1 << 20, npdt)), dev(np.ones(1 << 20, npdt))
    small = torch.zeros(4, dtype=tdt, device="cuda")
    count = ctypes.c_int(0)
    assert _lib.lib().lsk_axpy_sqnorm_f64 is not None
    rc = getattr(_lib.lib(), f"lsk_axpy_sqnorm_{suf}")(1 << 20, 1.0, P(xd), P(yd), P(small), 4, ctypes.byref(count), stream())
    assert rc == -2


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_sum_partials_two_stage(suf):
    """Counts on either side of the one-workgroup (8192), wide one-workgroup (65536) and two-stage paths,
    against an exactly rounded sum."""
    npdt, tdt = DT[suf]
    tol = 1e-13 if suf == "f64" else 2e-6
    rng = np.random.default_rng(19)
    for count in (0, 1, 255, 8192, 8193, 9000, 32768, 65535, 65536, 65537, 100_003, 4_194_304):
        v = rng.standard_normal(count).astype(npdt)
        print(f"case sum_partials_{suf} count={count}", flush=True)
        out = torch.full((1,), -1.0, dtype=tdt, device="cuda")
        vd = dev(v) if count else torch.zeros(1, dtype=tdt, device="cuda")
        ws = torch.empty(_lib.lib().lsk_dot_workspace_bytes(0), dtype=torch.uint8, device="cuda")
        _lib.call(f"lsk_sum_partials_{suf}", count, P(vd), P(out), P(ws), ws.numel(), stream())
        exact = float(np.sum(v.astype(np.float64)))
        assert abs(float(out) - exact) <= tol * float(np.sum(np.abs(v.astype(np.float64)))) + 1e-300
        out2 = torch.full((1,), -1.0, dtype=tdt, device="cuda")
        _lib.call(f"lsk_sum_partials_{suf}", count, P(vd), P(out2), None, 0, stream())
        assert float(out2) == float(out)


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("offset", [0, 1])
def test_axpy_xpay_fused(suf, offset):
    """z <- fma(a, p, z); p <- fma(b, p, r) in one pass: both bit-exact with the oracle's axpy
    followed by its xpay (alpha from 2 futures, beta from 2 futures, as CG passes them)."""
    npdt, tdt = DT[suf]
    rng = np.random.default_rng(19)
    for n in SIZES + [(1 << 22) + 11]:
        r = rng.standard_normal(n + offset).astype(npdt)
        p = rng.standard_normal(n + offset).astype(npdt)
        z = rng.standard_normal(n + offset).astype(npdt)
        fa = rng.standard_normal(2).astype(npdt)
        fb = rng.standard_normal(2).astype(npdt)
        print(f"case axpy_xpay_{suf} n={n} offset={offset}", flush=True)
        alpha = oracle.get_alpha([fa[0], fa[1]], npdt)
        beta = oracle.get_alpha([fb[0], fb[1]], npdt)
        want_z, want_p = z.copy(), p.copy()
        oracle.axpy(alpha, np.ascontiguousarray(p[offset:]), want_z[offset:])
        oracle.xpay(beta, np.ascontiguousarray(r[offset:]), want_p[offset:])
        rd, pd, zd, fad, fbd = dev(r), dev(p), dev(z), dev(fa), dev(fb)
        _lib.call(f"lsk_axpy_xpay_dev_{suf}", n, 2, P(fad[0:1]), P(fad[1:2]), None, None,
                  2, P(fbd[0:1]), P(fbd[1:2]), None, None, P(rd[offset:]), P(pd[offset:]), P(zd[offset:]), stream())
        assert np.array_equal(bits(zd.cpu().numpy()), bits(want_z)), n
        assert np.array_equal(bits(pd.cpu().numpy()), bits(want_p)), n
        assert np.array_equal(bits(rd.cpu().numpy()), bits(r)), n
    v = dev(np.ones(8, npdt))
    one = dev(np.ones(1, npdt))
    rc = getattr(_lib.lib(), f"lsk_axpy_xpay_dev_{suf}")(8, 1, P(one), None, None, None, 1, P(one), None, None, None,
                                                         P(v), P(v), P(dev(np.ones(8, npdt))), stream())
    assert rc == -2  # the three vectors must differ


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("offset", [0, 1])
def test_axpy_dot_fused(suf, offset):
    """y <- fma(alpha, x, y) with the partial sums of y_new . w for a third vector w: y bit-exact with
    the oracle's axpy, the sum within the dot tolerance and bit-reproducible; w == y is the squared
    norm, w == x is allowed."""
    npdt, tdt = DT[suf]
    tol = 1e-13 if suf == "f64" else 2e-6
    rng = np.random.default_rng(23)
    for n in SIZES + [(1 << 22) + 11]:
        x = rng.standard_normal(n + offset).astype(npdt)
        y = rng.standard_normal(n + offset).astype(npdt)
        w = rng.standard_normal(n + offset).astype(npdt)
        f = rng.standard_normal(1).astype(npdt)
        print(f"case axpy_dot_{suf} n={n} offset={offset}", flush=True)
        want = y.copy()
        oracle.axpy(f[0], np.ascontiguousarray(x[offset:]), want[offset:])
        fd = dev(f)
        for which in ("third", "self", "x"):
            xd, yd, wd = dev(x), dev(y), dev(w)
            other = {"third": wd, "self": yd, "x": xd}[which]
            ref = {"third": w, "self": want, "x": x}[which]
            exact = float(np.sum(want[offset:].astype(np.float64) * ref[offset:].astype(np.float64)))
            scale = float(np.sum(np.abs(want[offset:].astype(np.float64) * ref[offset:].astype(np.float64))))
            results = []
            for _ in range(2):
                yd = dev(y)
                other = yd if which == "self" else other
                partials = torch.full((2048,), 7.0, dtype=tdt, device="cuda")
                count = ctypes.c_int(-1)
                _lib.call(f"lsk_axpy_dot_dev_{suf}", n, 1, P(fd), None, None, None, P(xd[offset:]), P(yd[offset:]),
                          P(other[offset:]), P(partials), 2048, ctypes.byref(count), stream())
                assert np.array_equal(bits(yd.cpu().numpy()), bits(want)), (n, which)
                out = torch.full((1,), -1.0, dtype=tdt, device="cuda")
                _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(out), None, 0, stream())
                assert abs(float(out) - exact) <= tol * scale + 1e-300, (n, which)
                results.append(float(out))
            assert results[0] == results[1]


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_random_fill_bit_exact_and_partition_independent(gpu_ctx, suf):
    from legionsolvers_amd import IndexPartition, IndexSpace, PartitionedVector
    npdt, tdt = DT[suf]
    for n, seed, first in ((0, 1, 0), (1, 0, 0), (1000, 7, 123456789012), (70001, 2**63 + 5, 3)):
        x = torch.full((n,), -1.0, dtype=tdt, device="cuda")
        _lib.call(f"lsk_random_fill_{suf}", n, seed, first, P(x), stream())
        want = oracle.random_fill(n, seed, first, npdt)
        got = x.cpu().numpy()
        assert np.array_equal(bits(got), bits(want))
        assert n == 0 or (got.min() >= 0.0 and got.max() < 1.0)
    whole = None
    for pieces in (1, 3, 7):
        v = PartitionedVector(gpu_ctx, "v", IndexPartition.equal(IndexSpace(0, 9999), pieces), tdt)
        v.random_fill(seed=42)
        h = v.to_host()
        whole = h if whole is None else whole
        assert torch.equal(h, whole)
    assert 0.45 < float(whole.double().mean()) < 0.55


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_nontemporal_streams_give_the_same_bits(suf):
    """Option blas1.nontemporal = 1 (what vectors of 128 MiB and more get by default): every stream
    kernel's nontemporal instance gives the bits of the plain one."""
    npdt, tdt = DT[suf]
    rng = np.random.default_rng(31)
    lib = _lib.lib()
    results = {}
    for mode in (0, 1):
        _lib.check(lib.lsk_set_option(b"blas1.nontemporal", mode))
        try:
            got = []
            for n in (0, 5, 1024, 4099, (1 << 20) + 3):
                print(f"case nontemporal={mode} {suf} n={n}", flush=True)
                g = np.random.default_rng(n)
                x, y, z = (g.standard_normal(n).astype(npdt) for _ in range(3))
                f = dev(np.array([0.75, 1.5], dtype=npdt))
                xd, yd, zd = dev(x), dev(y), dev(z)
                _lib.call(f"lsk_axpy_{suf}", n, 0.5, P(xd), P(yd), stream())
                _lib.call(f"lsk_xpay_{suf}", n, -0.25, P(xd), P(yd), stream())
                _lib.call(f"lsk_scal_{suf}", n, 1.25, P(yd), stream())
                _lib.call(f"lsk_copy_{suf}", n, P(yd), P(zd), stream())
                out = torch.zeros(4, dtype=tdt, device="cuda")
                _lib.call(f"lsk_dot_{suf}", n, P(xd), P(yd), P(out[0:1]), None, 0, stream())
                _lib.call(f"lsk_nrm2_{suf}", n, P(yd), P(out[1:2]), None, 0, stream())
                partials = torch.zeros(2048, dtype=tdt, device="cuda")
                count = ctypes.c_int(0)
                _lib.call(f"lsk_axpy_dot_dev_{suf}", n, 2, P(f[0:1]), P(f[1:2]), None, None, P(xd), P(yd), P(zd),
                          P(partials), 2048, ctypes.byref(count), stream())
                _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(out[2:3]), None, 0, stream())
                if n:
                    r = dev(g.standard_normal(n).astype(npdt))
                    _lib.call(f"lsk_axpy_xpay_dev_{suf}", n, 1, P(f[0:1]), None, None, None, 1, P(f[1:2]), None, None,
                              None, P(r), P(yd), P(zd), stream())
                got.append((yd.cpu().numpy().tobytes(), zd.cpu().numpy().tobytes(), out.cpu().numpy().tobytes()))
            results[mode] = got
        finally:
            _lib.check(lib.lsk_set_option(b"blas1.nontemporal", -1))
    assert results[0] == results[1]
    v = ctypes.c_int(7)
    _lib.check(lib.lsk_get_option(b"blas1.nontemporal", ctypes.byref(v)))
    assert v.value == -1 and lib.lsk_set_option(b"blas1.nontemporal", 2) == _lib.LSK_ERR_ARG


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("offset", [0, 1])
def test_multi_dot_and_multi_axpy(suf, offset):
    """lsk_multi_dot_*: k dot products with one vector in one pass (groups of 8), each within the dot
    tolerance of the exactly rounded value and bit-reproducible; lsk_multi_axpy_*: bit for bit the
    chain of k axpys of the oracle, with the shares of the result's squared norm."""
    npdt, tdt = DT[suf]
    tol = 1e-13 if suf == "f64" else 2e-6
    rng = np.random.default_rng(41)
    lib = _lib.lib()
    for n in (0, 1, 3, 255, 1024, 4099, (1 << 20) + 3):
        for k in (1, 2, 7, 8, 9, 17, 31):
            if n > 5000 and k not in (1, 9, 31):
                continue
            print(f"case multi {suf} n={n} k={k} offset={offset}", flush=True)
            vs = [rng.standard_normal(n + offset).astype(npdt) for _ in range(k)]
            w = rng.standard_normal(n + offset).astype(npdt)
            vds, wd = [dev(v) for v in vs], dev(w)
            ptrs = (ctypes.c_void_p * k)(*[v[offset:].data_ptr() for v in vds])
            ws = torch.empty(int(lib.lsk_multi_dot_workspace_bytes(k)), dtype=torch.uint8, device="cuda")
            outs = []
            for _ in range(2):
                out = torch.full((k + 1,), 5.0, dtype=tdt, device="cuda")
                _lib.call(f"lsk_multi_dot_{suf}", n, k, ptrs, P(wd[offset:]), P(out), P(ws), ws.numel(), stream())
                outs.append(out.cpu().numpy())
            assert outs[0].tobytes() == outs[1].tobytes() and outs[0][k] == 5.0
            for j in range(k):
                a, b = vs[j][offset:].astype(np.float64), w[offset:].astype(np.float64)
                assert abs(float(outs[0][j]) - float(a @ b)) <= tol * float(np.abs(a * b).sum()) + 1e-300, (n, k, j)
            # the updates: y <- y - sum_j h_j v_j, h = the dots just computed
            y = rng.standard_normal(n + offset).astype(npdt)
            want = y.copy()
            for j in range(k):
                oracle.axpy(npdt(-outs[0][j]), np.ascontiguousarray(vs[j][offset:]), want[offset:])
            for with_norm in (False, True):
                yd = dev(y)
                partials = torch.full((2048,), 7.0, dtype=tdt, device="cuda") if with_norm else None
                count = ctypes.c_int(-1)
                coef = dev(outs[0][:k].copy())
                _lib.call(f"lsk_multi_axpy_{suf}", n, k, ptrs, P(coef), 1, P(yd[offset:]),
                          P(partials) if with_norm else None, 2048 if with_norm else 0,
                          ctypes.byref(count) if with_norm else None, stream())
                assert np.array_equal(bits(yd.cpu().numpy()), bits(want)), (n, k, with_norm)
                if with_norm:
                    res = torch.zeros(1, dtype=tdt, device="cuda")
                    _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(res), None, 0, stream())
                    exact = float(np.sum(want[offset:].astype(np.float64) ** 2))
                    assert abs(float(res) - exact) <= tol * max(exact, 1e-300) + 1e-300
    # refusals
    v = dev(np.ones(16, npdt))
    ptrs = (ctypes.c_void_p * 1)(v.data_ptr())
    out = torch.zeros(1, dtype=tdt, device="cuda")
    fn = getattr(lib, f"lsk_multi_dot_{suf}")
    assert fn(16, 1, ptrs, P(v), P(out), None, 0, stream()) == _lib.LSK_ERR_ARG          # no workspace
    assert fn(16, 65, ptrs, P(v), P(out), None, 0, stream()) == _lib.LSK_ERR_ARG         # too many vectors
    fa = getattr(lib, f"lsk_multi_axpy_{suf}")
    assert fa(16, 1, ptrs, P(out), 0, P(v), None, 0, None, stream()) == _lib.LSK_ERR_ARG  # a source is the destination


def test_torch_rccl_transport_single_rank():
    """The torch.distributed transport the planner uses between ranks (backend nccl = RCCL), on the one
    rank a one-GPU machine has: an all-reduce and a grouped send / receive with itself, ordered against
    kernels on the current stream (tools/nccl_single_rank_probe.py, in a process of its own so that this
    one keeps no process group)."""
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as probe:   # a port nobody holds right now
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    done = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_single_rank_probe.py")], env=env,
                          capture_output=True, text=True, timeout=240)
    print(done.stdout[-600:], flush=True)
    assert done.returncode == 0 and "results as expected: True" in done.stdout, done.stderr[-800:]
