"""CPU oracle for the Krylov inner loop -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's CPU leaf tasks (oracle.c, plain C) with numpy wrappers, plus
Python restatements of its partition arithmetic (partition.py) and solver sequences
(solvers.py).  It is the checker the HIP path is compared against; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

Pinning: tests/test_oracle_golden.py checks this oracle against every golden the reference
holds for the path (test_all.py's EXPECTED_CG_OUTPUT and EXPECTED_PARTITION_OUTPUT, the Test01 /
Test02 in-program expectations) and against fixtures generated from the reference's own
scripts/krylov.py (tests/golden/make_golden.py).  Results none of those cover (2-D/3-D
stencils, fp32, BiCGStab, GMRES, COO with unsorted entries) are "parity unpinned" by the
reference and are pinned only by this restatement cross-checked with scipy.sparse.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_DIR, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.run(["make", "-C", _DIR, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_i64 = ctypes.c_int64
_p = ctypes.c_void_p
_CT = {np.dtype(np.float32): ctypes.c_float, np.dtype(np.float64): ctypes.c_double}
_SUF = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            getattr(_lib, f"oracle_dot_{suf}").restype = ct
            getattr(_lib, f"oracle_scalar_op_{suf}").restype = ct
            getattr(_lib, f"oracle_scalar_op_{suf}").argtypes = [ctypes.c_int, ct, ct]
            getattr(_lib, f"oracle_cg_step_mt_{suf}").restype = ct
        for name in ("oracle_stencil_size", "oracle_stencil_fill", "oracle_task_id_t",
                     "oracle_task_id_di", "oracle_task_id_tdi", "oracle_task_id_tdddiii"):
            getattr(_lib, name).restype = _i64
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_p)


def _chk(a: np.ndarray, dtype=None) -> np.ndarray:
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    if dtype is not None:
        assert a.dtype == np.dtype(dtype), (a.dtype, dtype)
    return a


# ------------------------------------------------------------------ get_alpha / scalars --

def get_alpha(futures, dtype) -> np.generic:
    """src/LegionUtilities.cpp:72-97."""
    dt = np.dtype(dtype)
    f = np.array(list(futures), dtype=dt)
    out = np.zeros(1, dtype=dt)
    rc = getattr(lib(), f"oracle_get_alpha_{_SUF[dt]}")(ctypes.c_int(len(f)), _ptr(f), _ptr(out))
    if rc != 0:
        raise ValueError("get_alpha takes 0..4 futures")
    return out[0]


SOP = {"neg": 0, "add": 1, "sub": 2, "mul": 3, "div": 4, "sqrt": 5, "rsqrt": 6, "copy": 7}


def scalar_op(op: str, a, b=0.0, dtype=np.float64) -> np.generic:
    """src/UtilityTasks.cpp:34-99."""
    dt = np.dtype(dtype)
    ct = _CT[dt]
    return dt.type(getattr(lib(), f"oracle_scalar_op_{_SUF[dt]}")(SOP[op], ct(a), ct(b)))


# ------------------------------------------------------------------ BLAS-1 ---------------

def scal(alpha, x: np.ndarray) -> None:
    dt = x.dtype
    getattr(lib(), f"oracle_scal_{_SUF[dt]}")(_i64(x.size), _CT[dt](alpha), _ptr(_chk(x)))


def axpy(alpha, x: np.ndarray, y: np.ndarray) -> None:
    dt = y.dtype
    assert x.size == y.size and x.dtype == dt
    getattr(lib(), f"oracle_axpy_{_SUF[dt]}")(_i64(y.size), _CT[dt](alpha), _ptr(_chk(x)), _ptr(_chk(y)))


def xpay(alpha, x: np.ndarray, y: np.ndarray) -> None:
    dt = y.dtype
    assert x.size == y.size and x.dtype == dt
    getattr(lib(), f"oracle_xpay_{_SUF[dt]}")(_i64(y.size), _CT[dt](alpha), _ptr(_chk(x)), _ptr(_chk(y)))


def dot(v: np.ndarray, w: np.ndarray) -> np.generic:
    dt = v.dtype
    assert v.size == w.size and w.dtype == dt
    return dt.type(getattr(lib(), f"oracle_dot_{_SUF[dt]}")(_i64(v.size), _ptr(_chk(v)), _ptr(_chk(w))))


def xpay_nd(extent, alpha, x: np.ndarray, x_stride, y: np.ndarray, y_stride) -> None:
    dt = y.dtype
    ext = np.array(extent, dtype=np.int64)
    xs = np.array(x_stride, dtype=np.int64)
    ys = np.array(y_stride, dtype=np.int64)
    getattr(lib(), f"oracle_xpay_nd_{_SUF[dt]}")(ctypes.c_int(len(ext)), _ptr(ext), _CT[dt](alpha),
                                                _ptr(x), _ptr(xs), _ptr(y), _ptr(ys))


# ------------------------------------------------------------------ matvecs --------------

def coo_matvec(row, col, val, x, x_lo, x_hi, y, y_lo, y_hi) -> None:
    """y[row-y_lo] += val * x[col-x_lo] with the domain guards; src/COOMatrixTasks.cpp:67-74."""
    dt = val.dtype
    getattr(lib(), f"oracle_coo_matvec_{_SUF[dt]}")(
        _i64(val.size), _ptr(_chk(row, np.int64)), _ptr(_chk(col, np.int64)), _ptr(_chk(val)),
        _ptr(_chk(x, dt)), _i64(x_lo), _i64(x_hi), _ptr(_chk(y, dt)), _i64(y_lo), _i64(y_hi))


def csr_matvec(rects, k_base, col, val, x, x_lo, x_hi, y, beta=0, literal=False, mt=False) -> None:
    """rects: (rows, 2) int64 {lo, hi} global kernel ids; src/CSRMatrixTasks.cpp:73-91."""
    dt = val.dtype
    rects = _chk(rects, np.int64)
    rows = rects.shape[0]
    common = (_i64(rows), _i64(val.size), _ptr(rects), _i64(k_base), _ptr(_chk(col, np.int64)),
              _ptr(_chk(val)), _ptr(_chk(x, dt)), _i64(x_lo), _i64(x_hi), _ptr(_chk(y, dt)))
    if literal:
        if not beta:
            y[:] = 0
        getattr(lib(), f"oracle_csr_matvec_literal_{_SUF[dt]}")(*common)
    else:
        name = "oracle_csr_matvec_mt_" if mt else "oracle_csr_matvec_"
        getattr(lib(), name + _SUF[dt])(*common, ctypes.c_int(beta))


def coo_rmatvec(row, col, val, x, x_lo, x_hi, y, y_lo, y_hi) -> None:
    """y[col-y_lo] += val * x[row-x_lo], entries visited in ascending order: the loop of
    src/COOMatrixTasks.cpp:67-74 with the two index fields exchanged.  The reference's own
    COORmatvecTask body is assert(false) (src/COOMatrixTasks.cpp:78-83), so this definition is
    not pinned by any output of the reference: parity unpinned."""
    coo_matvec(col, row, val, x, x_lo, x_hi, y, y_lo, y_hi)


def csr_entry_rows(rects, k_base, nnz) -> np.ndarray:
    """Row (0-based in rects) of every kernel entry k_base..k_base+nnz-1, -1 for an entry inside
    no rect: the row field of the COO form of the same matrix."""
    rects = _chk(rects, np.int64)
    out = np.full(nnz, -1, dtype=np.int64)
    for r in range(rects.shape[0]):
        lo, hi = max(int(rects[r, 0]) - k_base, 0), min(int(rects[r, 1]) - k_base, nnz - 1)
        if hi >= lo:
            out[lo:hi + 1] = r
    return out


def csr_rmatvec(rects, k_base, col, val, x, x_lo, y, y_lo, y_hi, beta=0) -> None:
    """y <- A^T x + beta y for the CSR rows `rects` (row r reads x[r], the global index of x[0] is
    x_lo): coo_rmatvec over the entries in kernel order.  CSRRmatvecTask is assert(false) in the
    reference (src/CSRMatrixTasks.cpp:95-100): parity unpinned."""
    if not beta:
        y[:] = 0
    rows = csr_entry_rows(rects, k_base, val.size)
    live = rows >= 0
    coo_rmatvec(rows[live] + x_lo, _chk(col, np.int64)[live], val[live], x, x_lo, x_lo + rects.shape[0] - 1,
                y, y_lo, y_hi)


def rects_to_indptr(rects: np.ndarray) -> np.ndarray:
    """src/CuSPARSEHelpers.hpp:20-30."""
    rects = _chk(rects, np.int64)
    out = np.zeros(rects.shape[0] + 1, dtype=np.int64)
    lib().oracle_rects_to_indptr(_i64(rects.shape[0]), _ptr(rects), _ptr(out))
    return out


def set_threads(n: int) -> int:
    """Worker threads of the OpenMP (_mt) functions; returns the count in force."""
    lib().oracle_set_threads(int(n))
    return int(lib().oracle_get_max_threads())


def cg_step_mt(rects, col, val, sol, p, q, r, rr_old, shards: int):
    """One single-system CG step over `shards` row blocks with OpenMP (CPU-baseline timing)."""
    dt = val.dtype
    partial = np.zeros(shards, dtype=dt)
    fn = getattr(lib(), f"oracle_cg_step_mt_{_SUF[dt]}")
    return dt.type(fn(_i64(rects.shape[0]), _i64(val.size), _ptr(rects), _ptr(col), _ptr(val),
                      _ptr(sol), _ptr(p), _ptr(q), _ptr(r), _CT[dt](rr_old), ctypes.c_int(shards),
                      _ptr(partial)))


# ------------------------------------------------------------------ generators -----------

def sort_offsets(offsets, coeffs):
    """Offsets sorted as the fill tasks sort them: compare_row_major on the point, then the
    entry (src/StencilGenerator.cpp:40-52, src/StencilGenerator.hpp:174-183)."""
    pairs = sorted(zip([tuple(int(v) for v in o) for o in offsets], [float(c) for c in coeffs]))
    return [p[0] for p in pairs], [p[1] for p in pairs]


def stencil_size(extent, offsets) -> int:
    """calculate_stencil_size (src/StencilGenerator.hpp:270-323), by direct count."""
    ext = np.array(extent, dtype=np.int64)
    off = np.array(offsets, dtype=np.int64).reshape(len(offsets), len(extent))
    return int(lib().oracle_stencil_size(ctypes.c_int(len(ext)), _ptr(ext), ctypes.c_int(len(off)), _ptr(off)))


def linearized_stencil(extent, offsets, coeffs, dtype=np.float64, coo=False):
    """create_linearized_{csr,coo}_stencil_matrix, ROW_MAJOR (src/StencilGenerator.cpp:129-242,
    379-517).  Returns (rects, row|None, col, val); rects is (rows, 2) int64."""
    offs, cfs = sort_offsets(offsets, coeffs)
    ext = np.array(extent, dtype=np.int64)
    off = np.array(offs, dtype=np.int64).reshape(len(offs), len(ext))
    n = stencil_size(extent, offs)
    rows = int(np.prod(ext))
    col = np.zeros(n, dtype=np.int64)
    which = np.zeros(n, dtype=np.int32)
    row = np.zeros(n, dtype=np.int64) if coo else None
    rects = np.zeros((rows, 2), dtype=np.int64)
    got = lib().oracle_stencil_fill(ctypes.c_int(len(ext)), _ptr(ext), ctypes.c_int(len(off)), _ptr(off),
                                    _ptr(row) if coo else None, _ptr(col), _ptr(rects), _ptr(which))
    assert got == n
    val = np.array(cfs, dtype=dtype)[which]
    return rects, row, col, val


def laplacian_1d(n: int, dtype=np.float64):
    """Closed forms of the 1-D negative Laplacian fill tasks (src/ExampleSystems.cpp:311-321,
    382-389, 448-466).  Returns (rects, row, col, val)."""
    nnz = 3 * n - 2
    k = np.arange(nnz, dtype=np.int64)
    row = (k + 1) // 3
    col = k - 2 * ((k + 1) // 3)
    val = np.where(k % 3 != 0, -1.0, 2.0).astype(dtype)
    rects = np.zeros((n, 2), dtype=np.int64)
    r = np.arange(n, dtype=np.int64)
    rects[:, 0] = 3 * r - 1
    rects[:, 1] = 3 * r + 1
    rects[0] = (0, 1)
    rects[n - 1] = (3 * n - 4, 3 * n - 3)
    return rects, row, col, val


# The stencils test/BenchmarkStencil.cpp builds (:35-39, 48-54, 63-71, 80-124), by -dim flag.
def benchmark_stencil_offsets(dim_flag: int):
    if dim_flag == 1:
        return [(0,), (-1,), (1,)], [2.0, -1.0, -1.0]
    if dim_flag == 2:
        return [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)], [4.0, -1.0, -1.0, -1.0, -1.0]
    if dim_flag == 3:
        offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
        return offs, [6.0] + [-1.0] * 6
    if dim_flag == 4:
        offs, cfs = [], []
        for a in (-1, 0, 1):
            for b in (-1, 0, 1):
                for c in (-1, 0, 1):
                    nz = abs(a) + abs(b) + abs(c)
                    offs.append((a, b, c))
                    cfs.append({0: 88.0 / 26.0, 1: -6.0 / 26.0, 2: -3.0 / 26.0, 3: -2.0 / 26.0}[nz])
        return offs, cfs
    raise ValueError("INVALID DIM")


# ------------------------------------------------------------------ task ids -------------

def task_id_t(block, t):
    return int(lib().oracle_task_id_t(block, t))


def task_id_di(block, n, i):
    return int(lib().oracle_task_id_di(block, n, i))


def task_id_tdi(block, t, n, i):
    return int(lib().oracle_task_id_tdi(block, t, n, i))


def task_id_tdddiii(block, t, n1, n2, n3, i1, i2, i3):
    return int(lib().oracle_task_id_tdddiii(block, t, n1, n2, n3, i1, i2, i3))


def random_fill(n: int, seed: int, first_index: int, dtype=np.float64) -> np.ndarray:
    """What lsk_random_fill_* writes: one splitmix64 step of seed + (index + 1) * golden ratio per
    global index, top 53 bits / 2^53 (float64) or top 24 bits / 2^24 (float32).  (The reference's
    RandomFillTask, src/UtilityTasks.cpp:128-166, is seeded from std::random_device: there is
    nothing to be bit-exact with; this pins the replacement.)"""
    with np.errstate(over="ignore"):
        idx = np.arange(first_index + 1, first_index + 1 + n, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    if np.dtype(dtype) == np.float64:
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
