"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/lsk.h
declares, and its host-only entry points (get_alpha, task ids, argument validation) behave.
No compute call is made: there is no GPU here."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from legionsolvers_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) > 60
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/lsk.h but not exported"
    # the ctypes prototype table covers the header exactly
    assert sorted(_lib.PROTOTYPES) == declared
    assert lib.lsk_version().decode().startswith("lsk ")


def test_get_alpha_bit_exact_with_oracle():
    lib = _lib.lib()
    rng = np.random.default_rng(7)
    for dt, suf in ((np.float64, "f64"), (np.float32, "f32")):
        for nf in range(5):
            for _ in range(50):
                f = (rng.standard_normal(4) * 10.0 ** int(rng.integers(-3, 3))).astype(dt)
                out = np.zeros(1, dtype=dt)
                rc = getattr(lib, f"lsk_get_alpha_{suf}")(nf, f.ctypes.data, out.ctypes.data)
                assert rc == 0
                want = oracle.get_alpha(f[:nf], dt)
                assert out[0].tobytes() == np.asarray(want, dtype=dt).tobytes()
        out = np.zeros(1, dtype=dt)
        assert getattr(lib, f"lsk_get_alpha_{suf}")(5, f.ctypes.data, out.ctypes.data) == -2
        assert b"futures" in lib.lsk_last_error()


def test_task_ids_bit_exact_with_oracle():
    lib = _lib.lib()
    for block in (0, 11, 14, 18, 27):
        for t in range(2):
            assert lib.lsk_task_id_t(block, t) == oracle.task_id_t(block, t)
            for n in (1, 2, 3):
                for i in range(3):
                    assert lib.lsk_task_id_di(block, n, i) == oracle.task_id_di(block, n, i)
                    assert lib.lsk_task_id_tdi(block, t, n, i) == oracle.task_id_tdi(block, t, n, i)
            for combo in [(1, 1, 1, 2, 2, 2), (1, 2, 3, 0, 1, 2), (3, 3, 3, 2, 2, 2), (2, 1, 2, 1, 0, 1)]:
                assert lib.lsk_task_id_tdddiii(block, t, *combo) == oracle.task_id_tdddiii(block, t, *combo)
    # the ids LegionSolvers registers its double-precision 1-D variants under
    assert lib.lsk_task_id_tdi(12, 1, 1, 2) == 500001 + 5184 * 12 + 19        # axpy_double_1_longlong
    assert lib.lsk_task_id_tdddiii(18, 1, 1, 1, 1, 2, 2, 2) == 500001 + 5184 * 18 + 127  # csr_matvec


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before anything is launched."""
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    assert lib.lsk_scal_f64(-1, 1.0, null, null) == -2
    assert b"negative" in lib.lsk_last_error()
    assert lib.lsk_scal_f64(0, 1.0, null, null) == 0            # empty domain: nothing to do
    assert lib.lsk_axpy_f32(8, 1.0, null, null, null) == -2     # null vectors
    assert lib.lsk_csr_spmv_f64(4, 4, null, 0, null, null, null, 0, 3, null, 2, null) == -2  # beta
    assert lib.lsk_csr_spmv_f64(0, 0, null, 0, null, null, null, 0, -1, null, 0, null) == 0
    assert lib.lsk_coo_spmv_f64(-3, null, null, null, null, 0, 0, null, 0, 0, null) == -2
    assert lib.lsk_coo_spmv_f64(0, null, null, null, null, 0, 0, null, 0, 0, null) == 0
    assert lib.lsk_scalar_op_f64(99, null, null, null, null) == -2
    rects = np.zeros(9, dtype=np.int64)
    misaligned = ctypes.c_void_p(rects.ctypes.data + 8 if rects.ctypes.data % 16 == 0 else rects.ctypes.data)
    buf = ctypes.c_void_p(rects.ctypes.data)
    assert lib.lsk_csr_spmv_f64(2, 2, misaligned, 0, buf, buf, buf, 0, 1, buf, 0, null) == -3
    ext = (ctypes.c_int64 * 1)(8)
    offs = (ctypes.c_int64 * 2)(1, -1)  # not sorted
    assert lib.lsk_stencil_count(1, ext, 2, offs, 0, 7, buf, null) == -2
    with pytest.raises(_lib.LskError):
        _lib.check(-2)


def test_product_path_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from legionsolvers_amd import Context
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Context()


def test_abort_on_error_matches_the_reference_behaviour():
    """With lsk_set_abort_on_error(1) a failing call prints its message and aborts, as the
    reference's CHECK_* macros do (src/CUDAUtilities.cpp:13-63).  Run in a child process."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from legionsolvers_amd import _lib; "
            "L = _lib.lib(); L.lsk_set_abort_on_error(1); "
            "L.lsk_scal_f64(-1, 1.0, ctypes.c_void_p(0), ctypes.c_void_p(0)); print('survived')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == -6, (res.returncode, res.stderr[-500:])  # SIGABRT
    assert "[lsk]" in res.stderr and "negative" in res.stderr
    assert "survived" not in res.stdout


def test_header_is_plain_c_and_the_native_callers_compile():
    """include/lsk.h compiles as strict C99 (what a cgo / FFI binding would parse), and the C and
    C++ callers above it (examples/cg_c_abi.c, host_cpp/) compile against it; no GPU, no linking."""
    import shutil
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(repo, "include")
    if not (shutil.which("gcc") and shutil.which("g++")):
        pytest.skip("no host compiler")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                 os.path.join(inc, "lsk.h")],
                ["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", inc,
                 os.path.join(repo, "examples", "cg_c_abi.c")],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc,
                 "-I", os.path.join(repo, "host_cpp"), os.path.join(repo, "host_cpp", "benchmark_stencil.cpp")]):
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, " ".join(cmd) + "\n" + res.stderr[-3000:]


def test_callers_above_the_boundary_use_only_declared_entry_points():
    """The Legion task bodies (never compiled here: no Legion in this image), the C example and the C++ host name
    no library symbol that include/lsk.h does not declare: the text cannot drift from the ABI unnoticed."""
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(repo, "include", "lsk.h")).read()
    declared = set(_lib.declared_symbols()) | set(re.findall(r"\b(lsk_\w+_t)\b", header))
    own = {"lsk_legion", "lsk_legion_tasks", "lsk_legion_matrix_values_changed", "lsk_common"}   # the glue's own names
    for path in ("legion_glue/lsk_legion_tasks.cpp", "legion_glue/lsk_legion.h", "examples/cg_c_abi.c",
                 "host_cpp/legion_solvers_host.hpp", "host_cpp/benchmark_stencil.cpp", "host_cpp/stencil_generator.hpp"):
        text = open(os.path.join(repo, path)).read()
        for name in sorted(set(re.findall(r"\b(lsk_\w+)\b", text)) - own):
            if name.endswith("_"):   # a macro pastes the rest of the name (the entry type) behind it
                assert any(d.startswith(name) for d in declared), f"{path}: nothing declared begins with {name}"
            else:
                assert name in declared or (name + "_f32" in declared and name + "_f64" in declared), f"{path}: {name}"
