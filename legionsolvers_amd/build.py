"""Build recipe for liblsk.so: the HIP kernels and the C ABI of include/lsk.h, for gfx950.

hipcc cross-compiles without a GPU, so this runs in any ROCm container.  The library is
written next to this file (in-tree) so that it travels with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")
LIB_PATH = os.path.join(PKG_DIR, "liblsk.so")

SOURCES = ["shim.hip", "blas1.hip", "csr_spmv.hip", "sliced_spmv.hip", "coo_spmv.hip", "coo_binned.hip", "setup.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC_DIR, "lsk_common.hpp"), os.path.join(INCLUDE_DIR, "lsk.h")]

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # Nothing is fused behind the source's back: every fma in the kernels is written out.
    "-ffp-contract=off",
    # No packed-FP32 VALU forms (some have produced wrong results on these machines).
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: liblsk.so cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC_DIR, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into liblsk.so; returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    obj_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(obj_dir, exist_ok=True)
    objects = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, "-I", INCLUDE_DIR, "-I", CSRC_DIR, "-c",
               os.path.join(CSRC_DIR, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objects.append(obj)
    failed = False
    for src, proc in procs:
        out, _ = proc.communicate()
        text = out.decode(errors="replace")
        # hipcc's host pass does not know the packed-fp32 feature name: harmless noise.
        lines = [l for l in text.splitlines() if "packed-fp32-ops" not in l]
        if proc.returncode != 0:
            failed = True
            print(f"[build] {src} FAILED\n" + "\n".join(lines), file=sys.stderr)
        elif verbose and any(l.strip() for l in lines):
            print("\n".join(lines))
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objects, "-o", LIB_PATH,
            f"-L{ROCM}/lib", "-lrccl", f"-Wl,-rpath,{ROCM}/lib"]
    if verbose:
        print("[build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
