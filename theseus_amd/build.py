"""Builds libtheseus_hip.so (the C ABI of include/theseus_hip.h) with hipcc for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Usage:  python -m theseus_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtheseus_hip.so")
SOURCES = ["pg_kernels.hip", "chol_kernels.hip", "vjp_kernels.hip", "block_kernels.hip", "pg2_kernels.hip", "ba_kernels.hip", "vjp2_kernels.hip", "pgso3_kernels.hip", "pgso2_kernels.hip", "ba_vjp_kernels.hip", "vjpso3_kernels.hip", "vjp_unroll_kernels.hip", "vjp_unroll3_kernels.hip", "vjp_unroll_ba_kernels.hip"]
HEADERS = ["lie.cuh", "lie_se2.cuh", "lie_so3.cuh", "pg3_generic.cuh", "vjp_se3.cuh", "unroll_se3.cuh", "unroll_g3.cuh", "unroll_ba.cuh", "dual.cuh", "robust.cuh", "common.cuh", os.path.join("..", "..", "include", "theseus_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    newest_hdr = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _mtime(o) < max(_mtime(s), newest_hdr):
            cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print("[theseus_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB, *objs]
        if verbose:
            print("[theseus_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
