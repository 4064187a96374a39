"""Build the in-tree shared library (HIP kernels + C ABI + C++ host layer) for gfx950.

    python -m detqmc_amd.build            # or detqmc_amd.build.build()

hipcc cross-compiles without a GPU; the resulting detqmc_amd/lib/libdetqmc_amd.so is git-ignored but
travels to the GPU box with the source snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdetqmc_amd.so")

SOURCES = [
    "kernels_bmult.hip",
    "kernels_gemm.hip",
    "kernels_svd.hip",
    "kernels_update.hip",
    "kernels_qr.hip",
    "kernels_lu.hip",
    "kernels_measure.hip",
    "kernels_hubbard.hip",
    "dqmc_context.hip",
    os.path.join("host", "dsfmt19937.cpp"),
    os.path.join("host", "detsdw.cpp"),
    os.path.join("host", "dethubbard.cpp"),
]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "dqmc_internal.h"), os.path.join(CSRC, "host", "detsdw.h"),
               os.path.join(CSRC, "host", "dsfmt19937.h"),
               os.path.join(CSRC, "host", "dethubbard.h"),
               os.path.join(HERE, "..", "include", "dqmc_hip.h"), os.path.join(HERE, "..", "include", "detsdw_host.h"),
               os.path.join(HERE, "..", "include", "dethubbard_host.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    flags += os.environ.get("DQMC_BUILD_DEFINES", "").split()       # developer builds, e.g. -DDQMC_DECIDE_TIMING (use with force=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or hdr_time > os.path.getmtime(obj):
            cmd = [hipcc] + flags + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
