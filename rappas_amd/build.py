"""Build the native engine in-tree (gfx950 HIP kernels + C ABI).

`python -m rappas_amd.build` or `rappas_amd.build.build_engine()`.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rappas_amd", "csrc")
ENGINE_SO = os.path.join(ROOT, "rappas_amd", "librappas_place.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",  # bit parity: fl(v - T) and fl(S + d) must stay two roundings
    "-Wall", "-Wno-unused-function",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 engine)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def engine_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "rappas_place.h"))
    return srcs


def build_engine(force=False, verbose=False):
    """One object per .hip translation unit (compiled side by side), then one shared library."""
    srcs = engine_sources()
    if not force and not _stale(ENGINE_SO, srcs):
        return ENGINE_SO
    units = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "rk_kernels.hip"]  # rk_kernels.hip is #included by rk_engine.hip
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    procs = []
    for u in units:
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        cmd = [_hipcc()] + flags + ["-c", "-o", obj, os.path.join(CSRC, u)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=ROOT)))
    objs = []
    for cmd, obj, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", ENGINE_SO] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=ROOT)
    return ENGINE_SO


HOST_BIN = os.path.join(ROOT, "rappas_amd", "bin", "rk_place")
HOST_SRC = [os.path.join(CSRC, "host", f) for f in ("rk_place_main.cpp", "rk_hostio.hpp", "rk_javaser.hpp")]


def build_host_tools(force=False, verbose=False):
    """rk_place: the native (C++17) FASTA + --jsondb -> .jplace driver over librappas_place.so."""
    build_engine(force=False, verbose=verbose)
    if not force and not _stale(HOST_BIN, HOST_SRC + [ENGINE_SO, os.path.join(ROOT, "include", "rappas_place.h")]):
        return HOST_BIN
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", HOST_BIN, HOST_SRC[0], "-L" + os.path.dirname(ENGINE_SO),
           "-lrappas_place", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return HOST_BIN


if __name__ == "__main__":
    build_engine(force="--force" in sys.argv, verbose=True)
    print("built:", ENGINE_SO)
    print("built:", build_host_tools(force="--force" in sys.argv, verbose=True))
