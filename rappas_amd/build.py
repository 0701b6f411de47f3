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
# the same sources with -DRK_DEV_KNOBS: the developer / test knobs of DESIGN.md section 10 (environment variables, the shard-failure
# injector of the re-queue test) exist only in this build; the product library reads no environment variable
DEV_SO = os.path.join(ROOT, "rappas_amd", "librappas_place_dev.so")

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
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(ROOT, "include", "rappas_place.h"))
    return srcs


def build_engine(force=False, verbose=False):
    """One object per .hip translation unit (compiled side by side), then one shared library; product and developer build."""
    srcs = engine_sources()
    if not force and not _stale(ENGINE_SO, srcs) and not _stale(DEV_SO, srcs):
        return ENGINE_SO
    units = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "rk_kernels.hip"]  # rk_kernels.hip is #included by rk_engine.hip
    host_units = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".cpp")]  # host-only code (the SIMD read packer): plain C++
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    procs = []
    for tag, extra in (("", []), ("_dev", ["-DRK_DEV_KNOBS"])):
        for u in units:
            obj = os.path.join(objdir, u.replace(".hip", tag + ".o"))
            cmd = [_hipcc()] + flags + extra + ["-c", "-o", obj, os.path.join(CSRC, u)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((tag, cmd, obj, subprocess.Popen(cmd, cwd=ROOT)))
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    for u in host_units:  # no knobs in there: one object serves both libraries
        obj = os.path.join(objdir, u.replace(".cpp", ".o"))
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-c", "-o", obj, os.path.join(CSRC, u)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(("both", cmd, obj, subprocess.Popen(cmd, cwd=ROOT)))
    objs = {"": [], "_dev": [], "both": []}
    for tag, cmd, obj, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs[tag].append(obj)
    for tag, so in (("", ENGINE_SO), ("_dev", DEV_SO)):
        # -Bsymbolic: calls between the library's own entry points stay inside it when both builds are loaded into one process
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", so] + objs[tag] + objs["both"]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, check=True, cwd=ROOT)
        check_isa(so, verbose)
    return ENGINE_SO


def check_isa(so, verbose=False):
    """Refuse a build whose kernels hold the gfx950 64-bit-shift pattern that reads its count from v0 (tools/check_isa.py;
    hipcc emits it now and then, round 2's 5-bit packer had it): silent corruption is worse than a failed build."""
    import contextlib
    import io
    from .tools import check_isa as ci
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bad = ci.check([so])
    if verbose or bad:
        print(buf.getvalue(), end="", flush=True)
    if bad:
        os.replace(so, so + ".rejected")
        raise RuntimeError(f"{so}: {bad} 64-bit shift(s) with the count in the last allocated VGPR (gfx950 erratum, DESIGN.md 4.4): "
                           "change the source until the register allocation moves (the library was renamed to *.rejected)")


HOST_BIN = os.path.join(ROOT, "rappas_amd", "bin", "rk_place")
HOST_SRC = [os.path.join(CSRC, "host", f) for f in ("rk_place_main.cpp", "rk_hostio.hpp", "rk_javaser.hpp", "rk_fastio.hpp")]


def build_host_tools(force=False, verbose=False):
    """rk_place: the native (C++17) FASTA + --jsondb -> .jplace driver over librappas_place.so."""
    build_engine(force=False, verbose=verbose)
    if not force and not _stale(HOST_BIN, HOST_SRC + [ENGINE_SO, os.path.join(ROOT, "include", "rappas_place.h")]):
        return HOST_BIN
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-o", HOST_BIN, HOST_SRC[0], "-L" + os.path.dirname(ENGINE_SO),
           "-lrappas_place", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return HOST_BIN


if __name__ == "__main__":
    build_engine(force="--force" in sys.argv, verbose=True)
    print("built:", ENGINE_SO)
    print("built:", build_host_tools(force="--force" in sys.argv, verbose=True))
