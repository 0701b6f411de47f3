"""Build the native pieces in-tree: the HIP engine (gfx950) and, separately, the CPU oracle used by tests.

`python -m rappas_amd.build` or `rappas_amd.build.build_engine()`.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rappas_amd", "csrc")
ENGINE_SO = os.path.join(ROOT, "rappas_amd", "librappas_place.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")

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
    srcs = engine_sources()
    if not force and not _stale(ENGINE_SO, srcs):
        return ENGINE_SO
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", ENGINE_SO, os.path.join(CSRC, "rk_engine.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=ROOT)
    return ENGINE_SO


def build_oracle(force=False, verbose=False):
    """Test infrastructure only (see oracle/rappas_oracle.h)."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rappas_oracle.c", "rappas_oracle.h", "Makefile")]
    if not force and not _stale(ORACLE_SO, srcs):
        return ORACLE_SO
    cmd = ["make", "-C", ORACLE_DIR] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return ORACLE_SO


if __name__ == "__main__":
    build_engine(force="--force" in sys.argv, verbose=True)
    build_oracle(force="--force" in sys.argv, verbose=True)
    print("built:", ENGINE_SO, ORACLE_SO)
