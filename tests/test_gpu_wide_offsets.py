"""The 64-bit-offset code paths (row blobs >= 4 GiB: ITEM64 / OFF64 kernels) on small databases: a developer build of the engine
with the 32-bit limit lowered to 1000 bytes runs the parity suite in a subprocess."""
import os
import subprocess
import sys

import pytest

from rappas_amd import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_with_64bit_offsets_forced(tmp_path):
    so = str(tmp_path / "librk_off64.so")
    cmd = [build._hipcc()] + [f for f in build.HIPCC_FLAGS] + ["-DRK_FIT32_LIMIT=1000u", "-DRK_DEV_KNOBS", "-o", so,
                                                               os.path.join(build.CSRC, "rk_engine.hip"), os.path.join(build.CSRC, "rk_pack_host.cpp")]
    subprocess.run(cmd, check=True, cwd=ROOT)
    env = dict(os.environ, RK_LIB=so)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "-k", "c1_full or c2_scaled or c4_protein or large_tree or ambiguity or golden or randomised"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
    # and the variant really is the wide one
    probe = ("import rappas_amd as ra; from rappas_amd import synth; db = ra.PhyloKmerDB.from_synth(synth.make_config_db('C1')); "
             "print(db.kernel_name())")
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120).stdout
    assert "ITEM64" in out, out
