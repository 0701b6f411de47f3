"""CPU: the C-ABI shared library loads and exports every symbol include/rappas_place.h declares; argument
validation that needs no GPU.  No compute calls here."""
import ctypes as C
import os
import time
import re

import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import _lib, synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rappas_place.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    ra.build.build_engine()
    lib = C.CDLL(_lib.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rappas_place.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), "ctypes binding and header disagree"


def test_version_and_thresholds():
    lib = _lib.load()
    assert lib.rk_version() == 101
    for ns, k in [(4, 8), (4, 10), (20, 5), (4, 12)]:
        a, b = C.c_float(), C.c_float()
        lib.rk_thresholds(1.5, ns, k, C.byref(a), C.byref(b))
        p, t = O.thresholds(1.5, ns, k)
        assert np.float32(a.value) == p and np.float32(b.value) == t


def test_struct_layouts_match_header():
    # sizes the C compiler gives the header's structs (natural alignment, LP64)
    assert C.sizeof(_lib.rk_db_desc) == 72
    assert C.sizeof(_lib.rk_params) == 16
    assert C.sizeof(_lib.rk_result) == 40
    assert C.sizeof(_lib.rk_counters) == 48
    assert C.sizeof(_lib.rk_db_info) == 80


def test_no_cpu_fallback(gpu_available):
    """Without a device the product path must fail loudly (RK_ERR_NO_DEVICE), never compute on the CPU."""
    if gpu_available:
        pytest.skip("GPU present")
    sdb = synth.make_config_db("C1")
    with pytest.raises(ra.RkError) as ei:
        ra.PhyloKmerDB.from_synth(sdb)
    assert ei.value.code == _lib.RK_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_argument_validation_without_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rk_db_create(None, C.byref(h)) == _lib.RK_ERR_INVALID
    d = _lib.rk_db_desc(alphabet=7, k=8, n_branches=10)
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_INVALID
    assert b"alphabet" in lib.rk_last_error()
    d = _lib.rk_db_desc(alphabet=4, k=32, n_branches=10)  # 2 bits per base in a 64-bit code: k <= 31
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_UNSUPPORTED
    d = _lib.rk_db_desc(alphabet=4, k=8, n_branches=70000)
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_INVALID
    assert lib.rk_place_batch(None, None, 0, None, None, None, None) == _lib.RK_ERR_INVALID
    assert lib.rk_set_lanes_per_read(None, 16) == _lib.RK_ERR_INVALID
    lib.rk_db_destroy(None)  # no-op


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under rappas_amd/ or include/ may reference it."""
    for base in ("rappas_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".hpp")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "liboracle" not in txt and "rappas_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def _validate(sdb, **kw):
    args = dict(key_codes=sdb.key_codes, row_offsets=sdb.row_offsets, branch_ids=sdb.branch_ids, scores=sdb.scores)
    args.update(kw)
    return ra.validate_db(sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr_log10, sdb.thr, **args)


def test_db_validate_builds_image_without_device():
    """rk_db_validate runs the whole host-side image construction (no HIP): layout invariants of DESIGN.md section 3."""
    sdb = synth.make_config_db("C1")
    info = _validate(sdb)
    assert info.table_mode == _lib.RK_TABLE_DIRECT and info.bits_per_symbol == 2 and info.device == -1
    assert info.n_keys == sdb.n_keys and info.n_entries == sdb.n_entries
    lens = np.diff(sdb.row_offsets.astype(np.int64))
    assert info.max_row_len == lens.max()
    units = ((lens + 15) // 16).sum() + 1                   # rows padded to whole 128-byte units + the reserved unit 0
    assert info.rows_bytes == units * 128
    per = 24 if lens.max() <= 240 else 12                    # compact table: 16 bytes per 24 k-mers (4-bit unit counts) while no row
    assert info.table_bytes == ((4 ** 8 + per - 1) // per) * 16  # exceeds 15 units, per 12 k-mers (bytes) otherwise
    assert _validate(sdb, table_mode=_lib.RK_TABLE_DIRECT8).table_bytes == 4 ** 8 * 8
    h = _validate(sdb, table_mode=_lib.RK_TABLE_HASH)
    assert h.table_mode == _lib.RK_TABLE_HASH and h.table_slots >= 2 * sdb.n_keys and h.table_bytes == h.table_slots * 16
    aa = synth.make_db(20, 3, 30, 500, 2000, seed=1)
    assert _validate(aa).bits_per_symbol == 5
    big = synth.make_db(4, 6, 19999, 300, 150000, seed=2)    # large tree, long rows: indexed rows (+1 line per row), 8-byte descriptors
    bi = _validate(big)
    bl = np.diff(big.row_offsets.astype(np.int64))
    # [index line][u16 branch[lenp]][f32 score[lenp]] with lenp padded to 32 entries: 6 bytes per entry
    assert bi.table_mode == _lib.RK_TABLE_DIRECT8 and bi.rows_bytes == ((((bl + 31) // 32 * 32) * 6 // 64).sum() + len(bl) + 1) * 64
    long_rows = synth.make_db(4, 6, 8000, 30, 200000, seed=3)  # rows > 4080 entries do not fit the compact table
    assert np.diff(long_rows.row_offsets.astype(np.int64)).max() > 4080
    assert _validate(long_rows).table_mode == _lib.RK_TABLE_DIRECT8


def test_db_validate_rejects_bad_input():
    sdb = synth.make_db(4, 6, 31, 200, 900, seed=1)
    bad = sdb.branch_ids.copy(); bad[1] = bad[0]
    with pytest.raises(ra.RkError, match="repeated inside row"):
        _validate(sdb, branch_ids=bad)
    bad = sdb.branch_ids.copy(); bad[5] = 31
    with pytest.raises(ra.RkError, match=">= n_branches"):
        _validate(sdb, branch_ids=bad)
    dup = sdb.key_codes.copy(); dup[3] = dup[2]
    for mode in (_lib.RK_TABLE_DIRECT, _lib.RK_TABLE_DIRECT8, _lib.RK_TABLE_HASH):
        with pytest.raises(ra.RkError, match="duplicate k-mer"):
            _validate(sdb, key_codes=dup, table_mode=mode)
    nan = sdb.scores.copy(); nan[0] = np.inf
    with pytest.raises(ra.RkError, match="non-finite"):
        _validate(sdb, scores=nan)
    badcode = sdb.key_codes.copy(); badcode[0] = 4 ** 6
    with pytest.raises(ra.RkError, match="invalid k-mer code"):
        _validate(sdb, key_codes=badcode)
    off = sdb.row_offsets.copy(); off[3] = off[2]
    with pytest.raises(ra.RkError, match="empty row"):
        _validate(sdb, row_offsets=off)
    aa = synth.make_db(20, 3, 30, 100, 400, seed=1)
    codes = aa.key_codes.copy(); codes[0] = 21            # digit 21 >= 20
    with pytest.raises(ra.RkError, match="invalid k-mer code"):
        _validate(aa, key_codes=codes)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py --gpus N either starts N ranks itself or is started as one of N; anything else is an error, not a 1-rank run"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def _bench(args, extra_env=None, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "RK_BENCH_JOB_DIR")}
    env["RK_BENCH_ECHO_RANK"] = "1"
    env.update(extra_env or {})
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.monotonic() - t0


def test_bench_gpus_n_launches_n_ranks_and_fails_if_a_rank_fails():
    """without a GPU every rank exits with the 'needs an MI355X' error: the launcher must report that, not hide it"""
    p, _ = _bench(["--gpus", "2", "--reads", "1000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--verify", "0"])
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "ranks failed" in p.stderr and "needs an MI355X" in p.stderr
        assert p.stderr.count("bench rank") == 2 and "rank 0/2" in p.stderr and "rank 1/2" in p.stderr


@pytest.mark.parametrize("backend,n", [("none", 4), ("none", 8), ("gloo", 4), ("gloo", 8), ("auto", 4)])
def test_bench_rank_bookkeeping_with_4_and_8_ranks(backend, n):
    """--rehearse-launch: N fresh rank processes, rendezvous, two barriers, MAX over ranks, per-rank results -- no GPU, no
    placement.  Odd ranks take twice as long per step: the job's time must be the slow ranks' time.  `auto` has no RCCL
    here, so every rank must take the same fall-back to `none`."""
    import json
    p, _ = _bench(["--gpus", str(n), "--rehearse-launch", "--dist-backend", backend, "--steps", "20"])
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == n and line["value"] is None and "rehearsal" in line["metric"]
    assert line["dist_backend"].startswith("none" if backend == "auto" else backend)
    assert len(line["per_rank"]) == n and line["per_rank_min"] == min(line["per_rank"])
    assert line["per_rank"][0] > 1.5 * line["per_rank"][1]            # per-rank rates are the ranks' own
    assert 10.0 <= line["ms_per_step"] < 14.0                         # the job runs at the slow ranks' 10 ms per step
    assert sorted(int(x.split("/")[0]) for x in __import__("re").findall(r"bench rank (\d+/\d+)", p.stderr)) == list(range(n))


@pytest.mark.parametrize("backend", ["gloo", "none"])
def test_bench_a_rank_that_dies_before_the_rendezvous_ends_the_job_fast(backend):
    """rank 2 of 4 dies before the rendezvous: the others would wait for it until the store / file time-out; the parent
    must stop them and exit non-zero within seconds, naming the rank and showing its stderr"""
    p, dt = _bench(["--gpus", "4", "--rehearse-launch", "--dist-backend", backend, "--steps", "5"], {"RK_BENCH_REHEARSE_DIE_RANK": "2"})
    assert p.returncode != 0 and dt < 30.0, (p.returncode, dt)
    assert "rank 2 exited with code 3" in p.stderr and "dying before the rendezvous" in p.stderr and "were stopped" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_a_process_group_that_cannot_be_set_up_is_retried_without_one():
    """--dist-backend nccl without a GPU per rank: the set fails in the process-group set-up, the parent starts a fresh set
    of ranks with --dist-backend none and the job completes"""
    import json
    p, _ = _bench(["--gpus", "2", "--rehearse-launch", "--dist-backend", "nccl", "--steps", "5"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert "fresh set of ranks with --dist-backend none" in p.stderr
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dist_backend"] == "none" and len(line["per_rank"]) == 2


def test_bench_under_torch_distributed_run():
    """the driver's launch line (torch.distributed.run, env:// rendezvous), rehearsed on CPU with gloo and with no process group"""
    import json
    import socket
    import subprocess
    import sys
    for backend in ("gloo", "none"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RK_BENCH_JOB_DIR")}
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--dist-backend", backend,
                            "--steps", "5"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["dist_backend"] == backend and len(line["per_rank"]) == 2


def test_built_kernels_do_not_hold_the_gfx950_shift_count_erratum():
    """v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 with the count in the last allocated VGPR read it from v0 now and then on
    gfx950 (profiles/r03_shift_count_erratum.txt); the build refuses such code, and so does this test (disassembly, no GPU)"""
    from rappas_amd.tools import check_isa
    assert check_isa.check([ra.build.ENGINE_SO]) == 0


def test_placement_kernels_hold_no_64bit_shift_by_a_per_lane_count():
    """round 4: the placement kernels are written without 64-bit shifts whose count is a VGPR (ballots cut into lane groups, counts
    below a lane, window masks and k-mer codes by 32-bit funnel shifts: rk_kernels.hip group_bits / count_below / bits_below / funnel96
    / place_bits), so the gfx950 erratum has nothing to land on whatever the register allocation does; the scanner stays as a tripwire"""
    from rappas_amd.tools import check_isa
    census = check_isa.variable_shift_census(ra.build.ENGINE_SO)
    hot = {k: n for k, n in census.items() if any(t in k for t in ("place_", "pack_reads_kernel", "retile_", "fetch_row_kernel"))}
    assert len(hot) > 80 and sum(hot.values()) == 0, {k: n for k, n in hot.items() if n}


def test_the_erratum_checker_sees_the_pattern(tmp_path):
    """the checker on a code object that holds the pattern: hipcc's code for the 5-bit packer as round 2 had it"""
    import shutil
    import subprocess
    from rappas_amd.tools import check_isa
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pack_hazard")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-o", exe, os.path.join(ROOT, "scripts", "ubench", "pack_hazard.hip")], check=True)
    with __import__("tempfile").TemporaryDirectory() as d:
        hits = [h for co in check_isa.device_code_objects(exe, d) for h in check_isa.scan(co)[0]]
    # variants A, C, F, G: exactly the four that lose bits on the GPU (D, E, H never do)
    assert sorted(h[0] for h in hits) == ["_Z12pack_variantILi%dEEvPKhPKyyjS1_jPjS4_S4_S4_" % v for v in (65, 67, 70, 71)]
    assert all(h[2] == "v23" and h[3] == 24 for h in hits)


def test_the_erratum_checker_on_hand_written_probes_and_when_it_cannot_see(tmp_path):
    """the checker fails closed: the hand-written probe kernels of scripts/ubench/hazard (one 64-bit shift with its count in the last
    allocated VGPR -- the ones that differ on the hardware -- against the controls that never do), a library without device code, and
    a shift in code that has no register metadata"""
    import importlib.util
    import subprocess
    from rappas_amd.tools import check_isa
    spec = importlib.util.spec_from_file_location("make_shift_probe", os.path.join(ROOT, "scripts", "ubench", "hazard", "make_shift_probe.py"))
    msp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(msp)
    clang, lld = os.path.join(check_isa.LLVM_BIN, "clang"), os.path.join(check_isa.LLVM_BIN, "ld.lld")
    want = {"lshr_count_top_24": 1, "lshl_count_top": 1, "ashr_count_top": 1, "lshr_count_top_minus1": 0, "lshr_src_pair_top": 0, "lshr32_count_top": 0}
    for name, hits_wanted in want.items():
        n, setup, ref, test, between = msp.tests()[name]
        src, obj, co = (str(tmp_path / (name + e)) for e in (".s", ".o", ".co"))
        open(src, "w").write(msp.gen(n, setup, ref, test, between))
        subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj], check=True)
        subprocess.run([lld, "-shared", obj, "-o", co], check=True)
        hits, n_shifts, n_kernels = check_isa.scan(co)
        assert n_kernels == 1 and len(hits) == hits_wanted, (name, hits, n_shifts)
    with pytest.raises(check_isa.CheckerBlind):
        check_isa.check([os.path.join(ROOT, "oracle", "liboracle.so")])  # no device code: nothing could be checked
    # a shift inside a symbol without a metadata entry (a device function that was not inlined): the conservative 7-mod-8 rule
    bare = tmp_path / "bare.s"
    bare.write_text(".amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"\n.text\n.globl helper\n.type helper,@function\nhelper:\n"
                    "\tv_lshrrev_b64 v[4:5], v23, v[4:5]\n\tv_lshrrev_b64 v[6:7], v22, v[4:5]\n\ts_setpc_b64 s[30:31]\n")
    subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(bare), "-o", str(tmp_path / "bare.o")], check=True)
    subprocess.run([lld, "-shared", str(tmp_path / "bare.o"), "-o", str(tmp_path / "bare.co")], check=True)
    hits, n_shifts, n_kernels = check_isa.scan(str(tmp_path / "bare.co"))
    assert n_kernels == 0 and n_shifts == 2 and [h[2] for h in hits] == ["v23"], hits


def test_product_library_reads_no_environment_variable():
    """the developer / test knobs (and the shard-failure injector) exist only in the -DRK_DEV_KNOBS build: a JVM hands its whole
    environment to the libraries it loads"""
    product = open(ra.build.ENGINE_SO, "rb").read()
    dev = open(ra.build.DEV_SO, "rb").read()
    for knob in (b"RK_TEST_FAIL_SHARD", b"RK_WINDOW_ALWAYS", b"RK_WG_PASSES", b"RK_NO_WINDOW", b"RK_BUILD_BATCH_NODES", b"RK_CHUNK_READS"):
        assert knob not in product, knob
        assert knob in dev, knob
    # (the library's one remaining getenv reference is rocPRIM's own ROCPRIM_USE_ATOMIC_BLOCK_ID, in rocprim/device/detail/ordered_block_id.hpp)
    import re
    env_like = set(re.findall(rb"RK_[A-Z][A-Z0-9_]{3,}", product)) - {b"RK_TABLE_AUTO", b"RK_TABLE_HASH", b"RK_TABLE_DIRECT", b"RK_TABLE_DIRECT8"}
    assert not {e for e in env_like if not e.startswith((b"RK_ERR", b"RK_FLAG", b"RK_AMB", b"RK_ALPHABET", b"RK_OK"))}, env_like
    src = "".join(open(os.path.join(ROOT, "rappas_amd", "csrc", f)).read() for f in ("rk_engine.hip", "rk_build.hip", "rk_kernels.hip"))
    assert "getenv(" not in src                                   # every knob goes through rk_knob (rk_internal.h)


@pytest.mark.parametrize("alphabet,k,length", [(4, 10, 150), (4, 8, 31), (4, 12, 250), (4, 20, 1000), (20, 5, 100), (20, 3, 32), (20, 8, 250), (20, 2, 7)])
def test_host_packer_equals_the_reference_packer(alphabet, k, length, monkeypatch):
    """rk_pack_reads (no GPU, no handle): the vector path (AVX2 + BMI2 blocks of 32 symbols) and the table-driven path
    (RK_PACK_SCALAR on the developer build) both give the records of the numpy reference packer, the reads' lengths and the flags
    of AmbigSequenceKnife.java:103-130 (ambiguous -> state 0 + AMBIGUOUS, unsupported -> state 0 + BAD_CHAR), for ragged, empty,
    too-short and too-long reads"""
    seq, off = synth.make_reads(alphabet, 5000, length, seed=11 + length, amb_rate=0.004, bad_rate=0.02, var_len=length)
    bits = 2 if alphabet == 4 else 5
    lens_true = (off[1:] - off[:-1]).astype(np.int64)
    want, _ = synth.pack_reads_numpy(alphabet, seq, off)
    letters = synth.DNA_LETTERS if alphabet == 4 else synth.AA_LETTERS
    plain = np.zeros(256, bool)
    plain[letters] = True
    plain[letters + 32] = True
    if alphabet == 4:
        plain[[ord("U"), ord("u")]] = True
    known = plain.copy()
    known[list(b"NRYSWKMBDHVnryswkmbdhv-." if alphabet == 4 else b"XBZJxbzj*-!")] = True
    idx = np.repeat(np.arange(len(lens_true)), lens_true)
    n_unknown = np.bincount(idx, weights=~known[seq], minlength=len(lens_true))
    n_amb = np.bincount(idx, weights=known[seq] & ~plain[seq], minlength=len(lens_true))
    for scalar in (False, True):
        if scalar:
            monkeypatch.setenv("RK_PACK_SCALAR", "1")
            monkeypatch.setattr(_lib, "_LIB", _lib.load_dev())
        packed, lens, flags = ra.pack_reads(alphabet, k, seq, off, threads=3)
        assert np.array_equal(packed, want), scalar
        assert np.array_equal(lens, lens_true)
        assert np.array_equal((flags & ra.RK_FLAG_BAD_CHAR) != 0, n_unknown > 0)
        assert np.array_equal((flags & ra.RK_FLAG_AMBIGUOUS) != 0, n_amb > 0)
        assert np.array_equal((flags & ra.RK_FLAG_TOO_SHORT) != 0, lens_true < k)
        # records one word too short for the longest reads: cut at the record's capacity and flagged
        wpr = max(1, packed.shape[1] - 1)
        cap = wpr * 32 // bits
        p2, l2, f2 = ra.pack_reads(alphabet, k, seq, off, words_per_read=wpr, threads=2)
        assert np.array_equal(l2, np.minimum(lens_true, cap)) and np.array_equal((f2 & ra.RK_FLAG_TOO_LONG) != 0, lens_true > cap)
        full = lens_true <= cap
        assert np.array_equal(p2[full], want[full][:, :wpr])


def test_host_packer_follows_the_convertUO_switch():
    seq = np.frombuffer(b"RHKUOuoC", dtype=np.uint8)
    off = np.array([0, 8], dtype=np.uint64)
    _, _, f0 = ra.pack_reads(20, 3, seq, off)
    p1, _, f1 = ra.pack_reads(20, 3, seq, off, convert_uo=True)
    assert f0[0] & ra.RK_FLAG_BAD_CHAR and not (f1[0] & ra.RK_FLAG_BAD_CHAR)
    states = [(int(p1[0, 0]) | int(p1[0, 1]) << 32) >> (5 * i) & 31 for i in range(8)]
    assert states == [0, 1, 2, 9, 14, 9, 14, 9]       # U -> C (9), O -> L (14): AAStates.java:118-123
