"""GPU (-m gpu): bench.py's contract -- `--gpus N` really starts N ranks (one process per GPU), every config named in
BASELINE.json has a bench leg, and the JSON line carries roofline + cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=900, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("backend", ["gloo", "none"])
def test_gpus_2_starts_two_ranks_and_reports_the_whole_job(backend):
    """two ranks rehearsed on the one GPU of this box: gloo for the barrier / max-over-ranks (RCCL needs one GPU per rank), and
    with no process group at all (file rendezvous + monotonic stamps)"""
    common = ["--reads", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pcie", "--verify", "300"]
    one = run_bench("--gpus", "1", *common)
    two = run_bench("--gpus", "2", "--force-device", "0", "--dist-backend", backend, *common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["sharding"].startswith("reads x2") and two["dist_backend"] == backend
    # value = reads of ALL ranks / max-over-ranks time
    assert abs(two["value"] - 2 * 200000 * 3 / (two["ms_per_step"] * 3 / 1e3)) / two["value"] < 1e-6
    # both ranks share one GPU here, so the whole-job rate stays near the one-rank rate (it would double on two GPUs)
    assert 0.4 * one["value"] < two["value"] < 1.6 * one["value"], (one["value"], two["value"])
    # every rank reports its own rate; no rank can be slower than the whole job's per-rank share
    assert len(two["per_rank"]) == 2 and two["per_rank_min"] * 2 >= two["value"] * 0.999
    for line in (one, two):
        assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
        assert line["verified_vs_oracle"]["reads"] == 300


def test_auto_backend_with_two_ranks_on_one_gpu_ends_with_a_result():
    """`auto` tries RCCL first; two ranks on ONE GPU cannot form an RCCL communicator, so this is the fall-back the first
    real multi-GPU run would take if RCCL could not be set up: either in-process (vote) or by a fresh set of ranks"""
    line = run_bench("--gpus", "2", "--force-device", "0", "--reads", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                     "--no-pcie", "--verify", "0", timeout=900, RK_BENCH_PG_TIMEOUT="45")
    assert line["n_gpus"] == 2 and len(line["per_rank"]) == 2 and line["dist_backend"].split()[0] in ("none", "nccl")


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


@pytest.mark.parametrize("config,reads", [("C1", "1000"), ("C4", "100000"), ("C3", "2000000"), ("T4k", "200000"), ("T40k", "100000"), ("T64k", "100000"), ("P20k", "100000")])
def test_other_configs_have_a_bench_leg(config, reads):
    line = run_bench("--config", config, "--reads", reads, "--steps", "2", "--warmup", "1", "--cpu-sample", "2000", "--verify", "500")
    assert line["config"]["workload"].startswith(config + ":")
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["verified_vs_oracle"]["reads"] == 500


def test_single_process_mode_and_device_identity_fields():
    """what the first real multi-GPU record needs in order to prove N distinct GPUs (PCI bus ids per rank, visible-device count) and the
    mode RAPPAS itself would use -- ONE process, rk_db_clone to every device, rk_place_batch_multi per step -- here with three
    handles on the one GPU of this box; with two ranks on it, rank 0 measures the single-process mode after the ranks' timed region"""
    common = ["--reads", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pcie", "--no-clade", "--verify", "200", "--sp-reads", "150000"]
    one = run_bench("--gpus", "1", "--single-process", "--sp-handles", "3", *common)
    assert one["n_gpus_visible"] >= 1 and len(one["device_pci_bus_id"].split(":")) == 3
    sp = one["single_process"]
    assert "error" not in sp, sp
    assert sp["handles"] == 3 and sp["reads_per_step"] == 450000 and sp["distinct_devices"] == 1 and sp["placed_per_step"] > 440000
    assert sp["device_pci_bus_ids"] == [one["device_pci_bus_id"]] * 3 and sp["value"] > 1e7
    two = run_bench("--gpus", "2", "--force-device", "0", "--dist-backend", "none", *common)
    assert two["per_rank_pci_bus_id"] == [one["device_pci_bus_id"]] * 2 and two["distinct_gpus"] == 1   # (two ranks, ONE GPU: the line says so)
    assert "error" not in two["single_process"] and two["single_process"]["handles"] == 2


def test_device_ordinals_follow_hip_visible_devices():
    """ordinals are positions in the list of VISIBLE devices: with HIP_VISIBLE_DEVICES=0 the one GPU is ordinal 0 and ordinal 1 does not
    exist (rk_db_create refuses it; the engine never looks behind the runtime's remapping); the PCI bus id is the same GPU's"""
    probe = ("import json, rappas_amd as ra; from rappas_amd import synth, _lib; import bench\n"
             "sdb = synth.make_config_db('C1'); db = ra.PhyloKmerDB.from_synth(sdb, device=0)\n"
             "seq, off = synth.make_reads(4, 500, 150, seed=1); n = int((ra.PlacementProcess(db).processQueries(seq, off).flags & 1).sum())\n"
             "try:\n    ra.PhyloKmerDB.from_synth(sdb, device=1); second = 'ok'\nexcept _lib.RkError as e:\n    second = e.code\n"
             "print(json.dumps(dict(placed=n, second=second, pci=bench.pci_bus_id(0), info_device=db.info.device)))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    outs = []
    for vis in (None, "0"):
        e = dict(env)
        if vis is not None:
            e["HIP_VISIBLE_DEVICES"] = vis
        p = subprocess.run([sys.executable, "-c", probe], env=e, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    assert outs[0]["placed"] == outs[1]["placed"] > 400 and outs[0]["pci"] == outs[1]["pci"] and outs[1]["info_device"] == 0
    assert outs[1]["second"] == -1   # RK_ERR_INVALID: device 1 is out of range when one device is visible
