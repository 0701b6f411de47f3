#!/usr/bin/env python3
"""bench.py -- reads placed / second of the phylo-kmer placement hot path on MI355X.

One "step" = one pass of the placement kernel over one batch of synthetic reads that is already resident in
HBM (packed 2-bit), against the BASELINE C2 database (DNA k=10, 999 branches, ~1e7 entries, seed 42).
Multi-GPU: reads shard across ranks, DB replicated per GPU, no collective on the data path ("weak" scaling:
every rank places its own `--reads`).  The CPU oracle is used here ONLY as the cpu_baseline leg and as the
pre-timing checker of a sample; the timed path is the HIP engine through the C ABI.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: enough for about 1-10 s of GPU time: 200 for C1/C2/C4, 10 for C3/C5s/C5m, 3 for C5)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: 10, or 1 for the long configs)")
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5", "C5mini", "C5s", "C5m", "T4k", "T8k", "T20k", "T40k", "T64k", "P20k", "L9k", "L16k"],
                    help="C2 = the configuration BASELINE's metric is quoted on (default); C3 = C2's DB with 1.25e8 reads per GPU "
                         "(1e9 reads over 8 GPUs); C5 = 10k-leaf tree, k=12, ~200 GB DB generated on the device, 1.25e7 x 250 bp per GPU; "
                         "T4k / T8k / T20k / T40k / T64k = C2's database and reads on trees of 3 999 / 7 999 / 19 999 / 39 999 / 65 535 branches; P20k = amino acids k=5, C4-like rows, 19 999 branches; L9k / L16k = rows of 400 / 1 000 entries "
                         "on 9 001 / 15 999 branches (not BASELINE configs)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (default: the config's count)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per read (0=auto, 8/16/32/64)")
    ap.add_argument("--table", default="auto", choices=["auto", "direct", "direct8", "hash"])
    ap.add_argument("--verify", type=int, default=2000, help="reads checked against the oracle before timing")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clade", action="store_true", help="C2 and T4k ... T64k: skip the clade-shaped variant of the line (reads cut from a genome whose k-mers "
                    "make up the database: the K best branches are neighbours, every k-mer is present)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg (host buffers through rk_place_batch / rk_place_batch_packed)")
    ap.add_argument("--pcie-reads", type=int, default=4_000_000, help="reads of the PCIe-inclusive leg (rank 0, N=1)")
    ap.add_argument("--db-scale", type=float, default=1.0)
    ap.add_argument("--dist-backend", default="auto", choices=["auto", "nccl", "gloo", "none"],
                    help="how ranks do the barriers / max-over-ranks (there is no collective on the data path): auto = nccl (RCCL), "
                         "falling back to none if RCCL cannot be set up; gloo = CPU tensors (rehearsals); none = no process group, "
                         "file rendezvous + CLOCK_MONOTONIC stamps (one node)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: put every rank on this device")
    ap.add_argument("--single-process", action="store_true",
                    help="add the mode RAPPAS itself would use -- ONE host process driving the GPUs: rk_db_clone to every device, one rk_place_batch_multi call per "
                         "step over page-locked buffers -- as the field `single_process` (never `value`).  On by default when --gpus > 1 (rank 0 measures it after "
                         "the ranks' timed region); with --gpus 1 it runs --sp-handles handles on the one device")
    ap.add_argument("--sp-handles", type=int, default=0, help="--single-process: handles (default: --gpus; on fewer visible devices they share them round-robin)")
    ap.add_argument("--sp-reads", type=int, default=2_000_000, help="--single-process: reads per handle and step")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher / rank bookkeeping rehearsal WITHOUT a GPU: no engine, no placement, a step is a 5 ms sleep; "
                         "the JSON line says so and carries no rate (tests of --gpus N on CPU)")
    a = ap.parse_args()
    long_cfg = {"C3": 10, "C5": 3, "C5s": 10, "C5m": 5, "C5mini": 20, "T4k": 40, "T8k": 30, "T20k": 30, "T40k": 20, "T64k": 20, "P20k": 20, "L9k": 20, "L16k": 10}
    if a.steps <= 0:
        a.steps = long_cfg.get(a.config, 200)
    if a.warmup < 0:
        a.warmup = 1 if a.config in long_cfg else 10
    return a


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def fasta_to_jplace_leg(ra, db, h_seq, h_off, n, rlen, n_branches, K):
    """FASTA file in the page cache -> rk_place (--dbimage: the engine's image of this database with a random tree of n_branches nodes in
    its user blob) -> .jplace file; the tool's own --timing line gives the passes.  Three runs, the best one is reported."""
    import shutil
    import subprocess
    import tempfile
    from rappas_amd import build, hostio, synth
    exe = build.build_host_tools()
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    d = tempfile.mkdtemp(prefix="rk_f2j_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        tree = hostio.parse_newick(synth.make_newick(n_branches, seed=3))
        img = os.path.join(d, "db.rkimg")
        db.save(img, user=hostio.tree_to_blob(tree))
        # >r0000001 + 150 bases: fixed-width records written in one piece
        hw = len(str(n - 1))
        rec = np.empty((n, 2 + hw + 1 + rlen + 1), np.uint8)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r")
        idx = np.arange(n)
        for j in range(hw):
            rec[:, 2 + hw - 1 - j] = ord("0") + (idx // 10 ** j) % 10
        rec[:, 2 + hw] = ord("\n")
        rec[:, 3 + hw:3 + hw + rlen] = np.asarray(h_seq[:n * rlen]).reshape(n, rlen)
        rec[:, -1] = ord("\n")
        fa = os.path.join(d, "q.fasta")
        rec.tofile(fa)
        best = None
        for _ in range(3):
            r = subprocess.run([exe, "--dbimage", img, "--fasta", fa, "--out", os.path.join(d, "q.jplace"), "--keep-at-most", str(K), "--threads", str(threads),
                                "--timing"], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-300:])
            t = json.loads(r.stdout.strip().splitlines()[-1])
            if best is None or t["fasta_to_jplace_s"] < best["fasta_to_jplace_s"]:
                best = t
        best = {k_: (round(v, 6) if isinstance(v, float) else v) for k_, v in best.items()}
        return {"value": best["reads"] / best["fasta_to_jplace_s"], "unit": "reads/s", "host_threads": threads, "passes": best,
                "what": f"rk_place --dbimage: {n} x {rlen} bp FASTA records ({best['fasta_bytes']} bytes, page cache) -> scan, dedup, gather, rk_place_batch, jplace "
                        f"({best['jplace_bytes']} bytes written); best of 3 runs; database load ({best['db_s']:.2f} s) and process start not counted"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def unpack_to_ascii(alphabet, packed_np, length):
    """packed u32 [n, wpr] -> ASCII (for the oracle legs)."""
    from rappas_amd import synth
    bits = 2 if alphabet == 4 else 5
    letters = synth.DNA_LETTERS if alphabet == 4 else synth.AA_LETTERS
    n, wpr = packed_np.shape
    big = np.zeros((n, wpr + 1), dtype=np.uint64)
    big[:, :wpr] = packed_np.astype(np.uint64)
    out = np.zeros((n, length), dtype=np.uint8)
    for i in range(length):
        bit = i * bits
        w, sh = bit >> 5, bit & 31
        v = (big[:, w] | (big[:, w + 1] << np.uint64(32))) >> np.uint64(sh)
        out[:, i] = letters[(v & np.uint64((1 << bits) - 1)).astype(np.int64)]
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    return out.reshape(-1), off


def row_length_table(sdb):
    """int32[sigma^k]: row length per dense k-mer index (0 = absent) of a host-side SynthDB."""
    k, sigma = sdb.k, sdb.alphabet
    lens = (sdb.row_offsets[1:] - sdb.row_offsets[:-1]).astype(np.int64)
    if sigma == 4:
        dense = sdb.key_codes.astype(np.int64)
    else:
        dense = np.zeros(sdb.n_keys, dtype=np.int64)
        for i in range(k):
            dense += ((sdb.key_codes >> np.uint64(5 * i)) & np.uint64(31)).astype(np.int64) * (20 ** i)
    table = np.zeros(sigma ** k, dtype=np.int32)
    table[dense] = lens
    return table


def count_entries_torch(torch, table, sigma, k, bits, packed, length, chunk=1 << 20):
    """Exact H = sum over reads and k-mer positions of the matched row length, with plain torch ops
    (independent of the engine): dense row-length table gather over every k-mer code."""
    dev = packed.device
    Q = length - k + 1
    table_t = torch.from_numpy(table).to(dev)
    n, wpr = packed.shape
    total = 0
    hits = 0
    for a in range(0, n, chunk):
        p = packed[a:a + chunk].to(torch.int64) & 0xFFFFFFFF
        m = p.shape[0]
        sym = torch.empty((m, length), dtype=torch.int64, device=dev)
        for i in range(length):
            bit = i * bits
            w, sh = bit >> 5, bit & 31
            v = p[:, w] >> sh
            if sh + bits > 32 and w + 1 < wpr:
                v = v | (p[:, w + 1] << (32 - sh))
            sym[:, i] = v & ((1 << bits) - 1)
        idx = torch.zeros((m, Q), dtype=torch.int64, device=dev)
        for i in range(k):
            idx += sym[:, i:i + Q] * (sigma ** i)
        ln = table_t[idx]
        total += int(ln.sum().item())
        hits += int((ln > 0).sum().item())
        del p, sym, idx, ln
    return total, hits


# ---------------------------------------------------------------------------------------------------------
# Rank bookkeeping.  The data path has no collective (reads are independent: PlacementProcess.java:1067-1075),
# so the only cross-rank steps are the two barriers around the timed region, the MAX over ranks of the elapsed
# time and the collection of per-rank rates.  Three ways to do them:
#   nccl  = RCCL through torch.distributed (what the driver's torch.distributed.run launch gets by default),
#   gloo  = the same calls on CPU tensors (rehearsals on one GPU / without a GPU),
#   none  = no process group at all: ranks rendezvous through files in a directory private to the job and
#           stamp start / end with CLOCK_MONOTONIC (system-wide on Linux, all ranks are on ONE node); the
#           job's elapsed time is max(end) - min(start), which is >= every rank's own bracketed time.
#   auto  = nccl, and if its set-up raises on any rank every rank falls back to `none` (a vote through the
#           same directory, so all ranks take the same branch).
# Whatever the backend, every rank leaves `result.<rank>.json` in the job directory and rank 0 folds the
# per-rank rates into the one JSON line.
# ---------------------------------------------------------------------------------------------------------
EX_PG_FAILED = 75  # a rank that could not set up its process group exits with this (EX_TEMPFAIL)


def job_dir():
    """Directory shared by the ranks of ONE job on this node.  The self-launcher creates it and passes it down;
    under torch.distributed.run every worker is a child of the same agent process, so (agent pid, MASTER_PORT)
    names the job."""
    d = os.environ.get("RK_BENCH_JOB_DIR")
    if not d:
        import tempfile
        d = os.path.join(tempfile.gettempdir(), f"rk_bench_{os.getuid()}_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}")
    os.makedirs(d, exist_ok=True)
    return d


def _put(path, text):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)  # atomic: a reader sees the whole file or none of it


class RankSync:
    """barrier / max-over-ranks / per-rank results for bench.py (see the block comment above)."""

    def __init__(self, rank, world, backend, device=None, timeout_s=600.0):
        self.rank, self.world, self.device, self.timeout_s = rank, world, device, timeout_s
        self.dir = job_dir() if world > 1 else None
        self.backend = backend if world > 1 else "single"
        self.note = None
        self._n = 0
        self.dist = None
        if world == 1:
            return
        if backend in ("nccl", "gloo", "auto"):
            err = self._init_pg("nccl" if backend == "auto" else backend)
            if backend == "auto":
                # all ranks take the same branch: nccl only if it came up on every one of them
                _put(os.path.join(self.dir, f"pg.{rank}"), "ok" if err is None else f"failed: {err}")
                votes = self._wait_all("pg", read=True)
                bad = {r: v for r, v in votes.items() if v != "ok"}
                if bad:
                    if err is None:
                        self._destroy_pg()
                    r0 = min(bad)
                    self.backend, self.note = "none", f"nccl set-up failed on rank(s) {sorted(bad)} ({bad[r0][:200]}); fell back to --dist-backend none"
                    if rank == 0:
                        print("bench.py: " + self.note, file=sys.stderr, flush=True)
                else:
                    self.backend = "nccl"
            elif err is not None:
                _put(os.path.join(self.dir, f"pg_failed.{rank}"), err)
                print(f"bench.py: rank {rank}: {backend} process group failed: {err}", file=sys.stderr, flush=True)
                raise SystemExit(EX_PG_FAILED)
        _put(os.path.join(self.dir, f"pg_done.{rank}"), self.backend)

    # -- torch.distributed -------------------------------------------------------------------------------
    def _init_pg(self, backend):
        try:
            import datetime
            import torch
            import torch.distributed as dist
            kw = dict(timeout=datetime.timedelta(seconds=float(os.environ.get("RK_BENCH_PG_TIMEOUT", "180"))))
            if os.environ.get("RK_BENCH_INIT_FILE"):  # self-launched ranks rendezvous through a file: no port to race for
                kw.update(init_method="file://" + os.environ["RK_BENCH_INIT_FILE"], rank=self.rank, world_size=self.world)
            else:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                if self.device is None:
                    raise RuntimeError("nccl needs a GPU per rank")
                dist.init_process_group("nccl", device_id=self.device, **kw)
            else:
                dist.init_process_group("gloo", **kw)
            self.dist = dist
            if dist.get_world_size() != self.world:
                raise RuntimeError(f"process group has {dist.get_world_size()} ranks, --gpus says {self.world}")
            # one real collective now, so that a broken transport shows here and not inside the timed region
            t = torch.ones(1, dtype=torch.float64, device=self.device if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            if int(t.item()) != self.world:
                raise RuntimeError(f"all_reduce over {self.world} ranks returned {t.item()}")
            return None
        except Exception as e:  # noqa: BLE001 -- reported, then either a fallback (auto) or a non-zero exit
            self._destroy_pg()
            return f"{type(e).__name__}: {e}"

    def _destroy_pg(self):
        try:
            if self.dist is not None and self.dist.is_initialized():
                self.dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        self.dist = None

    # -- files -------------------------------------------------------------------------------------------
    def _wait_all(self, stem, read=False, ranks=None):
        """wait until `<stem>.<r>` exists for every rank; RuntimeError naming the missing ranks after timeout_s"""
        ranks = list(range(self.world)) if ranks is None else ranks
        t_end = time.monotonic() + self.timeout_s
        missing = ranks
        while True:
            missing = [r for r in missing if not os.path.exists(os.path.join(self.dir, f"{stem}.{r}"))]
            if not missing:
                break
            if time.monotonic() > t_end:
                raise RuntimeError(f"bench.py: rank {self.rank} waited {self.timeout_s:.0f} s at '{stem}' for rank(s) {missing}")
            time.sleep(0.0005)
        if not read:
            return None
        return {r: open(os.path.join(self.dir, f"{stem}.{r}")).read() for r in ranks}

    def barrier(self):
        if self.world == 1:
            return
        if self.backend in ("nccl", "gloo"):
            self.dist.barrier()
            return
        self._n += 1
        _put(os.path.join(self.dir, f"barrier{self._n}.{self.rank}"), "")
        self._wait_all(f"barrier{self._n}")

    def finish(self, t_start, t_end, mine):
        """every rank: publish its own record; rank 0: (whole-job elapsed seconds, list of per-rank records)"""
        elapsed = t_end - t_start
        if self.world == 1:
            return elapsed, [mine]
        if self.backend in ("nccl", "gloo"):
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        _put(os.path.join(self.dir, f"result.{self.rank}"), json.dumps(dict(mine, rank=self.rank, t_start=t_start, t_end=t_end, pid=os.getpid())))
        if self.rank != 0:
            return elapsed, None
        recs = [json.loads(v) for _, v in sorted(self._wait_all("result", read=True).items())]
        if self.backend == "none":
            elapsed = max(r["t_end"] for r in recs) - min(r["t_start"] for r in recs)
        return elapsed, recs

    def close(self):
        if self.world == 1 or getattr(self, "_closed", False):
            return
        self._closed = True
        if self.backend in ("nccl", "gloo"):
            try:
                self.dist.barrier()
            finally:
                self._destroy_pg()
        _put(os.path.join(self.dir, f"closed.{self.rank}"), "")
        if self.rank == 0 and not os.environ.get("RK_BENCH_JOB_DIR"):  # under a launcher the job directory is ours to remove
            import shutil
            self._wait_all("closed")
            shutil.rmtree(self.dir, ignore_errors=True)


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one fresh process per GPU).

    Runs before torch is imported or any HIP call is made in this process: the parent never touches the GPU.  It
    polls its children; when one exits non-zero it stops the others within seconds and exits non-zero with that
    rank's stderr tail (rank 0 would otherwise sit in the rendezvous until its timeout).  If the ranks die, or
    hang, while setting up the nccl / gloo process group, a fresh set is started with `--dist-backend none`.
    Children get RANK / LOCAL_RANK / WORLD_SIZE as `torch.distributed.run` would set them; the process group
    rendezvous goes through a file in the job directory (no port to race for)."""
    import shutil
    import subprocess
    import tempfile

    def run_set(backend):
        d = tempfile.mkdtemp(prefix="rk_bench_job_")
        argv = [x for x in sys.argv[1:]]
        for i, x in enumerate(argv):  # the set's backend replaces whatever the command line said
            if x == "--dist-backend":
                del argv[i:i + 2]
                break
            if x.startswith("--dist-backend="):
                del argv[i]
                break
        argv += ["--dist-backend", backend]
        procs = []
        for r in range(a.gpus):
            # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; without it RCCL's
            # hipIpcGetMemHandle fails ("invalid argument").  The image exports it already; keep it if a caller dropped it.
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       RK_BENCH_JOB_DIR=d, RK_BENCH_INIT_FILE=os.path.join(d, "pg_store"))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=open(os.path.join(d, f"rank{r}.out"), "wb"),
                                          stderr=open(os.path.join(d, f"rank{r}.err"), "wb")))

        def tail(r, n=4000):
            try:
                return open(os.path.join(d, f"rank{r}.err"), errors="replace").read()[-n:]
            except OSError:
                return ""

        def stop_all():
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 5.0
            for p in procs:
                try:
                    p.wait(max(0.0, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()

        pg_deadline = time.monotonic() + float(os.environ.get("RK_BENCH_PG_TIMEOUT", "180")) + 120.0
        failed = None
        try:
            while True:
                codes = [p.poll() for p in procs]
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                if bad:
                    time.sleep(0.3)  # ranks that are failing for the same reason get to say so themselves
                    failed = [(r, p.poll()) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                    break
                if all(c == 0 for c in codes):
                    break
                pg_up = all(os.path.exists(os.path.join(d, f"pg_done.{r}")) for r in range(a.gpus))
                if not pg_up and backend != "none" and time.monotonic() > pg_deadline:
                    failed = "pg-timeout"
                    break
                time.sleep(0.05)
        finally:
            stop_all()
        # a failure of the process-group set-up itself (a rank said so with EX_PG_FAILED, or the group never came up)
        in_pg = backend != "none" and (failed == "pg-timeout" or (failed is not None and any(c == EX_PG_FAILED for _, c in failed)))
        if failed is None:
            for r in range(a.gpus):  # the ranks' diagnostics, in rank order
                sys.stderr.write(open(os.path.join(d, f"rank{r}.err"), errors="replace").read())
            sys.stdout.write(open(os.path.join(d, "rank0.out"), errors="replace").read())
            sys.stdout.flush()
            shutil.rmtree(d, ignore_errors=True)
            return 0, False
        if failed == "pg-timeout":
            sys.stderr.write(f"bench.py: the {backend} process group did not come up on every rank in time\n")
        else:
            for r, c in failed:
                sys.stderr.write(f"bench.py: rank {r} exited with code {c}; its stderr ends:\n{tail(r)}\n")
            sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {failed}; the other ranks were stopped\n")
        sys.stderr.flush()
        shutil.rmtree(d, ignore_errors=True)
        return 1, in_pg

    rc, in_pg = run_set(a.dist_backend)
    if rc != 0 and in_pg and a.dist_backend in ("auto", "nccl"):
        sys.stderr.write("bench.py: starting a fresh set of ranks with --dist-backend none (no process group; the data path has no collective)\n")
        rc, _ = run_set("none")
    return rc


def pci_bus_id(device_index):
    """hipDeviceGetPCIBusId of a visible device ordinal ("0000:c1:00.0"): what tells N ranks on N distinct GPUs from N ranks on one"""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
            return buf.value.decode()
    except Exception:
        pass
    return None


def single_process_leg(ra, torch, db, h_seq, h_off, rlen, K, n_handles, reads_per_handle, steps=5):
    """ONE process, n_handles device handles of the same database (rk_db_clone: device to device, over xGMI between GPUs), one
    rk_place_batch_multi call per step: contiguous shards, one host thread per handle, no collective."""
    n_vis = torch.cuda.device_count()
    devices = [i % max(1, n_vis) for i in range(n_handles)]
    clones = []
    try:
        t0 = time.perf_counter()
        for dv in devices[1:]:
            clones.append(db.clone(device=dv))
        clone_s = time.perf_counter() - t0
        dbs = [db] + clones
        n = reads_per_handle * n_handles
        base = len(h_off) - 1
        rep = (n + base - 1) // base
        seq = ra.host_alloc(n * rlen, np.uint8)
        for r in range(rep):  # (the batch's own reads, repeated to the size asked for)
            a0, a1 = r * base * rlen, min(n, (r + 1) * base) * rlen
            seq[a0:a1] = h_seq[:a1 - a0]
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(rlen)
        out = ra.Placements(ra.host_alloc(n, np.uint8), ra.host_alloc((n, K), np.uint16), ra.host_alloc((n, K), np.float32),
                            ra.host_alloc((n, K), np.float64), ra.host_alloc(n, np.uint32), {})
        pp = ra.PlacementProcess(db)
        pp.processQueriesMulti(dbs, seq, off, keepAtMost=K, out=out)
        pp.processQueriesMulti(dbs, seq, off, keepAtMost=K, out=out)
        t0 = time.perf_counter()
        for _ in range(steps):
            pp.processQueriesMulti(dbs, seq, off, keepAtMost=K, out=out)
        dt = (time.perf_counter() - t0) / steps
        placed = int(out.counters["placed"])
        return {"value": n / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "handles": n_handles, "reads_per_step": n, "steps": steps,
                "devices": devices, "device_pci_bus_ids": [pci_bus_id(dv) for dv in devices], "distinct_devices": len(set(devices)),
                "clone_s": clone_s, "placed_per_step": placed, "entry": "rk_db_clone x (handles - 1) + rk_place_batch_multi over page-locked buffers (characters in, results out)"}
    finally:
        for c in clones:
            c.close()


def per_rank_fields(sync, recs):
    """what rank 0 adds to the JSON line about the individual ranks (their own reads/s: own reads / own time for its steps)"""
    out = {"dist_backend": sync.backend if sync.note is None else f"{sync.backend} ({sync.note})"}
    if recs and len(recs) > 1:
        rates = [r["reads_per_s"] for r in recs]
        out.update(per_rank=rates, per_rank_min=min(rates), per_rank_max=max(rates),
                   per_rank_kernel_ms=[r["kernel_ms"] for r in recs], per_rank_device=[r.get("device_index") for r in recs],
                   per_rank_pci_bus_id=[r.get("pci_bus_id") for r in recs], distinct_gpus=len({r.get("pci_bus_id") for r in recs if r.get("pci_bus_id")}))
    return out


def rehearse(a, rank, world):
    """--rehearse-launch: everything bench.py does around the timed region (rank start, rendezvous, barriers, per-rank
    results, the one JSON line) with a sleep in place of the placement step.  No GPU, no engine, no rate."""
    die = os.environ.get("RK_BENCH_REHEARSE_DIE_RANK")  # tests: this rank dies before the rendezvous
    if die is not None and int(die) == rank:
        print(f"bench rank {rank}: dying before the rendezvous (RK_BENCH_REHEARSE_DIE_RANK)", file=sys.stderr, flush=True)
        raise SystemExit(3)
    sync = RankSync(rank, world, a.dist_backend, timeout_s=float(os.environ.get("RK_BENCH_SYNC_TIMEOUT", "600")))
    sync.barrier()
    t0 = time.monotonic()
    for _ in range(a.steps):
        time.sleep(0.005 * (1 + rank % 2))  # odd ranks are slower: the job's time must be theirs
    t_own = time.monotonic()
    sync.barrier()
    t1 = time.monotonic()
    n_reads = a.reads or 1000
    elapsed, recs = sync.finish(t0, t1, dict(reads_per_s=n_reads * a.steps / (t_own - t0), kernel_ms=(t_own - t0) / a.steps * 1e3, device_index=None))
    if rank == 0:
        line = {"metric": "launcher rehearsal (no placement ran)", "value": None, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "data": "none: --rehearse-launch", "scaling": "weak"}
        line.update(per_rank_fields(sync, recs))
        print(json.dumps(line), flush=True)
    sync.close()
    return 0


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself; torch.distributed.run must use --nproc-per-node N)")
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("RK_BENCH_ECHO_RANK"):  # tests: proof that N separate rank processes were started
        print(f"bench rank {rank}/{world} pid {os.getpid()}", file=sys.stderr, flush=True)
    if a.rehearse_launch:
        return rehearse(a, rank, world)
    import torch

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the placement engine has no CPU fallback")
    if a.force_device >= 0:
        local_rank = a.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sync = RankSync(rank, world, a.dist_backend, device=dev)
    n_gpus = world

    import rappas_amd as ra
    from rappas_amd import synth

    mode = {"auto": ra.RK_TABLE_AUTO, "direct": ra.RK_TABLE_DIRECT, "direct8": ra.RK_TABLE_DIRECT8, "hash": ra.RK_TABLE_HASH}[a.table]
    spec = None  # set when the DB exists only in HBM (generated on the device from the seed)
    if a.config in synth.SPEC_CONFIGS:
        alphabet, k, leaves, _, _, rlen, n_reads_cfg = synth.SPEC_CONFIGS[a.config]
        spec = synth.make_spec(a.config, seed=42)
        if a.db_scale != 1.0:
            spec.mean_row_len = max(1.0, spec.mean_row_len * a.db_scale)
        t_db = time.perf_counter()
        db = ra.PhyloKmerDB.synthetic(spec, device=local_rank, table_mode=mode)
        t_db = time.perf_counter() - t_db
        n_branches, bits = spec.n_branches, spec.bits
        lens_table = spec.row_length_table()

        def oracle_db_for(seq, off):  # the rows these reads touch, regenerated on the host by the generator's numpy twin
            return O.OracleDB.from_synth(spec.subset_db(synth.codes_of_reads(alphabet, k, seq, off)))
        db_desc = (f"{db.info.n_keys} keys / {db.info.n_entries} entries ({db.info.rows_bytes / 1e9:.1f} GB image) generated on the "
                   f"device from seed 42 in {t_db:.1f} s")
    else:
        cfg_name = "C2" if a.config == "C3" else a.config
        alphabet, k, leaves, n_keys, n_entries, rlen, n_reads_cfg = synth.CONFIGS[cfg_name]
        if a.config == "C3":
            n_reads_cfg = 125_000_000  # 1e9 reads over 8 GPUs
        sdb = synth.make_config_db(cfg_name, seed=42, scale=a.db_scale)
        db = ra.PhyloKmerDB.from_synth(sdb, device=local_rank, table_mode=mode)
        n_branches, bits = sdb.n_branches, sdb.bits
        lens_table = row_length_table(sdb)

        def oracle_db_for(seq, off):
            return O.OracleDB.from_synth(sdb)
        db_desc = f"{sdb.n_keys} keys / {sdb.n_entries} entries (seed 42)"
    n_reads = a.reads or n_reads_cfg
    if a.lanes:
        db.set_lanes_per_read(a.lanes)
    pp = ra.PlacementProcess(db)
    K = 7
    wpr = db.packed_words(rlen)
    from oracle import oracle as O  # checker + cpu_baseline leg only; never inside the timed region

    # synthetic reads, uniform i.i.d. symbols, generated on the device straight into the packed layout
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    if alphabet == 4:
        packed = torch.randint(-2**31, 2**31, (n_reads, wpr), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        tail_bits = rlen * bits - 32 * (wpr - 1)
        if tail_bits < 32:
            packed[:, wpr - 1] &= (1 << tail_bits) - 1
    else:
        sym = torch.randint(0, 20, (n_reads, rlen), dtype=torch.int64, device=dev, generator=gen)
        acc = torch.zeros((n_reads, wpr + 1), dtype=torch.int64, device=dev)
        for i in range(rlen):
            bit = i * bits
            w, sh = bit >> 5, bit & 31
            v = sym[:, i] << sh
            acc[:, w] |= v & 0xFFFFFFFF
            acc[:, w + 1] |= v >> 32
        packed = acc[:, :wpr].to(torch.int32)
        del sym, acc
    out = dict(n_rows=torch.empty(n_reads, dtype=torch.uint8, device=dev),
               branch=torch.empty((n_reads, K), dtype=torch.int16, device=dev),
               score=torch.empty((n_reads, K), dtype=torch.float32, device=dev),
               lwr=torch.empty((n_reads, K), dtype=torch.float64, device=dev),
               flags=torch.empty(n_reads, dtype=torch.int32, device=dev))

    def step():
        pp.place_packed(packed, fixed_len=rlen, out=out, keepAtMost=K)

    # ---- correctness gate on a sample before any timing (oracle = checker only) ----
    verified = None
    if a.verify and rank == 0:
        from tests.util import compare_with_oracle
        nv = min(a.verify, n_reads, 48 if spec is not None else a.verify)
        step()
        torch.cuda.synchronize()
        seq, off = unpack_to_ascii(alphabet, packed[:nv].cpu().numpy().view(np.uint32), rlen)
        odb = oracle_db_for(seq, off)
        ref = odb.place(seq, off, keep_at_most=K)
        got = ra.Placements(out["n_rows"][:nv].cpu().numpy(), out["branch"][:nv].cpu().numpy().view(np.uint16),
                            out["score"][:nv].cpu().numpy(), out["lwr"][:nv].cpu().numpy(),
                            out["flags"][:nv].cpu().numpy().view(np.uint32), {})
        st = compare_with_oracle(got, ref, odb, seq, off)
        verified = dict(reads=nv, ties=st["ties"], max_lwr_rel=st["max_lwr_rel"])

    # ---- algorithmic bytes: B = ceil(R*b/8) + Q*8 + H*6 + (2 + K*14)   ----
    Q = rlen - k + 1
    H_total, hit_kmers = count_entries_torch(torch, lens_table, alphabet, k, bits, packed, rlen)
    H_mean = H_total / n_reads
    B = math.ceil(rlen * bits / 8) + Q * 8 + H_mean * 6 + (2 + K * 14)
    # the engine's own count of the same quantities (rk_count_work_device: a kernel of its own) next to the torch count above
    try:
        work = pp.count_work(packed, fixed_len=rlen)
        work["equals_independent_count"] = bool(work["entries"] == H_total and work["kmers_hit"] == hit_kmers and work["kmers_probed"] == n_reads * Q)
    except AttributeError:  # (RK_LIB pointing at a build older than the entry point)
        work = None

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    sync.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.monotonic()
    for s in range(a.steps):
        evs[s][0].record()
        step()
        evs[s][1].record()
    torch.cuda.synchronize()
    t_own = time.monotonic()
    sync.barrier()
    torch.cuda.synchronize()
    t1 = time.monotonic()
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    kern_avg_s = sum(kern_ms) / len(kern_ms) / 1e3
    # t0 .. t1 is this rank's barrier-to-barrier time (MAX over ranks = the job's time); t_own is when its own steps were done
    elapsed, per_rank = sync.finish(t0, t1, dict(reads_per_s=n_reads * a.steps / (t_own - t0), kernel_ms=kern_avg_s * 1e3,
                                                  device=torch.cuda.get_device_name(local_rank), device_index=local_rank,
                                                  pci_bus_id=pci_bus_id(local_rank)))

    if rank == 0:
        value = n_gpus * n_reads * a.steps / elapsed
        achieved = (B * n_reads / kern_avg_s) / 1e9  # GB/s, algorithmic bytes / average kernel duration
        peak = 8000.0
        traffic = None
        # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of THIS configuration and batch size
        # (profiles/pmc_traffic_<config>.json, written by scripts/profile.sh + prof_summary.py; null when none matches)
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.config}.json")
        if not os.path.exists(pmc_path):
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                pj = json.load(open(pmc_path))
                if pj.get("config") == a.config and pj.get("reads") == n_reads:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        sym = "bp DNA" if alphabet == 4 else "aa"
        metric = {"C2": "reads placed/sec (whole node), 150 bp DNA vs 10^7-kmer DB",
                  "C3": "reads placed/sec (whole node), 150 bp DNA vs 10^7-kmer DB"}.get(
                      a.config, f"reads placed/sec (whole node), {rlen} {sym} reads, config {a.config}")
        line = {
            "metric": metric,
            "value": value, "unit": "reads/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.config}: {'DNA' if alphabet == 4 else 'AA'} k={k}, {n_branches} branches, "
                                   f"phylo-kmer DB of {db_desc}, replicated per GPU; "
                                   f"{n_reads} x {rlen} symbol reads per GPU per step (uniform, seed 1+rank), keep_at_most=7",
                       "table": {ra.RK_TABLE_DIRECT: "direct (compact blocks: 0.67 B/k-mer with 4-bit unit counts, 1.33 B/k-mer with bytes; see config.kernel)", ra.RK_TABLE_DIRECT8: "direct8", ra.RK_TABLE_HASH: "hash"}[db.info.table_mode],
                       "kernel": db.kernel_name(), "reads_per_gpu": n_reads, "sharding": f"reads x{n_gpus}, DB replicated"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic, "bytes_per_read": B, "entries_per_read": H_mean,
                         "kernel_ms": kern_avg_s * 1e3, "kernel": db.kernel_name().split("<")[0]},
            "verified_vs_oracle": verified,
            "work_counters": work,
        }
        line.update(per_rank_fields(sync, per_rank))
        line["n_gpus_visible"] = torch.cuda.device_count()
        line["device_pci_bus_id"] = pci_bus_id(local_rank)
        if (a.single_process or n_gpus > 1) and alphabet == 4 and spec is None:
            # ---- ONE process over all GPUs (what a single JVM would do), after every rank's timed region; never `value` ----
            sync.close()  # (the other ranks leave now: their processes must not sit in a barrier on the GPUs this leg is about to use)
            try:
                sp_n = min(a.sp_reads, n_reads)
                sp_seq, sp_off = unpack_to_ascii(alphabet, packed[:sp_n].cpu().numpy().view(np.uint32), rlen)
                line["single_process"] = single_process_leg(ra, torch, db, sp_seq, sp_off, rlen, K, a.sp_handles or n_gpus, sp_n)
            except Exception as e:
                line["single_process"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        pinned_bufs = None
        if not a.no_pcie and n_gpus == 1:
            # the page-locked caller buffers of the boundary legs below, allocated here, as a caller would at start-up.  (Allocated after
            # the clade leg -- seconds of numpy on the host in between -- the same leg ran at 1.0 - 1.9e8 reads/s instead of 3.0e8, five
            # runs of five: wherever the thread was by then, its page-locked memory landed further from the GPU.)
            npc_ = min(a.pcie_reads, n_reads)
            pinned_bufs = (ra.host_alloc((npc_, wpr), np.uint32),
                           ra.Placements(ra.host_alloc(npc_, np.uint8), ra.host_alloc((npc_, K), np.uint16), ra.host_alloc((npc_, K), np.float32),
                                         ra.host_alloc((npc_, K), np.float64), ra.host_alloc(npc_, np.uint32), {}))
        if a.config in ("C2", "T4k", "T8k", "T20k", "T40k", "T64k") and not a.no_clade and n_gpus == 1:
            # ---- the same tree and row statistics with clade-shaped reads (what real placements look like): never `value` ----
            cdb_s, genome = synth.make_clade_db(k=k, n_branches=n_branches)
            nc = min(2_000_000, n_reads)
            cseq, coff = synth.make_clade_reads(genome, nc, rlen)
            cdb = ra.PhyloKmerDB.from_synth(cdb_s, device=local_rank)
            cpp = ra.PlacementProcess(cdb)
            cpk = torch.from_numpy(cpp.pack_reads_host(cseq, coff)[0].view(np.int32)).to(dev)
            cout = cpp.place_packed(cpk, fixed_len=rlen, keepAtMost=K)
            torch.cuda.synchronize()
            cH, _ = count_entries_torch(torch, row_length_table(cdb_s), alphabet, k, bits, cpk, rlen)
            cB = math.ceil(rlen * bits / 8) + Q * 8 + cH / nc * 6 + (2 + K * 14)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                cpp.place_packed(cpk, fixed_len=rlen, out=cout, keepAtMost=K)
            e1.record()
            torch.cuda.synchronize()
            c_s = e0.elapsed_time(e1) / 1e3 / 10
            nv = min(500, nc)
            codb = O.OracleDB.from_synth(cdb_s)
            cgot = ra.Placements(cout["n_rows"][:nv].cpu().numpy(), cout["branch"][:nv].cpu().numpy().view(np.uint16), cout["score"][:nv].cpu().numpy(),
                                 cout["lwr"][:nv].cpu().numpy(), cout["flags"][:nv].cpu().numpy().view(np.uint32), {})
            from tests.util import compare_with_oracle as _cmp
            _cmp(cgot, codb.place(cseq[:int(coff[nv])], coff[:nv + 1], keep_at_most=K), codb, cseq[:int(coff[nv])], coff[:nv + 1])
            line["clade"] = {"value": nc / c_s, "unit": "reads/s", "kernel_ms": c_s * 1e3, "reads": nc, "entries_per_read": cH / nc, "bytes_per_read": cB,
                             "roofline_frac": cB * nc / c_s / 1e9 / peak, "verified_vs_oracle": nv,
                             "workload": f"rappas_amd.synth.make_clade_db: {cdb_s.n_keys} keys / {cdb_s.n_entries} entries on {n_branches} branches, rows of one 500-bp "
                                         f"stretch of a 700 kbp genome share a neighbourhood of the tree; {nc} x {rlen} bp reads cut from the genome "
                                         "(every k-mer present: a third more row entries per read than the uniform reads of `value`)"}
            cdb.close()
        if not a.no_pcie and n_gpus == 1:
            # ---- the boundary RAPPAS would call: host buffers in, host buffers out (never `value`) ----
            # pageable numpy arrays (what a JVM heap array looks like to the library), result arrays allocated once and reused
            npc = min(a.pcie_reads, n_reads)
            h_packed = packed[:npc].cpu().numpy().view(np.uint32)
            h_seq, h_off = unpack_to_ascii(alphabet, h_packed, rlen)
            reuse = ra.Placements(np.zeros(npc, np.uint8), np.zeros((npc, K), np.uint16), np.zeros((npc, K), np.float32),
                                  np.zeros((npc, K), np.float64), np.zeros(npc, np.uint32), {})

            def timed(fn, reps=5):
                fn()
                fn()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                return npc * reps / (time.perf_counter() - t0)
            r_packed = timed(lambda: pp.processQueriesPacked(h_packed, fixed_len=rlen, keepAtMost=K, out=reuse))
            dev_n = out["n_rows"][:npc].cpu().numpy()
            same = bool(np.array_equal(reuse.n_rows, dev_n) and np.array_equal(reuse.branch, out["branch"][:npc].cpu().numpy().view(np.uint16)))
            r_ascii = timed(lambda: pp.processQueries(h_seq, h_off, keepAtMost=K, out=reuse))
            # the same with page-locked caller buffers (rk_host_alloc; a JVM wraps them as direct ByteBuffers): no staging copies
            p_packed, p_out = pinned_bufs
            p_packed[:] = h_packed
            r_pinned = timed(lambda: pp.processQueriesPacked(p_packed, fixed_len=rlen, keepAtMost=K, out=p_out))
            same = same and bool(np.array_equal(p_out.n_rows, dev_n))
            pk_out = pp.pack_reads_host(h_seq, h_off, max_len=rlen)           # (first call: the output pages are touched here)
            same = same and bool(np.array_equal(pk_out[0], h_packed))
            t0 = time.perf_counter()
            for _ in range(5):
                pp.pack_reads_host(h_seq, h_off, max_len=rlen, out=pk_out)
            r_pack = 5 * npc / (time.perf_counter() - t0)
            out_b = 1 + K * 14 + 4
            line["pcie_inclusive"] = {
                "packed_host": {"value": r_packed, "unit": "reads/s", "entry": "rk_place_batch_packed", "bytes_in_per_read": wpr * 4, "bytes_out_per_read": out_b},
                "packed_host_page_locked": {"value": r_pinned, "unit": "reads/s", "entry": "rk_place_batch_packed over rk_host_alloc buffers",
                                            "bytes_in_per_read": wpr * 4, "bytes_out_per_read": out_b},
                "ascii_host": {"value": r_ascii, "unit": "reads/s", "entry": "rk_place_batch", "bytes_in_per_read": rlen + 8, "bytes_out_per_read": out_b},
                "host_packer": {"value": r_pack, "unit": "reads/s", "entry": "rk_pack_reads_host (AVX2 + BMI2 blocks of 32 symbols), output arrays reused, 5 calls",
                                "threads": min(os.cpu_count() or 1, 16)},
                "sample": f"first {npc} reads of the batch, pageable host arrays unless named page-locked, result arrays reused, 5 calls after two warm-ups",
                "equals_device_path": same}
        if not a.no_pcie and n_gpus == 1 and a.config == "C2" and alphabet == 4:
            # ---- the stand-alone tool around the kernel: FASTA bytes (page cache) -> .jplace bytes written, through rappas_amd/bin/rk_place
            #      (scan + dedup + gather + rk_place_batch + the jplace text, every pass on all of the tool's host threads; DB image
            #      loading and process start-up are outside the figure).  Never `value`. ----
            try:
                line["fasta_to_jplace"] = fasta_to_jplace_leg(ra, db, h_seq, h_off, npc, rlen, n_branches, K)
            except Exception as e:  # the leg must not take the bench line down
                line["fasta_to_jplace"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not a.no_cpu_baseline and n_gpus == 1:
            # bounded sample: about 10 s of single-thread work (the oracle does ~1.6e7 row entries per second)
            ns = min(a.cpu_sample, n_reads, max(64 if spec is not None else 2000, int(1.6e8 / max(1.0, H_mean))))
            seq, off = unpack_to_ascii(alphabet, packed[:ns].cpu().numpy().view(np.uint32), rlen)
            odb = oracle_db_for(seq, off)
            nw = min(ns, 2000)
            odb.place(seq[:int(off[nw])], off[:nw + 1], keep_at_most=K)  # warm caches
            c0 = time.perf_counter()
            odb.place(seq, off, keep_at_most=K)
            cdt = time.perf_counter() - c0
            line["cpu_baseline"] = {"value": ns / cdt, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "nproc": os.cpu_count(),
                                    "usable_cpus": len(os.sched_getaffinity(0)),
                                    "sample": f"first {ns} reads of rank 0's batch, "
                                              + ("the DB rows those reads touch regenerated on the host, " if spec is not None else "same DB, ")
                                              + f"oracle/rappas_oracle.c (single thread, like the reference's placement loop), {cdt:.1f} s"}
            # the same oracle on every host core (reads split into contiguous chunks, one thread each; ctypes drops the GIL)
            from concurrent.futures import ThreadPoolExecutor
            ncore = max(1, min(os.cpu_count() or 1, 64))
            nall = ns if spec is not None else min(n_reads, ns * min(ncore, 8))  # (a generated DB exists on the host only for the sample)
            seq_a, off_a = (seq, off) if nall == ns else unpack_to_ascii(alphabet, packed[:nall].cpu().numpy().view(np.uint32), rlen)
            bounds = [nall * i // ncore for i in range(ncore + 1)]

            def work(i):
                a0, a1 = bounds[i], bounds[i + 1]
                if a1 > a0:
                    odb.place(seq_a[int(off_a[a0]):int(off_a[a1])], off_a[a0:a1 + 1] - off_a[a0], keep_at_most=K)
            c0 = time.perf_counter()
            with ThreadPoolExecutor(ncore) as ex:
                list(ex.map(work, range(ncore)))
            cdt2 = time.perf_counter() - c0
            line["cpu_baseline_all_cores"] = {"value": nall / cdt2, "unit": "reads/s", "cores": ncore, "kind": "port",
                                              "sample": f"first {nall} reads, {ncore} threads, {cdt2:.1f} s"}
        print(json.dumps(line), flush=True)
    sync.close()


if __name__ == "__main__":
    main()
