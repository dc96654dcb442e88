#!/usr/bin/env python
"""bench.py -- FBS iterations/sec + achieved HBM GB/s, dense LASSO (soft-threshold prox), MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one FBS iteration of the product solver (fasta_python_amd.FBSolver, driven as fasta() drives it: the library's
host-side loop fh_iterate issues the launches and takes the reference's decisions) over a device-resident
synthetic matrix: by default ONE launch of the one-pass kernel (both directions from a single read of A);
`--fused off` gives the north-star structure of one K-fwd + one K-adj launch (+ one K-fwd per backtrack).
N = 1: BASELINE.json configs[1], A = 65536 x 65536 float64 (32 GiB).  N > 1: the SAME matrix row-sharded
over N ranks (one process per GPU, strong scaling), one RCCL all-reduce of the A^T partial sums per
iteration; torch.distributed (gloo) is used only for rendezvous and barriers.  `--gpus N` without a
torch.distributed.run environment makes this script start its own N workers (as a child process, before
anything touches HIP) and relay rank 0's line.

Prints ONE JSON line (rank 0): the driver's contract fields, `roofline` (achieved = the bytes the EXECUTED
algorithm must move / the dominant kernel's HIP-event time, so frac <= 1), `cpu_baseline`, and `extra`:
the other BASELINE configs timed by the same code in the same process -- nnls (config 3), tv and
tv_accelerated (config 4; timed over >= 100 iterations after warming up INTO the backtracking regime, whatever --steps says),
lasso_two_launch (the two-launch structure), lasso_f32_storage (opt-in float32 storage of A), lasso_wide_131072 (the widest
single-team-of-16 shape), natural_run (configs 2 and 3 to tolerance 1e-5: iterations, loop and whole-call seconds),
inproc_8_row_blocks (the same matrix as 8 row blocks driven from this one process: the single-call multi-device form,
here with all blocks on this GPU), and for N > 1 `config5_shard` (32768 rows per rank = BASELINE config 5's per-GPU shape).
`--inproc` runs the whole N-GPU job from ONE process (ShardedDenseMatrixMap over devices 0..N-1, or --devices).

Rendezvous between ranks is torch.distributed (gloo) when torch is importable and a dependency-free TCP rendezvous
(`SocketGroup`) otherwise; every wait is bounded, a rank that dies takes the job down (non-zero exit) instead of parking the
others in a collective, and `ranks_seen != world` is an error.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
FUSED_OPT = {"auto": "auto", "on": True, "off": False}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="the headline's timed region (a fresh solve: setup, W warm-up steps, K timed steps) is run "
                    "this many times; value = the median run, the spread is reported (SURVEY.md 8(d): median of >= 5 runs)")
    ap.add_argument("--extra-repeats", type=int, default=3, help="repeats of every sub-result's timed region")
    ap.add_argument("--rows", dest="m", type=int, default=65536, help="total rows of A (default: BASELINE config 2)")
    ap.add_argument("--cols", dest="n", type=int, default=65536)
    ap.add_argument("--workload", default="lasso", choices=["lasso", "nnls", "tv"],
                    help="lasso = BASELINE config 2 (default, the headline); nnls = config 3; tv = config 4 (8192^2 image)")
    ap.add_argument("--image", type=int, default=8192, help="TV image side (workload tv and the tv sub-results)")
    ap.add_argument("--accelerate", action="store_true", help="FISTA (workload tv / lasso / nnls)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the nnls / tv / two-launch sub-results")
    ap.add_argument("--skip-extra", default="", help="comma list of sub-results to skip (e.g. inproc: the two in-process row-block runs, "
                                                     "whose launches of the headline kernel on SMALLER row blocks would otherwise be "
                                                     "averaged into the same rocprofv3 kernel-stats row)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of A the CPU baseline runs on (0 = all)")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--tune", default="", help="comma list key=value of FH_TUNE_* integers, e.g. 3=2,0=8")
    ap.add_argument("--fused", default="auto", choices=["auto", "on", "off"],
                    help="one-pass iteration kernel (fh_step): auto = when the shape supports it")
    ap.add_argument("--storage", default="f64", choices=["f64", "f32"],
                    help="device storage of A: f64 (default, the headline) or f32 (opt-in throughput mode; arithmetic stays float64)")
    ap.add_argument("--preflight-only", action="store_true",
                    help="run the multi-GPU preflight (fasta_python_amd/preflight.py: devices, RCCL, communicator, two checked all-reduces, "
                         "co-residency probe, ranks_seen) and stop; `--gpus N` always runs it first")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rendezvous, broadcast, barrier and max-over-ranks only -- no GPU work (CPU rehearsal of --gpus N)")
    ap.add_argument("--inproc", action="store_true",
                    help="drive all --gpus devices from ONE process (ShardedDenseMatrixMap, fh_create_ex ndev > 1) instead of one process per GPU")
    ap.add_argument("--devices", default="", help="--inproc: comma list of device ids, one per row block (default 0..gpus-1; "
                                                  "a repeated id, e.g. 0,0,0,0, puts every block on that GPU)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torch", "socket"],
                    help="how `--gpus N` from a bare shell starts its workers: torch.distributed.run, or the built-in spawner with the "
                         "TCP rendezvous (auto: torch when importable)")
    ap.add_argument("--job-timeout", type=float, default=3000.0, help="own spawner: wall-clock limit of the whole N-rank job; the ranks are ended "
                    "and the exit status is non-zero when it passes (a rank stuck in a GPU collective is not covered by --rdv-timeout)")
    ap.add_argument("--rdv-timeout", type=float, default=600.0, help="seconds any rendezvous wait may take before the job is abandoned")
    ap.add_argument("--hang-at-rank", type=int, default=-1, help=argparse.SUPPRESS)     # tests: this rank sleeps forever after the first barrier
    ap.add_argument("--die-at-rank", type=int, default=-1, help=argparse.SUPPRESS)      # tests: this rank exits(3) after the first barrier
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------
# multi-process plumbing
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_env(extra=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.update(extra or {})
    return env


def _torch_run_available():
    import importlib.util
    try:
        return importlib.util.find_spec("torch") is not None and importlib.util.find_spec("torch.distributed.run") is not None
    except (ImportError, ValueError):
        return False


def self_launch(args, argv):
    """`python bench.py --gpus N` from a bare shell: start N workers as CHILD processes (this process has made no HIP call and
    makes none) and let rank 0 print the line on the inherited stdout.  With torch: under torch.distributed.run.  Without (or
    --launcher socket): N plain children that meet over the TCP rendezvous of SocketGroup; the parent watches them and, as
    torch.distributed.run does, ends the others as soon as one fails -- a dead rank must not leave its peers in a collective."""
    use_torch = args.launcher == "torch" or (args.launcher == "auto" and _torch_run_available())
    if use_torch:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        return subprocess.run(cmd, env=_worker_env()).returncode
    port = _free_port()
    token = os.urandom(16).hex()                       # the ranks prove they belong to THIS job (the port is open to every local user)
    procs = []
    for rank in range(args.gpus):
        env = _worker_env({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(args.gpus),
                           "FASTA_BENCH_RDV": f"127.0.0.1:{port}", "FASTA_BENCH_TOKEN": token})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = list(procs)
    deadline = time.time() + args.job_timeout          # a rank stuck inside a GPU collective is not covered by the rendezvous timeouts
    while pending:
        time.sleep(0.05)
        if time.time() > deadline:
            print(f"bench.py: the job exceeded --job-timeout {args.job_timeout:.0f} s; ending its {len(pending)} remaining rank(s)", file=sys.stderr)
            for other in pending:
                other.terminate()
            time.sleep(2.0)
            for other in pending:
                if other.poll() is None:
                    other.kill()
            return rc or 124
        for pr in list(pending):
            code = pr.poll()
            if code is None:
                continue
            pending.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:                  # exactly the processes started above
                    other.terminate()
    return rc


class quiet_stdout:
    """Park the C-level stdout (fd 1) on stderr for the duration: gloo announces its peers and RCCL prints a version banner
    there, and the driver reads ONE JSON line from stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.keep, 1)
        os.close(self.keep)


class SocketGroup:
    """Dependency-free rendezvous for the ranks of one node: rank 0 listens on FASTA_BENCH_RDV, the others connect; barrier,
    byte broadcast and max-reduce are one round trip through rank 0.  Every wait is bounded by `timeout` seconds.
    Wire format: fixed binary frames -- [kind: 1 byte][length: 8 bytes little endian][payload] with kind N (nothing), F (one float64),
    B (raw bytes) -- nothing that is received is ever unpickled or evaluated.  A connecting peer first sends a hello frame (the job's
    token from FASTA_BENCH_TOKEN, then its rank as 4 bytes); rank 0 drops connections whose token is not this job's."""
    MAX_FRAME = 1 << 20

    def __init__(self, rank, world, addr, timeout):
        import hmac
        self.rank, self.world, self.local_rank = rank, world, int(os.environ.get("LOCAL_RANK", rank))
        self.force, self.dist = False, self
        self.token = os.environ.get("FASTA_BENCH_TOKEN", "").encode()
        if not self.token:              # the port is open to every local user: without a job token any connection would pass for a rank
            raise SystemExit("SocketGroup: FASTA_BENCH_TOKEN is empty -- the rendezvous needs the job's token (bench.py's own spawner sets it)")
        host, port = addr.rsplit(":", 1)
        self.peers = []
        if rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((host, int(port)))
            srv.listen(world + 4)
            deadline = time.time() + timeout
            by_rank = {}
            while len(by_rank) < world - 1:
                srv.settimeout(max(0.05, deadline - time.time()))
                conn, _ = srv.accept()                  # socket.timeout after `timeout` seconds in all
                conn.settimeout(min(timeout, 1.0))     # the hello frame follows the connect at once: a silent stranger costs the (single-threaded) accept loop 1 s, not 10
                try:
                    hello = self._recv(conn)
                    ok = (isinstance(hello, bytes) and len(hello) == len(self.token) + 4 and hmac.compare_digest(hello[:-4], self.token))
                    peer = int.from_bytes(hello[-4:], "little") if ok else -1
                    if not (1 <= peer < world) or peer in by_rank:
                        raise ConnectionError("not a rank of this job")
                except (OSError, ValueError, ConnectionError):
                    conn.close()                        # a stranger (or a duplicate): ignored, the job goes on waiting for its ranks
                    continue
                conn.settimeout(timeout)
                by_rank[peer] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    self.conn = socket.create_connection((host, int(port)), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            self.conn.settimeout(timeout)
            self._send(self.conn, self.token + int(rank).to_bytes(4, "little"))

    @staticmethod
    def _send(conn, obj):
        import struct
        if obj is None:
            kind, data = b"N", b""
        elif isinstance(obj, (bytes, bytearray)):
            kind, data = b"B", bytes(obj)
        else:
            kind, data = b"F", struct.pack("<d", float(obj))
        conn.sendall(kind + len(data).to_bytes(8, "little") + data)

    @classmethod
    def _recv(cls, conn):
        import struct

        def exact(k):
            buf = b""
            while len(buf) < k:
                chunk = conn.recv(k - len(buf))
                if not chunk:
                    raise ConnectionError("a rank closed its rendezvous connection (it died?)")
                buf += chunk
            return buf
        head = exact(9)
        kind, size = head[:1], int.from_bytes(head[1:], "little")
        if size > cls.MAX_FRAME or kind not in (b"N", b"B", b"F") or (kind == b"F" and size != 8) or (kind == b"N" and size):
            raise ValueError("malformed rendezvous frame")
        data = exact(size)
        return None if kind == b"N" else (data if kind == b"B" else struct.unpack("<d", data)[0])

    def _gather_scatter(self, value, combine):
        """rank 0 collects one value per rank, combines, sends the result back"""
        if self.world == 1:
            return combine([value])
        if self.rank == 0:
            out = combine([value] + [self._recv(c) for c in self.peers])
            for c in self.peers:
                self._send(c, out)
            return out
        self._send(self.conn, value)
        return self._recv(self.conn)

    def barrier(self):
        self._gather_scatter(None, lambda vs: None)

    def broadcast_bytes(self, payload):
        return self._gather_scatter(payload, lambda vs: vs[0])

    def max(self, value):
        return float(self._gather_scatter(float(value), max))

    def close(self):
        for c in self.peers + ([self.conn] if self.rank else []):
            c.close()


class Group:
    """Rendezvous/barrier plumbing: torch.distributed over gloo when launched with >1 rank (bounded waits)."""

    def __init__(self, timeout=600.0):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # FASTA_BENCH_FORCE_DIST=1 rehearses the multi-process plumbing (gloo + RCCL communicator) with one rank
        self.force = os.environ.get("FASTA_BENCH_FORCE_DIST") == "1"
        if self.world > 1 or self.force:
            import datetime
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            with quiet_stdout():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(seconds=timeout))
                dist.barrier()
            self.dist, self.torch = dist, torch

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def broadcast_bytes(self, payload):
        if not self.dist:
            return payload
        box = [payload]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def max(self, value):
        if not self.dist:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def make_group(timeout=600.0):
    """The rendezvous this process was started under: FASTA_BENCH_RDV (bench.py's own spawner) -> SocketGroup, else torch/gloo."""
    rdv = os.environ.get("FASTA_BENCH_RDV")
    if rdv:
        return SocketGroup(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), rdv, timeout)
    return Group(timeout)


def plumbing_only(args, grp):
    """CPU rehearsal of the N-rank launch: everything bench.py does between ranks except the GPU work."""
    token = grp.broadcast_bytes(bytes(range(128)) if grp.rank == 0 else None)
    assert token == bytes(range(128))
    grp.barrier()
    if args.die_at_rank == grp.rank:                       # tests: a rank that dies mid-job must end the job, not hang it
        os._exit(3)
    if args.hang_at_rank == grp.rank:                      # tests: a rank stuck for good (as inside a GPU collective): --job-timeout ends the job
        time.sleep(1e6)
    t0 = time.perf_counter()
    time.sleep(0.01 * (grp.rank + 1))
    grp.barrier()
    elapsed = grp.max(time.perf_counter() - t0)
    if grp.rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": grp.world, "ranks": grp.world, "max_elapsed_s": elapsed,
                          "rows_per_rank": args.m // grp.world, "rendezvous": type(grp).__name__,
                          "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
    grp.close()


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle is the checker / the baseline, never the product)
# ------------------------------------------------------------------------------------------------------------
def _blas_info():
    try:
        from threadpoolctl import threadpool_info
        blas = [d for d in threadpool_info() if d.get("user_api") == "blas"]
        if blas:
            return int(blas[0]["num_threads"]), blas[0].get("internal_api", "?") + " " + str(blas[0].get("version", ""))
    except Exception:
        pass
    return os.cpu_count() or 1, "?"


def cpu_baseline(A_map, b, mu, n, m_total, rows, iters, repeats):
    """The NumPy oracle loop (oracle/fasta_np.py, parity-pinned to the reference) on this box's host cores, on the same
    matrix pulled back from HBM -- all rows by default -- `repeats` times; the median rate is reported."""
    import numpy as np
    from oracle import fasta_np as fo
    from oracle import problems as pr
    rows = min(rows or m_total, A_map.Wshape[0])
    t0 = time.perf_counter()
    A = A_map.host_rows(0, rows)
    pull_s = time.perf_counter() - t0
    P = pr.sparse_least_squares_from(A, b[:rows], mu)
    rates, passes = [], None
    for _ in range(repeats):
        np.random.seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = fo.fasta(*P.args7(), max_iters=iters, tolerance=0.0)
        rates.append(c.iteration_count / (c.times[c.iteration_count] - c.times[0]))
        passes = c.passes
    rates.sort()
    median = rates[len(rates) // 2]
    threads, blas_name = _blas_info()
    scaled = "" if rows == m_total else f", scaled by {rows}/{m_total}"
    return {
        "value": median * rows / m_total,
        "unit": "iterations/s",
        "cores": threads,
        "kind": "port",
        "runs": [r * rows / m_total for r in rates],
        "sample": (f"oracle NumPy loop ({blas_name}, {threads} BLAS threads, {os.cpu_count()} logical CPUs), median of {repeats} runs of "
                   f"{iters} iterations each on rows 0..{rows} of the same {m_total}x{n} float64 matrix{scaled}; "
                   f"per run A passes={passes['A']} AH={passes['AH']} incl. the setup passes, which lie outside the timed span "
                   f"(times[k]-times[0], as the reference's print_info derives it); D2H of the matrix took {pull_s:.1f} s"),
    }


def pmc_traffic(kernel_name):
    """HBM bytes per launch of EXACTLY the instantiation this run launched, from the newest committed rocprofv3 PMC summary of this same
    command (profiles/*_pmc_summary.json, produced by scripts/profile_bench.sh + summarize_profile.py).  `kernel_name` is the full
    instantiation as rocprofv3 prints it (see fused_kernel_name); a summary that does not hold that very kernel is REFUSED -- the
    field is null and `traffic_source` says why -- instead of lending another instantiation's bytes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, "no profiles/*_pmc_summary.json"
    path = files[-1]                                   # the newest summary only: an older round's kernels are not this build's
    with open(path) as fh:
        data = json.load(fh)
    rec = data["kernels"].get(kernel_name)
    if rec is None:
        return None, f"{os.path.basename(path)} holds no launch of {kernel_name}: re-run scripts/profile_bench.sh"
    return rec["traffic_bytes"], os.path.basename(path)


def fused_kernel_name(n, storage, tuning, ncu):
    """The one-pass instantiation fh_step launches for rows of n columns, spelled as rocprofv3 spells it."""
    from fasta_python_amd import hip
    (ppt, pipe, team, xlds, nbo), inst = hip.fused_shape(n, storage, tuning.get(hip.TUNE_FUSED_VARIANT, 2 | 32) & 0xFFFF, ncu)
    return f"void k_fused_dense<{ppt}, 1, {pipe}, {team}, {xlds}, {nbo}, {1 if storage == 'f32' else 0}>(FusedP)" if inst else None


# ------------------------------------------------------------------------------------------------------------
# timed loops
# ------------------------------------------------------------------------------------------------------------
def timed_steps(make_solver, ctx, grp, warmup, steps, repeats=1, trace=False):
    """`repeats` runs of: a fresh solve (make_solver().setup()), W untimed steps, exactly K timed steps between barrier + device sync on
    both sides (max over ranks).  Returns the MEDIAN run (its wall clock, its HIP-event kernel times, its backtracks) with the spread
    over the runs next to it -- the first run separately, so that a cold start (first touch of a fresh buffer, clocks) is visible and
    not averaged in -- mirroring the reference's harness, which runs every mode and reads the times off each run
    (fasta/examples/__init__.py:54-91).  trace: wall time of every step of the first run, warm-up included."""
    import numpy as np
    from fasta_python_amd import hip
    runs = []
    for rep in range(max(1, repeats)):
        solver = make_solver()
        step_ms = []
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            solver.setup()
            if trace and rep == 0:
                for _ in range(warmup):
                    ts = time.perf_counter()
                    solver.step()
                    step_ms.append((time.perf_counter() - ts) * 1e3)
            else:
                solver.advance(warmup)
            ctx.timing_reset()
            ctx.timing_enable(True)
            bt0 = solver.total_backtracks
            fs0 = solver.fused_steps
            grp.barrier(); ctx.sync()
            t0 = time.perf_counter()
            if trace and rep == 0:
                for _ in range(steps):
                    ts = time.perf_counter()
                    solver.step()
                    step_ms.append((time.perf_counter() - ts) * 1e3)
            else:
                solver.advance(steps)        # the product's default driver: the library's host-side loop (fh_iterate), K iterations, no Python in between
            ctx.sync(); grp.barrier()
            t1 = time.perf_counter()
            ctx.timing_enable(False)
        elapsed = grp.max(t1 - t0)
        k = {name: ctx.timing_get(kid) for name, kid in
             (("fwd", hip.K_FWD), ("adj", hip.K_ADJ), ("comm", hip.K_COMM), ("fused", hip.K_FUSED), ("aux", hip.K_AUX), ("level", hip.K_LEVEL))}
        runs.append({"elapsed": elapsed, "backtracks": solver.total_backtracks - bt0, "fused_steps": solver.fused_steps - fs0, "k": k,
                     "solver": solver, "step_ms": step_ms})
    order = sorted(range(len(runs)), key=lambda i: runs[i]["elapsed"])
    med = runs[order[len(order) // 2]]
    per_step = [r["elapsed"] / steps * 1e3 for r in runs]
    med["spread"] = {"repeats": len(runs), "ms_per_step": {"first": per_step[0], "min": min(per_step), "median": med["elapsed"] / steps * 1e3,
                                                            "max": max(per_step)},
                     "runs_ms_per_step": per_step, "backtracks_per_run": [r["backtracks"] for r in runs]}
    if trace:
        med["spread"]["first_run_step_ms"] = [round(v, 4) for v in runs[0]["step_ms"]]
        med["spread"]["first_run_step_ms_note"] = f"wall time of each step of the first run: {warmup} warm-up steps, then the {steps} timed ones"
    return med


def kernel_table(per):
    """per: name -> (total_ms, launches, algorithmic bytes per launch).  Returns (dominant name, table)."""
    per = {k: v for k, v in per.items() if v[1]}
    table = {k: {"launches": v[1], "avg_ms": v[0] / v[1], "algorithmic_bytes_per_launch": v[2],
                 "GB/s": v[2] / (v[0] / v[1] * 1e-3) / 1e9} for k, v in per.items()}
    dom = max(per, key=lambda k: per[k][0]) if per else None
    return dom, table


def dense_bytes(m, n, esize=8):
    """Algorithmic HBM bytes per launch of the three dense kernels (SURVEY.md 8(d) vector terms):
    K-fwd reads A, x0, g0, b and writes xhat, xprox, z; K-adj reads A, z, b, x0, xprox, xhat, g... ; the one-pass kernel
    reads A ONCE and moves the union of both vector sets (3m + 7n).  esize = bytes per stored element of A (vectors: 8)."""
    return {"fwd": m * n * esize + (4 * n + 2 * m) * 8, "adj": m * n * esize + (2 * m + 5 * n) * 8,
            "fused": m * n * esize + (3 * m + 7 * n) * 8}


def run_dense(args, grp, A, m_total, n, workload, fused, steps, warmup, accelerate=False, plain=False, repeats=None, trace=False):
    """One dense workload (LASSO or NNLS) on the resident matrix `A` (this rank's row block); the three modes of the reference's
    test_modes (examples/__init__.py:66-91): adaptive (default), accelerated (FISTA), plain (neither)."""
    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd import synthetic
    ctx = A.ctx
    m_local = A.Wshape[0]              # rows this PROCESS holds (all of them for an in-process sharded map)
    row0 = A.rows[0]
    blocks = ctx.shard_count() if hasattr(ctx, "shard_count") else 1
    mu, sigma = 0.02, (0.005 if workload == "nnls" else 0.01)     # nn_least_squares.py:49 uses 0.005
    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=sigma, row0=row0, m_total=m_total)
    loss = fa.LeastSquares(b)
    if workload == "l1ball":        # examples/lasso.py:45,66: proxg = project_L1_ball(x, mu), mu = 0.8 ||x_true||_1 (sort-free level search on the device)
        mu = 0.8 * float(np.abs(x_true).sum())
    reg = {"lasso": fa.Shrink, "nnls": lambda _: fa.NonNeg(), "l1ball": fa.L1Ball, "linf": fa.LinfProx}[workload](mu)
    def make_solver():
        np.random.seed(3)       # same Lipschitz probes on every rank, in every repeat
        return fa.FBSolver(A, loss, reg, np.zeros(n), adaptive=not (accelerate or plain), accelerate=accelerate, verbose=False,
                           max_iters=warmup + steps, tolerance=0.0, backtrack=True, evaluate_objective=False, fused=fused)
    t = timed_steps(make_solver, ctx, grp, warmup, steps, args.extra_repeats if repeats is None else repeats, trace)
    solver = t["solver"]
    # per LAUNCH: a multi-device context launches once per row block (its timers add up launches and time over the blocks)
    by = dense_bytes(m_local // blocks, n, 4 if getattr(A, "storage", "f64") == "f32" else 8)
    per = {"fasta_fwd(k_fwd_dense)": t["k"]["fwd"] + (by["fwd"],), "fasta_adj(k_adj_dense)": t["k"]["adj"] + (by["adj"],),
           "fasta_step(k_fused_dense)": t["k"]["fused"] + (by["fused"],),
           # the clipping-level search in front of every forward launch of the l-infinity prox / l1-ball kinds: reads x0 and g0
           "fasta_level(k_level_search)": t["k"]["level"] + (2 * n * 8,)}
    dom, table = kernel_table(per)
    if blocks > 1 and getattr(ctx, "devices", None) and len(set(ctx.devices)) == 1:
        # row blocks that share ONE device are timed on one block (the middle one) and scaled by the number of blocks: an estimate
        for row in table.values():
            row["timing"] = f"1 of {blocks} blocks x {blocks} (sampled, fh_timing_enable)"
    loop_bytes = sum(v[1] * v[2] for v in per.values())
    # SURVEY.md 8(d) prices every iteration at TWO passes over A (N_A = iters + backtracks, N_AH = iters)
    model_bytes = ((steps + t["backtracks"]) * by["fwd"] + steps * by["adj"])
    comm_ms, comm_cnt = t["k"]["comm"]
    return {
        "value": steps / t["elapsed"], "ms_per_step": t["elapsed"] / steps * 1e3, "elapsed": t["elapsed"],
        "backtracks": t["backtracks"], "fused_steps": t["fused_steps"], "dominant": dom, "per_kernel": table,
        "loop_GB/s_wallclock": loop_bytes / t["elapsed"] / 1e9,
        "vs_two_pass_model": {"bytes_per_iteration": by["fwd"] + by["adj"], "GB/s": model_bytes / t["elapsed"] / 1e9,
                              "frac": model_bytes / t["elapsed"] / 1e9 / HBM_PEAK_GBS,
                              "note": "SURVEY.md 8(d) byte model (A read twice per iteration) / wall-clock: exceeds the spec peak "
                                      "when the one-pass kernel reads A once"},
        "comm_avg_ms": comm_ms / comm_cnt if comm_cnt else None, "comm_launches": comm_cnt,
        "level_search_share_of_kernel_time": (t["k"]["level"][0] / sum(v[0] for v in per.values()) if t["k"]["level"][1] else None),
        "solver": solver, "b": b, "mu": mu, "spread": t["spread"],
    }


def natural_runs(A, n, m_total):
    """the natural runs of configs 2 and 3 -- default options, tolerance 1e-5, through fasta() itself."""
    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd import synthetic
    out = {}
    x_true = synthetic.sparse_signal(n, seed=1)
    for name, reg, sigma in (("lasso", fa.Shrink(0.02), 0.01), ("nnls", fa.NonNeg(), 0.005)):
        b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=sigma, row0=A.rows[0], m_total=m_total)
        ls = fa.LeastSquares(b)
        np.random.seed(3)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, tolerance=1e-5, backend="hip")
        wall = time.perf_counter() - t0
        k = c.iteration_count
        loop = c.times[k] - c.times[0]
        out[name] = {"iterations": int(k), "backtracks": int(c.backtracks), "loop_s": loop, "whole_call_s": wall,
                     "iterations_per_s": k / loop, "final_residual": float(c.residuals[k - 1]),
                     # (plain sums, not np.linalg.norm: a BLAS call here wakes OpenBLAS's worker threads, whose busy-wait under the box's
                     # CPU quota stalls the main thread for tens of ms a few iterations into the NEXT solve -- scripts/probes/natural_run.py)
                     "rel_error_vs_x_true": math.sqrt(float(np.sum((c.solution - x_true) ** 2)) / float(np.sum(x_true ** 2)))}
    out["note"] = ("fasta(A, ls.f, ls.gradf, reg.g, reg.prox, x0, tolerance=1e-5) on the resident matrix: loop_s = times[k] - times[0] "
                   "(the reference's print_info span), whole_call_s adds the setup passes (Lipschitz probes, init) and the D2H of the solution")
    return out


def device_loop_runs(grp, sizes=((512, 1024), (2048, 2048), (4096, 4096), (6000, 6000)), iters=512, per_launch=64, repeats=3):
    """fasta(..., device_iters=K) -- the FBS loop itself in persistent launches (fh_run, csrc/fh_run.h) -- against the per-iteration path on
    the reference's own problem sizes (its examples are 200 x 1000 ... 1000 x 2000; SURVEY.md 8(d) config 1 is 512 x 1024), where a launch and
    the host round trip cost more than the iteration's arithmetic.  Same synthetic LASSO recipe as the headline; best of `repeats` solves."""
    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd import synthetic
    out = {}
    for m, n in sizes:
        A = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n), device=grp.local_rank)
        try:
            x_true = synthetic.sparse_signal(n, seed=1)
            b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
            loss, reg = fa.LeastSquares(b), fa.Shrink(0.02)
            rates = {}
            for name, kw in (("python_driver", {"device_iters": 0}), ("per_iteration_launches", {}), ("device_loop", {"device_iters": per_launch})):
                best, steps_on_device = 0.0, 0
                for _ in range(repeats):
                    np.random.seed(3)
                    solver = fa.FBSolver(A, loss, reg, np.zeros(n), verbose=False, max_iters=iters, tolerance=0.0, **kw)
                    with warnings.catch_warnings(), np.errstate(all="ignore"):
                        warnings.simplefilter("ignore")
                        solver.setup()
                        A.ctx.sync()
                        t0 = time.perf_counter()
                        solver.run()
                        A.ctx.sync()
                        best = max(best, solver.i / (time.perf_counter() - t0))
                    steps_on_device = solver.device_steps
                rates[name] = {"iterations/s": best, "us_per_iteration": 1e6 / best, "iterations_inside_persistent_launches": steps_on_device}
            rates["speedup"] = rates["device_loop"]["iterations/s"] / rates["per_iteration_launches"]["iterations/s"]
            rates["library_loop_vs_python_driver"] = rates["per_iteration_launches"]["iterations/s"] / rates["python_driver"]["iterations/s"]
            if (m, n) == (512, 1024):
                # BASELINE config 1 as BASELINE states it: the same problem on the NumPy CPU path -- the product's generic host loop with the
                # reference's closure forms (examples/sparse_least_squares.py:41-44) on a host copy of the matrix; no GPU involved
                Ah = A.host_rows(0, m)
                mu = 0.02
                best = 0.0
                for _ in range(repeats):
                    np.random.seed(3)
                    t0 = time.perf_counter()
                    with warnings.catch_warnings(), np.errstate(all="ignore"):
                        warnings.simplefilter("ignore")
                        r = fa.fasta(Ah, Ah.T, lambda z: .5 * np.linalg.norm((z - b).ravel()) ** 2, lambda z: z - b,
                                     lambda x: mu * np.linalg.norm(x.ravel(), 1), lambda x, t: fa.proximal.shrink(x, t * mu),
                                     np.zeros(n), verbose=False, max_iters=iters, tolerance=0.0, backend="numpy")
                    k = r.iteration_count
                    best = max(best, k / (r.times[k] - r.times[0]))
                rates["numpy_host_loop"] = {"iterations/s": best, "us_per_iteration": 1e6 / best,
                                            "note": "BASELINE config 1: fasta(A, A.T, f, gradf, g, proxg, x0) with Python closures on host arrays (generic loop, bit-identical to the reference)"}
            out[f"{m}x{n}"] = rates
        finally:
            A.close()
    out["note"] = (f"{iters} iterations, tolerance 0, adaptive FBS with backtracking; python_driver = device_iters=0 (FBSolver.step between all launches, rounds 1-5); "
                   "per_iteration_launches = the default (one launch per iteration issued by the library's host-side loop fh_iterate: same launches, same decisions, "
                   f"bit-identical histories); device_loop = fasta(..., device_iters={per_launch}): backtracking test, "
                   "Barzilai-Borwein step, residuals, best iterate and the stop rule are decided on the device, histories come back once per launch; opt-in")
    return out


def settle_after_free(gib):
    """The driver clears freed device memory in the background: for ~30 ms per GiB freed a read-only stream over OTHER, resident memory runs
    3 % slow (profiles/r05_free_aftermath.txt: 7.16 -> 6.95 TB/s for 2.9 s after 96 GiB were freed).  A sub-result timed right behind a
    large hipFree would carry that; wait it out (about 10 s over the whole default run)."""
    time.sleep(0.2 + 0.035 * gib)


def wait_for_driver_clearing(device, limit_s=20.0):
    """Before the first large allocation of this process: has somebody else's memory just been freed on this device (the test-suite that ran a moment
    ago, an earlier bench process)?  The driver clears it in the background for 24-30 ms per GiB, and everything timed meanwhile runs 1-3 % slow
    (profiles/r05_free_aftermath.txt; the headline of profiles/r06_bench_under_kernel_trace.json, which started 2 s behind the previous bench process:
    4.87 ms per launch, its sub-results 4.78).  hipMemGetInfo reports that memory free at once; the device's sysfs counter mem_info_vram_used keeps it until
    it HAS been cleared (profiles/r06_placement.txt, section 10).  Wait until the two agree (at most limit_s).  Returns what it saw, or None where the
    counter cannot be read."""
    import ctypes as C
    try:
        from fasta_python_amd import hip
        hip.load_library()                                    # (libamdhip64 is in the process from here on)
        rt = None
        for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            try:
                rt = C.CDLL(name); break
            except OSError:
                continue
        if rt is None:
            return None
        buf = C.create_string_buffer(64)
        if rt.hipDeviceGetPCIBusId(buf, C.c_int(64), C.c_int(device)) != 0:
            return None
        path = f"/sys/bus/pci/devices/{buf.value.decode().strip().lower()}/mem_info_vram_used"
        if not os.path.exists(path) or rt.hipSetDevice(C.c_int(device)) != 0:
            return None
        free, total = C.c_size_t(0), C.c_size_t(0)
        t0, first = time.perf_counter(), None
        while True:
            with open(path) as f:
                used_driver = int(f.read())
            if rt.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
                return None
            excess = used_driver - (total.value - free.value)
            first = excess if first is None else first
            if excess < (1 << 30) or time.perf_counter() - t0 > limit_s:
                break
            time.sleep(0.05)
        return {"freed_but_not_yet_cleared_GiB_at_start": round(max(first, 0) / 2 ** 30, 2), "waited_s": round(time.perf_counter() - t0, 2),
                "note": "sysfs mem_info_vram_used minus what hipMemGetInfo counts as in use; the timed regions start after the two agree"}
    except Exception:          # (a probe of the environment: never a reason for the bench to fail)
        return None


def sub_result(r, workload):
    d = r["per_kernel"].get(r["dominant"], {}) if r["dominant"] else {}
    return {"workload": workload, "value": r["value"], "unit": "iterations/s", "ms_per_step": r["ms_per_step"],
            "backtracks_in_timed_steps": r["backtracks"], "kernel": r["dominant"], "avg_launch_ms": d.get("avg_ms"),
            "achieved_GB/s": d.get("GB/s"), "frac": d.get("GB/s") / HBM_PEAK_GBS if d else None,
            "per_kernel": r["per_kernel"], "spread": r.get("spread"),
            **({"level_search_share_of_kernel_time": r["level_search_share_of_kernel_time"]} if r.get("level_search_share_of_kernel_time") is not None else {})}


def tv_bytes(P, accelerate, zfree=True):
    """Algorithmic bytes per launch of the stencil kernels as executed: neither the gradient nor -- in the
    z-free one-pass kernel -- z is materialised.  One-pass: reads x0 16 + b 8, writes xprox 16 per pixel; with FISTA the iterate is
    kept as (prox output, previous prox output, coefficient): reads 2 x 16 + 8, writes 16.  The z-streaming one-pass kernels
    (FH_TUNE_TV_ZFREE = 0) add a read and a write of z: 56 / 80.  Two launches: 56 + 56 (+48 FISTA)."""
    fused = (56 if accelerate else 40) if zfree else (80 if accelerate else 56)
    return {"fwd": 56 * P, "adj": (56 + (48 if accelerate else 0)) * P, "fused": fused * P}


def tv_kernel_name(accelerate, zfree):
    if zfree:
        return "fasta_step_accel(k_tv_onepass<accel>)" if accelerate else "fasta_step(k_tv_onepass)"
    return "fasta_step_accel(k_fused_tv_accel)" if accelerate else "fasta_step(k_fused_tv_step)"


def run_tv(args, grp, steps, warmup, fused, accelerate, repeats=3):
    """BASELINE config 4: TV denoising dual on an image of side --image, 1 GPU."""
    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd.examples.tv_denoising import checkerboard
    side = args.image
    np.random.seed(7)
    M = checkerboard(side, side, max(1, side // 32))
    M += 0.1 * np.random.standard_normal(M.shape)
    mu = 0.1
    A = fa.GradDivMap(M.shape, device=grp.local_rank)
    zfree = True
    for item in filter(None, args.tune.split(",")):
        k, v = item.split("=")
        A.ctx.set_tuning(int(k), int(v))
        if int(k) == 10:                               # FH_TUNE_TV_ZFREE: which one-pass kernel runs, hence its bytes and its name
            zfree = bool(int(v))
    try:
        loss, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        def make_solver():
            np.random.seed(3)
            return fa.FBSolver(A, loss, reg, np.zeros(M.shape + (2,)), adaptive=not accelerate, accelerate=accelerate,
                               verbose=False, max_iters=warmup + steps, tolerance=0.0, fused=fused)
        t = timed_steps(make_solver, A.ctx, grp, warmup, steps, repeats)
    finally:
        A.close()
    P = side * side
    by = tv_bytes(P, accelerate, zfree)
    per = {"fasta_fwd(k_fwd_tv_step)": t["k"]["fwd"] + (by["fwd"],), "fasta_adj(k_adj_tv_step)": t["k"]["adj"] + (by["adj"],),
           tv_kernel_name(accelerate, zfree): t["k"]["fused"] + (by["fused"],)}
    dom, table = kernel_table(per)
    model_bytes = (steps * 136 + t["backtracks"] * 64) * P
    return {
        "value": steps / t["elapsed"], "ms_per_step": t["elapsed"] / steps * 1e3, "elapsed": t["elapsed"],
        "steps": steps, "warmup": warmup,
        # what a stretch of iterations WITHOUT backtracking runs at: every launch (one per iteration + one per backtrack) costs the same
        "backtrack_free_value": (steps + t["backtracks"]) / t["elapsed"],
        "backtracks": t["backtracks"], "fused_steps": t["fused_steps"], "dominant": dom, "per_kernel": table,
        "loop_GB/s_wallclock": sum(v[1] * v[2] for v in per.values()) / t["elapsed"] / 1e9,
        "vs_materialised_model": {"bytes_per_iteration": 136 * P, "GB/s": model_bytes / t["elapsed"] / 1e9,
                                  "frac": model_bytes / t["elapsed"] / 1e9 / HBM_PEAK_GBS,
                                  "note": "SURVEY.md 8(d) materialised-vector model (136*P per iteration + 64*P per backtrack) / wall-clock"},
        "side": side, "spread": t["spread"],
    }


def tv_small_runs(grp, side=512, iters=600, repeats=3):
    """The stencil at the REFERENCE's own size (fasta/examples/tv_denoising.py:113-125 denoises the 512 x 512 `ascent()`; here the same
    checkerboard + noise as the 8192^2 runs): one sweep moves 40 P = 10.5 MB (~2 us of HBM time), so an iteration is what it costs to
    launch one kernel and get 16 scalars back.  Python driver (device_iters=0) and the library's host-side loop (default) side by side, the
    sweep's own HIP-event time next to both, adaptive and FISTA; plus the natural run (default options, tolerance 1e-5)."""
    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd import hip
    from fasta_python_amd.examples.tv_denoising import checkerboard
    np.random.seed(7)
    M = checkerboard(side, side, max(1, side // 32))
    M += 0.1 * np.random.standard_normal(M.shape)
    mu = 0.1
    A = fa.GradDivMap(M.shape, device=grp.local_rank)
    out = {}
    try:
        loss, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        x0 = np.zeros(M.shape + (2,))
        for mode, acc in (("adaptive", False), ("accelerated", True)):
            res = {}
            for name, kw in (("python_driver", {"device_iters": 0}), ("library_loop", {})):
                best, sweep_us, launches, bts = 0.0, None, 0, 0
                for rep in range(repeats + 1):                   # the last pass carries the HIP events (two records per launch cost a few us: kept out of the rate)
                    timed = rep == repeats
                    np.random.seed(3)
                    solver = fa.FBSolver(A, loss, reg, x0, adaptive=not acc, accelerate=acc, verbose=False, max_iters=iters, tolerance=0.0, **kw)
                    with warnings.catch_warnings(), np.errstate(all="ignore"):
                        warnings.simplefilter("ignore")
                        solver.setup()
                        solver.advance(60)                       # into the backtracking regime, as the 8192^2 runs
                        A.ctx.timing_reset(); A.ctx.timing_enable(timed)
                        bt0, i0 = solver.total_backtracks, solver.i
                        A.ctx.sync()
                        t0 = time.perf_counter()
                        solver.advance(iters)
                        A.ctx.sync()
                        el = time.perf_counter() - t0
                        A.ctx.timing_enable(False)
                    if timed:
                        ms, cnt = A.ctx.timing_get(hip.K_FUSED)
                        sweep_us, launches = (ms / cnt * 1e3 if cnt else None), cnt
                    elif (solver.i - i0) / el > best:
                        best, bts = (solver.i - i0) / el, solver.total_backtracks - bt0
                per_launch = 1e6 * (iters - 60) / best / launches if launches else None
                res[name] = {"iterations/s": best, "us_per_iteration": 1e6 / best, "backtracks": int(bts), "launches": int(launches),
                             "us_per_launch_wallclock": per_launch, "sweep_us_hip_events": sweep_us,
                             "launch_to_sweep_ratio": per_launch / sweep_us if sweep_us and per_launch else None}
            res["library_loop_vs_python_driver"] = res["library_loop"]["iterations/s"] / res["python_driver"]["iterations/s"]
            out[mode] = res
        np.random.seed(3)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = fa.fasta(A, loss.f, loss.gradf, reg.g, reg.prox, x0, verbose=False, tolerance=1e-5, backend="hip")
        wall = time.perf_counter() - t0
        k = c.iteration_count
        out["natural_run"] = {"iterations": int(k), "backtracks": int(c.backtracks), "loop_s": c.times[k] - c.times[0], "whole_call_s": wall,
                              "iterations_per_s": k / (c.times[k] - c.times[0])}
    finally:
        A.close()
    P = side * side
    out["side"], out["algorithmic_bytes_per_sweep"] = side, {"adaptive": 40 * P, "accelerated": 56 * P}
    out["note"] = (f"TV denoising {side}x{side} float64 -- the reference example's own image size (tv_denoising.py:113-125) -- {iters - 60} timed iterations after 60 "
                   "warm-up ones, tolerance 0, best of 3; k_tv_onepass moves 40 P (adaptive) / 56 P (FISTA) bytes per sweep; at this size the iteration is the launch + "
                   "the one synchronisation, not the sweep: launch_to_sweep_ratio = wall clock per launch / the sweep's HIP-event time")
    return out


def tv_line(args, r, accelerate):
    d = r["per_kernel"][r["dominant"]]
    side = r["side"]
    return {
        "metric": "FBS iterations/sec, TV denoising dual (div/grad stencil, unit-ball prox)",
        "value": r["value"], "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "spread": r["spread"],
        "config": {"workload": f"TV denoising {side}x{side} float64 (BASELINE config 4), {'FISTA' if accelerate else 'adaptive FBS'} with backtracking",
                   "backtracks_in_timed_steps": r["backtracks"], "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": d["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["GB/s"] / HBM_PEAK_GBS,
                     # (the sweep's default instantiations: plain 2-row trips in one buffer, FISTA 4-row trips in three; non-temporal stores)
                     "traffic": pmc_traffic(("void k_tv_onepass<0, 1, 4, 2, 3, 0>(TvZP)" if accelerate else "void k_tv_onepass<0, 0, 2, 2, 1, 0>(TvZP)"))[0]
                                if side == 8192 and "k_tv_onepass" in r["dominant"] and not args.tune else None,
                     "kernel": r["dominant"], "avg_launch_ms": d["avg_ms"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                     "per_kernel": r["per_kernel"], "loop_GB/s_wallclock": r["loop_GB/s_wallclock"],
                     "vs_materialised_model": r["vs_materialised_model"]},
    }


# ------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    # every launch mode (own spawner, torch.distributed.run started by somebody else, one process): before anything loads HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC; RCCL fails without it
    if args.gpus > 1 and not args.inproc and "WORLD_SIZE" not in os.environ and os.environ.get("FASTA_BENCH_FORCE_DIST") != "1":
        raise SystemExit(self_launch(args, argv))          # before any import that could initialise HIP
    grp = make_group(args.rdv_timeout)
    if grp.world != (1 if args.inproc else args.gpus):
        raise SystemExit(f"--gpus {args.gpus}{' --inproc (one process)' if args.inproc else ''} but WORLD_SIZE={grp.world}")
    if args.plumbing_only:
        return plumbing_only(args, grp)

    import numpy as np
    import fasta_python_amd as fa
    from fasta_python_amd import hip, synthetic

    # device of this rank: LOCAL_RANK, folded onto the devices that exist (a node with fewer GPUs than ranks -- e.g. a rehearsal of
    # `--gpus 2` on a one-GPU box -- puts several ranks on one device; whether RCCL accepts that is RCCL's call)
    ndev = max(1, hip.device_count())
    ranks_on_my_device = len([r for r in range(grp.world) if r % ndev == grp.local_rank % ndev]) if grp.world > ndev else 1
    grp.local_rank = grp.local_rank % ndev
    fused = FUSED_OPT[args.fused]
    before_start = wait_for_driver_clearing(grp.local_rank)
    # ---- multi-GPU preflight: before anything large is allocated, its verdict is the first thing on stderr -------------------------
    if args.gpus > 1 or grp.force or args.preflight_only:
        from fasta_python_amd import preflight
        if args.inproc:
            line = preflight.inproc_check([int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus)), args.n)
        else:
            share = max(32, hip.device_cus(grp.local_rank) // ranks_on_my_device // 32 * 32) if ranks_on_my_device > 1 else 0
            line = preflight.rank_check(grp, args.n, share, quiet_stdout)
        if grp.rank == 0:
            print(line, file=sys.stderr, flush=True)
        if args.preflight_only:
            return grp.close()
    if args.workload == "tv":
        if grp.world != 1:
            raise SystemExit("the TV workload is single-GPU (BASELINE config 4)")
        print(json.dumps(tv_line(args, run_tv(args, grp, args.steps, args.warmup, fused, args.accelerate, args.repeats), args.accelerate)))
        return grp.close()

    m_total, n = args.m, args.n
    assert m_total % grp.world == 0
    m_local = m_total // grp.world
    row0 = grp.rank * m_local
    tuning = {}
    for item in filter(None, args.tune.split(",")):
        k, v = item.split("=")
        tuning[int(k)] = int(v)
    if ranks_on_my_device > 1 and hip.TUNE_FUSED_CUS not in tuning:
        # ranks that SHARE a device each take their share of its CUs for the one-pass kernel (whole-CU workgroups, all resident at
        # once): the ranks' grids are then co-resident by construction instead of timing each other out
        tuning[hip.TUNE_FUSED_CUS] = max(32, hip.device_cus(grp.local_rank) // ranks_on_my_device // 32 * 32)
    inproc_devices = None
    if args.inproc:
        inproc_devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
        assert len(inproc_devices) == args.gpus, "--devices must name one device per row block (--gpus of them)"

    def shard(m_all, storage=None):
        """This rank's row block of the synthetic (m_all x n) matrix, generated in HBM, with the RCCL communicator attached
        (--inproc: the whole matrix as row blocks over the devices of this one process)."""
        if inproc_devices is not None:
            return fa.ShardedDenseMatrixMap.synthetic(m_all, n, seed=0, scale=synthetic.lasso_scale(m_all, n), devices=inproc_devices,
                                                      tuning=tuning, storage=storage or args.storage)
        rows = m_all // grp.world
        A = fa.DenseMatrixMap.synthetic(rows, n, seed=0, scale=synthetic.lasso_scale(m_all, n), row0=grp.rank * rows,
                                        m_total=m_all, device=grp.local_rank, tuning=tuning, storage=storage or args.storage)
        if grp.world > 1 or grp.force:
            with quiet_stdout():                       # (RCCL's version banner goes to stdout)
                uid = grp.broadcast_bytes(hip.comm_unique_id() if grp.rank == 0 else None)
                A.ctx.comm_init(grp.world, grp.rank, uid)
        return A

    A = shard(m_total)
    ctx = A.ctx
    ranks_seen = ctx.comm_count()
    want_ranks = args.gpus if (args.inproc or grp.world > 1) else 1
    if ranks_seen != want_ranks:
        raise SystemExit(f"row sharding over {want_ranks} GPUs was asked for but the communicator reports {ranks_seen} rank(s)")
    main_r = run_dense(args, grp, A, m_total, n, args.workload, fused, args.steps, args.warmup, args.accelerate, repeats=args.repeats)
    dom = main_r["dominant"]
    d = main_r["per_kernel"][dom]
    ncu = ctx.cu_count()[0]
    dom_kernel = (fused_kernel_name(n, args.storage, tuning, ctx.cu_count()[1]) if "fused" in dom else
                  ("void k_adj_dense<4, 1, 0>(AdjP)" if "adj" in dom else "void k_fwd_dense<8, 1, 1, 0>(FwdP)"))
    traffic, traffic_src = (pmc_traffic(dom_kernel) if (m_total, n, args.gpus, args.storage) == (65536, 65536, 1, "f64") and dom_kernel
                            else (None, "PMC summaries are taken at the default configuration (65536 x 65536 float64, 1 GPU) only"))
    # read-only ceiling on the same buffer: the probe with one and with two persistent workgroups per CU (the one-pass kernel itself
    # can only have one: it uses the whole register file), the better of the two is the ceiling quoted
    ceilings = {}
    for wg_per_cu in (1, 2):
        ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, ncu * wg_per_cu)
        ms, ceil_bytes = ctx.stream_read_ms(3)
        ceilings[wg_per_cu] = ceil_bytes / ms / 1e6
    ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, tuning.get(hip.TUNE_FWD_GRID_CAP, 0))
    fused_kind = ctx.fused_supported()

    names = {"lasso": ("LASSO", "soft-threshold"), "nnls": ("NNLS", "non-negativity")}[args.workload]
    result = {
        "metric": "FBS iterations/sec + achieved HBM GB/s, dense A m=n=65536, LASSO prox",
        "value": main_r["value"],
        "unit": "iterations/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": main_r["ms_per_step"],
        "spread": main_r["spread"],
        "protocol": (f"value = the MEDIAN of {args.repeats} runs of the timed region; each run is a fresh solve: setup, {args.warmup} untimed steps, exactly "
                     f"{args.steps} timed steps between barrier + device sync on both sides; spread.ms_per_step gives first / min / median / max"),
        "before_start": before_start,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64" if args.storage == "f64" else "f32-storage (A stored float32; vectors, accumulation, scalars float64)",
        "data": "synthetic",
        "config": {"workload": f"{names[0]} dense A {m_total}x{n} {'float64' if args.storage == 'f64' else 'float32-storage'}, {names[1]} prox, "
                               f"{'FISTA' if args.accelerate else 'adaptive FBS'} with backtracking"
                               + (f", row-sharded over {args.gpus} GPUs ({m_total // args.gpus} rows each"
                                  f"{', one process driving all devices ' + str(inproc_devices) if args.inproc else ', one process per GPU'})" if args.gpus > 1 else ""),
                   "m": m_total, "n": n, "prox": "shrink" if args.workload == "lasso" else "nonneg", "mu": main_r["mu"],
                   "backtracks_in_timed_steps": main_r["backtracks"],
                   "iteration_structure": ("one launch per iteration (one-pass kernel: both directions from a single read of A)"
                                           if main_r["fused_steps"] else "two launches per iteration (K-fwd, K-adj)"),
                   "fused_supported": fused_kind,        # 0 = this shape has no one-pass kernel (two passes over A per iteration)
                   "parallelism": (f"row-shard x{args.gpus}" + (" in-process" if args.inproc else "")) if args.gpus > 1 else "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": d["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": d["GB/s"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": dom, "kernel_instantiation": dom_kernel, "avg_launch_ms": d["avg_ms"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                     "stream_read_ceiling_GB/s": max(ceilings.values()),
                     "stream_read_GB/s_by_workgroups_per_cu": ceilings,
                     "stream_read_probe": "k_stream_probe<16,1> over the same device copy of A: one or two persistent workgroups per CU (best of both), three rotating "
                                          "register buffers of 16 non-temporal 16-byte loads per lane (32 loads in flight), loads + adds only.  A reference stream, not a bound: "
                                          "its workgroups run free, the one-pass kernel's teams advance in lockstep over one window of rows (rows dealt cyclically, round 6) "
                                          "and read the same bytes 1-1.5 % faster than this probe",
                     "per_kernel": main_r["per_kernel"],
                     "loop_GB/s_wallclock": main_r["loop_GB/s_wallclock"],
                     "vs_two_pass_model": main_r["vs_two_pass_model"],
                     "comm_avg_ms": main_r["comm_avg_ms"], "comm_launches": main_r["comm_launches"],
                     "ranks_seen": ranks_seen,
                     "fused_one_pass_steps": main_r["fused_steps"],
                     "note": "achieved = bytes the executed algorithm must move per launch (A once for the one-pass kernel, "
                             "plus the 3m+7n vector terms) / HIP-event launch time"},
    }

    extra = {}
    if not args.no_extra and not args.accelerate and args.workload == "lasso" and args.storage == "f64":
        if fused is not False and main_r["fused_steps"]:
            r = run_dense(args, grp, A, m_total, n, "lasso", False, args.steps, args.warmup)
            extra["lasso_two_launch"] = sub_result(r, f"LASSO {m_total}x{n}, two launches per iteration (K-fwd + K-adj, the north-star structure)")
        r = run_dense(args, grp, A, m_total, n, "nnls", fused, args.steps, args.warmup)
        extra["nnls"] = sub_result(r, f"NNLS {m_total}x{n} (BASELINE config 3), non-negativity prox, same matrix")
        # the reference's other two modes (examples/__init__.py:66-91 test_modes): accelerated = FISTA with restart, plain = neither
        r = run_dense(args, grp, A, m_total, n, "lasso", fused, args.steps, args.warmup, accelerate=True)
        extra["lasso_accelerated"] = sub_result(r, f"LASSO {m_total}x{n}, FISTA (accelerate=True, adaptive=False), one launch per iteration (fh_step_accel)")
        r = run_dense(args, grp, A, m_total, n, "lasso", fused, args.steps, args.warmup, plain=True)
        extra["lasso_plain"] = sub_result(r, f"LASSO {m_total}x{n}, plain FBS (adaptive=False, accelerate=False)")
        # the two sort-free prox kinds (north_star's "l-infinity ball"; fasta/proximal.py:12-41, examples/lasso.py:45): every forward launch is
        # preceded by the clipping-level search, timed as its own per_kernel row
        r = run_dense(args, grp, A, m_total, n, "l1ball", fused, args.steps, args.warmup)
        extra["lasso_l1ball"] = sub_result(r, f"l1-ball constrained LASSO {m_total}x{n} (examples/lasso.py: proxg = project_L1_ball(x, 0.8 ||x_true||_1)), same matrix")
        r = run_dense(args, grp, A, m_total, n, "linf", fused, args.steps, args.warmup)
        extra["linf"] = sub_result(r, f"least squares + {0.02} ||x||_inf on the same {m_total}x{n} matrix (proxg = project_Linf_ball(x, t mu), fasta/proximal.py:12-31)")
        if args.gpus > 1 or grp.force:                 # (FASTA_BENCH_FORCE_DIST=1 rehearses this branch with one rank)
            # BASELINE config 5's per-GPU shape: 32768 rows per rank (N = 8 gives the 262144 x 65536 matrix itself)
            A.close()
            settle_after_free(m_total // grp.world * n * 8 / 2 ** 30)
            A = shard(32768 * args.gpus)
            ctx = A.ctx
            r = run_dense(args, grp, A, 32768 * args.gpus, n, "lasso", fused, args.steps, args.warmup)
            s = sub_result(r, f"LASSO {32768 * args.gpus}x{n} row-sharded over {args.gpus} GPUs, 32768 rows each "
                              f"(BASELINE config 5 is this at 8 GPUs)")
            s["comm_avg_ms"], s["ranks_seen"] = r["comm_avg_ms"], ctx.comm_count()
            extra["config5_shard"] = s
        else:
            # opt-in float32-storage mode on the same synthetic matrix (rounded to float32): a SEPARATE line, never the headline
            A32 = shard(m_total, "f32")
            nat32 = None
            try:
                r = run_dense(args, grp, A32, m_total, n, "lasso", fused, args.steps, args.warmup)
                nat32 = natural_runs(A32, n, m_total)["lasso"]        # (its set-up is one read of the float32 matrix since round 6)
            finally:
                A32.close()
                settle_after_free(m_total * n * 4 / 2 ** 30)
            s32 = sub_result(r, f"LASSO {m_total}x{n}, A stored float32 (opt-in; vectors, accumulation and scalars float64)")
            s32["dtype"] = "f32-storage"
            s32["natural_run"] = nat32
            s32["tolerance"] = ("iterates equal the reference's run on A.astype(float32) to the float64 path's tolerances; against the "
                                "float64-matrix run they differ by the rounding of A (<= 3e-7 relative away from the chaotic regime)")
            extra["lasso_f32_storage"] = s32
            # wide rows (n in (65536, 131072]: x slice in LDS, posting one row ahead): 32768 x 131072, the same 32 GiB
            if (m_total, n) == (65536, 65536):
                Aw = fa.DenseMatrixMap.synthetic(32768, 131072, seed=0, scale=synthetic.lasso_scale(32768, 131072), device=grp.local_rank)
                try:
                    # (the driver's round-3 line had this sub-result at 5.28 ms / launch against 4.78-4.88 in five builder runs, one sample
                    # each: now `extra_repeats` fresh solves, the first reported separately, and every step of the first run traced)
                    r = run_dense(args, grp, Aw, 32768, 131072, "lasso", fused, args.steps, args.warmup, trace=True)
                    s = sub_result(r, "LASSO 32768x131072 float64 (wide rows: 16 members x 16 pieces, x slice in LDS)")
                    s["fused_supported"] = Aw.ctx.fused_supported()
                    extra["lasso_wide_131072"] = s
                finally:
                    Aw.close()
                    settle_after_free(32)
            # TV: warmed up INTO the backtracking regime (the adaptive run starts backtracking after ~40 iterations and then does so
            # every second or third one) and timed over 100 iterations, whatever --steps / --warmup say
            for key, acc in (("tv", False), ("tv_accelerated", True)):
                r = run_tv(args, grp, max(args.steps, 100), max(args.warmup, 60), "auto", acc, args.extra_repeats)
                s = sub_result(r, f"TV denoising {args.image}x{args.image} (BASELINE config 4), "
                                  f"{'FISTA (test_modes accelerated)' if acc else 'adaptive FBS'}")
                s["steps"], s["warmup"] = r["steps"], r["warmup"]
                s["backtrack_free_value"] = r["backtrack_free_value"]
                s["note"] = ("value = iterations/s over the timed steps INCLUDING their backtracking launches; backtrack_free_value = "
                             "launches/s = the rate of a stretch without backtracking (what round 2 reported)")
                s["vs_materialised_model"] = r["vs_materialised_model"]
                extra[key] = s
            if "tv_512" not in args.skip_extra.split(","):
                extra["tv_512"] = tv_small_runs(grp)
            extra["natural_run"] = natural_runs(A, n, m_total)
            if "device_loop" not in args.skip_extra.split(","):
                extra["device_loop"] = device_loop_runs(grp)
            # the single-call multi-device form (ShardedDenseMatrixMap, fh_create_ex ndev > 1) on this one GPU: the same matrix as 8
            # row blocks of 8192 x 65536, all on this device -- what the row-sharded plumbing (8 local launches, the sum over the
            # blocks, 8 n-side epilogues, one synchronisation) costs next to the single launch of the headline
            if (m_total, n) == (65536, 65536) and "inproc" not in args.skip_extra.split(","):
                A.close()
                settle_after_free(32)
                A8 = fa.ShardedDenseMatrixMap.synthetic(m_total, n, seed=0, scale=synthetic.lasso_scale(m_total, n),
                                                        devices=[grp.local_rank] * 8, tuning=tuning)
                try:
                    r = run_dense(args, grp, A8, m_total, n, "lasso", fused, args.steps, args.warmup)
                    s = sub_result(r, f"LASSO {m_total}x{n} as 8 row blocks of {m_total // 8} rows driven from one process "
                                      f"(ShardedDenseMatrixMap, all blocks on this GPU, sums in block order by an in-library kernel)")
                    s["row_blocks"], s["comm_avg_ms"] = A8.ctx.comm_count(), r["comm_avg_ms"]
                    # the natural run on the row blocks: its set-up is ONE launch of the two-right-hand-side kernel per block + one exchange (round 6)
                    s["natural_run"] = natural_runs(A8, n, m_total)["lasso"]
                    extra["inproc_8_row_blocks"] = s
                finally:
                    A8.close()
                    settle_after_free(32)
                # BASELINE config 5's matrix ITSELF (262144 x 65536 float64 = 128 GiB: it fits one MI355X) as its 8 per-GPU shards of
                # 32768 x 65536, all on this GPU one after the other.  NOT a scaling number: it is the full config-5 problem solved
                # through the row-sharded code path on the hardware a one-GPU box has; an 8-GPU run does each block on its own device.
                A5 = fa.ShardedDenseMatrixMap.synthetic(262144, n, seed=0, scale=synthetic.lasso_scale(262144, n),
                                                        devices=[grp.local_rank] * 8, tuning=tuning)
                try:
                    r = run_dense(args, grp, A5, 262144, n, "lasso", fused, args.steps, args.warmup)
                    s = sub_result(r, "LASSO 262144x65536 (BASELINE config 5's matrix) as its 8 row blocks of 32768 rows, all eight on THIS one GPU "
                                      "(ShardedDenseMatrixMap with a repeated device id): every iteration runs the 8 per-GPU launches back to back")
                    s["row_blocks"], s["comm_avg_ms"] = A5.ctx.comm_count(), r["comm_avg_ms"]
                    s["per_block_launch_ms"] = s["avg_launch_ms"]
                    extra["config5_matrix_on_one_gpu"] = s
                finally:
                    A5.close()
                    settle_after_free(128)
                A = shard(m_total)                     # (the CPU baseline below pulls the matrix back from HBM)
    if extra:
        result["extra"] = extra
    if grp.rank == 0 and grp.world == 1 and not args.no_cpu_baseline and args.workload == "lasso" and not args.accelerate and args.storage == "f64":
        result["cpu_baseline"] = cpu_baseline(A, main_r["b"], main_r["mu"], n, m_total, args.cpu_rows, args.cpu_iters, args.cpu_repeats)
    if grp.rank == 0:
        print(json.dumps(result))
    A.close()
    grp.close()


if __name__ == "__main__":
    main()
