#!/usr/bin/env python
"""bench.py -- FBS iterations/sec + achieved HBM GB/s, dense LASSO (soft-threshold prox), MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one FBS iteration of the product solver (fasta_python_amd.FBSolver.step): one K-fwd
launch + one K-adj launch (+ one K-fwd per backtrack) over a device-resident synthetic matrix.
N = 1: BASELINE.json configs[1], A = 65536 x 65536 float64 (32 GiB).  N > 1: the SAME matrix
row-sharded over N ranks (one process per GPU, strong scaling), one RCCL all-reduce of the A^T
partial sums per iteration; torch.distributed (gloo) is used only for rendezvous and barriers.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

# the HIP library is loaded before anything else that might pull a second HIP runtime into the process
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

HBM_PEAK_GBS = 8000.0     # MI355X spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="m", type=int, default=65536, help="total rows of A (default: BASELINE config 2)")
    ap.add_argument("--cols", dest="n", type=int, default=65536)
    ap.add_argument("--workload", default="lasso", choices=["lasso", "nnls", "tv"],
                    help="lasso = BASELINE config 2 (default, the headline); nnls = config 3; tv = config 4 (8192^2 image)")
    ap.add_argument("--image", type=int, default=8192, help="TV image side (workload tv)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=8192)
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--tune", default="", help="comma list key=value of FH_TUNE_* integers, e.g. 3=2,0=8")
    ap.add_argument("--fused", default="auto", choices=["auto", "on", "off"],
                    help="one-pass iteration kernel (fh_step): auto = when the shape supports it")
    return ap.parse_args()


class Group:
    """Rendezvous/barrier plumbing: torch.distributed over gloo when launched with >1 rank."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # FASTA_BENCH_FORCE_DIST=1 rehearses the multi-process plumbing (gloo + RCCL communicator) with one rank
        self.force = os.environ.get("FASTA_BENCH_FORCE_DIST") == "1"
        if self.world > 1 or self.force:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
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


def cpu_baseline(A_map, b, mu, n, m_total, sample_rows, iters):
    """The NumPy oracle loop (oracle/fasta_np.py, parity-pinned to the reference) on the first
    `sample_rows` rows of the same matrix, on this box's host cores; scaled to the full row count."""
    from oracle import fasta_np as fo
    from oracle import problems as pr
    rows = min(sample_rows, A_map.Wshape[0])
    A = A_map.host_rows(0, rows)
    P = pr.sparse_least_squares_from(A, b[:rows], mu)
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = fo.fasta(*P.args7(), max_iters=iters, tolerance=0.0)
    dt = c.times[c.iteration_count] - c.times[0]
    it_s_sample = c.iteration_count / dt
    try:
        from threadpoolctl import threadpool_info
        blas = [d for d in threadpool_info() if d.get("user_api") == "blas"]
        threads = int(blas[0]["num_threads"]) if blas else (os.cpu_count() or 1)
        blas_name = (blas[0].get("internal_api", "?") + " " + str(blas[0].get("version", ""))) if blas else "?"
    except Exception:
        threads, blas_name = os.cpu_count() or 1, "?"
    return {
        "value": it_s_sample * rows / m_total,
        "unit": "iterations/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"oracle NumPy loop ({blas_name}), {c.iteration_count} iterations on rows 0..{rows} of the same "
                   f"{m_total}x{n} matrix ({it_s_sample:.3f} it/s on the sample, scaled by {rows}/{m_total}; "
                   f"passes A={c.passes['A']} AH={c.passes['AH']} incl. setup are outside the timed span)"),
    }


def pmc_traffic(kernel_substr):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this same command
    (profiles/*_pmc_summary.json, produced by scripts/profile_bench.sh + summarize_profile.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    for path in reversed(files):                       # newest summary that profiled this kernel
        with open(path) as fh:
            data = json.load(fh)
        for name, rec in data["kernels"].items():
            if kernel_substr in name:
                return rec["traffic_bytes"], os.path.basename(path)
    return None, None


def run_tv(args, grp):
    """BASELINE config 4: TV denoising dual on an image of side --image, 1 GPU."""
    from fasta_python_amd.examples.tv_denoising import checkerboard
    side = args.image
    np.random.seed(7)
    M = checkerboard(side, side, max(1, side // 32))
    M += 0.1 * np.random.standard_normal(M.shape)
    mu = 0.1
    A = fa.GradDivMap(M.shape, device=grp.local_rank)
    ctx = A.ctx
    loss, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
    total = args.warmup + args.steps
    solver = fa.FBSolver(A, loss, reg, np.zeros(M.shape + (2,)), adaptive=True, accelerate=False, verbose=False,
                         max_iters=total, tolerance=0.0, fused={"auto": "auto", "on": True, "off": False}[args.fused])
    np.random.seed(3)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        solver.setup()
        for _ in range(args.warmup):
            solver.step()
        ctx.timing_reset(); ctx.timing_enable(True)
        bt0 = solver.total_backtracks
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver.step()
        ctx.sync()
        t1 = time.perf_counter()
        ctx.timing_enable(False)
    elapsed = t1 - t0
    P = side * side
    fwd_ms, fwd_cnt = ctx.timing_get(hip.K_FWD)
    adj_ms, adj_cnt = ctx.timing_get(hip.K_ADJ)
    fus_ms, fus_cnt = ctx.timing_get(hip.K_FUSED)
    per = {"fasta_fwd(k_fwd_tv_step)": (fwd_ms, fwd_cnt, 64 * P), "fasta_adj(k_adj_tv_step)": (adj_ms, adj_cnt, 72 * P),
           "fasta_step(k_fused_tv_step)": (fus_ms, fus_cnt, 136 * P)}
    per = {k: v for k, v in per.items() if v[1]}
    dom = max(per, key=lambda k: per[k][0])
    dms, dcnt, dbytes = per[dom]
    achieved = dbytes / (dms / dcnt * 1e-3) / 1e9
    result = {
        "metric": "FBS iterations/sec, TV denoising dual (div/grad stencil, unit-ball prox)",
        "value": args.steps / elapsed, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"TV denoising {side}x{side} float64 (BASELINE config 4), adaptive FBS with backtracking",
                   "backtracks_in_timed_steps": solver.total_backtracks - bt0, "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("k_fused_tv_step" if "fused" in dom else ("k_fwd_tv_step" if "fwd" in dom else "k_adj_tv_step"))[0]
                     if side == 8192 else None,
                     "kernel": dom,
                     "avg_launch_ms": dms / dcnt, "algorithmic_bytes_per_launch": dbytes,
                     "note": "priced at the materialised-vector model of SURVEY.md 8(d) (64*P / 72*P); this build never "
                             "materialises the gradient: the two step kernels move 56*P each, the one-pass kernel ~60*P for both (fh_tv.h)",
                     "per_kernel": {k: {"launches": v[1], "avg_ms": v[0] / v[1], "GB/s": v[2] / (v[0] / v[1] * 1e-3) / 1e9}
                                    for k, v in per.items() if v[1]},
                     "loop_GB/s_wallclock": (fwd_cnt * 64 * P + adj_cnt * 72 * P + fus_cnt * 136 * P) / elapsed / 1e9},
    }
    print(json.dumps(result))
    A.close()


def main():
    args = parse()
    grp = Group()
    if args.workload == "tv":
        if grp.world != 1:
            raise SystemExit("the TV workload is single-GPU (BASELINE config 4)")
        return run_tv(args, grp)
    args.prox = "nonneg" if args.workload == "nnls" else "shrink"
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={grp.world}: launch with torch.distributed.run")
    m_total, n = args.m, args.n
    assert m_total % grp.world == 0
    m_local = m_total // grp.world
    row0 = grp.rank * m_local
    mu, sigma = 0.02, (0.005 if args.workload == "nnls" else 0.01)     # nn_least_squares.py:49 uses 0.005

    tuning = {}
    for item in filter(None, args.tune.split(",")):
        k, v = item.split("=")
        tuning[int(k)] = int(v)

    scale = synthetic.lasso_scale(m_total, n)
    A = fa.DenseMatrixMap.synthetic(m_local, n, seed=0, scale=scale, row0=row0, m_total=m_total,
                                    device=grp.local_rank, tuning=tuning)
    ctx = A.ctx
    if grp.world > 1 or grp.force:
        uid = grp.broadcast_bytes(hip.comm_unique_id() if grp.rank == 0 else None)
        ctx.comm_init(grp.world, grp.rank, uid)

    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=sigma, row0=row0, m_total=m_total)
    loss = fa.LeastSquares(b)
    reg = fa.Shrink(mu) if args.prox == "shrink" else fa.NonNeg()

    total = args.warmup + args.steps
    solver = fa.FBSolver(A, loss, reg, np.zeros(n), adaptive=True, accelerate=False, verbose=False,
                         max_iters=total, tolerance=0.0, backtrack=True, evaluate_objective=False,
                         fused={"auto": "auto", "on": True, "off": False}[args.fused])
    np.random.seed(3)           # same Lipschitz probes on every rank
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        solver.setup()
        for _ in range(args.warmup):
            solver.step()
        ctx.timing_reset()
        ctx.timing_enable(True)
        bt0 = solver.total_backtracks
        grp.barrier(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver.step()
        ctx.sync(); grp.barrier()
        t1 = time.perf_counter()
        ctx.timing_enable(False)
    elapsed = grp.max(t1 - t0)
    backtracks = solver.total_backtracks - bt0

    fwd_ms, fwd_cnt = ctx.timing_get(hip.K_FWD)
    adj_ms, adj_cnt = ctx.timing_get(hip.K_ADJ)
    comm_ms, comm_cnt = ctx.timing_get(hip.K_COMM)
    fus_ms, fus_cnt = ctx.timing_get(hip.K_FUSED)
    # algorithmic bytes per launch (DESIGN.md section "byte model")
    bytes_fwd = (m_local * n + 2 * n + m_local + 2 * n + m_local) * 8
    bytes_adj = (m_local * n + 2 * m_local + 4 * n + n) * 8
    per = {"fasta_fwd(k_fwd_dense)": (fwd_ms, fwd_cnt, bytes_fwd), "fasta_adj(k_adj_dense)": (adj_ms, adj_cnt, bytes_adj),
           # one launch = both directions: priced at the two-pass algorithmic bytes of SURVEY.md 8(d); it MOVES half
           "fasta_step(k_fused_dense)": (fus_ms, fus_cnt, bytes_fwd + bytes_adj)}
    dom = max(per, key=lambda k: per[k][0])
    dms, dcnt, dbytes = per[dom]
    achieved = dbytes / (dms / dcnt * 1e-3) / 1e9 if dcnt else 0.0
    loop_bytes = fwd_cnt * bytes_fwd + adj_cnt * bytes_adj + fus_cnt * (bytes_fwd + bytes_adj)
    traffic, traffic_src = (pmc_traffic("k_fused_dense" if "fused" in dom else ("k_adj_dense" if "adj" in dom else "k_fwd_dense<8, 1, 1>"))
                            if (m_total, n, grp.world) == (65536, 65536, 1) else (None, None))
    ceil_ms, ceil_bytes = ctx.stream_read_ms(2)

    result = {
        "metric": "FBS iterations/sec + achieved HBM GB/s, dense A m=n=65536, LASSO prox",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": grp.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{'NNLS' if args.workload == 'nnls' else 'LASSO'} dense A {m_total}x{n} float64, "
                               f"{'non-negativity' if args.workload == 'nnls' else 'soft-threshold'} prox, adaptive FBS with backtracking"
                               + (f", row-sharded over {grp.world} GPUs ({m_local} rows each)" if grp.world > 1 else ""),
                   "m": m_total, "n": n, "prox": args.prox, "mu": mu, "backtracks_in_timed_steps": backtracks,
                   "parallelism": f"row-shard x{grp.world}" if grp.world > 1 else "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "stream_read_ceiling_GB/s": ceil_bytes / ceil_ms / 1e6,
                     "kernel": dom, "avg_launch_ms": dms / dcnt if dcnt else None,
                     "algorithmic_bytes_per_launch": dbytes,
                     "per_kernel": {k: {"launches": v[1], "avg_ms": v[0] / v[1] if v[1] else None,
                                        "GB/s": v[2] / (v[0] / v[1] * 1e-3) / 1e9 if v[1] else None} for k, v in per.items()},
                     "loop_GB/s_wallclock": loop_bytes / elapsed / 1e9,
                     "comm_avg_ms": comm_ms / comm_cnt if comm_cnt else None,
                     "fused_one_pass_steps": solver.fused_steps,
                     "note": ("achieved = algorithmic (two-pass) bytes / launch time; the fused one-pass kernel reads A once, "
                              "so its HBM traffic is about half of algorithmic_bytes_per_launch" if "fused" in dom else None)},
    }
    if grp.rank == 0 and grp.world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(A, b, mu, n, m_total, args.cpu_sample_rows, args.cpu_iters)
    if grp.rank == 0:
        print(json.dumps(result))
    A.close()
    grp.close()


if __name__ == "__main__":
    main()
