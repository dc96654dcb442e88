"""Fail-fast multi-GPU preflight for the row-sharded operator (the op being sharded is `A @ x` / `A.T @ x`, fasta/linalg.py:41).

A row-sharded run first allocates its row block (16 GiB per GPU at BASELINE config 5) and only then meets RCCL.  This module makes the
first contact cheap and its failures legible: device count, the RCCL library that gets loaded and its version, communicator
initialisation over all N ranks, one all-reduce of 1 KiB and one of n + 3 doubles (the per-iteration exchange of the one-pass kernel)
checked against their closed-form sums, the co-residency probe of the one-pass kernel on every device, and `ranks_seen == N` -- a few
milliseconds of GPU time.  On any failure it prints ONE line naming the step and exits non-zero before anything large is allocated.

    python -m fasta_python_amd.preflight N                # one process per GPU (spawns N ranks through bench.py's launcher)
    python -m fasta_python_amd.preflight N --inproc       # one process driving N devices (fh_create_ex, ndev > 1) [--devices 0,0,...]

`bench.py --gpus N` runs the same checks first (both forms); their verdict is the first thing on its stderr.
"""
import contextlib
import os
import sys
import time

import numpy as np

from . import hip

EXIT_CODE = 3


class PreflightError(SystemExit):
    def __init__(self, step, why, rank=None):
        who = "" if rank is None else f" (rank {rank})"
        print(f"fasta preflight FAILED at step '{step}'{who}: {why}", file=sys.stderr, flush=True)
        super().__init__(EXIT_CODE)


def _step(name, rank, fn):
    try:
        return fn()
    except PreflightError:
        raise
    except (hip.HipError, AssertionError, OSError, ValueError) as exc:
        raise PreflightError(name, f"{type(exc).__name__}: {exc}", rank)


def _exchange_checks(ctx, world, n, rank):
    """ranks_seen and the two all-reduces against their closed forms (collective: every rank runs this)."""
    seen = _step("ranks_seen", rank, ctx.comm_count)
    if seen != world:
        raise PreflightError("ranks_seen", f"the communicator reports {seen} rank(s), {world} were asked for", rank)
    for count, what in ((128, "1 KiB"), (n + 3, f"n + 3 = {n + 3} doubles")):
        err, blocks = _step(f"all-reduce of {what}", rank, lambda c=count: ctx.comm_selftest(c))
        if err != 0.0 or blocks != world:
            raise PreflightError(f"all-reduce of {what}", f"sum over {blocks} block(s) differs from its closed form by {err:g}", rank)


def rank_check(grp, n=65536, one_pass_cus=0, quiet=contextlib.nullcontext):
    """One process per GPU: every rank calls this right after the rendezvous (`grp`: rank, world, local_rank, broadcast_bytes, barrier).
    Returns the summary line (rank 0 prints it)."""
    t0 = time.perf_counter()
    rank, world = grp.rank, grp.world
    ndev = _step("device count", rank, hip.device_count)
    if ndev < 1:
        raise PreflightError("device count", "no HIP device is visible", rank)
    device = grp.local_rank % ndev
    ctx = _step("context", rank, lambda: hip.HipContext(device))
    try:
        if one_pass_cus:
            ctx.set_tuning(hip.TUNE_FUSED_CUS, one_pass_cus)
        with quiet():
            uid = grp.broadcast_bytes(_step("RCCL library + unique id", rank, hip.comm_unique_id) if rank == 0 else None)
            _step(f"communicator init over {world} rank(s)", rank, lambda: ctx.comm_init(world, rank, uid))
        _exchange_checks(ctx, world, n, rank)
        dev_cus, used = ctx.cu_count()
        if not _step("co-residency probe", rank, lambda: ctx.coresident_probe(used)):
            raise PreflightError("co-residency probe", f"{used} whole-CU workgroups do not run side by side on device {device} "
                                 f"({dev_cus} CUs reported): the one-pass kernel would time out (CU mask, partition mode or a co-tenant?)", rank)
        grp.barrier()
        line = (f"fasta preflight ok: {world} rank(s) x 1 GPU, {ndev} device(s) visible, RCCL {hip.comm_library()} (version {hip.comm_version()}), "
                f"all-reduce of 128 and {n + 3} doubles exact, ranks_seen {world}, co-residency probe ok on {used} of {dev_cus} CUs, "
                f"{time.perf_counter() - t0:.2f} s")
        ctx.comm_destroy()
        return line
    finally:
        ctx.close()


def inproc_check(devices, n=65536):
    """One process driving all devices (fh_create_ex with ndev > 1).  Besides the exchange checks: a 256 x 4096 LASSO run through the
    multi-device context against the same run on a single device (the distinct-device branch -- per-device hipSetDevice, grouped
    all-reduce on ncclCommInitAll communicators -- when the ids differ)."""
    import warnings
    from . import DenseMatrixMap, LeastSquares, ShardedDenseMatrixMap, Shrink, fasta
    t0 = time.perf_counter()
    devices = [int(d) for d in devices]
    ndev = _step("device count", None, hip.device_count)
    if max(devices) >= ndev or min(devices) < 0:
        raise PreflightError("device count", f"devices {devices} were asked for, {ndev} device(s) are visible")
    ctx = _step(f"multi-device context over {devices}", None, lambda: hip.HipContext(devices=devices) if len(devices) > 1 else hip.HipContext(devices[0]))
    try:
        _exchange_checks(ctx, len(devices), n, None)
        for k in range(ctx.shard_count()):
            shard = ctx.shard(k)[0]
            dev_cus, used = shard.cu_count()
            if not _step("co-residency probe", None, lambda s=shard, u=used: s.coresident_probe(u)):
                raise PreflightError("co-residency probe", f"{used} whole-CU workgroups do not run side by side on device {devices[k]}")
    finally:
        ctx.close()
    rng = np.random.RandomState(5)
    m, ncols = 256, 4096
    A = rng.randn(m, ncols) / (np.sqrt(m) + np.sqrt(ncols))
    xt = np.zeros(ncols)
    xt[rng.permutation(ncols)[:40]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    runs = []
    for op in (DenseMatrixMap(A, device=devices[0]), ShardedDenseMatrixMap(A, devices=devices) if len(devices) > 1 else DenseMatrixMap(A, device=devices[0])):
        try:
            ls, reg = LeastSquares(b), Shrink(0.02)
            np.random.seed(11)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runs.append(_step("small solve through the multi-device context", None,
                                  lambda: fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(ncols), verbose=False, max_iters=8,
                                                tolerance=0.0, backend="hip")))
        finally:
            op.close()
    one, many = runs
    k = one.iteration_count
    if many.iteration_count != k or many.backtracks != one.backtracks or \
            not np.allclose(many.residuals[:k], one.residuals[:k], rtol=1e-9, atol=0.0) or \
            not np.allclose(many.solution, one.solution, rtol=1e-9, atol=1e-13):
        raise PreflightError("small solve through the multi-device context", "the row-sharded 256 x 4096 solve differs from the single-device solve")
    return (f"fasta preflight ok: 1 process x {len(devices)} row block(s) on devices {devices}, {ndev} device(s) visible, "
            f"exchange = {'RCCL ' + hip.comm_library() + ' (version ' + str(hip.comm_version()) + ')' if len(set(devices)) > 1 else 'in-library sum'}, "
            f"sums of 128 and {n + 3} doubles exact, 256 x 4096 solve equals the single-device solve, {time.perf_counter() - t0:.2f} s")


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m fasta_python_amd.preflight", description=__doc__.split("\n\n")[0])
    ap.add_argument("gpus", type=int, nargs="?", default=1)
    ap.add_argument("--inproc", action="store_true", help="one process driving all devices instead of one process per GPU")
    ap.add_argument("--devices", default="", help="--inproc: comma list of device ids (default 0..gpus-1)")
    ap.add_argument("--cols", type=int, default=65536, help="n of the run being prepared (the second all-reduce carries n + 3 doubles)")
    args = ap.parse_args(argv)
    if args.inproc:
        devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
        print(inproc_check(devices, args.cols), file=sys.stderr, flush=True)
        return 0
    # one process per GPU: the ranks are started (and a dead one is noticed) by bench.py's launcher
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    if not os.path.exists(bench):
        raise SystemExit("the one-process-per-GPU preflight uses bench.py's launcher, which is not next to the package; use --inproc")
    import subprocess
    env = dict(os.environ)
    if args.gpus == 1:
        env["FASTA_BENCH_FORCE_DIST"] = "1"            # a one-rank communicator on real RCCL
    return subprocess.run([sys.executable, bench, "--gpus", str(args.gpus), "--cols", str(args.cols), "--preflight-only"], env=env).returncode


if __name__ == "__main__":
    raise SystemExit(main())
