"""Worker for tests/test_gpu_multiprocess.py: one of WORLD_SIZE processes that ALL use GPU 0.  Each holds one row block of A in its
own HipContext, the communicator is formed through fh_comm_unique_id / fh_comm_init exactly as in a one-process-per-GPU run -- with
FASTA_RCCL_LIB pointing at tests/mock_rccl (real RCCL refuses two ranks on one device) -- and the solve goes through the
product's `fasta()`.  Rendezvous (unique id broadcast, barriers): bench.py's SocketGroup."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                    # noqa: E402
import fasta_python_amd as fa                   # noqa: E402
from fasta_python_amd import hip                # noqa: E402


def main():
    out_dir, mode, m, n, fused = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    probe_no_rank = int(sys.argv[6]) if len(sys.argv) > 6 else -1     # this rank's co-residency probe is made to answer "no"
    driver = sys.argv[7] if len(sys.argv) > 7 else "record"           # "record": record_iterates (Python between iterations); "library" / "python": that driver
    grp = bench.make_group(120.0)
    rng = np.random.RandomState(7)              # the same problem on every rank
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 40)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    lo = grp.rank * m // grp.world
    hi = (grp.rank + 1) * m // grp.world
    op = fa.DenseMatrixMap(np.ascontiguousarray(A[lo:hi]), device=0)
    # The ranks SHARE this GPU, and a one-pass launch needs all of its workgroups resident at once: each rank takes its share of the
    # CUs (world x (CUs / world) whole-CU workgroups are co-resident by construction), so the one-pass kernel must serve every launch.
    dev_cus, _ = op.ctx.cu_count()
    op.ctx.set_tuning(hip.TUNE_FUSED_CUS, max(32, dev_cus // grp.world // 32 * 32))
    if grp.rank == probe_no_rank:
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_PROBE_SAYS_NO)
    uid = grp.broadcast_bytes(hip.comm_unique_id() if grp.rank == 0 else None)
    op.ctx.comm_init(grp.world, grp.rank, uid)
    assert op.ctx.comm_count() == grp.world and op.ctx.sharded
    ls, reg = fa.LeastSquares(b[lo:hi]), fa.Shrink(0.02)
    opts = dict(tolerance=1e-7, evaluate_objective=True, record_iterates=(driver == "record"), max_iters=40, verbose=False,
                adaptive=(mode != "fista"), accelerate=(mode == "fista"),
                fused={"auto": "auto", "on": True, "off": False}[fused])
    if mode == "forced_backtracking":
        opts.update(L=1.0, tau0=5000.0)
    np.random.seed(9)                           # same Lipschitz probes on every rank
    raised = ""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            solver = fa.FBSolver(op, ls, reg, np.zeros(n), **opts, **({} if driver == "record" else dict(driver=driver, device_iters=7))).setup()
        except ValueError as exc:                  # fused=True without a one-pass kernel: must happen on EVERY rank or on none
            raised = str(exc)
    if raised:
        np.savez(os.path.join(out_dir, f"rank{grp.rank}.npz"), raised=raised)
    else:
        op.ctx.timing_reset()
        op.ctx.timing_enable(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = solver.run()
        comm_ms, comm_launches = op.ctx.timing_get(hip.K_COMM)      # exchanges of the loop alone
        np.savez(os.path.join(out_dir, f"rank{grp.rank}.npz"), residuals=c.residuals, stepsizes=c.stepsizes, objectives=c.objectives,
                 iterates=c.iterates if c.iterates is not None else np.zeros(0), solution=c.solution, library_steps=c.library_steps, iteration_count=c.iteration_count, backtracks=c.backtracks,
                 fused_steps=solver.fused_steps, use_fused=int(solver.use_fused), solver_mode=str(solver.mode),
                 backoff=solver._fused_backoff, comm_launches=int(comm_launches), cus=np.array(op.ctx.cu_count()),
                 comm_library=hip.comm_library())
    grp.barrier()
    op.ctx.comm_destroy()
    op.close()
    grp.close()


if __name__ == "__main__":
    main()
