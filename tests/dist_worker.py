"""Worker for tests/test_sharded_cpu.py: one of WORLD_SIZE gloo ranks.  Exercises (1) bench.py's rendezvous plumbing
(unique-id style byte broadcast, barrier, max-over-ranks) and (2) the PRODUCT under row sharding, two ways:

  path "driver"   fasta_python_amd.FBSolver -- the host driver of the device loop -- over a NumPy stand-in for a row-sharded
                  device context (tests/fake_ctx.py with this rank's row block; the loss sum and the A_k^T r_k partials are
                  all-reduced over gloo exactly where fasta_hip.hip calls ncclAllReduce);
  path "generic"  fasta_python_amd.fasta's generic host loop with sharded closures (local forward rows, all-reduced adjoint).

Every rank must take the same branches and end with the same replicated iterate; the parent test compares rank 0 with a
single-process run."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                    # noqa: E402  (Group = the product's rendezvous helper)
import fasta_python_amd as fa                   # noqa: E402
from fasta_python_amd import hip                # noqa: E402
from fasta_python_amd.linalg import LinearMap, _DeviceMap   # noqa: E402
from tests.fake_ctx import FakeContext          # noqa: E402


def problem(seed, M=96, N=160, K=6, sigma=0.01, mu=0.02):
    """examples/sparse_least_squares.py:62-74 recipe (same draws as oracle.problems.sparse_least_squares)."""
    np.random.seed(seed)
    x = np.zeros(N)
    x[np.random.permutation(N)[:K]] = 1
    A = np.random.randn(M, N)
    A /= np.linalg.norm(A, 2)
    b = A @ x + sigma * np.random.randn(M)
    return A, b, mu


class ShardedFakeContext(FakeContext):
    """Row block of A on this rank; sums that fasta_hip.hip all-reduces with RCCL are all-reduced here with gloo.

    The one-pass entry points (step / step_accel) model csrc/fasta_hip.hip:dense_step: the launch's timeout word rides in the same
    all-reduce as g1 and the loss sums, so EVERY rank sees the summed verdict and raises -- `inject_timeout_at` makes this rank's
    k-th one-pass launch report a (local) timeout."""
    sharded = True

    def __init__(self, Ak, allreduce, fused_kind=0, inject_timeout_at=None):
        self.allreduce = allreduce
        self.inject_timeout_at = inject_timeout_at
        self.one_pass_launches = 0
        self.timeout_raised_at = None           # index of the one-pass launch whose summed timeout word was non-zero
        FakeContext.__init__(self, lambda x: Ak @ x, lambda r: allreduce(Ak.T @ r), (Ak.shape[1],), (Ak.shape[0],), fused_kind)

    def _f_sum(self, z):                        # fh_fwd: reduce_fsq_over_ranks; fh_adj: the 1-double all-reduce next to g1
        return float(self.allreduce(np.array([FakeContext._f_sum(self, z)]))[0])

    def _verdict(self, scalars):
        local = 1.0 if self.one_pass_launches == self.inject_timeout_at else 0.0
        k = self.one_pass_launches
        self.one_pass_launches += 1
        if float(self.allreduce(np.array([local]))[0]) != 0.0:       # the word every rank reads back (scalars[15])
            self.timeout_raised_at = k
            raise hip.HipTimeout("fused one-pass kernel: team hand-off timed out (injected)")
        return scalars

    def step(self, tau):
        return self._verdict(FakeContext.step(self, tau))

    def step_accel(self, tau, coef, restart):
        return self._verdict(FakeContext.step_accel(self, tau, coef, restart))


class ShardedFakeMap(_DeviceMap):
    def __init__(self, Ak, allreduce, fused_kind=0, inject_timeout_at=None):
        self.shape = Ak.shape
        self.ctx = ShardedFakeContext(Ak, allreduce, fused_kind, inject_timeout_at)
        LinearMap.__init__(self, self.ctx.fwd_op, self.ctx.adj_op, (Ak.shape[1],), (Ak.shape[0],))


def main():
    out_dir, mode, path = sys.argv[1], sys.argv[2], sys.argv[3]
    grp = bench.Group()
    import torch
    dist = grp.dist
    # ---- plumbing ------------------------------------------------------------------------------
    token = grp.broadcast_bytes(bytes(range(128)) if grp.rank == 0 else None)
    assert token == bytes(range(128))
    assert grp.max(float(grp.rank)) == float(grp.world - 1)
    grp.barrier()

    # ---- sharded FBS through the product ------------------------------------------------------------
    A, b, mu = problem(5)
    m = A.shape[0]
    assert m % grp.world == 0
    lo = grp.rank * (m // grp.world)
    hi = lo + m // grp.world
    Ak, bk = A[lo:hi], b[lo:hi]

    def allreduce(v):
        t = torch.from_numpy(np.array(v, dtype=np.float64, copy=True).reshape(-1))
        dist.all_reduce(t)
        return t.numpy().reshape(np.shape(v))

    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True, verbose=False,
                adaptive=(mode != "accelerated"), accelerate=(mode == "accelerated"))
    x0 = np.zeros(A.shape[1])
    np.random.seed(9)                                   # same Lipschitz probes on every rank
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        extra = {}
        if path == "driver":
            op = ShardedFakeMap(Ak, allreduce)
            ls, reg = fa.LeastSquares(bk), fa.Shrink(mu)
            c = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, x0, **opts)
            assert op.ctx.calls["fwd"] + op.ctx.calls["pair"] > 0
        elif path == "timeout":
            # one-pass kernel on every launch (fused kind 1); rank 1's 4th one-pass launch times out LOCALLY.  Both ranks must
            # drop the one-pass kernel in that same launch (solver.py:_forward) and go on with K-fwd / K-adj in lock step.
            op = ShardedFakeMap(Ak, allreduce, fused_kind=1, inject_timeout_at=3 if grp.rank == 1 else None)
            ls, reg = fa.LeastSquares(bk), fa.Shrink(mu)
            solver = fa.FBSolver(op, ls, reg, x0, **opts).setup()
            assert solver.mode == "always"
            c = solver.run()
            assert not solver.use_fused and op.ctx.timeout_raised_at == 3
            assert op.ctx.calls["fwd"] > 0 and op.ctx.calls["adj"] > 0          # the two-launch path took over
            extra = dict(timeout_raised_at=op.ctx.timeout_raised_at, fused_steps=solver.fused_steps,
                         one_pass_launches=op.ctx.one_pass_launches, two_launch_fwd=op.ctx.calls["fwd"])
        else:
            Ashard = fa.LinearMap(lambda x: Ak @ x, lambda r: allreduce(Ak.T @ r), (A.shape[1],), (hi - lo,))
            f = lambda z: .5 * np.sqrt(float(allreduce(np.sum((z - bk) ** 2)))) ** 2
            gradf = lambda z: z - bk
            g = lambda x: mu * np.abs(x).sum()
            proxg = lambda x, t: fa.proximal.shrink(x, t * mu)
            c = fa.fasta(Ashard, f, gradf, g, proxg, x0, **opts)
    np.savez(os.path.join(out_dir, f"rank{grp.rank}.npz"), residuals=c.residuals, stepsizes=c.stepsizes,
             objectives=c.objectives, iterates=c.iterates, solution=c.solution,
             iteration_count=c.iteration_count, backtracks=c.backtracks, **extra)
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
