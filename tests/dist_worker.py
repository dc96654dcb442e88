"""Worker for tests/test_sharded_cpu.py: one of WORLD_SIZE gloo ranks.  Exercises (1) bench.py's
rendezvous plumbing (unique-id style byte broadcast, barrier, max-over-ranks) and (2) the row-sharding
scheme the HIP path implements -- local forward rows, all-reduced ||r||^2, all-reduced A_k^T r_k -- restated
on the NumPy oracle so it can run without a GPU."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                    # noqa: E402  (Group = the product's rendezvous helper)
from oracle import fasta_np as fo               # noqa: E402
from oracle import problems as pr               # noqa: E402


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    grp = bench.Group()
    import torch
    dist = grp.dist
    # ---- plumbing ------------------------------------------------------------------------------
    token = grp.broadcast_bytes(bytes(range(128)) if grp.rank == 0 else None)
    assert token == bytes(range(128))
    assert grp.max(float(grp.rank)) == float(grp.world - 1)
    grp.barrier()

    # ---- sharded FBS on the oracle ----------------------------------------------------------------
    np.random.seed(5)
    P = pr.sparse_least_squares(M=96, N=160, K=6)
    A, b, mu = P.data["A"], P.data["b"], P.data["mu"]
    m = A.shape[0]
    assert m % grp.world == 0
    lo = grp.rank * (m // grp.world)
    hi = lo + m // grp.world
    Ak, bk = A[lo:hi], b[lo:hi]

    def allreduce(v):
        t = torch.from_numpy(np.array(v, dtype=np.float64, copy=True).reshape(-1))
        dist.all_reduce(t)
        return t.numpy().reshape(np.shape(v))

    Ashard = fo.LinearMap(lambda x: Ak @ x, lambda r: allreduce(Ak.T @ r), (A.shape[1],), (hi - lo,))
    f = lambda z: .5 * np.sqrt(float(allreduce(np.sum((z - bk) ** 2)))) ** 2
    gradf = lambda z: z - bk
    g = lambda x: mu * np.abs(x).sum()
    proxg = lambda x, t: fo.shrink(x, t * mu)
    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True,
                adaptive=(mode != "accelerated"), accelerate=(mode == "accelerated"))
    np.random.seed(9)                                   # same Lipschitz probes on every rank
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = fo.fasta(Ashard, f, gradf, g, proxg, P.x0, **opts)
    np.savez(os.path.join(out_dir, f"rank{grp.rank}.npz"), residuals=c.residuals, stepsizes=c.stepsizes,
             objectives=c.objectives, iterates=c.iterates, solution=c.solution,
             iteration_count=c.iteration_count, backtracks=c.backtracks)
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
