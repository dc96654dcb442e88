"""One process per GPU -- with two and three REAL processes, all on the one GPU a lease has.

Real RCCL refuses two ranks on one device, so the ranks' communicator is formed by tests/mock_rccl (a stand-in with RCCL's entry
points whose all-reduce stages through POSIX shared memory; FASTA_RCCL_LIB points the library at it).  Everything else is the
product as an 8-GPU run executes it: one HipContext per process holding its row block, fh_comm_unique_id shipped to the peers,
fh_comm_init, per iteration the local one-pass launch (mode 2) -> ONE all-reduce of n + 3 doubles (g1 partials, loss sums, timeout
word) -> n-side epilogue -> scalars; K-fwd / K-adj with their own exchanges when `fused` is off.  Every rank must take the same
branches and hold the same iterate, and the solve must match the oracle (fasta/__init__.py:95-320 restated) and a single-process run."""
import json
import os
import socket
import subprocess
import sys
import warnings

import numpy as np
import pytest

from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mock_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mock_rccl") / "libmock_rccl.so")
    src = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is needed to build the test-only RCCL stand-in")
    subprocess.run([hipcc, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src, "-lrt"], check=True, capture_output=True, timeout=300)
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(world, args, mock, timeout=300):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), FASTA_BENCH_RDV=f"127.0.0.1:{port}",
                   FASTA_BENCH_TOKEN=f"{port:032x}", FASTA_RCCL_LIB=mock, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_gpu_worker.py")] + [str(a) for a in args],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1500:] + se[-3000:]


@pytest.mark.parametrize("world,mode,m,n,fused", [
    (2, "adaptive", 96, 160, "auto"), (2, "fista", 96, 160, "on"), (2, "forced_backtracking", 96, 160, "on"), (2, "adaptive", 96, 160, "off"),
    (3, "adaptive", 300, 4096, "on"), (2, "fista", 256, 20000, "on"), (2, "adaptive", 64, 70000, "on"), (4, "fista", 200, 9000, "on")])
def test_real_processes_row_shard_a_solve_on_one_gpu(tmp_path, mock_rccl, world, mode, m, n, fused):
    _run_ranks(world, [tmp_path, mode, m, n, fused], mock_rccl)
    rng = np.random.RandomState(7)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 40)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    opts = dict(tolerance=1e-7, evaluate_objective=True, record_iterates=True, max_iters=40,
                adaptive=(mode != "fista"), accelerate=(mode == "fista"))
    if mode == "forced_backtracking":
        opts.update(L=1.0, tau0=5000.0)
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks[1:]:                                   # replicated state: bit-identical on every rank
        for key in ("residuals", "stepsizes", "objectives", "solution", "iteration_count", "backtracks", "fused_steps", "use_fused",
                    "comm_launches", "solver_mode"):
            assert np.array_equal(r[key], ranks[0][key]), key
    r0 = ranks[0]
    assert int(r0["iteration_count"]) == want.iteration_count and int(r0["backtracks"]) == want.backtracks
    if mode == "forced_backtracking":
        assert want.backtracks >= 4
    assert str(r0["comm_library"]) == mock_rccl                                   # the substitution is visible, not silent
    if fused == "on":
        # Each rank's one-pass grid is capped to its share of the CUs (FH_TUNE_FUSED_CUS = CUs / world), so the ranks' grids are
        # co-resident on the shared GPU by construction: NO hand-off timeout, no fall-back -- the sharded one-pass path (local launch ->
        # ONE all-reduce of n + 3 doubles -> epilogue) must have served the solve.
        launches = int(r0["iteration_count"]) + int(r0["backtracks"])
        assert int(r0["use_fused"]) == 1 and int(r0["backoff"]) == 64, "a one-pass launch timed out"
        assert int(r0["cus"][1]) * world <= int(r0["cus"][0])
        if str(r0["solver_mode"]) == "always":                                 # the one-pass kernel takes every launch, retries included
            assert int(r0["fused_steps"]) == launches and int(r0["comm_launches"]) == launches
        else:                                                                    # small matrix ("speculative"): K-fwd / K-adj after a backtrack
            assert str(r0["solver_mode"]) == "speculative" and 1 <= int(r0["fused_steps"]) <= launches
            assert int(r0["comm_launches"]) >= launches
    k = want.iteration_count
    np.testing.assert_allclose(r0["residuals"][:k], want.residuals[:k], rtol=1e-6)
    np.testing.assert_allclose(r0["objectives"][:k + 1], want.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(r0["iterates"][:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("world,mode,m,n", [(2, "adaptive", 256, 20000), (3, "fista", 300, 4096), (2, "forced_backtracking", 96, 160)])
def test_every_rank_runs_the_library_loop_and_they_stay_in_step(tmp_path, mock_rccl, world, mode, m, n):
    """Round 6: on a rank of a row-sharded run the loop is driven by fh_iterate too (the default) -- every rank calls it with the same arguments,
    takes the same decisions from the all-reduced scalars and issues the same collectives.  Real rank processes (calls of 7 iterations) against
    the same ranks driven by Python: EQUAL histories and solutions, on every rank; one-read set-up per row block where the shape has one."""
    runs = {}
    for driver in ("library", "python"):
        out = tmp_path / driver
        out.mkdir()
        _run_ranks(world, [out, mode, m, n, "auto", -1, driver], mock_rccl)
        runs[driver] = [np.load(out / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        lib, py = runs["library"][r], runs["python"][r]
        assert int(lib["library_steps"]) == int(lib["iteration_count"]) and int(py["library_steps"]) == 0
        for key in ("residuals", "stepsizes", "objectives", "solution", "iteration_count", "backtracks", "fused_steps", "comm_launches"):
            assert np.array_equal(lib[key], py[key]), (r, key)
            assert np.array_equal(lib[key], runs["library"][0][key]), (r, key)
    if mode == "forced_backtracking":
        assert int(runs["library"][0]["backtracks"]) >= 4


@pytest.mark.parametrize("fused", ["auto", "on"])
def test_one_rank_whose_probe_says_no_takes_every_rank_off_the_one_pass_kernel(tmp_path, mock_rccl, fused):
    """the co-residency verdict behind fh_fused_supported is per context and timing-based; ranks that disagreed
    would issue mismatched collectives (fh_step: one exchange of n + 3; fh_fwd / fh_adj: 1, then n + 1) and hang.  The ranks now AGREE
    on the verdict (a sum over the communicator inside fh_fused_supported).  Rank 1's probe is made to say no (FH_TUNE_FUSED_VARIANT
    bit 128): with fused="auto" BOTH ranks must run the two-launch path and match the oracle; with fused=True BOTH must raise."""
    world, mode, m, n = 2, "adaptive", 256, 20000
    _run_ranks(world, [tmp_path, mode, m, n, fused, 1], mock_rccl)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    if fused == "on":
        assert all("raised" in r.files and "fused=True" in str(r["raised"]) for r in ranks)
        return
    rng = np.random.RandomState(7)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 40)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), tolerance=1e-7, evaluate_objective=True, record_iterates=True, max_iters=40)
    for r in ranks:
        assert int(r["fused_steps"]) == 0 and int(r["use_fused"]) == 0          # nobody launched the one-pass kernel
        assert int(r["iteration_count"]) == want.iteration_count and int(r["backtracks"]) == want.backtracks
        assert int(r["comm_launches"]) >= want.iteration_count and str(r["solver_mode"]) in ("pair", "None")    # K-fwd / K-adj exchanges
    assert np.array_equal(ranks[0]["solution"], ranks[1]["solution"])
    k = want.iteration_count
    np.testing.assert_allclose(ranks[0]["residuals"][:k], want.residuals[:k], rtol=1e-6)
    np.testing.assert_allclose(ranks[0]["iterates"][:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)


def test_bench_with_two_real_ranks_on_one_gpu(mock_rccl):
    """`bench.py --gpus 2` end to end -- its own spawner, the TCP rendezvous, two worker processes, one communicator -- with both
    ranks folded onto the one GPU: the line must say n_gpus 2 / ranks_seen 2, and carry the N > 1 sub-results."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FASTA_BENCH_RDV")}
    env["FASTA_RCCL_LIB"] = mock_rccl
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher", "socket", "--rows", "8192", "--cols", "8192",
                          "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["roofline"]["ranks_seen"] == 2 and out["roofline"]["comm_launches"] >= 5
    assert out["config"]["parallelism"] == "row-shard x2" and out["value"] > 0
    # each rank runs its one-pass grid on half of the CUs (bench.py sets FH_TUNE_FUSED_CUS for ranks that share a device): the main run
    # keeps the one-pass kernel, so the two-launch comparison is measured as well
    assert out["roofline"]["fused_one_pass_steps"] == 5 and out["roofline"]["comm_launches"] == 5
    assert set(out["extra"]) >= {"nnls", "config5_shard", "lasso_two_launch"} and out["extra"]["config5_shard"]["ranks_seen"] == 2


def test_bench_two_real_ranks_at_the_config5_shard_shape(mock_rccl):
    """The headline matrix (65536 x 65536) row-sharded over two real rank processes on the one GPU: 32768 x 65536 per rank is BASELINE
    config 5's per-GPU shard shape, the default one-pass kernel serves every launch, ONE exchange per iteration."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FASTA_BENCH_RDV")}
    env["FASTA_RCCL_LIB"] = mock_rccl
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher", "socket", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["roofline"]["ranks_seen"] == 2
    assert out["config"]["m"] == 65536 and "32768 rows each" in out["config"]["workload"]
    assert out["config"]["backtracks_in_timed_steps"] == 0
    assert out["roofline"]["fused_one_pass_steps"] == 4 and out["roofline"]["comm_launches"] == 4      # one exchange per iteration, no fall-back


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FASTA_BENCH_RDV",
                                                            "FASTA_BENCH_TOKEN", "FASTA_BENCH_FORCE_DIST")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [2, 3, 4])
def test_preflight_with_several_real_ranks(mock_rccl, world):
    """round 5: `bench.py --gpus N` meets RCCL on a few KiB first (fasta_python_amd/preflight.py): device count, library + version,
    communicator over N ranks, all-reduces of 128 and n + 3 doubles against their closed forms, co-residency probe, ranks_seen == N.
    Here with N real rank processes on the one GPU (the stand-in library in RCCL's place); the verdict is the first line on stderr."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--launcher", "socket", "--preflight-only"],
                         capture_output=True, text=True, timeout=300, env=_clean_env(FASTA_RCCL_LIB=mock_rccl), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    ok = [ln for ln in res.stderr.splitlines() if ln.startswith("fasta preflight")]
    assert len(ok) == 1 and ok[0].startswith(f"fasta preflight ok: {world} rank(s)") and f"ranks_seen {world}" in ok[0] and mock_rccl in ok[0], res.stderr[-3000:]
    assert res.stdout.strip() == ""


def test_preflight_with_one_rank_on_real_rccl():
    """`python -m fasta_python_amd.preflight 1`: a one-rank communicator on the system's RCCL -- the library, its version and the two
    checked all-reduces are the real thing."""
    res = subprocess.run([sys.executable, "-m", "fasta_python_amd.preflight", "1"], capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    ok = [ln for ln in res.stderr.splitlines() if ln.startswith("fasta preflight")]
    assert len(ok) == 1 and "ok: 1 rank(s)" in ok[0] and "librccl" in ok[0] and "version -1" not in ok[0], res.stderr[-3000:]


def test_preflight_of_the_in_process_form_and_its_failure_line():
    """The in-process form on the one GPU (`--devices 0,0,0`: three row blocks, sums by the in-library kernel, and the 256 x 4096 solve
    against the single-device one); and a device that does not exist: ONE line naming the step, exit status 3, nothing on stdout."""
    res = subprocess.run([sys.executable, "-m", "fasta_python_amd.preflight", "3", "--inproc", "--devices", "0,0,0"], capture_output=True, text=True,
                         timeout=300, env=_clean_env(), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "fasta preflight ok: 1 process x 3 row block(s)" in res.stderr and "equals the single-device solve" in res.stderr
    res = subprocess.run([sys.executable, "-m", "fasta_python_amd.preflight", "2", "--inproc", "--devices", "0,63"], capture_output=True, text=True,
                         timeout=300, env=_clean_env(), cwd=ROOT)
    lines = [ln for ln in res.stderr.splitlines() if ln.startswith("fasta preflight")]
    assert res.returncode == 3 and len(lines) == 1 and lines[0].startswith("fasta preflight FAILED at step 'device count'"), res.stderr[-3000:]
    assert res.stdout.strip() == ""


INPROC_RCCL = r"""
import json, sys, warnings
import numpy as np
sys.path.insert(0, {root!r})
import fasta_python_amd as fa
from fasta_python_amd import hip
shards, mode, m, n = {shards}, {mode!r}, {m}, {n}
rng = np.random.RandomState(7)
A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
xt = np.zeros(n); xt[rng.permutation(n)[:max(1, n // 40)]] = 1
b = A @ xt + 0.01 * rng.randn(m)
op = fa.ShardedDenseMatrixMap(A, devices=[0] * shards, _rccl_shell=True)      # the RCCL form although the device id repeats
assert op.ctx.shard_count() == shards and op.ctx.comm_count() == shards
ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
opts = dict(tolerance=1e-7, evaluate_objective=True, max_iters=40, verbose=False, adaptive=(mode != "fista"), accelerate=(mode == "fista"), fused={fused})
np.random.seed(9)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    c = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), backend="hip", **opts)
ms, launches = op.ctx.timing_get(hip.K_COMM)
op.close()
k = c.iteration_count
print("RESULT " + json.dumps(dict(iteration_count=int(k), backtracks=int(c.backtracks), residuals=c.residuals[:k].tolist(),
                                  objectives=c.objectives[:k + 1].tolist(), solution=c.solution.tolist())))
"""


@pytest.mark.parametrize("shards,mode,m,n,fused", [(2, "adaptive", 96, 160, True), (8, "fista", 96, 160, True), (4, "adaptive", 300, 4096, True),
                                                   (3, "adaptive", 200, 20000, False)])
def test_grouped_rccl_branch_of_the_in_process_form_with_several_shards(mock_rccl, shards, mode, m, n, fused):
    """The in-process multi-device form with DISTINCT device ids exchanges through ncclCommInitAll communicators: ncclGroupStart, one
    ncclAllReduce per shard, ncclGroupEnd, from one host thread.  FH_CREATE_RCCL_SHELL selects that branch although the id repeats
    (all shards on this GPU, one stream), and the stand-in -- which, unlike real RCCL, accepts several ranks on one device -- completes
    each group at ncclGroupEnd.  So the whole grouped-exchange code path runs with 2..8 shards; only the devices are not distinct."""
    code = INPROC_RCCL.format(root=ROOT, shards=shards, mode=mode, m=m, n=n, fused=fused)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, FASTA_RCCL_LIB=mock_rccl))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    got = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    rng = np.random.RandomState(7)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 40)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), tolerance=1e-7, evaluate_objective=True, max_iters=40, adaptive=(mode != "fista"), accelerate=(mode == "fista"))
    assert got["iteration_count"] == want.iteration_count and got["backtracks"] == want.backtracks
    k = want.iteration_count
    np.testing.assert_allclose(got["residuals"], want.residuals[:k], rtol=1e-6)
    np.testing.assert_allclose(got["objectives"], want.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(got["solution"], want.solution, rtol=1e-5, atol=1e-9)
