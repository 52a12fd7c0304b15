"""GPU (-m gpu): the multi-rank path end to end on ONE GPU -- several processes, each with its own engine over its
shard of events and injections, records exchanged through the node's shared-memory segment (gwi_shm_comm_init: the
exchange bench.py prefers on a real multi-GPU node too) with a gloo rendezvous; the in-engine RCCL exchange needs one
GPU per rank, its world-1 form is covered in test_gpu_parity.py.  bench.py itself checks the sharded result against an
unsharded engine and reports the difference."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])


def _check(d, n):
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["ranks"] == n and len(mg["per_rank"]) == n
    assert mg["exchange"].startswith("host shared-memory")
    chk = mg["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] < 1e-12 and chk["grad_max_err_over_scale"] < 1e-12, chk
    assert mg["independent_chains"]["evals_per_s"] > 0
    assert all(r["avg_kernel_us"]["scan"] > 0 for r in mg["per_rank"])
    assert d["median_ms_per_step"] > 0 and d["p5_ms"] <= d["median_ms_per_step"] <= d["p95_ms"]


def test_two_ranks_share_one_gpu_under_torch_distributed_run():
    env = dict(os.environ, GWI_BENCH_BACKEND="gloo", GWI_BENCH_DEVICE="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--config", "c1", "--spin", "0.05"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    _check(_last_json(out.stdout), 2)


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus N` as the driver types it: the script spawns its ranks itself (a child
    torch.distributed.run) and relays ONE JSON line; on this 1-GPU box the ranks share the device."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--spin", "0.05", "--also", "c3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly the JSON line on stdout
    d = json.loads(lines[0])
    _check(d, 2)
    assert d["steps"] == 20 and d["warmup"] == 5
    c3 = d["configs"]["c3"]  # the B-spline configuration sharded the same way
    assert c3["multi_gpu"]["sharded_vs_single_gpu"]["log_likelihood_rel_err"] < 1e-12
    assert c3["roofline"]["timed_launches"] >= 20


def test_three_ranks_torch_collective_fallback():
    """GWI_BENCH_EXCHANGE=torch: the torch.distributed all_gather variant (gloo here) still works end to end."""
    env = dict(os.environ, GWI_BENCH_BACKEND="gloo", GWI_BENCH_DEVICE="0", GWI_BENCH_EXCHANGE="torch")
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--config", "c1", "--spin", "0.05"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 3 and d["multi_gpu"]["exchange"].startswith("torch.distributed")
    chk = d["multi_gpu"]["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] < 1e-12 and chk["grad_max_err_over_scale"] < 1e-12, chk


def test_forced_sharded_world1_runs_the_rccl_variant():
    """GWI_FORCE_SHARDED=1: one GPU, the N > 1 code path with a nccl (RCCL) process group of one rank -- the in-engine
    ncclAllGather exchange measured first, in a child process started before the parent touches the GPU, then the
    shared-memory exchange for the headline; the two are reported as peers and `rccl_ranks` counts the ranks of the
    communicator that carried the HEADLINE's records (0 for shared memory)."""
    env = dict(os.environ, GWI_FORCE_SHARDED="1", MASTER_PORT=str(29800 + os.getpid() % 90))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--config", "c1", "--spin", "0.05", "--also", "none"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    mg = d["multi_gpu"]
    assert mg["exchange"].startswith("host shared-memory") and mg["rccl_ranks"] == 0
    ex = mg["exchanges"]
    assert ex["shm"]["headline"] and ex["shm"]["ms_per_step"] == d["ms_per_step"]
    leg = ex["rccl_allgather"]
    assert leg["child_exit_code"] == 0 and leg["evals_per_s"] > 0 and leg["rccl_ranks"] == 1 and leg["identical_on_all_ranks"]
    assert abs(leg["last_log_likelihood"]) > 0
    assert mg["sharded_vs_single_gpu"]["log_likelihood_rel_err"] < 1e-12


def test_rccl_leg_that_does_not_finish_is_killed_and_reported():
    """A hung RCCL exchange is a time-out of the CHILD: it is killed by its PID, reported with a non-zero exit code, and the
    main measurement (started only afterwards) is complete."""
    env = dict(os.environ, GWI_FORCE_SHARDED="1", GWI_BENCH_RCCL_TIMEOUT="0.05", MASTER_PORT=str(29700 + os.getpid() % 90))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--config", "c1", "--spin", "0.05", "--also", "none",
           "--k-batch", "0", "--chains", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    leg = d["multi_gpu"]["exchanges"]["rccl_allgather"]
    assert leg["child_exit_code"] == 124 and "killed" in leg["error"] and "evals_per_s" not in leg
    assert d["value"] > 0 and d["multi_gpu"]["exchanges"]["shm"]["headline"]
