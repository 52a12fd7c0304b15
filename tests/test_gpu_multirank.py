"""GPU (-m gpu): the multi-GPU path with a REAL exchange between ranks -- one process per GPU, events and injections
sharded (9,9,9,9,9,8,8,8 / 8 x 25, equal injection slices: SURVEY 8e; reference contract pipeline/analysis.py:78-86,
:126-134), ONE ncclAllGather (RCCL over xGMI) of the partial records per evaluation on each engine's own stream
(``gwi_comm_init`` + ``gwi_eval_sharded``).

The build boxes have one GPU, so the R >= 2 tests are gated on the visible device count: they run the first time the suite
lands on a box with two (eight) GPUs and are SKIPPED, not passed, elsewhere.  The one-rank test runs the same child script
with a communicator of one rank on any box, so that the script, the rendezvous and the result files are exercised
everywhere.

Children are fresh processes (tests/multirank_child.py) started with subprocess -- never an exec of this process -- and a
child that does not finish in time is killed by its PID and fails the test.
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "multirank_child.py")


def _device_count():
    import torch  # counting devices does not initialise the GPU

    return torch.cuda.device_count()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(world, configs, out_prefix, limit_s):
    """Start `world` children, wait for all of them until the deadline; kill stragglers by PID.  Returns the per-rank
    result files' contents, or fails the test with the children's stderr tails."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GWI_QUIET"] = "1"
    port = _free_port()
    # output to files, not pipes: a rank blocked on a full pipe while this process waits for another rank would stall the group
    logs = [open(f"{out_prefix}.{r}.log", "w") for r in range(world)]
    kids = [subprocess.Popen([sys.executable, CHILD, str(r), str(world), str(port), out_prefix, ",".join(configs)], env=env, stdout=logs[r], stderr=subprocess.STDOUT)
            for r in range(world)]
    deadline = time.monotonic() + limit_s
    hung = []
    for r, k in enumerate(kids):
        try:
            k.wait(timeout=max(1.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            k.kill()  # this exact PID
            k.wait()
            hung.append(r)
    for f in logs:
        f.close()
    tails = "\n".join(f"[rank {r}] rc={k.returncode}\n{open(f'{out_prefix}.{r}.log').read()[-1500:]}" for r, k in enumerate(kids) if k.returncode != 0)
    assert not hung, f"ranks {hung} of {world} did not finish within {limit_s:.0f} s and were killed\n{tails}"
    assert all(k.returncode == 0 for k in kids), f"a rank failed\n{tails}"
    return [np.load(f"{out_prefix}.{r}.npz") for r in range(world)]


def check(results, configs, world):
    from golden_util import rel_err

    r0 = results[0]
    for cfg in configs:
        n_eval = r0[f"{cfg}/theta"].shape[0]
        # the shards tile the catalog: contiguous, balanced event blocks and injection slices (engine.shard_bounds)
        ev = [tuple(int(v) for v in r[f"{cfg}/events"]) for r in results]
        inj = [tuple(int(v) for v in r[f"{cfg}/injections"]) for r in results]
        assert ev[0][0] == 0 and inj[0][0] == 0
        for a, b in zip(ev[:-1], ev[1:]):
            assert a[1] == b[0]
        for a, b in zip(inj[:-1], inj[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in ev]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        if cfg == "c3" and world == 8:
            assert sizes == [9, 9, 9, 9, 9, 8, 8, 8]
        if cfg == "c5" and world == 8:
            assert sizes == [25] * 8
        for i in range(n_eval):
            ll = float(r0[f"{cfg}/{i}/sharded_ll"])
            g = r0[f"{cfg}/{i}/sharded_grad"]
            assert np.isfinite(ll) and np.all(np.isfinite(g)) and np.any(g != 0.0)
            for r in results[1:]:  # every rank assembles identical bits from the gathered records
                assert float(r[f"{cfg}/{i}/sharded_ll"]) == ll
                assert np.array_equal(r[f"{cfg}/{i}/sharded_grad"], g)
                assert float(r[f"{cfg}/{i}/sharded_log_mu"]) == float(r0[f"{cfg}/{i}/sharded_log_mu"])
            # against the unsharded engine over the whole catalog (each rank ran one on its own GPU)
            for r in results:
                f_ll, f_g = float(r[f"{cfg}/{i}/full_ll"]), r[f"{cfg}/{i}/full_grad"]
                assert rel_err(ll, f_ll) < 1e-10
                scale = max(1.0, float(np.max(np.abs(f_g))))
                assert float(np.max(np.abs(g - f_g))) / scale < 1e-8
                assert abs(float(r[f"{cfg}/{i}/sharded_log_mu"]) - float(r[f"{cfg}/{i}/full_log_mu"])) < 1e-10 * abs(float(r[f"{cfg}/{i}/full_log_mu"]))
                # this rank's per-event sites are the unsharded engine's for the same events
                assert np.allclose(r[f"{cfg}/{i}/sharded_log_bfs"], r[f"{cfg}/{i}/full_log_bfs"], rtol=1e-12, atol=1e-11)
            # ... and against the C oracle (rank 0 ran it): north_star's <= 1e-9 in fp64, the whole gradient at 1e-8 of its scale
            o_ll, o_g = float(r0[f"{cfg}/{i}/oracle_ll"]), r0[f"{cfg}/{i}/oracle_grad"]
            assert rel_err(ll, o_ll) < 1e-9
            scale = max(1.0, float(np.max(np.abs(o_g))))
            assert float(np.max(np.abs(g - o_g))) / scale < 1e-8
        # the C loop of sequential sharded evaluations gives the values of the one-by-one calls
        for i in range(2):
            for r in results:
                assert rel_err(float(r[f"{cfg}/seq_ll"][i]), float(r0[f"{cfg}/{i}/sharded_ll"])) < 1e-12


def test_child_script_with_a_communicator_of_one_rank(tmp_path):
    """Any box: the child with R = 1 (ncclCommInitRank of one rank, the all-gather of one record)."""
    configs = ["small", "c3"]
    results = run_ranks(1, configs, str(tmp_path / "w1"), limit_s=420)
    check(results, configs, 1)


@pytest.mark.skipif(_device_count() < 2, reason="needs >= 2 GPUs: RCCL cannot place two ranks of one communicator on one device")
def test_rccl_two_ranks_on_two_gpus(tmp_path):
    """Two ranks on two GPUs: ragged small catalog, config 2, and the B-spline configs 3 and 5 (BASELINE config 4 is config 3
    sharded), each against the unsharded engine and the C oracle; bit-identical results across ranks."""
    configs = ["small", "c2", "c3", "c5"]
    results = run_ranks(2, configs, str(tmp_path / "w2"), limit_s=900)
    check(results, configs, 2)


@pytest.mark.skipif(_device_count() < 8, reason="needs 8 GPUs")
def test_rccl_eight_ranks_on_eight_gpus(tmp_path):
    """BASELINE config 4 / 5 as stated: 8 ranks, 9,9,9,9,9,8,8,8 and 8 x 25 events, one all-gather per evaluation."""
    configs = ["c3", "c5"]
    results = run_ranks(8, configs, str(tmp_path / "w8"), limit_s=1200)
    check(results, configs, 8)


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def check_compact_line(stdout, world):
    """The LAST line of bench.py's stdout: strict JSON, at most 4 KB, the contract's keys, and for the N > 1 path a short `multi_gpu` block."""
    last = stdout.strip().splitlines()[-1]
    assert last.startswith("{") and len(last.encode()) <= 4096, (len(last), last[:200])

    def refuse(name):
        raise ValueError(name)

    line = json.loads(last, parse_constant=refuse)
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line["n_gpus"] == world and line["value"] > 0 and line["ms_per_step"] > 0 and line["dtype"] == "f64"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and roof["avg_kernel_us"]["scan"] > 0
    mg = line["multi_gpu"]
    assert mg["ranks"] == world and mg["rccl_ranks"] in (0, world) and len(mg["per_rank_events"]) == world and sum(mg["per_rank_events"]) == 69
    chk = mg["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] <= 1e-9 and chk["grad_max_err_over_scale"] <= 1e-8
    return line


def test_bench_rccl_leg_compares_with_the_unsharded_engine(tmp_path):
    """bench.py's in-engine RCCL leg (a child per rank; here a world of one rank) reports `sharded_vs_single_gpu` like the
    shared-memory path: the gathered-record result against an unsharded engine, value and whole gradient."""
    import json

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), GWI_QUIET="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-leg", "--config", "c2", "--steps", "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True)
    try:
        out, err = child.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        child.kill()  # this exact PID
        out, err = child.communicate()
        pytest.fail("the RCCL leg did not finish within 300 s and was killed\n" + (err or "")[-1500:])
    assert child.returncode == 0, (err or "")[-1500:]
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    chk = line["sharded_vs_single_gpu"]
    assert chk["within_tolerance"] and chk["log_likelihood_rel_err"] <= 1e-9 and chk["grad_max_err_over_scale"] <= 1e-8
    assert line["rccl_ranks"] == 1 and line["identical_on_all_ranks"]


def test_bench_multi_gpu_headline_is_the_rccl_exchange(tmp_path):
    """bench.py's N > 1 path with a world of ONE rank (GWI_FORCE_SHARDED=1: every box has that): the probe of the in-engine
    ncclAllGather runs first in a child process; when it comes back clean the HEADLINE's records travel through that exchange
    (`multi_gpu.rccl_ranks == world`, what BASELINE.json's north_star names), the shared-memory exchange is reported beside it,
    and the sharded result is compared with the unsharded engine."""
    env = dict(os.environ, GWI_FORCE_SHARDED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    detail_file = str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "20", "--also", "none", "--no-cpu-baseline", "--k-batch", "0", "--chains", "0",
           "--spin", "0.1", "--detail", detail_file]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        child.kill()
        out, err = child.communicate()
        pytest.fail("bench.py did not finish within 600 s and was killed\n" + (err or "")[-1500:])
    assert child.returncode == 0, (err or "")[-2000:]
    compact = check_compact_line(out, world=1)
    assert compact["multi_gpu"]["rccl_ranks"] == 1 and compact["multi_gpu"]["rccl_probe"]["child_exit_code"] == 0 and compact["detail"] == detail_file
    line = json.load(open(detail_file))  # everything else: the side file
    assert line["value"] == pytest.approx(compact["value"], rel=1e-5)
    mg = line["multi_gpu"]
    assert mg["ranks"] == 1 and mg["rccl_ranks"] == 1, mg
    ex = mg["exchanges"]
    assert ex["rccl_allgather"]["headline"] is True and ex["rccl_allgather"]["evals_per_s"] == line["value"]
    assert ex["rccl_allgather_probe"]["child_exit_code"] == 0 and ex["rccl_allgather_probe"]["probe_clean_on_every_rank"] is True
    # (the two exchanges sum the records on different paths -- device-side final reduce vs the host's -- : last bits may differ)
    assert ex["shm"]["evals_per_s"] > 0 and abs(ex["shm"]["last_log_likelihood"] - line["last_log_likelihood"]) <= 1e-12 * abs(line["last_log_likelihood"])
    chk = mg["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] <= 1e-9 and chk["grad_max_err_over_scale"] <= 1e-8
    # ... and with the probe switched off the shared-memory exchange carries the headline, and says so
    env["GWI_BENCH_RCCL_VARIANT"] = "0"
    env["MASTER_PORT"] = str(_free_port())
    out2 = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert check_compact_line(out2.stdout, world=1)["multi_gpu"]["rccl_ranks"] == 0
    mg2 = json.load(open(detail_file))["multi_gpu"]
    assert mg2["rccl_ranks"] == 0 and mg2["exchanges"]["shm"]["headline"] is True


def test_bench_in_the_drivers_launch_form_with_two_ranks_on_this_box(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` -- the form the driver launches for N > 1 --
    on whatever this box has.  With two GPUs the in-engine RCCL all-gather carries the headline; on a one-GPU box both ranks
    share the device, RCCL is skipped WITH ITS REASON (it cannot place two ranks of a communicator on one device) and the
    shared-memory exchange carries it: either way two processes evaluate their shards of events and injections on the GPU,
    exchange the partial records once per evaluation, and the result equals the unsharded engine's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GWI_QUIET="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "20"]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path), start_new_session=True)
    try:
        out, err = child.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, 9)  # the launcher and its two ranks: this exact process group
        out, err = child.communicate()
        pytest.fail("the two-rank bench did not finish within 900 s and was killed\n" + (err or "")[-1500:])
    assert child.returncode == 0, (err or "")[-2500:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-1500:]  # rank 0 prints ONE line
    compact = check_compact_line(out, world=2)
    assert compact["steps"] == 200 and compact["warmup"] == 20 and compact["scaling"] == "strong"
    # rccl_ranks: the ranks of the communicator that carried the HEADLINE's records -- the world with a GPU per rank, 0 when the ranks share one
    assert compact["multi_gpu"]["rccl_ranks"] == (2 if _device_count() >= 2 else 0)
    assert compact["multi_gpu"]["per_rank_events"] == [35, 34] and set(compact["configs"]) == {"c3", "c5"}
    for blk in compact["configs"].values():  # the B-spline configs sharded the same way, each against the unsharded engine
        chk = blk["multi_gpu"]["sharded_vs_single_gpu"]
        assert blk["value"] > 0 and chk["log_likelihood_rel_err"] <= 1e-9 and chk["grad_max_err_over_scale"] <= 1e-8
    assert compact["detail"] == "bench_detail.json"  # written to the working directory (here: tmp_path)
    line = json.load(open(tmp_path / "bench_detail.json"))
    assert line["n_gpus"] == 2 and line["value"] == pytest.approx(compact["value"], rel=1e-5)
    mg = line["multi_gpu"]
    assert mg["ranks"] == 2 and len(mg["per_rank"]) == 2
    assert sorted(r["n_ev"] for r in mg["per_rank"]) == [34, 35] and all(r["n_inj"] == 25000 for r in mg["per_rank"])  # SURVEY 8e's partition of 69 events / 50 k injections
    chk = mg["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] <= 1e-9 and chk["grad_max_err_over_scale"] <= 1e-8
    ex = mg["exchanges"]
    if _device_count() >= 2:
        assert mg["rccl_ranks"] == 2 and ex["rccl_allgather"]["headline"] is True
    else:
        assert mg["rccl_ranks"] == 0 and ex["shm"]["headline"] is True and ex["shm"]["evals_per_s"] == line["value"]
        assert "one device" in ex["rccl_allgather"]["skipped"] and mg["devices_shared_between_ranks"] == 1
