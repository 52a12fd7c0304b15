"""GPU (-m gpu): the multi-rank path end to end on ONE GPU -- two processes (torch.distributed.run), each
with its own engine over its shard of events and injections on device 0, records exchanged through a gloo
group (the in-engine RCCL exchange needs one GPU per rank; its world-1 form is covered in test_gpu_parity.py).
bench.py itself checks the sharded result against an unsharded engine and reports the difference."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu():
    env = dict(os.environ, GWI_BENCH_BACKEND="gloo", GWI_BENCH_DEVICE="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--config", "c1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    chk = d["sharded_vs_single_gpu"]
    assert chk["log_likelihood_rel_err"] < 1e-12 and chk["grad_max_err_over_scale"] < 1e-12, chk
    assert d["independent_chains"]["evals_per_s"] > 0
