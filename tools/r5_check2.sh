#!/bin/bash
mkdir -p gpurun_out/r5_check2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5_check2/gpu_tests.log 2>&1
tail -4 gpurun_out/r5_check2/gpu_tests.log
s=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5_check2/bench.json 2> gpurun_out/r5_check2/bench.err
echo "bench wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_check2/bench.json").read().strip().splitlines()[-1])
print("c2", d["value"], d["roofline"]["frac"], d["batched"]["evals_per_s"])
ls = d["native_nuts_lockstep"]
print({k: ls[k] for k in ("evals_per_s", "mean_points_per_batch", "wall_s")}, {k: ls["chains_of_similar_length"][k] for k in ("evals_per_s", "mean_points_per_batch", "wall_s", "evals_per_chain_min_max")})
PY
