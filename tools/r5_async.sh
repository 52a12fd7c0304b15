#!/bin/bash
# round 5: the two-halves batched entry -- its test, then the default bench line and config 3's (sets in flight, one thread)
mkdir -p gpurun_out/r5_async
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_halves or batched_evaluation" > gpurun_out/r5_async/test.log 2>&1
tail -5 gpurun_out/r5_async/test.log
timeout 600 python bench.py > gpurun_out/r5_async/bench_c2.json 2> gpurun_out/r5_async/bench_c2.err
timeout 600 python bench.py --config c3 > gpurun_out/r5_async/bench_c3.json 2> gpurun_out/r5_async/bench_c3.err
python - <<'PY'
import json
for c in ("c2", "c3"):
    try:
        d = json.loads(open(f"gpurun_out/r5_async/bench_{c}.json").read().strip().splitlines()[-1])
        print(c, d["value"], d["roofline"]["frac"], json.dumps(d.get("batched", {}).get("sets_in_flight")), d.get("batched", {}).get("evals_per_s"))
    except Exception as e:
        print(c, "failed", e)
PY
