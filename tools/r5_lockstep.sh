#!/bin/bash
# round 5: lock-step chains -- the two tests, then the default bench line
mkdir -p gpurun_out/r5_lockstep
timeout 900 python -m pytest tests/test_gpu_dropin_api.py tests/test_sampling_cpu.py -x -q -k "lockstep or native_nuts" > gpurun_out/r5_lockstep/test.log 2>&1
tail -15 gpurun_out/r5_lockstep/test.log
timeout 900 python bench.py > gpurun_out/r5_lockstep/bench.json 2> gpurun_out/r5_lockstep/bench.err
tail -3 gpurun_out/r5_lockstep/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_lockstep/bench.json").read().strip().splitlines()[-1])
print("c2", d["value"], json.dumps(d.get("native_nuts_lockstep")), d["native_nuts"]["evals_per_s"])
for c in ("c3", "c5"):
    x = d["configs"][c]
    print(c, x["value"], json.dumps(x.get("native_nuts_lockstep")), x["native_nuts"]["evals_per_s"])
PY
