#!/bin/bash
set -u
mkdir -p gpurun_out
(time python -c "import __graft_entry__ as g; g.build(); g.smoke()") > gpurun_out/r5_smoke.txt 2>&1; tail -6 gpurun_out/r5_smoke.txt
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/r5_tests_final.txt 2>&1; tail -4 gpurun_out/r5_tests_final.txt
(time timeout 1500 python bench.py) > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; tail -3 gpurun_out/r5_bench_default.err; head -c 600 gpurun_out/r5_bench_default.json
echo
(time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r5_bench_driver_form.json 2> gpurun_out/r5_bench_driver_form.err; tail -3 gpurun_out/r5_bench_driver_form.err; head -c 400 gpurun_out/r5_bench_driver_form.json
