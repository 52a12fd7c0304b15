#!/bin/bash
set -u
mkdir -p gpurun_out
export GWI_JIT_CACHE=/tmp/gwi_jit_cache
(time timeout 2400 python -m pytest tests -m gpu -x -q) > gpurun_out/r5_tests3.txt 2>&1; tail -8 gpurun_out/r5_tests3.txt
{
BT_KS=16 timeout 600 python tools/batch_time.py c3 "GWI_BATCH_MFMA=0" "GWI_BATCH_MFMA=1" "" 2>&1 | grep -v amdgpu.ids
BT_KS=16 timeout 600 python tools/batch_time.py c5 "GWI_BATCH_MFMA=0" "GWI_BATCH_MFMA=1" "" 2>&1 | grep -v amdgpu.ids
BT_KS=16 timeout 600 python tools/batch_time.py c2 "GWI_PBATCH=0" "" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5_batch3.txt 2>&1
cat gpurun_out/r5_batch3.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_form.json 2> gpurun_out/r5_bench_driver_form.err; tail -c 1500 gpurun_out/r5_bench_driver_form.json
