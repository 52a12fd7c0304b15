#!/bin/bash
set -u
mkdir -p gpurun_out
export GWI_JIT_CACHE=/tmp/gwi_jit_cache
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "batch or one_load" > gpurun_out/r5_tests5.txt 2>&1; tail -5 gpurun_out/r5_tests5.txt
{
BT_KS=4,16 timeout 600 python tools/batch_time.py c2 "GWI_FUSED_BATCH_TAIL=0" "" 2>&1 | grep -v amdgpu.ids
BT_KS=16 timeout 600 python tools/batch_time.py c3 "GWI_FUSED_BATCH_TAIL=0" "" 2>&1 | grep -v amdgpu.ids
BT_KS=16 timeout 600 python tools/batch_time.py c1 "GWI_FUSED_BATCH_TAIL=0" "" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5_batch5.txt 2>&1
cat gpurun_out/r5_batch5.txt
