#!/bin/bash
# round 5, first GPU call: the new paths' parity, then timings (run from the repo root on the GPU box)
set -u
mkdir -p gpurun_out
export GWI_JIT_CACHE=/tmp/gwi_jit_cache
timeout 900 python -m pytest tests/test_gpu_generic.py -x -q > gpurun_out/r5_generic.txt 2>&1; tail -5 gpurun_out/r5_generic.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or one_load or marginalised or smoke" > gpurun_out/r5_batch.txt 2>&1; tail -5 gpurun_out/r5_batch.txt
{
BT_KS=16 timeout 600 python tools/batch_time.py c2 "GWI_PBATCH=0" "GWI_PBATCH_PTS=16" "GWI_PBATCH_PTS=8" "GWI_PBATCH_PTS=4" "GWI_PBATCH_PTS=2" ""
BT_KS=4,8,64 GWI_MAX_BATCH=64 timeout 600 python tools/batch_time.py c2 "GWI_PBATCH=0" ""
BT_KS=16 timeout 600 python tools/batch_time.py c1 "GWI_PBATCH=0" "GWI_PBATCH_PTS=16" ""
} > gpurun_out/r5_pbatch_time.txt 2>&1
tail -30 gpurun_out/r5_pbatch_time.txt
{
for c in c2 c3 c5; do
  timeout 300 python tools/lib_time.py $c
  GWI_FORCE_JIT=1 timeout 300 python tools/lib_time.py $c
  GWI_FORCE_GENERIC=1 GWI_QUIET=1 timeout 600 python tools/lib_time.py $c
done
} > gpurun_out/r5_generic_time.txt 2>&1
cat gpurun_out/r5_generic_time.txt
