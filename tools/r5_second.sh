#!/bin/bash
set -u
mkdir -p gpurun_out
export GWI_JIT_CACHE=/tmp/gwi_jit_cache
timeout 1200 python -m pytest tests/test_gpu_generic.py tests/test_gpu_parity.py tests/test_gpu_terms.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r5_tests2.txt 2>&1; tail -5 gpurun_out/r5_tests2.txt
{
for r in 1 2; do
for L in _lib_r5a _lib; do
  GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so timeout 300 python tools/lib_time.py c2 c1 2>&1 | grep -v amdgpu.ids
done
GWI_FORCE_JIT=1 timeout 300 python tools/lib_time.py c2 2>&1 | grep -v amdgpu.ids
done
for L in _lib_few _lib_few_u4 _lib_few_ref _lib_few_u4ref; do
  echo "== $L"
  GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so BT_KS=16 timeout 600 python tools/batch_time.py c2 "GWI_PBATCH=0" "GWI_PBATCH_PTS=16" "GWI_PBATCH_PTS=8" "GWI_PBATCH_PTS=4" "" 2>&1 | grep -v amdgpu.ids
  GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so BT_KS=16 timeout 600 python tools/batch_time.py c1 "GWI_PBATCH=0" "" 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r5_ab2.txt 2>&1
cat gpurun_out/r5_ab2.txt
