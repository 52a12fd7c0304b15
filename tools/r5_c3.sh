#!/bin/bash
set -u
mkdir -p gpurun_out
{
for r in 1 2; do
for L in _lib_few _lib_abl_noscatter _lib_abl_toreg _lib_abl_rep64; do
  GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so timeout 300 python tools/lib_time.py c3 c5 2>&1 | grep -v amdgpu.ids
done
done
GWI_ENGINE_LIB=$PWD/gwinferno_amd/_lib_abl_stamps/libgwi_engine.so GWI_AQL=0 timeout 300 python tools/stamp_phases.py c3 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5_c3_abl.txt 2>&1
cat gpurun_out/r5_c3_abl.txt
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
