#!/bin/bash
set -u
mkdir -p gpurun_out
export GWI_JIT_CACHE=/tmp/gwi_jit_cache
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/r5_tests4.txt 2>&1; tail -8 gpurun_out/r5_tests4.txt
