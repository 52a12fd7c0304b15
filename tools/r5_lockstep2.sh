#!/bin/bash
mkdir -p gpurun_out/r5_lockstep
export GWI_LOCKSTEP_STATS=1
for a in "c2 1 16" "c2 2 16" "c2 3 16" "c2 2 8" "c3 2 16" "c5 1 16 30 10"; do
  timeout 600 python tools/lockstep_time.py $a 2>&1 | grep -v amdgpu.ids
done
