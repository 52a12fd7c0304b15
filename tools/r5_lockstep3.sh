#!/bin/bash
export GWI_LOCKSTEP_STATS=1
for st in 1 2 3; do
  export LOCKSTEP_SAME_START=$st
  for a in "c2 2 16 300 200" "c2 3 16 300 200"; do
    timeout 600 python tools/lockstep_time.py $a 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
export LOCKSTEP_SAME_START=1
timeout 600 python tools/lockstep_time.py c3 2 16 60 30 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/lockstep_time.py c5 1 16 30 10 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/lockstep_time.py c5 2 16 30 10 2>&1 | grep -v amdgpu.ids | tail -2
