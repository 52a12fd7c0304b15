#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds (directories under gwinferno_amd/) on one box, one process per build and measurement,
# interleaved over rounds:   bash tools/ab_libs.sh "_lib_base _lib_taylor _lib" [rounds] [configs]
# Per build: the batched scan of config 2 at K = 16 (tools/batch_time.py) and the single scans of the given configs (bench.py's
# live kernel durations, dispatch timestamps).
LIBS=${1:-"_lib_base _lib"}
ROUNDS=${2:-3}
CFGS=${3:-"c2 c3 c5"}
for r in $(seq $ROUNDS); do
  for L in $LIBS; do
    export GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so
    BT_KS=16 python3 tools/batch_time.py c2 2>/dev/null | grep "K=16" | tail -1 | sed "s/^/round $r $L /"
    for c in $CFGS; do
      python3 bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline --k-batch 0 --also none --detail '' 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['avg_kernel_us']
print('round $r $L $c single scan us %.3f  combine %.2f  step us %.2f  frac %.4f' % (k['scan'], k['combine'], 1e3*d['ms_per_step'], d['roofline']['frac']))"
    done
  done
done
