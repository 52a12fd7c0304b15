#!/bin/bash
# Run ON THE GPU BOX: round-4 A/B of scan-kernel builds / settings on one box, interleaved over rounds (tools/lib_time.py).
#   bash tools/ab_r4.sh "c3 c5" 3 "base e1 . noscatter toreg zrep64 allrep64"     ("." = gwinferno_amd/_lib)
CFGS=${1:-"c3 c5"}
ROUNDS=${2:-2}
LIBS=${3:-"base ."}
L=$PWD/gwinferno_amd
for r in $(seq $ROUNDS); do
  for name in $LIBS; do
    dir=_lib_$name; [ "$name" = "." ] && dir=_lib
    env GWI_QUIET=1 GWI_ENGINE_LIB=$L/$dir/libgwi_engine.so python3 tools/lib_time.py $CFGS 2>&1 | grep -E "loop us" | sed "s/^/$(printf '%-10s' $name) | /"
  done
done
