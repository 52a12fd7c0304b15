#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds on one box, one process per build, interleaved over rounds.
#   bash tools/ab_libs.sh "c3 c5" "_lib_old _lib" [rounds]      (directories under gwinferno_amd/)
# Variants of the launch geometry per config come from AB_VARIANTS_<cfg> (quoted strings as for tools/geometry_sweep.py).
CFGS=${1:-"c3 c5"}
LIBS=${2:-"_lib_old _lib"}
ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for L in $LIBS; do
    for c in $CFGS; do
      v="AB_VARIANTS_$c"
      eval "set -- ${!v:-\"\"}"
      GWI_ENGINE_LIB=$PWD/gwinferno_amd/$L/libgwi_engine.so python3 tools/geometry_sweep.py $c "$@" 2>&1 | grep "scan us" | sed "s/^/$L /"
    done
  done
done
