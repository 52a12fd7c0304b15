#!/bin/bash
# Run ON THE GPU BOX: batched launches (K = 16) of two library builds, interleaved: bash tools/ab_batch_r4.sh "mfmaread2 ." 3
LIBS=${1:-"base ."}
ROUNDS=${2:-2}
L=$PWD/gwinferno_amd
for r in $(seq $ROUNDS); do
  for name in $LIBS; do
    dir=_lib_$name; [ "$name" = "." ] && dir=_lib
    for c in c3 c5; do
      for flag in "--mfma" ""; do
        [ -z "$flag" ] && export GWI_BATCH_MFMA=0 || unset GWI_BATCH_MFMA
        env GWI_QUIET=1 GWI_ENGINE_LIB=$L/$dir/libgwi_engine.so python3 tools/batch_run.py --config $c --k 16 --n 60 $flag 2>/dev/null | sed "s/^/$(printf '%-10s' $name) | /"
      done
    done
    unset GWI_BATCH_MFMA
    env GWI_QUIET=1 GWI_ENGINE_LIB=$L/$dir/libgwi_engine.so python3 tools/defaults_batch_time.py 2>/dev/null | sed "s/^/$(printf '%-10s' $name) | defaults (12 tiles) /"
  done
done
