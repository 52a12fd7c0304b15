#!/bin/bash
# round 5: what the per-point reductions of scan_pbatch_kernel cost (timing-only builds: results wrong by construction)
L=$PWD/gwinferno_amd
for r in 1 2; do
  for name in pb_base pb_nomax pb_nosum pb_neither; do
    GWI_QUIET=1 BT_KS=16 GWI_ENGINE_LIB=$L/_lib_$name/libgwi_engine.so python3 tools/batch_time.py c2 "" 2>/dev/null | tail -1 | sed "s/^/$(printf '%-11s' $name) | /"
  done
done
