#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh [configs...]
# Produces, under gpurun_out/prof/<config>/{trace,fetch,write}, the rocprofv3 kernel-trace stats
# and the two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC slots,
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter passes use only --pmc.
# tools/summarize_profiles.py turns the CSVs into the files committed under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CONFIGS=${@:-c2 c3 c5}
cd /tmp && export TMPDIR=/tmp
for c in $CONFIGS; do
  out=$R/gpurun_out/prof/$c
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --config $c --steps 400 --warmup 50 --no-cpu-baseline --k-batch 0 --chains 0 > $out/bench_under_trace.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --k-batch 0 --chains 0 --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --k-batch 0 --chains 0 --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS --output-format csv -d $out/sq_a -- python3 $R/bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --k-batch 0 --chains 0 --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $out/sq_b -- python3 $R/bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --k-batch 0 --chains 0 --timing-every 0 > /dev/null 2>&1
  python3 $R/bench.py --config $c --steps 2000 --warmup 200 > $out/bench.json 2> /dev/null
done
