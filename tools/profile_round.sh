#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh [configs...]
# Produces, under gpurun_out/prof/<config>/{trace,fetch,write,sq_a,sq_b,lds}, the rocprofv3 kernel-trace stats and the
# PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC slots, /opt/skills/guides/MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), and under gpurun_out/prof/batch_<config>_{taps,mfma}/ the same for the batched kernels
# (K = 16 points per launch; 4-tap gradient vs the v_mfma_f64_16x16x4 gradient GEMM).  Counter passes use only --pmc;
# the profiled program comes directly after `--`.  tools/summarize_profiles.py turns the CSVs into profiles/<round>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CONFIGS=${@:-c2 c3 c5 b50k def50k}
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --k-batch 0 --chains 0 --also none --spin 0.2 --detail /dev/null"
for c in $CONFIGS; do
  out=$R/gpurun_out/prof/$c
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --config $c --steps 200 --warmup 20 $B > $out/bench_under_trace.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --config $c --steps 60 --warmup 10 $B --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --config $c --steps 60 --warmup 10 $B --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS --output-format csv -d $out/sq_a -- python3 $R/bench.py --config $c --steps 60 --warmup 10 $B --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $out/sq_b -- python3 $R/bench.py --config $c --steps 60 --warmup 10 $B --timing-every 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/lds -- python3 $R/bench.py --config $c --steps 60 --warmup 10 $B --timing-every 0 > /dev/null 2>&1
  python3 $R/bench.py --config $c --steps 2000 --warmup 200 --also none --detail $out/bench_detail.json > $out/bench.json 2> /dev/null
done
for c in c3 c5 b50k def50k; do
  for path in taps mfma rows; do
    out=$R/gpurun_out/prof/batch_${c}_$path
    mkdir -p $out
    flag=""; [ $path = mfma ] && flag="--mfma"; [ $path = rows ] && flag="--rows"; [ $path = taps ] && export GWI_BATCH_MFMA=0 || unset GWI_BATCH_MFMA
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/batch_run.py --config $c --k 16 --n 40 $flag > $out/run_under_trace.json 2> /dev/null
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $out/mfma -- python3 $R/tools/batch_run.py --config $c --k 16 --n 10 $flag > /dev/null 2>&1
    python3 $R/tools/batch_run.py --config $c --k 16 --n 100 $flag > $out/run.json 2> /dev/null
  done
done
# ---- (a) the same single evaluations through a chain compiled at gwi_create (GWI_FORCE_JIT=1: hipRTC) and through the
# generic kernel (GWI_FORCE_GENERIC=1: run-time term loop) -- what a product of densities without an ahead-of-time chain runs;
# (b) batched launches of the parametric config 2: one load per sample (scan_pbatch_kernel) against one grid row per point
# (GWI_PBATCH=0), with the FETCH_SIZE / WRITE_SIZE passes that show what each streams from beyond L2
for mode in jit generic; do
  [ $mode = jit ] && export GWI_FORCE_JIT=1 || export GWI_FORCE_GENERIC=1
  export GWI_QUIET=1
  for c in $CONFIGS; do
    out=$R/gpurun_out/prof/chain_${c}_$mode
    mkdir -p $out
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --config $c --steps 200 --warmup 20 $B > $out/bench_under_trace.json 2> /dev/null
    python3 $R/bench.py --config $c --steps 1000 --warmup 100 --also none $B > $out/bench.json 2> /dev/null
  done
  unset GWI_FORCE_JIT GWI_FORCE_GENERIC GWI_QUIET
done
for path in pbatch rows4 rowsperpoint; do  # one load per sample, balanced units (GWI_PBATCH=1) / round 5's four grid rows of four points / one grid row per point (the default)
  out=$R/gpurun_out/prof/batch_c2_$path
  mkdir -p $out
  unset GWI_PBATCH GWI_PBATCH_BALANCED
  [ $path = pbatch ] && export GWI_PBATCH=1
  [ $path = rows4 ] && export GWI_PBATCH=1 GWI_PBATCH_BALANCED=0
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/batch_run.py --config c2 --k 16 --n 40 > $out/run_under_trace.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/tools/batch_run.py --config c2 --k 16 --n 10 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/tools/batch_run.py --config c2 --k 16 --n 10 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $out/mfma -- python3 $R/tools/batch_run.py --config c2 --k 16 --n 10 > /dev/null 2>&1
  python3 $R/tools/batch_run.py --config c2 --k 16 --n 100 > $out/run.json 2> /dev/null
done
unset GWI_PBATCH GWI_PBATCH_BALANCED
# the driver's own command, for the record
cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/prof/bench_driver_form_detail.json > gpurun_out/prof/bench_driver_form.json 2> /dev/null
# summarise on the box (the raw traces exceed what gpurun carries back) and keep only the summary
unset GWI_BATCH_MFMA
python3 tools/summarize_profiles.py ${ROUND:-round6} $R/gpurun_out/profile_summary > $R/gpurun_out/profile_summary.log 2>&1
cp gpurun_out/prof/bench_driver_form.json gpurun_out/prof/bench_driver_form_detail.json gpurun_out/profile_summary/ 2>/dev/null
rm -rf $R/gpurun_out/prof
ls -la $R/gpurun_out/profile_summary
