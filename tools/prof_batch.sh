#!/bin/bash
# Run ON THE GPU BOX: kernel trace + SQ counters of a batched run.  bash tools/prof_batch.sh c5 [--mfma]  -> gpurun_out/pb_<cfg><flag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
c=$1; flag=$2
out=$R/gpurun_out/pb_$c$flag.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb && mkdir -p /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb/trace -- python3 $R/tools/batch_run.py --config $c --k 16 --n 30 $flag > /dev/null 2>&1
python3 - <<PY > $out
import csv,glob
f=glob.glob('/tmp/pb/trace/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'gwi::' in r['Name']: print(r['Name'][:90], r['Calls'], 'avg_us', float(r['AverageNs'])/1e3)
PY
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pb/pmc
  rocprofv3 --pmc $set --output-format csv -d /tmp/pb/pmc -- python3 $R/tools/batch_run.py --config $c --k 16 --n 6 $flag > /dev/null 2>&1
  python3 - <<PY >> $out
import csv,glob,collections
fs=glob.glob('/tmp/pb/pmc/*/*_counter_collection.csv')
acc=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'scan' in r['Kernel_Name']: acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k[0], k[1], sum(v)/len(v))
PY
done
cat $out
