# usage: bash tools/gpu_prof_quick.sh <config>  -- scan-kernel counters of one config (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
c=${1:-c3}
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --k-batch 0 --chains 0 --also none --spin 0.1 --steps 60 --warmup 10 --timing-every 0"
out=/tmp/q_$c; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/a -- python3 $R/bench.py --config $c $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python3 $R/bench.py --config $c $B > /dev/null 2>&1
python3 - $out <<'PY'
import csv,glob,sys,collections
for part in ("a","b"):
    f=glob.glob(sys.argv[1]+"/"+part+"/*/*_counter_collection.csv")
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "scan_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
