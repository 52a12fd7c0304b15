cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
B="python bench.py --also none --no-cpu-baseline --steps 1000 --warmup 50 --config c2"
for i in 1 2; do
$B > $O/c2_default_$i.json 2>/dev/null
GWI_COMBINE_THREADS=256 $B > $O/c2_ct256_$i.json 2>/dev/null
GWI_AQL_READBACK=0 $B > $O/c2_rb0_$i.json 2>/dev/null
GWI_AQL=0 $B > $O/c2_hip_$i.json 2>/dev/null
done
