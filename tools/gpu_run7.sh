cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --also none --no-cpu-baseline --steps 2000 --warmup 100"
for i in 1 2; do
$B --config c2 > $O/c2_$i.json 2>/dev/null
$B --config c3 --chains 0 > $O/c3_$i.json 2>/dev/null
done
GWI_AQL_READBACK=0 $B --config c2 > $O/c2_rb0.json 2>/dev/null
tail -3 $O/pytest.log
