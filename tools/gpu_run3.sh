cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
./tools/microbench/mfma_f64_layout > $O/mfma_layout.txt 2>&1
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python bench.py --also none --no-cpu-baseline --chains 0 --steps 600 --warmup 50"
for c in c3 c5; do
  $B --config $c > $O/${c}_mfma.json 2>/dev/null
  GWI_BATCH_MFMA=0 $B --config $c > $O/${c}_taps.json 2>/dev/null
done
$B --config c2 > $O/c2.json 2>/dev/null
cat $O/mfma_layout.txt; tail -5 $O/pytest.log
