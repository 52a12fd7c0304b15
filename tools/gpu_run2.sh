cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
B="python bench.py --also none --no-cpu-baseline --chains 0 --steps 1000 --warmup 50"
$B --config c2 > $O/c2.json 2>/dev/null
GWI_AQL_READBACK=0 $B --config c2 > $O/c2_rb0.json 2>/dev/null
for r in 8 16 32 64; do
  GWI_GACC_REP=$r $B --config c5 > $O/c5_rep$r.json 2>/dev/null
  GWI_GACC_REP=$r $B --config c3 > $O/c3_rep$r.json 2>/dev/null
done
$B --config c5 > $O/c5_default.json 2>/dev/null
$B --config c3 > $O/c3_default.json 2>/dev/null
GWI_SAMPLES_PER_LANE=2 $B --config c5 > $O/c5_u2.json 2>/dev/null
GWI_SAMPLES_PER_LANE=1 $B --config c3 > $O/c3_u1.json 2>/dev/null
GWI_DETERMINISTIC=1 $B --config c5 > $O/c5_det.json 2>/dev/null
tail -5 $O/pytest.log; tail -3 $O/smoke.log
