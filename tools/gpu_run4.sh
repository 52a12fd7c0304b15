cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for c in c2 c3 c5; do
  python tools/batch_run.py --config $c --k 16 --n 200 > $O/batch_$c.json 2>/dev/null
  GWI_AQL_BATCH=0 python tools/batch_run.py --config $c --k 16 --n 200 > $O/batch_${c}_hip.json 2>/dev/null
done
GWI_MAX_BATCH=64 python tools/batch_run.py --config c2 --k 64 --n 100 > $O/batch_c2_k64.json 2>/dev/null
python bench.py --also none --no-cpu-baseline --steps 1000 --warmup 50 --config c2 > $O/c2.json 2>/dev/null
tail -4 $O/pytest.log; cat $O/batch_*.json
