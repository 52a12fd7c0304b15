cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r2a/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2a/smoke.log
(time python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r2a/bench_driver.json 2> gpurun_out/r2a/bench_driver.err
python bench.py --steps 2000 --warmup 100 --also none --no-cpu-baseline > gpurun_out/r2a/bench_c2_rb1.json 2>/dev/null
GWI_AQL_READBACK=0 python bench.py --steps 2000 --warmup 100 --also none --no-cpu-baseline > gpurun_out/r2a/bench_c2_rb0.json 2>/dev/null
python bench.py --steps 2000 --warmup 100 --also none --no-cpu-baseline > gpurun_out/r2a/bench_c2_rb1b.json 2>/dev/null
(time python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/r2a/bench_n2.json 2> gpurun_out/r2a/bench_n2.err
tail -3 gpurun_out/r2a/pytest.log; tail -4 gpurun_out/r2a/smoke.log
