cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
GWI_FUZZ_POINTS=400 python -m pytest tests/test_gpu_fuzz.py -q > $O/fuzz400.log 2>&1; echo "rc=$?" >> $O/fuzz400.log
GWI_DETERMINISTIC=1 GWI_FUZZ_POINTS=48 python -m pytest tests/test_gpu_fuzz.py -q > $O/fuzz_det.log 2>&1; echo "rc=$?" >> $O/fuzz_det.log
GWI_BATCH_MFMA=2 python -m pytest tests/test_gpu_parity.py -q -k "batched" > $O/batched_mfma_all.log 2>&1; echo "rc=$?" >> $O/batched_mfma_all.log
python tools/setup_time.py c3 c5 c5x10 > $O/setup_time.txt 2>&1
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver.json 2> $O/bench_driver.err
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/fuzz400.log $O/fuzz_det.log $O/batched_mfma_all.log $O/pytest.log; cat $O/setup_time.txt; tail -4 $O/bench_driver.err
