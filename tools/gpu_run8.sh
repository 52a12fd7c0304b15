cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
B="python bench.py --also none --no-cpu-baseline --chains 0 --k-batch 0 --steps 1000 --warmup 50"
for spb in 512 768 1024 1536 2048; do
  GWI_SAMPLES_PER_BLOCK=$spb $B --config c3 > $O/c3_spb$spb.json 2>/dev/null
done
for spb in 512 1024 1536 2048; do
  GWI_SAMPLES_PER_BLOCK=$spb $B --config c2 > $O/c2_spb$spb.json 2>/dev/null
done
for spb in 768 1024 1280 1792 2560; do
  GWI_SAMPLES_PER_BLOCK=$spb $B --config c5 > $O/c5_spb$spb.json 2>/dev/null
done
GWI_SAMPLES_PER_LANE=1 GWI_SAMPLES_PER_BLOCK=768 $B --config c3 > $O/c3_u1_spb768.json 2>/dev/null
GWI_SAMPLES_PER_LANE=1 GWI_SAMPLES_PER_BLOCK=512 $B --config c3 > $O/c3_u1_spb512.json 2>/dev/null
