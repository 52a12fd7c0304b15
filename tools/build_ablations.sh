#!/bin/bash
# Timing-only ablation builds of the engine (results are wrong by construction): tools/build_ablations.sh NAME -DFLAG...
# -> gwinferno_amd/_lib_NAME/{libgwi_engine.so,gwi_kernels.hsaco}; run with GWI_ENGINE_LIB=gwinferno_amd/_lib_NAME/libgwi_engine.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$R/gwinferno_amd/_lib_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 -I$R/include -I$R/gwinferno_amd/csrc $@"
/opt/rocm/bin/hipcc $F -fPIC -shared -o $out/libgwi_engine.so $R/gwinferno_amd/csrc/gwi_engine.hip $R/gwinferno_amd/csrc/gwi_sampler.cpp -ldl -lpthread 2>/dev/null &
/opt/rocm/bin/hipcc $F --cuda-device-only --no-gpu-bundle-output -o $out/gwi_kernels.hsaco $R/gwinferno_amd/csrc/gwi_engine.hip 2>/dev/null &
wait
ls -la $out
