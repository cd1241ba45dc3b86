#!/bin/bash
# libq3tts_stamps.so (kernels.hip under -DQ3_STAMPS, every other object from the product build) + scripts/bin/ubench_chain linked against it
set -e
cd "$(dirname "$0")/.."
PKG=qwen3-tts-rust_amd
FLAGS="-mllvm -amdgpu-kernarg-preload-count=16 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
mkdir -p scripts/bin
/opt/rocm/bin/hipcc $FLAGS ${Q3_UB_DEFS:-} -DQ3_STAMPS=${Q3_STAMPS:-2} -x hip -c $PKG/csrc/kernels.hip -o scripts/bin/kernels_stamps.o
OBJS=$(ls $PKG/build/*.o | grep -v "/kernels.hip.o\|/kernels_fused.hip.o")
/opt/rocm/bin/hipcc $FLAGS ${Q3_UB_DEFS:-} -DQ3_STAMPS=${Q3_STAMPS:-2} -x hip -c $PKG/csrc/kernels_fused.hip -o scripts/bin/kernels_fused_stamps.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/bin/libq3tts_stamps.so scripts/bin/kernels_stamps.o scripts/bin/kernels_fused_stamps.o $OBJS -ldl
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DQ3_STAMPS=1 -I$PKG/csrc -o scripts/bin/ubench_chain scripts/ubench_chain.hip -Lscripts/bin -lq3tts_stamps -Wl,-rpath,'$ORIGIN'
echo built scripts/bin/ubench_chain
