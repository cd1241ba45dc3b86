#!/bin/bash
# Builds libq3tts.so (HIP kernels + C ABI, both boundaries) in-tree for gfx950.  No cmake: plain hipcc.
set -e
cd "$(dirname "$0")"
ARCH=${Q3_ARCH:-gfx950}
# -amdgpu-kernarg-preload-count: the first 16 kernel-argument dwords arrive in SGPRs with the dispatch instead of through an s_load at kernel
# start (measured 2.918 -> 2.893 ms/frame at B=1); Q3_EXTRA_FLAGS is for experiments
FLAGS="${Q3_EXTRA_FLAGS:-} -mllvm -amdgpu-kernarg-preload-count=16 -O3 -std=c++17 -fPIC --offload-arch=$ARCH -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result"
mkdir -p build
OBJS=""
PIDS=""
for f in csrc/gguf.cpp csrc/host_logic.cpp csrc/kernels.hip csrc/kernels_fused.hip csrc/ggml_mode.hip csrc/sampler.hip csrc/transformer.cpp csrc/engine.cpp csrc/codec.hip csrc/mel.hip csrc/onnx_reader.cpp csrc/onnx_exec.hip csrc/tokenizer.cpp csrc/capi.cpp csrc/capi_ops.cpp csrc/group.cpp csrc/llama_shim.cpp; do
  [ -f "$f" ] || continue
  o=build/$(basename "$f").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find csrc ../include -newer "$o" \( -name '*.h' \) -print -quit)" ]; then
    echo "hipcc $f"
    rm -f "$o"   # a failed compile must not leave a stale object behind for the link
    /opt/rocm/bin/hipcc $FLAGS -x hip -c "$f" -o "$o" &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
FAIL=0
for p in $PIDS; do wait "$p" || FAIL=1; done
if [ "$FAIL" != 0 ]; then echo "build failed" >&2; exit 1; fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=$ARCH -o libq3tts.so $OBJS -ldl
mkdir -p runtime && cp -f libq3tts.so runtime/libllama.so
# host side: C++ mirror of the reference API + the Boundary-A replay harness (plain g++, no HIP)
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -shared -o libq3tts_host.so host/tts_engine.cpp -L. -lq3tts -Wl,-rpath,'$ORIGIN'
g++ -O2 -std=c++17 -ffp-contract=off -o ref_replay host/ref_replay.cpp -ldl
g++ -O2 -std=c++17 -I../include -o ../tools/q3onnx_dump ../tools/q3onnx_dump.cpp -L. -lq3tts -Wl,-rpath,'$ORIGIN/../qwen3-tts-rust_amd'
echo "built $(pwd)/libq3tts.so"
