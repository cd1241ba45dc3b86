#!/bin/bash
# Boundary A (zero-Rust-change path) timing on the full synthetic model: the reference's loop replayed call for call through
# runtime/libllama.so (host/ref_replay.cpp), 43-row prompt, 32 frames, no codec.  Run on the GPU box after bench.py has written the model.
set -e
M=${Q3_BENCH_MODEL:-/tmp/q3tts_synth_full}
cd "$(dirname "$0")/../qwen3-tts-rust_amd"
python3 - <<PY
import sys, os, json, numpy as np
sys.path.insert(0, "python"); sys.path.insert(0, "..")
import q3tts as Q, bench
a = Q.Assets("$M/gguf_q8_0/qwen3_assets.gguf")
spk = np.array(json.load(open("../tests/golden/speakers/vivian.json"))["spk_emb"], np.float32)
bench.build_prompt(a, spk).tofile("/tmp/replay_prompt.f32")
PY
./ref_replay $M/gguf_q8_0 /tmp/replay_prompt.f32 43 32 /tmp/replay_codes.i32 1
