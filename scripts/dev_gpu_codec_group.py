"""codec-only throughput of grouped decoding on the full-size synthetic codec: audio seconds per second for G streams x 4 frames per pass"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
import q3tts as Q
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/q3tts_synth_full"
path = out + "/onnx/q3tts_codec.gguf"
if not os.path.exists(path):
    import subprocess
    tool = os.path.join(ROOT, "tools", "q3synth")
    if not os.path.exists(tool):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    subprocess.check_call([tool, "--out", out, "--preset", "full", "--quant", "q8_0", "--what", "4"])
rng = np.random.default_rng(1)
for G in (1, 4, 16, 32):
    gd = Q.Decoder(path, n_streams=G, max_frames=4, max_group=G)
    for s in range(G):
        gd.reset(s)
    codes = rng.integers(0, 2048, (G, 4, 16))
    for _ in range(3):
        gd.decode_group(list(range(G)), codes)
    n = 20
    t0 = time.time()
    for _ in range(n):
        gd.decode_group(list(range(G)), codes)
    dt = (time.time() - t0) / n
    print("G=%d: %.3f ms per pass, %.0f audio-s/s codec-only (incl. host copies)" % (G, dt * 1e3, G * 4 * 0.08 / dt))
    gd.close()
