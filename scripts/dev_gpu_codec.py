import os, sys, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
import q3oracle as O, q3tts as Q
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/q3tiny"
preset = sys.argv[2] if len(sys.argv) > 2 else "tiny"
if not os.path.exists(out + "/onnx/q3tts_codec.gguf"):
    subprocess.check_call([os.path.join(ROOT, "tools", "q3synth"), "--out", out, "--preset", preset, "--quant", "q8_0", "--what", "4"])
path = out + "/onnx/q3tts_codec.gguf"
rng = np.random.default_rng(1)
nf = 10 if preset == "tiny" else 6
codes = rng.integers(0, 2048, (nf, 16))
oc = O.Codec(path); oc.reset()
t0 = time.time(); ref = oc.decode(codes).copy(); print("oracle %.2fs" % (time.time() - t0))
gd = Q.Decoder(path, 1); gd.reset()
t0 = time.time()
chunks = [codes[:4], codes[4:5], codes[5:]]
got = np.concatenate([gd.decode(c, i == len(chunks) - 1) for i, c in enumerate(chunks)])
print("gpu %.3fs" % (time.time() - t0), got.shape, ref.shape)
err = got - ref
print("rms err %.3e max %.3e ; ref rms %.3f" % (np.sqrt(np.mean(err ** 2)), np.abs(err).max(), np.sqrt(np.mean(ref ** 2))))
for i in range(0, got.size, got.size // 6):
    seg = slice(i, i + got.size // 6)
    print("  seg", i, "rms %.2e" % np.sqrt(np.mean(err[seg] ** 2)))
gd.reset(); t0 = time.time(); gd.decode(codes[:4]); t1 = time.time() - t0; t0 = time.time(); gd.decode(codes[:4]); print("4-frame chunk: %.3f ms, %.3f ms" % (t1 * 1e3, (time.time() - t0) * 1e3))
