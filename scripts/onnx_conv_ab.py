"""Times decoder-shaped convolutions through the ONNX executor (q3tts_onnx_session_*): run once as is and once with Q3_ONNX_NAIVE=1 (the
one-output-per-thread kernels) to see what the matrix-core path (k_mm_mfma) buys.  Graph: Conv1d 512 -> 512, k = 7, dilation 3 over 4096 positions;
Conv1d 512 -> 512, k = 1; ConvTranspose1d 512 -> 256, k = 16, stride 8 over 512 positions; MatMul [1024 x 1024] x [1024 x 1024]."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3tts as Q
import onnx_writer as W
rng = np.random.default_rng(0)
f = lambda *s: (rng.standard_normal(s) * 0.05).astype(np.float32)
cases = {
    "conv k7 d3 512->512 x4096": ([W.node("Conv", ["x", "w"], ["y"], attrs=[W.attr_ints("dilations", [3]), W.attr_ints("pads", [9, 9])])], {"w": f(512, 512, 7)}, {"x": f(1, 512, 4096)}, [1, 512, 4096], 2 * 512 * 512 * 7 * 4096),
    "conv k1 512->512 x4096": ([W.node("Conv", ["x", "w"], ["y"])], {"w": f(512, 512, 1)}, {"x": f(1, 512, 4096)}, [1, 512, 4096], 2 * 512 * 512 * 4096),
    "convT k16 s8 512->256 x512": ([W.node("ConvTranspose", ["x", "w"], ["y"], attrs=[W.attr_ints("strides", [8]), W.attr_ints("pads", [4, 4])])], {"w": f(512, 256, 16)}, {"x": f(1, 512, 512)}, [1, 256, 4096], 2 * 512 * 256 * 16 * 512),
    "matmul 1024^3": ([W.node("MatMul", ["x", "w"], ["y"])], {"w": f(1024, 1024)}, {"x": f(1024, 1024)}, [1024, 1024], 2 * 1024 ** 3),
}
for name, (nodes, inits, feeds, oshape, flops) in cases.items():
    path = "/tmp/onnx_ab_%d.onnx" % abs(hash(name))
    ins = [W.value_info(k, W.F32, list(v.shape)) for k, v in feeds.items()]
    open(path, "wb").write(W.model(nodes, [W.tensor(k, v) for k, v in inits.items()], ins, [W.value_info("y", W.F32, oshape)], opset=17))
    s = Q.OnnxSession(path)
    for _ in range(2):
        s.run(feeds, ["y"])
    n = 5
    t0 = time.time()
    for _ in range(n):
        s.run(feeds, ["y"])
    dt = (time.time() - t0) / n
    s.close()
    print("%-30s %8.3f ms per run (incl. feed upload + fetch)  %.2f TFLOP/s" % (name, dt * 1e3, flops / dt / 1e12))
