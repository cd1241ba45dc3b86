"""Developer GPU check: op-level and end-to-end parity of the HIP path against the oracle on the tiny model."""
import os, sys, json, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
import q3oracle as O
import q3tts as Q

out = "/tmp/q3tiny"
if not os.path.exists(out + "/gguf_q8_0/qwen3_tts_talker.gguf"):
    subprocess.check_call([os.path.join(ROOT, "tools", "q3synth"), "--out", out, "--preset", "tiny", "--quant", "q8_0"])
print("devices", Q.device_count())
rng = np.random.default_rng(0)
# --- gemv op
L = O.lib()
for (n, k, ntok) in [(64, 256, 1), (96, 2048, 1), (256, 6144, 3), (2048, 1024, 2), (128, 3072, 9)]:
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    # encode Q8_0 like ggml
    wb = w.reshape(n, k // 32, 32)
    amax = np.abs(wb).max(-1)
    d = (amax / 127).astype(np.float32)
    idv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0).astype(np.float32)
    q = np.rint(wb * idv[..., None]).astype(np.int8)
    raw = np.zeros((n, k // 32, 34), np.uint8)
    raw[..., :2] = d.astype(np.float16).view(np.uint8).reshape(n, k // 32, 2)
    raw[..., 2:] = q.view(np.uint8)
    x = rng.standard_normal((ntok, k)).astype(np.float32)
    xq = np.zeros((ntok, k), np.int8); xd = np.zeros((ntok, k // 32), np.uint16)
    for t in range(ntok):
        L.q3o_quant_act(x[t].ctypes.data, k, xq[t].ctypes.data, xd[t].ctypes.data)
    yo = np.zeros((ntok, n), np.float32)
    for t in range(ntok):
        L.q3o_matvec(8, raw.ctypes.data, n, k, xq[t].ctypes.data, xd[t].ctypes.data, None, yo[t].ctypes.data)
    for lpr in (2, 4, 8):
        yg = Q.op_gemv_q8(raw, n, k, xq, xd, lpr)
        print("gemv", n, k, ntok, lpr, "bitexact" if np.array_equal(yo.view(np.uint32), yg.view(np.uint32)) else "MISMATCH max|d|=%g" % np.abs(yo - yg).max())
# --- rmsnorm
for d in (256, 1024, 2048):
    x = rng.standard_normal((3, d)).astype(np.float32) * 3
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    xq, xd, xn = Q.op_rmsnorm_quant(x, g, 1e-6)
    ok = True
    for t in range(3):
        y = np.zeros(d, np.float32); L.q3o_rmsnorm(x[t].ctypes.data, g.ctypes.data, d, 1e-6, y.ctypes.data)
        q = np.zeros(d, np.int8); dd = np.zeros(d // 32, np.uint16); L.q3o_quant_act(y.ctypes.data, d, q.ctypes.data, dd.ctypes.data)
        ok &= np.array_equal(y.view(np.uint32), xn[t].view(np.uint32)) and np.array_equal(q, xq[t]) and np.array_equal(dd, xd[t])
    print("rmsnorm", d, "bitexact" if ok else "MISMATCH")
# --- transformer layer-level
tp = out + "/gguf_q8_0/qwen3_tts_talker.gguf"
om = O.Model(tp, 4096)
gm = Q.TfContext(tp, 4096, 16)
xs = rng.standard_normal((20, 2048)).astype(np.float32) * 0.3
pos = np.array([[t, t, t, 0] for t in range(20)], np.int32)
hg, lg = gm.eval(xs[:12], pos[:12], 0, 2160)
ok = True
for t in range(12):
    ho, lo = om.eval(xs[t], pos[t], 2048, 0, 2160)
    e1 = np.array_equal(ho.view(np.uint32), hg[t].view(np.uint32)); e2 = np.array_equal(lo.view(np.uint32), lg[t].view(np.uint32))
    if not (e1 and e2):
        print("tok", t, "hidden", e1, np.abs(ho - hg[t]).max(), "logits", e2, np.abs(lo - lg[t]).max()); ok = False
for t in range(12, 20):
    h1, l1 = gm.eval(xs[t:t + 1], pos[t:t + 1], 0, 2160)
    ho, lo = om.eval(xs[t], pos[t], 2048, 0, 2160)
    if not (np.array_equal(ho.view(np.uint32), h1[0].view(np.uint32)) and np.array_equal(lo.view(np.uint32), l1[0].view(np.uint32))):
        print("dec tok", t, np.abs(ho - h1[0]).max(), np.abs(lo - l1[0]).max()); ok = False
print("transformer prefill(12)+decode(8):", "bitexact" if ok else "MISMATCH")
# --- end to end
spk = np.array(json.load(open(os.path.join(ROOT, "tests/golden/speakers/vivian.json")))["spk_emb"], np.float32)
oe = O.Engine(out + "/gguf_q8_0", None, 8)
text = rng.integers(0, 4000, 8).astype(np.int32)
prompt = oe.assets.build_core(text, spk_emb=spk)
t0 = time.time(); oc, _ = oe.generate(prompt, max_steps=12); t1 = time.time()
ge = Q.Engine(out, "q8_0", max_batch=1, max_steps=64, load_codec=False)
p2 = ge.assets.build_core(text, spk_emb=spk)
print("prompt equal:", np.array_equal(prompt, p2))
t2 = time.time(); r = ge.generate_batch([prompt], max_steps=12)[0]; t3 = time.time()
print("oracle %.2fs gpu %.3fs" % (t1 - t0, t3 - t2))
print("codes equal:", np.array_equal(oc, r["codes"]), oc.shape, r["codes"].shape)
if not np.array_equal(oc, r["codes"]):
    print(oc[:2]); print(r["codes"][:2])
r2 = ge.generate_batch([prompt], max_steps=12)[0]
print("repeat equal:", np.array_equal(r2["codes"], r["codes"]))
print(ge.stats())
