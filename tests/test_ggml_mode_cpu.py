"""the oracle's "ggml-CPU" arithmetic mode (oracle/q3o_ggml.c, Q3_SPEC=ggml) -- llama.cpp's portable CPU kernels restated
[EXT] -- against the engine's own arithmetic specification, on the seeded tiny model.  llama.cpp itself is not in this image, so "token ids
match the reference" is turned into a MEASURED statement: how far apart two faithful arithmetics put the logits (noise), how large the
greedy top-2 margins are, and that tokens agree wherever margin > noise.  Random synthetic weights have far smaller margins than a trained
model, so the agreement rate printed here is a lower bound for real weights, not an estimate of it."""
import os
import numpy as np
import pytest


def test_ggml_kernels_match_dequantised_reference(oracle, tiny_model):
    """q3o_matvec_ggml for Q8_0 / Q5_K / Q6_K rows vs a float64 dot product of the DEQUANTISED weights with the DEQUANTISED (Q8_0 / Q8_K)
    activations -- exact up to f32 accumulation noise, which pins block layouts, scale/min unpacking and the bsums * mins term."""
    import ctypes as C
    import ggml_ref as G
    L = oracle.lib()
    L.q3o_matvec_ggml.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.q3o_dequant_row.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(3)
    for sub, name in (("gguf_q8_0", "blk.0.attn_q.weight"), ("gguf_q5_k_m", "blk.0.attn_q.weight"), ("gguf_q5_k_m", "blk.0.ffn_down.weight"),
                      ("gguf_q5_k_m", "blk.1.attn_v.weight")):
        _, t = G.read_gguf(os.path.join(tiny_model, sub, "qwen3_tts_talker.gguf"))
        ty, shape, raw = t[name]
        raw = np.frombuffer(raw, np.uint8).copy() if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8)
        k, n = int(shape[0]), 24
        rb = raw.size // int(shape[1])
        x = (rng.standard_normal(k) * 0.7).astype(np.float32)
        y = np.zeros(n, np.float32)
        L.q3o_matvec_ggml(int(ty), raw.ctypes.data, n, k, x.ctypes.data, y.ctypes.data)
        # reference: dequantise the rows; quantise / dequantise the activations the way the weight type asks
        w = np.zeros((n, k), np.float32)
        for r in range(n):
            L.q3o_dequant_row(int(ty), raw[r * rb:].ctypes.data, k, w[r].ctypes.data)
        if int(ty) == 8:
            xb = x.reshape(-1, 32); d = (np.abs(xb).max(1) / 127).astype(np.float32)
            q = np.where(d[:, None] > 0, np.sign(xb) * np.floor(np.abs(xb / np.where(d[:, None] > 0, d[:, None], 1)) + 0.5), 0)
            xdq = (q * d.astype(np.float16).astype(np.float32)[:, None]).reshape(-1)
        else:
            xb = x.reshape(-1, 256); idx = np.abs(xb).argmax(1); mx = xb[np.arange(xb.shape[0]), idx]
            isc = (-127.0 / mx).astype(np.float32)
            q = np.minimum(np.rint((isc[:, None] * xb).astype(np.float32)), 127)
            xdq = (q * (1.0 / isc)[:, None].astype(np.float32)).reshape(-1)
        ref = w.astype(np.float64) @ xdq.astype(np.float64)
        assert np.abs(y - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), (sub, name, np.abs(y - ref).max())


@pytest.mark.parametrize("sub", ["gguf_q8_0", "gguf_q5_k_m"])
def test_token_agreement_and_margins_spec_vs_ggml(oracle, tiny_model, vivian, sub):
    qdir = os.path.join(tiny_model, sub)
    if not os.path.exists(os.path.join(qdir, "qwen3_assets.gguf")):
        os.symlink(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"), os.path.join(qdir, "qwen3_assets.gguf"))
    eng = oracle.Engine(qdir, None, 4)
    prompt = eng.assets.build_core(np.arange(100, 108, dtype=np.int32), lang_id=2055, spk_emb=vivian)
    n = 10
    try:
        oracle.set_arith_mode(0)
        spec_codes, spec_own, spec_m = eng.generate_measured(prompt, n)
        assert np.array_equal(spec_codes, spec_own)
        # logits of both arithmetics at the same state: last prompt token through the talker
        m = oracle.Model(os.path.join(qdir, "qwen3_tts_talker.gguf"), 64)
        def last_logits():
            m.clear()
            lg = None
            for t in range(prompt.shape[0]):
                _, lg = m.eval(prompt[t], [t, t, t, 0], 2048, 0, 2160)
            return lg.copy()
        l_spec = last_logits()
        oracle.set_arith_mode(1)
        l_ggml = last_logits()
        m.close()
        noise = float(np.abs(l_spec - l_ggml).max())
        scale = float(np.abs(l_spec).max())
        assert 0.0 < noise < 0.05 * scale, (noise, scale)     # two faithful arithmetics: different bits, same numbers
        # teacher-forced along the spec trajectory: per-code agreement, and the rule "margin > 2 x noise => same token"
        _, ggml_own, ggml_m = eng.generate_measured(prompt, n, forced=spec_codes)
    finally:
        oracle.set_arith_mode(0)
        eng.close()
    agree0 = spec_codes[:, 0] == ggml_own[:, 0]
    agree_all = spec_codes == ggml_own
    safe = spec_m[:, 0] > 2.0 * noise
    assert np.all(agree0[safe]), (spec_m[:, 0], noise)
    rate = float(agree_all.mean())
    print("\n[%s] logit noise spec vs ggml %.3e (|logit| max %.2f); code_0 margins median %.3e min %.3e; predictor min-margins median %.3e; "
          "codes equal under teacher forcing: %.1f %% (code_0 %.0f %%), frames with margin > 2 x noise: %d / %d"
          % (sub, noise, scale, float(np.median(spec_m[:, 0])), float(spec_m[:, 0].min()), float(np.median(spec_m[:, 1])), 100 * rate,
             100 * float(agree0.mean()), int(safe.sum()), n))
    assert rate > 0.5   # random-weight margins are tiny; real weights separate far better (this is a floor, not a claim)
