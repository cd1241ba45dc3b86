"""GGUF container + ggml block formats: oracle C code vs an independent numpy restatement (tests/ggml_ref.py)."""
import ctypes as C
import os
import numpy as np
import ggml_ref as G


def _deq(oracle, ty, raw, k):
    out = np.zeros(k, np.float32)
    raw = np.ascontiguousarray(raw)
    oracle.lib().q3o_dequant_row(ty, raw.ctypes.data, k, out.ctypes.data)
    return out


def test_gguf_files_parse_and_metadata(tiny_model, oracle):
    kv, tensors = G.read_gguf(os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf"))
    assert kv["general.architecture"] == "qwen3-tts-talker"
    assert kv["qwen3-tts-talker.embedding_length"] == 2048 and kv["qwen3-tts-talker.rope.dimension_sections"] == [24, 20, 20, 0]
    assert tensors["blk.0.attn_q.weight"][0] == 8 and tensors["blk.0.attn_norm.weight"][0] == 0
    m = oracle.Model(os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf"), 64)
    m.close()
    # the assets file has no array-typed KV (the reference's reader rejects type 9: assets_manager.rs:93-97)
    kv, tensors = G.read_gguf(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"))
    assert not any(isinstance(v, list) for v in kv.values())
    assert set(tensors) >= {"proj.weight", "proj.bias", "text_embd"} | {"codec_embd.%d" % i for i in range(16)}
    assert all(t[0] == 0 for t in tensors.values())  # F32 only (assets_manager.rs:163-167)


def test_bad_gguf_is_rejected(tmp_path, oracle):
    p = tmp_path / "bad.gguf"
    p.write_bytes(b"GGUX" + b"\0" * 64)
    err = C.create_string_buffer(256)
    assert not oracle.lib().q3o_gguf_open(str(p).encode(), err, 256)
    assert b"Not a GGUF file" in err.value
    p.write_bytes(b"GGUF" + (1).to_bytes(4, "little") + b"\0" * 64)
    assert not oracle.lib().q3o_gguf_open(str(p).encode(), err, 256)
    assert b"Unsupported GGUF version" in err.value


def test_dequant_q8_0_q5_k_q6_k_match_public_formulas(tiny_model, oracle):
    for sub, names in (("gguf_q8_0", ["blk.0.attn_q.weight"]), ("gguf_q5_k_m", ["blk.0.attn_q.weight", "blk.0.attn_v.weight", "output.weight"])):
        _, tensors = G.read_gguf(os.path.join(tiny_model, sub, "qwen3_tts_talker.gguf"))
        for name in names:
            ty, ne, raw = tensors[name]
            k = ne[0]
            rb = G.ROW_BYTES[ty](k)
            ref_fn = {8: G.deq_q8_0, 13: G.deq_q5_k, 14: G.deq_q6_k}[ty]
            for r in (0, 1, ne[1] - 1):
                row = np.array(raw[r * rb:(r + 1) * rb])
                got = _deq(oracle, ty, row, k)
                ref = ref_fn(row, k)[0]
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, ty, r)
    # the Q5_K_M mix really contains both K-quant types
    _, tensors = G.read_gguf(os.path.join(tiny_model, "gguf_q5_k_m", "qwen3_tts_talker.gguf"))
    assert {t[0] for n, t in tensors.items() if n.endswith("weight") and len(t[1]) == 2} >= {13, 14}


def test_quantised_matvec_close_to_float_reference(tiny_model, oracle):
    """spec S3 sanity: the int8-block dot equals the dequantised f64 dot up to activation-quantisation noise."""
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for sub in ("gguf_q8_0", "gguf_q5_k_m"):
        _, tensors = G.read_gguf(os.path.join(tiny_model, sub, "qwen3_tts_talker.gguf"))
        for name in ("blk.0.attn_q.weight", "blk.0.ffn_down.weight"):
            ty, ne, raw = tensors[name]
            k, n = ne[0], 16
            rb = G.ROW_BYTES[ty](k)
            w = np.array(raw[: n * rb])
            x = rng.standard_normal(k).astype(np.float32)
            xq = np.zeros(k, np.int8); xd = np.zeros(k // 32, np.uint16)
            L.q3o_quant_act(x.ctypes.data, k, xq.ctypes.data, xd.ctypes.data)
            y = np.zeros(n, np.float32)
            L.q3o_matvec(ty, w.ctypes.data, n, k, xq.ctypes.data, xd.ctypes.data, x.ctypes.data, y.ctypes.data)
            wf = np.stack([_deq(oracle, ty, w[r * rb:(r + 1) * rb], k) for r in range(n)]).astype(np.float64)
            ref = wf @ x.astype(np.float64)
            xdeq = xq.astype(np.float64) * np.repeat(xd.view(np.float16).astype(np.float64), 32)
            exact = wf @ xdeq  # same quantised operands, exact arithmetic
            assert np.abs(y - exact).max() < 1e-5 * np.abs(exact).max() + 1e-6
            assert np.abs(y - ref).max() < 0.02 * np.abs(ref).max() + 1e-3
