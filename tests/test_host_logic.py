"""Source-pinned host semantics: sampler, prompt layouts, chunker, gathers, projection order.
Expected values are derived by hand from the reference source (file:line in each test); both the oracle (C) and the
product's host logic (C++ behind the C ABI, CPU-only) must reproduce them."""
import json
import os
import numpy as np
import pytest


# ---------------- sampler: /root/reference/src/models/llama/mod.rs:666-776 ----------------
def _both(oracle, q3, logits, start, end, **kw):
    o, _ = oracle.sample(logits, start, end, **kw)
    s = q3.Sampler(kw.get("temperature", 0.0), kw.get("top_k", 0), kw.get("top_p", 1.0), kw.get("seed", 42))
    p = s.sample(logits, start, end)
    s.close()
    assert o == p
    return o


def test_greedy_first_max_and_range(oracle, q3):
    lg = np.zeros(3072, np.float32)
    lg[[5, 9, 2150, 2500]] = [3.0, 3.0, 9.0, 100.0]
    assert _both(oracle, q3, lg, 0, 2160) == 2150          # 2500 is outside [0,2160) (engine.rs:555)
    lg[2150] = 3.0
    assert _both(oracle, q3, lg, 0, 2160) == 5             # ties: first max, strict '>' (mod.rs:695)
    assert _both(oracle, q3, lg, 6, 2160) == 9             # range start respected
    assert _both(oracle, q3, np.full(64, -np.inf, np.float32), 10, 40) == 10   # nothing > -inf: max_idx stays `start` (:692)
    nan = np.full(64, np.nan, np.float32); nan[20] = -5.0
    assert _both(oracle, q3, nan, 0, 64) == 20             # NaN never compares greater
    assert _both(oracle, q3, lg, 0, 99999) == 2500         # end clamps to n_vocab (:687)


def test_predictor_slices(oracle, q3):
    rng = np.random.default_rng(0)
    lg = rng.standard_normal(30720).astype(np.float32)
    for q in (1, 7, 15):                                   # engine.rs:588-596
        s, e = (q - 1) * 2048, q * 2048
        assert _both(oracle, q3, lg, s, e) - s == int(np.argmax(lg[s:e]))


def test_temperature_topk_topp_edges(oracle, q3):
    lg = np.array([0.0, 5.0, 4.0, -2.0, 4.0, 1.0], np.float32)
    # top_k = 1 => only the best candidate survives whatever r is (mod.rs:711-713)
    for seed in range(5):
        assert _both(oracle, q3, lg, 0, 6, temperature=1.0, top_k=1, top_p=1.0, seed=seed) == 1
    # tiny top_p => cut after the first candidate (cumsum >= top_p inclusive, :737-744)
    for seed in range(5):
        assert _both(oracle, q3, lg, 0, 6, temperature=0.7, top_k=0, top_p=1e-6, seed=seed) == 1
    # stable sort keeps index 2 before index 4 on equal logits (:708), so top_k=1 keeps index 2
    for seed in range(5):
        assert _both(oracle, q3, lg, 2, 6, temperature=1.0, top_k=1, top_p=1.0, seed=seed) == 2
    # very low temperature: only the two tied maxima carry probability
    assert {_both(oracle, q3, lg, 2, 6, temperature=1e-3, top_k=0, top_p=1.0, seed=s) for s in range(20)} == {2, 4}
    # negative top_k is cast to a huge usize => disabled (:646,711)
    got = {_both(oracle, q3, lg, 0, 6, temperature=5.0, top_k=-3, top_p=1.0, seed=s) for s in range(40)}
    assert len(got) >= 4
    # default config (0.7/40/0.9, engine.rs:25-34) draws only from the nucleus
    rng = np.random.default_rng(1)
    big = rng.standard_normal(2160).astype(np.float32) * 3
    order = np.argsort(-big, kind="stable")
    p = np.exp((big[order[:40]] - big[order[0]]) / np.float32(0.7)); p /= p.sum()
    nucleus = set(order[: int(np.searchsorted(np.cumsum(p), 0.9) + 1)].tolist())
    for s in range(30):
        assert _both(oracle, q3, big, 0, 2160, temperature=0.7, top_k=40, top_p=0.9, seed=s) in nucleus


def test_chacha12_known_answer(oracle):
    """Pins the RNG core (rand StdRng = ChaCha12) against a published known-answer vector: all-zero 256-bit key, zero counter/nonce,
    12 rounds (J. Strombergson's ChaCha test vectors, TC1) -- first keystream block.  The 20-round RFC 7539-style vector
    76b8e0ad... differs, so the round count is pinned too.  The product's host and device samplers are tied to this
    implementation by the sampler parity tests."""
    import ctypes as C
    L = oracle.lib()
    L.q3o_rng_next_u32.restype = C.c_uint32
    L.q3o_rng_next_u32.argtypes = [C.c_void_p]

    class Rng(C.Structure):
        _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64), ("buf", C.c_uint32 * 16), ("idx", C.c_int)]
    r = Rng()
    r.idx = 16          # buffer exhausted -> the next call generates block 0
    words = [L.q3o_rng_next_u32(C.byref(r)) for _ in range(16)]
    stream = b"".join(int(w).to_bytes(4, "little") for w in words).hex()
    assert stream == ("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                      "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")
    assert L.q3o_rng_next_u32(C.byref(r)) != words[0] and r.counter == 2   # block counter advances in words 12-13


def test_rng_stream_is_shared_and_seeded(oracle):
    import ctypes as C
    L = oracle.lib()
    a = C.create_string_buffer(256); b = C.create_string_buffer(256)
    L.q3o_rng_seed(a, 42); L.q3o_rng_seed(b, 42)
    s1 = [L.q3o_rng_next_u32(a) for _ in range(40)]
    assert s1 == [L.q3o_rng_next_u32(b) for _ in range(40)]
    L.q3o_rng_seed(b, 43)
    assert s1 != [L.q3o_rng_next_u32(b) for _ in range(40)]
    assert len(set(s1)) == 40


# ---------------- gathers / fallback / tts_pad: assets_manager.rs:244-249,419-460 ----------------
def test_gathers_and_fallbacks(tiny_model, oracle, q3):
    path = os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf")
    oa, pa = oracle.Assets(path), q3.Assets(path)
    import ggml_ref as G
    _, t = G.read_gguf(path)
    e0 = np.array(t["codec_embd.0"][2]).view(np.float32).reshape(-1, 2048)
    e3 = np.array(t["codec_embd.3"][2]).view(np.float32).reshape(-1, 2048)
    txt = np.array(t["text_embd"][2]).view(np.float32).reshape(-1, 2048)
    for A in (oa, pa):
        assert np.array_equal(A.codec_embedding(0, 3065), e0[3065])
        assert np.array_equal(A.codec_embedding(3, 2047), e3[2047])
        assert np.array_equal(A.codec_embedding(3, -7), e3[0])           # negative code -> 0 (:422)
        assert not A.codec_embedding(3, 2048).any()                       # OOB -> zeros (:436)
        assert not A.codec_embedding(16, 0).any()                         # q out of range -> zeros
        assert np.array_equal(A.text_embedding(17), txt[17])
        tok = 151644                                                      # beyond the reduced table -> fallback (:454-460)
        exp = np.array([np.float32(np.float32(tok * 17 + i) % np.float32(2.0)) - np.float32(1.0) for i in range(2048)], np.float32)
        assert np.array_equal(A.text_embedding(tok), exp)
    assert not pa.tts_pad().any()                                         # table shorter than 151672 rows -> zeros (:248)
    oa.close(); pa.close()


# ---------------- prompt layouts: prompt.rs:28-118,141-277 ----------------
def test_prompt_layouts(tiny_model, oracle, q3, vivian):
    path = os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf")
    oa, pa = oracle.Assets(path), q3.Assets(path)
    T, Cd = pa.text_embedding, lambda c: pa.codec_embedding(0, c)
    marker, pad0 = T(151671), Cd(2148)
    text = np.array([11, 22, 33], np.int32)
    # preset path (engine.rs:398-412): lang 2055, speaker embedding row = marker + spk_emb
    exp = [T(151644), T(77091), T(198)] + [marker + Cd(c) for c in (2154, 2156, 2055, 2157)] + [marker + vivian]
    exp += [T(151672) + pad0] + [T(int(t)) + pad0 for t in text] + [T(151673) + pad0, marker + Cd(2149)]
    exp = np.stack(exp)
    for A in (oa, pa):
        got = A.build_core(text, lang_id=2055, spk_emb=vivian)
        assert got.shape == (11 + 3, 2048) and np.array_equal(got, exp)
    # no language => NOTHINK block (prompt.rs:192-204); speaker by id (:207-214); instruct block (:154-169)
    ins = np.array([7, 8], np.int32)
    exp2 = [T(151644), T(872), T(198), T(7), T(8), T(151645), T(198), T(151644), T(77091), T(198)]
    exp2 += [marker + Cd(c) for c in (2155, 2156, 2157)] + [marker + Cd(3065)]
    exp2 += [T(151672) + pad0] + [T(int(t)) + pad0 for t in text] + [T(151673) + pad0, marker + Cd(2149)]
    for A in (oa, pa):
        assert np.array_equal(A.build_core(text, lang_id=None, spk_id=3065, instr_ids=ins), np.stack(exp2))
    # clone prompt (prompt.rs:28-118): mid = BOS/ref/EOS text + pad, marker+E0[2160], per-frame marker+sum_q, marker+pad
    rng = np.random.default_rng(5)
    ref_codes = rng.integers(0, 2048, 3 * 16 + 5).astype(np.int32)   # 3 whole frames, 5 stray codes ignored (:79)
    ref_text = np.array([5, 6], np.int32)
    mid = [T(151672) + pad0, T(5) + pad0, T(6) + pad0, T(151673) + pad0, marker + Cd(2160)]
    for s in range(3):
        acc = np.zeros(2048, np.float32)
        for q in range(16):
            acc = acc + pa.codec_embedding(q, int(ref_codes[s * 16 + q]))
        mid.append(marker + acc)
    mid.append(marker + pad0)
    exp3 = [T(151644), T(77091), T(198)] + [marker + Cd(c) for c in (2154, 2156, 2055, 2157)] + [marker + vivian] + mid
    exp3 += [T(151672) + pad0] + [T(int(t)) + pad0 for t in text] + [T(151673) + pad0, marker + Cd(2149)]
    for A in (oa, pa):
        got = A.build_clone(text, ref_codes, ref_text, vivian)
        assert np.array_equal(got, np.stack(exp3))
    oa.close(); pa.close()


# ---------------- chunker: engine.rs:505-541 ----------------
def _oracle_chunker_trace(oracle, msgs):
    import ctypes as C
    L = oracle.lib()
    calls = []
    CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int)
    cb = CB(lambda u, codes, n, fin: calls.append(([codes[i] for i in range(n)], bool(fin))))
    st = C.create_string_buffer(4096 * 8 + 4 + 4 + 1024 * 8 + 64)
    for codes, fin in msgs:
        a = np.ascontiguousarray(codes, np.int64)
        L.q3o_chunker_push(st, a.ctypes.data if a.size else None, a.size, 1 if fin else 0, C.cast(cb, C.c_void_p), None)
    return calls


def _product_chunker_trace(q3, msgs):
    ch = q3.Chunker()
    for codes, fin in msgs:
        ch.push(codes, fin)
    out = ch.calls
    ch.close()
    return out


@pytest.mark.parametrize("n_frames,expect", [
    (10, [(4, False), (4, False), (2, True)]),   # leftover flushed with is_last=1
    (8, [(4, False), (4, False)]),               # QUIRK: frame count multiple of 4 -> no is_last call (engine.rs:510-536)
    (3, [(3, True)]),
    (0, []),
])
def test_chunker_traces(oracle, q3, n_frames, expect):
    frames = [np.arange(16, dtype=np.int64) + 100 * f for f in range(n_frames)]
    msgs = [(f, False) for f in frames] + [(np.zeros(0, np.int64), True)]
    for calls in (_oracle_chunker_trace(oracle, msgs), _product_chunker_trace(q3, msgs)):
        assert [(len(c) // 16, fin) for c, fin in calls] == expect
        flat = [v for c, _ in calls for v in c]
        assert flat == [int(min(max(v, 0), 2047)) for f in frames for v in f]


def test_chunker_clamps_and_handles_ragged_messages(oracle, q3):
    msgs = [(np.array([-5, 3000] + [7] * 14 + [1, 2, 3], np.int64), False), (np.arange(45, dtype=np.int64), False),
            (np.array([9] * 4, np.int64), True)]
    for calls in (_oracle_chunker_trace(oracle, msgs), _product_chunker_trace(q3, msgs)):
        # 19 codes (<64, no call), +45 = 64 -> one call of 4 frames; final: 4 stray codes -> valid_len 0 -> cleared, no call
        assert [(len(c), fin) for c, fin in calls] == [(64, False)]
        assert calls[0][0][:2] == [0, 2047]


# ---------------- projection order + feedback: assets_manager.rs:383-399, engine.rs:622-631 ----------------
def test_project_order_on_vivian(tiny_model, oracle, vivian):
    import ggml_ref as G
    path = os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf")
    _, t = G.read_gguf(path)
    W = np.array(t["proj.weight"][2]).view(np.float32).reshape(-1, 2048)
    b = np.array(t["proj.bias"][2]).view(np.float32)
    oa = oracle.Assets(path)
    got = oa.project(vivian)[: b.size]
    for o in (0, 1, 17, b.size - 1):                       # bias first, then ascending i, separate mul and add, f32
        s = np.float32(b[o])
        for i in range(2048):
            s = np.float32(s + np.float32(vivian[i] * W[o, i]))
        assert got[o] == s
    oa.close()


def test_voice_file_fixture_shape():
    root = os.path.dirname(os.path.abspath(__file__))
    for name in ("vivian", "serena"):
        v = json.load(open(os.path.join(root, "golden", "speakers", name + ".json")))
        assert len(v["spk_emb"]) == 2048 and v["name"] == name            # voice_file.rs alias spk_emb; spk_id is dropped
