"""The oracle must keep reproducing the committed golden vectors (tests/golden/oracle_tiny_v1.npz, made by
tests/golden/make_golden.py from the seeded tiny model)."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "oracle_tiny_v1.npz")


def test_oracle_reproduces_golden(tiny_model, oracle, vivian):
    g = np.load(GOLD)
    eng = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), os.path.join(tiny_model, "onnx", "q3tts_codec.gguf"), 4)
    prompt = eng.assets.build_core(np.arange(100, 108, dtype=np.int32), lang_id=2055, spk_emb=vivian)
    assert np.allclose([prompt.astype(np.float64).sum(), np.abs(prompt).astype(np.float64).sum()], g["prompt_checksum"], rtol=0, atol=0)
    codes, pcm = eng.generate(prompt, max_steps=12, temperature=0.0, seed=42, mask_eos=True, want_pcm=True)
    assert np.array_equal(codes, g["greedy_codes"])
    assert np.array_equal(pcm[:4096], g["greedy_pcm_head"]) and pcm.size == int(g["greedy_pcm_stats"][0])
    codes_s, _ = eng.generate(prompt, max_steps=12, temperature=0.7, top_k=40, top_p=0.9, seed=42, mask_eos=True)
    assert np.array_equal(codes_s, g["sampled_codes"])
    assert not np.array_equal(codes_s, codes)
    assert np.array_equal(eng.assets.project(vivian)[:256], g["project_vivian"])
    eng.close()
    m = oracle.Model(os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf"), 64)
    for t in range(3):
        h, l = m.eval(prompt[t], [t, t, t, 0], 2048, 0, 2160)
        assert np.array_equal(h, g["talker_hidden3"][t]) and np.array_equal(l, g["talker_logits3"][t])
    m.close()
    lg = np.sin(np.arange(2160, dtype=np.float32) * 0.37) * 3
    kat = [oracle.sample(lg, 0, 2160, temperature=0.7, top_k=40, top_p=0.9, seed=s)[0] for s in range(16)]
    assert kat == g["sampler_kat"].tolist()


def test_q5_k_m_and_bf16_oracle_paths_run(tiny_model, oracle, vivian):
    """configs[0] (Q5_K_M, CPU) and the bf16 path of config 5 run through the oracle; different quantisations of the same
    seeded weights give different but equally long greedy runs."""
    runs = {}
    for sub in ("gguf_q8_0", "gguf_q5_k_m", "gguf_bf16"):
        qdir = os.path.join(tiny_model, sub)
        if not os.path.exists(os.path.join(qdir, "qwen3_assets.gguf")):
            os.symlink(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"), os.path.join(qdir, "qwen3_assets.gguf"))
        eng = oracle.Engine(qdir, None, 4)
        prompt = eng.assets.build_core(np.arange(100, 108, dtype=np.int32), lang_id=2055, spk_emb=vivian)
        runs[sub], _ = eng.generate(prompt, max_steps=4, temperature=0.0, seed=42, mask_eos=True)
        eng.close()
        assert runs[sub].shape == (4, 16) and runs[sub].min() >= 0 and runs[sub][:, 1:].max() < 2048 and runs[sub][:, 0].max() < 2160


TF_GGUF = os.path.join(ROOT, "tests", "golden", "qwen3_tf_f16.gguf")
TF_EXP = os.path.join(ROOT, "tests", "golden", "qwen3_tf_expected.npz")
TF_TOL = 2e-3   # f32 torch (f32 softmax/KV) vs spec arithmetic with an f16 KV cache; relative to max(1, |ref|_inf)


def test_oracle_vs_transformers_fixture(oracle):
    """The committed fixture was computed by `transformers` Qwen3Model (tests/golden/make_transformers_fixture.py), not by oracle/:
    it pins the oracle's transformer math to an independent implementation; the same fixture pins the HIP path on the GPU box."""
    g = np.load(TF_EXP)
    D, L, H, HKV, FF, V, N = [int(v) for v in g["meta"]]
    om = oracle.Model(TF_GGUF, 64)
    for i in range(N):
        h, lg = om.eval(g["x"][i], [i, i, i, 0], D, 0, V)
        assert np.abs(h - g["hidden"][i]).max() < TF_TOL * max(1.0, np.abs(g["hidden"][i]).max()), i
        assert np.abs(lg - g["logits"][i]).max() < TF_TOL * max(1.0, np.abs(g["logits"][i]).max()), i
    om.close()


def test_oracle_predictor_loop_vs_transformers_fixture(oracle):
    """The code-predictor loop (/root/reference/src/tts/engine.rs:596-640: two prompt rows, then 15 greedy passes, pass i reading slice i of the stacked
    output matrix and feeding its code back through codebook table i) against what `transformers`' own `generate()` emits for its public implementation of
    that loop (tests/golden/make_predictor_fixture.py): every pass's logits within TF_TOL and the 15 codes equal (smallest argmax margin of the fixture 0.15)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "predictor_tf_expected.npz"))
    D, L, H, HKV, FF, V, G, _ = [int(v) for v in g["meta"]]
    om = oracle.Model(os.path.join(ROOT, "tests", "golden", "predictor_tf_f16.gguf"), 64)
    om.eval(g["x"][0], [0, 0, 0, 0], D, 0, V)
    _, lg = om.eval(g["x"][1], [1, 1, 1, 0], D, 0, V)
    codes = []
    for i in range(G - 1):
        assert np.abs(lg - g["logits"][i]).max() < TF_TOL * max(1.0, np.abs(g["logits"][i]).max()), i
        codes.append(int(np.argmax(lg)))
        if i < G - 2:
            _, lg = om.eval(g["tables"][i][codes[-1]].astype(np.float32), [i + 2, i + 2, i + 2, 0], D, (i + 1) * V, (i + 2) * V)
    om.close()
    assert codes == g["codes"].tolist()


def test_oracle_q8_path_vs_transformers_fixture(oracle):
    """same numbers, the model quantised to Q8_0: the oracle's int8 path (block quantisation of activations, integer dots, scale chain)
    stays within the 8-bit quantisation noise of the float32 transformers result (measured 3e-2; a wrong scale, block order or sign
    shows up as O(1))."""
    g = np.load(TF_EXP)
    D, L, H, HKV, FF, V, N = [int(v) for v in g["meta"]]
    om = oracle.Model(os.path.join(ROOT, "tests", "golden", "qwen3_tf_q8_0.gguf"), 64)
    for i in range(N):
        h, lg = om.eval(g["x"][i], [i, i, i, 0], D, 0, V)
        assert np.abs(h - g["hidden"][i]).max() < 6e-2 * max(1.0, np.abs(g["hidden"][i]).max()), i
        assert np.abs(lg - g["logits"][i]).max() < 6e-2 * max(1.0, np.abs(g["logits"][i]).max()), i
    om.close()


def test_oracle_q5_k_m_path_vs_transformers_fixture(oracle):
    """the Q5_K_M copy (Q5_K q / k / o / gate / up, Q6_K v / down / output; encoded by tests/ggml_ref.py) against `transformers` run on the
    DEQUANTISED weights: what is left between the two is the activation quantisation of the K-quant block arithmetic (measured 2e-2; bar 6e-2)"""
    g = np.load(TF_EXP)
    D, L, H, HKV, FF, V, N = [int(v) for v in g["meta"]]
    om = oracle.Model(os.path.join(ROOT, "tests", "golden", "qwen3_tf_q5_k_m.gguf"), 64)
    worst = 0.0
    for i in range(N):
        h, lg = om.eval(g["x"][i], [i, i, i, 0], D, 0, V)
        worst = max(worst, np.abs(h - g["hidden_q5"][i]).max() / max(1.0, np.abs(g["hidden_q5"][i]).max()), np.abs(lg - g["logits_q5"][i]).max() / max(1.0, np.abs(g["logits_q5"][i]).max()))
    om.close()
    assert worst < 6e-2, worst


C2W_GGUF = os.path.join(ROOT, "tests", "golden", "code2wav_tf.gguf")
C2W_EXP = os.path.join(ROOT, "tests", "golden", "code2wav_tf_expected.npz")
C2W_SKIP_FRAMES = 6   # frames left out at the start: the two signals differ inside the receptive field of the left edge (see make_code2wav_fixture.py)


def code2wav_compare(ours, g):
    """expected[n] == ours[n + shift] away from the left edge; returns (rms, max abs) of the difference over the compared span"""
    shift, spf = int(g["meta"][7]), ours.size // g["codes"].shape[0]
    exp = g["wav"]
    n0 = C2W_SKIP_FRAMES * spf
    n1 = min(exp.size, ours.size - shift)
    assert n1 - n0 > 10 * spf
    d = ours[n0 + shift:n1 + shift].astype(np.float64) - exp[n0:n1]
    return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())


def test_oracle_codec_vs_transformers_code2wav_fixture(oracle):
    """The committed waveform was computed by `transformers` Qwen3OmniMoeCode2Wav (tests/golden/make_code2wav_fixture.py), not by oracle/: the
    whole decoder -- RVQ sum, sliding-window transformer with layer scales, ConvNeXt up-sampling, SnakeBeta / transposed-conv / residual-unit
    blocks, output conv and clamp -- streamed in 4-frame chunks like engine.rs:505-541, against an independent implementation."""
    g = np.load(C2W_EXP)
    c = oracle.Codec(C2W_GGUF)
    c.reset()
    codes = g["codes"]
    ours = np.concatenate([c.decode(codes[o:o + 4], o + 4 >= codes.shape[0]).copy() for o in range(0, codes.shape[0], 4)])
    c.close()
    assert ours.size == codes.shape[0] * c.spf
    rms, mx = code2wav_compare(ours, g)
    assert rms < 2e-6 and mx < 3e-5, (rms, mx)
    assert 0.05 < g["wav"].std() < 0.5 and np.abs(g["wav"]).max() < 1.0   # a live, unclipped signal
