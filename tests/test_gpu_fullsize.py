"""Full-size ("Q3TTS-1.7B-synth", BASELINE.json configs[1..3] shapes) checks on the GPU: a short oracle comparison plus
size-independent properties (determinism, batch invariance, code ranges, chunk-count bookkeeping)."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full_model(synth_tool):
    out = os.environ.get("Q3_BENCH_MODEL", "/tmp/q3tts_synth_full")
    marker = os.path.join(out, ".complete_q8_0")
    if not os.path.exists(marker):
        subprocess.check_call([synth_tool, "--out", out, "--preset", "full", "--quant", "q8_0", "--seed", "1234"])
        open(marker, "w").write("ok")
    return out


def test_fullsize_parity_and_properties(gpu, oracle, full_model, vivian):
    ge = gpu.Engine(full_model, "q8_0", max_batch=2, max_steps=64, load_codec=True)
    rng = np.random.default_rng(42)
    prompt = ge.assets.build_core(rng.integers(0, 4000, 32).astype(np.int32), lang_id=2055, spk_emb=vivian)
    assert prompt.shape == (43, 2048)
    r = ge.generate_batch([prompt], max_steps=24, mask_eos=True, want_pcm=True)[0]
    assert r["codes"].shape == (24, 16) and r["codes"][:, 0].max() < 2160 and r["codes"][:, 1:].max() < 2048 and r["codes"].min() >= 0
    assert r["pcm"].size == 24 * 1920 and np.isfinite(r["pcm"]).all() and np.abs(r["pcm"]).max() <= 1.0
    # oracle on the host for the first frames (~0.3 s/frame + 43-token prefill)
    oe = oracle.Engine(os.path.join(full_model, "gguf_q8_0"), None, 8)
    oc, _ = oe.generate(prompt, max_steps=6, mask_eos=True)
    oe.close()
    assert np.array_equal(oc, r["codes"][:6])
    # determinism + batch invariance at full size
    r2 = ge.generate_batch([prompt, prompt], max_steps=24, mask_eos=True)
    assert np.array_equal(r2[0]["codes"], r["codes"]) and np.array_equal(r2[1]["codes"], r["codes"])
    # streaming codec == the codec oracle on the generated codes (first 5 frames: ~1 s of CPU)
    oc2 = oracle.Codec(os.path.join(full_model, "onnx", "q3tts_codec.gguf")); oc2.reset()
    ref = oc2.decode(np.clip(r["codes"][:5], 0, 2047)).copy(); oc2.close()
    assert np.sqrt(np.mean((ref - r["pcm"][: ref.size]) ** 2)) < 1e-4
    ge.close()
