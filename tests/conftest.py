import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _run(cmd):
    subprocess.check_call(cmd)


@pytest.fixture(scope="session")
def oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    import q3oracle
    q3oracle.lib()
    return q3oracle


@pytest.fixture(scope="session")
def synth_tool():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    return os.path.join(ROOT, "tools", "q3synth")


@pytest.fixture(scope="session")
def tiny_model(synth_tool):
    """Seeded tiny 'Q3TTS-synth' model (talker d=2048 L=2, predictor d=256 L=2, small codec), Q8_0 + Q5_K_M + f32."""
    out = os.environ.get("Q3_TINY_MODEL", "/tmp/q3tts_pytest_tiny")
    if not os.path.exists(os.path.join(out, ".complete3")):
        _run([synth_tool, "--out", out, "--preset", "tiny", "--quant", "q8_0"])
        _run([synth_tool, "--out", out, "--preset", "tiny", "--quant", "q5_k_m", "--what", "3"])
        _run([synth_tool, "--out", out, "--preset", "tiny", "--quant", "bf16", "--what", "3"])
        open(os.path.join(out, ".complete3"), "w").write("ok")
    for sub in ("gguf_q5_k_m", "gguf_bf16"):   # the assets file is F32 in every quantisation directory (assets_manager.rs:163-167)
        dst = os.path.join(out, sub, "qwen3_assets.gguf")
        if not os.path.exists(dst):
            os.symlink(os.path.join(out, "gguf_q8_0", "qwen3_assets.gguf"), dst)
    return out


@pytest.fixture(scope="session")
def vivian():
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "speakers", "vivian.json")))
    return np.array(v["spk_emb"], np.float32)


@pytest.fixture(scope="session")
def q3():
    """The product library through its C ABI.  Missing library => the test fails (no fallback)."""
    import q3tts
    q3tts.lib()
    return q3tts


@pytest.fixture(scope="session")
def gpu(q3):
    if q3.device_count() < 1:
        pytest.fail("gpu-marked test started without a HIP device")
    return q3
