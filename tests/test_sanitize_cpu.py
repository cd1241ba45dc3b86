"""Host-side logic under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_under_asan_ubsan(tiny_model, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_logic_asan")
    src = [os.path.join(ROOT, "tests", "sanitize", "host_logic_asan.cpp"),
           os.path.join(ROOT, "qwen3-tts-rust_amd", "csrc", "host_logic.cpp"), os.path.join(ROOT, "qwen3-tts-rust_amd", "csrc", "gguf.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe] + src
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    assert r.stdout.startswith("ok rows=")
