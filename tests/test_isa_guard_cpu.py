"""Build-time guard on the gfx950 ISA of every HIP source (cross-compiled here, no GPU needed).

The arithmetic spec (include/q3tts_spec.h) rounds an f32 result and THEN converts it to f16 where a value is stored as f16 (K/V cache,
activation scales).  The AMDGPU backend may fold `fptrunc(fma(a, b, c))` into v_fma_mixlo_f16 / v_fma_mixhi_f16, which rounds the exact
a*b+c once to f16 -- one f16 ulp away from the spec on exact ties (seen on MI355X as a K-cache mismatch in ~1 of 8000 elements).
kdev.h's f2h() makes its operand opaque so the fold cannot happen; this test fails if any kernel ever contains the fused forms again."""
import glob
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qwen3-tts-rust_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["-mllvm", "-amdgpu-kernarg-preload-count=16", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-S", "-x", "hip"]


def _asm(src, out_dir):
    out = os.path.join(out_dir, os.path.basename(src) + ".s")
    subprocess.run([HIPCC] + FLAGS + [src, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_single_rounding_f16_fma_in_any_kernel(tmp_path):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert srcs
    with ThreadPoolExecutor(4) as ex:
        texts = list(ex.map(lambda s: _asm(s, str(tmp_path)), srcs))
    for src, text in zip(srcs, texts):
        bad = re.findall(r"v_(?:fma|mad)_mix(?:lo|hi)_f16[^\n]*", text)
        assert not bad, "%s: %d f16-result mixed fma instructions, e.g. %s" % (os.path.basename(src), len(bad), bad[0])
        assert "v_cvt_f16_f32" in text or "f2h" not in open(src).read()   # the two-step rounding is what is there instead
    # the float-weight matmul must be on the K = 1 f32 matrix instructions (exact fma chains), not on a K > 1 form that sums products first
    k = texts[[os.path.basename(s) for s in srcs].index("kernels.hip")]
    assert "v_mfma_f32_16x16x1_4b_f32" in k and "v_mfma_f32_32x32x1_2b_f32" in k
    # ... with one exception: k_gemm_q8_tile1 builds its per-block scale tile d_x * d_w with the K = 2 form whose second k is fed zeros (one exact
    # product of two f16 values per output, + 0 * 0): allowed there and nowhere else
    funcs = re.split(r"\n(?=_Z\w+:)", k)
    for fn in funcs:
        if re.search(r"v_mfma_f32_(?:32x32x2|16x16x4)_f32", fn):
            assert fn.startswith("_ZN2q315k_gemm_q8_tile1"), fn.split(":")[0]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_q8_kernels_use_no_scratch(tmp_path):
    """The batched int8 GEMMs sit at their 128-register budget (two 512-thread workgroups per CU); one spilled register gives the kernel a
    private segment, and a kernel with scratch was measured ~2 us slower PER LAUNCH on MI355X (430 GEMM launches per frame).  Any edit that
    tips the allocation over must fail here, not in the next bench."""
    text = _asm(os.path.join(CSRC, "kernels.hip"), str(tmp_path))
    meta = text[text.index("amdhsa.kernels"):]
    seen = 0
    for blk in meta.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "k_gemm_q8_mfma" in name or "k_gemm_q8_tile1" in name or "k_gemm_q8_tok" in name or "k_gemv_q8I" in name:
            seen += 1
            assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0, name
            assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 128 or "k_gemm_q8_t" in name and "tok" in name or "k_gemv_q8I" in name, name
    assert seen >= 5
