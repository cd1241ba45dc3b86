"""Pins the building blocks of the codec oracle against the installed `transformers` Code2Wav modules (the public analogue of the
reference's qwen3_tts_decoder.onnx): SnakeBeta, the causal conv, the decoder residual unit (dilations 1/3/9) and the
ConvNeXt upsample block must compute what oracle/q3o_codec.c computes.  The oracle's full stack is tied to the same formulas by
test_codec_mel_oracle.py::test_codec_streaming_equals_full_and_matches_torch; the only deliberate difference is the transposed
convolution's trimming (the analogue trims k-s samples on both sides; the streamable form keeps the first T*s outputs), which is
checked here as well so the difference is exactly that shift."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

M = pytest.importorskip("transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe")


def _snake(x, a, b):  # oracle/q3o_codec.c snake: x + 1/(exp(beta)+1e-9) * sin^2(x*exp(alpha))
    return x + (1.0 / (torch.exp(b) + 1e-9))[None, :, None] * torch.sin(x * torch.exp(a)[None, :, None]) ** 2


def _causal(x, w, b, dil=1, groups=1):  # left pad (k-1)*dil, no look-ahead
    return F.conv1d(F.pad(x, ((w.shape[-1] - 1) * dil, 0)), w, b, dilation=dil, groups=groups)


def test_snake_residual_unit_and_convnext_match_transformers():
    torch.manual_seed(0)
    C, T = 32, 50
    x = torch.randn(1, C, T)
    with torch.no_grad():
        sb = M.Qwen3OmniMoeSnakeBeta(C)
        sb.alpha.copy_(torch.randn(C) * 0.3); sb.beta.copy_(torch.randn(C) * 0.3)
        assert torch.allclose(sb(x), _snake(x, sb.alpha, sb.beta), atol=1e-6)
        for dil in (1, 3, 9):
            ru = M.Qwen3OmniMoeCode2WavDecoderResidualUnit(C, dil)
            for p in ru.parameters():
                p.copy_(torch.randn_like(p) * 0.2)
            y = _causal(_snake(x, ru.act1.alpha, ru.act1.beta), ru.conv1.conv.weight, ru.conv1.conv.bias, dil)
            y = _causal(_snake(y, ru.act2.alpha, ru.act2.beta), ru.conv2.conv.weight, ru.conv2.conv.bias)
            assert torch.allclose(ru(x), x + y, atol=2e-5), dil
        cn = M.Qwen3OmniMoeConvNeXtBlock(C)
        for p in cn.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
        d = _causal(x, cn.dwconv.conv.weight, cn.dwconv.conv.bias, groups=C)
        d = F.layer_norm(d.transpose(1, 2), (C,), cn.norm.weight, cn.norm.bias, 1e-6)
        d = F.gelu(d @ cn.pwconv1.weight.T + cn.pwconv1.bias) @ cn.pwconv2.weight.T + cn.pwconv2.bias
        assert torch.allclose(cn(x), x + (cn.gamma * d).transpose(1, 2), atol=2e-5)


def test_transposed_conv_differs_from_analogue_only_by_the_trim():
    torch.manual_seed(1)
    cin, cout, s, T = 16, 8, 4, 12
    tc = M.Qwen3OmniMoeCausalTransConvNet(cin, cout, 2 * s, s)
    x = torch.randn(1, cin, T)
    with torch.no_grad():
        full = F.conv_transpose1d(x, tc.conv.weight, tc.conv.bias, stride=s)     # length (T+1)*s
        ours = full[..., : T * s]                                                # streamable: first T*s outputs (oracle, HIP decoder)
        theirs = tc(x)                                                           # analogue: drops k-s = s samples on both sides
        assert theirs.shape[-1] == (T - 1) * s
        assert torch.equal(theirs, ours[..., s:])                                # same samples, shifted by one stride
