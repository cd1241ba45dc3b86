"""Generates the ORACLE-FREE codec-decoder fixture: run in the build container (needs `transformers`), commit the two outputs.

    python tests/golden/make_code2wav_fixture.py

Writes tests/golden/code2wav_tf.gguf (a small `Qwen3OmniMoeCode2Wav` -- the public analogue of the reference's qwen3_tts_decoder.onnx, SURVEY 8c,
/root/reference/src/models/onnx.rs:342-458 -- with seeded random weights, stored under this repo's codec tensor names) and
tests/golden/code2wav_tf_expected.npz (codes [T][16] and the waveform the locally installed `transformers` model computes for them, float32).
tests/test_gpu_parity.py::test_codec_vs_transformers_code2wav_fixture runs csrc/codec.hip on that file and compares; nothing from oracle/ takes
part, so an error shared by the oracle and the kernels cannot hide there (tests/test_golden_cpu.py checks the oracle against the same numbers).

How the analogue maps onto this repo's decoder (the two differences are representational, the waveform is the same):
  * code embedding: the analogue averages ONE table's rows (code + q * 2048) over the 16 quantizers and has no input convolution; here the 16
    codebooks are summed and a causal k = 3 `pre_conv` follows.  Fixture: codebook_q = table rows of quantizer q / 16, pre_conv = identity (last tap).
  * transposed convolutions of the 4 decoder blocks (kernel 2r, stride r): the analogue drops r samples on BOTH sides, the streamable form keeps the
    first T*r outputs, i.e. analogue[n] = ours[n + r] per block (tests/test_codec_blocks_vs_transformers.py pins exactly that).  Through the
    fixture's rates (8, 5) the shifts add up to 8*5 + 5 = 45 samples (555 for the full-size (8, 5, 4, 3)); causal convolutions are shift-equivariant
    away from the left edge, so expected[n] == ours[n + 45] once the receptive field lies inside both signals (the test skips the first frames).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from gguf_writer import write_gguf  # noqa: E402

CB, H, NH, FFN, NL, WIN, DEC = 256, 32, 2, 64, 2, 8, 64   # small on purpose: the committed GGUF stays ~0.7 MB
UP, RATES, T = (2, 2), (8, 5), 24
SHIFT = 8 * 5 + 5                                          # sum over decoder blocks of rate_i * prod(later rates)


def main():
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    torch.manual_seed(4321)
    cfg = Qwen3OmniMoeCode2WavConfig(codebook_size=CB, hidden_size=H, num_attention_heads=NH, num_key_value_heads=NH, sliding_window=WIN,
                                     intermediate_size=FFN, num_hidden_layers=NL, num_quantizers=16, upsample_rates=RATES, upsampling_ratios=UP,
                                     decoder_dim=DEC, rms_norm_eps=1e-5, max_position_embeddings=512,
                                     rope_parameters={"rope_type": "default", "rope_theta": 10000.0})
    cfg._attn_implementation = "eager"
    m = Qwen3OmniMoeCode2Wav(cfg).eval().float()
    with torch.no_grad():
        for name, p in m.named_parameters():   # lively but bounded values everywhere (layer scales / gammas / snake parameters included)
            if name.endswith("alpha") or name.endswith("beta"):
                p.copy_(torch.randn_like(p) * 0.1)
            elif "layer_scale" in name or name.endswith("gamma") or ".scale" in name:
                p.copy_(0.1 + torch.randn_like(p) * 0.02)
            elif p.ndim == 1 and ("norm" in name):
                p.copy_(1.0 + torch.randn_like(p) * 0.1)
            elif p.ndim == 1:
                p.copy_(torch.randn_like(p) * 0.02)
            elif name.startswith("code_embedding"):
                p.copy_(torch.randn_like(p) * 1.0)
            else:
                fan_in = p[0].numel() if "convt" not in name and "ConvTranspose" not in name else p.shape[0] * p.shape[-1]
                p.copy_(torch.randn_like(p) / np.sqrt(max(fan_in, 1)))
        m.decoder[-1].conv.weight.mul_(0.06)   # keep the waveform inside the final clamp (a clipped sample compares equal whatever produced it)
    sd = {k: v.detach().float().numpy() for k, v in m.state_dict().items()}
    t = {}
    emb = sd["code_embedding.weight"]
    for q in range(16):
        t["codec.codebook.%d" % q] = emb[q * CB:(q + 1) * CB] / np.float32(16.0)
    pre = np.zeros((H, H, 3), np.float32)
    pre[np.arange(H), np.arange(H), 2] = 1.0
    t["codec.pre_conv.weight"] = pre
    t["codec.pre_conv.bias"] = np.zeros(H, np.float32)
    for l in range(NL):
        s, d = "pre_transformer.layers.%d." % l, "codec.tf.%d." % l
        t[d + "attn_norm"] = sd[s + "input_layernorm.weight"]
        t[d + "wq"] = sd[s + "self_attn.q_proj.weight"]; t[d + "wk"] = sd[s + "self_attn.k_proj.weight"]
        t[d + "wv"] = sd[s + "self_attn.v_proj.weight"]; t[d + "wo"] = sd[s + "self_attn.o_proj.weight"]
        t[d + "ls_attn"] = sd[s + "self_attn_layer_scale.scale"]
        t[d + "ffn_norm"] = sd[s + "post_attention_layernorm.weight"]
        t[d + "w_gate"] = sd[s + "mlp.gate_proj.weight"]; t[d + "w_up"] = sd[s + "mlp.up_proj.weight"]; t[d + "w_down"] = sd[s + "mlp.down_proj.weight"]
        t[d + "ls_ffn"] = sd[s + "mlp_layer_scale.scale"]
    t["codec.tf.norm"] = sd["pre_transformer.norm.weight"]
    for i in range(len(UP)):
        s, d = "upsample.%d." % i, "codec.up.%d." % i
        t[d + "convt.weight"] = sd[s + "0.conv.weight"]; t[d + "convt.bias"] = sd[s + "0.conv.bias"]
        t[d + "dw.weight"] = sd[s + "1.dwconv.conv.weight"].reshape(H, 7); t[d + "dw.bias"] = sd[s + "1.dwconv.conv.bias"]
        t[d + "ln.weight"] = sd[s + "1.norm.weight"]; t[d + "ln.bias"] = sd[s + "1.norm.bias"]
        t[d + "pw1.weight"] = sd[s + "1.pwconv1.weight"]; t[d + "pw1.bias"] = sd[s + "1.pwconv1.bias"]
        t[d + "pw2.weight"] = sd[s + "1.pwconv2.weight"]; t[d + "pw2.bias"] = sd[s + "1.pwconv2.bias"]
        t[d + "gamma"] = sd[s + "1.gamma"]
    t["codec.dec.conv_in.weight"] = sd["decoder.0.conv.weight"]; t["codec.dec.conv_in.bias"] = sd["decoder.0.conv.bias"]
    for b in range(len(RATES)):
        s, d = "decoder.%d.block." % (b + 1), "codec.dec.%d." % b
        t[d + "snake.alpha"] = sd[s + "0.alpha"]; t[d + "snake.beta"] = sd[s + "0.beta"]
        t[d + "convt.weight"] = sd[s + "1.conv.weight"]; t[d + "convt.bias"] = sd[s + "1.conv.bias"]
        for u in range(3):
            r, q = s + "%d." % (2 + u), d + "ru.%d." % u
            t[q + "snake1.alpha"] = sd[r + "act1.alpha"]; t[q + "snake1.beta"] = sd[r + "act1.beta"]
            t[q + "conv1.weight"] = sd[r + "conv1.conv.weight"]; t[q + "conv1.bias"] = sd[r + "conv1.conv.bias"]
            t[q + "snake2.alpha"] = sd[r + "act2.alpha"]; t[q + "snake2.beta"] = sd[r + "act2.beta"]
            t[q + "conv2.weight"] = sd[r + "conv2.conv.weight"]; t[q + "conv2.bias"] = sd[r + "conv2.conv.bias"]
    n = len(RATES)
    t["codec.dec.snake_out.alpha"] = sd["decoder.%d.alpha" % (n + 1)]; t["codec.dec.snake_out.beta"] = sd["decoder.%d.beta" % (n + 1)]
    t["codec.dec.conv_out.weight"] = sd["decoder.%d.conv.weight" % (n + 2)]; t["codec.dec.conv_out.bias"] = sd["decoder.%d.conv.bias" % (n + 2)]
    kv = {"general.architecture": "q3tts-codec", "codec.n_codebooks": 16, "codec.codebook_size": CB, "codec.codebook_dim": H, "codec.hidden": H,
          "codec.n_layers": NL, "codec.n_heads": NH, "codec.head_dim": H // NH, "codec.ffn": FFN, "codec.window": WIN, "codec.dec_dim": DEC,
          "codec.n_up": len(UP), "codec.n_dec": len(RATES), "codec.rope_base": 10000.0, "codec.eps": 1e-5}
    for i, f in enumerate(UP):
        kv["codec.up_ratio.%d" % i] = f
    for i, r in enumerate(RATES):
        kv["codec.dec_rate.%d" % i] = r
    write_gguf(os.path.join(HERE, "code2wav_tf.gguf"), kv, t)
    codes = torch.from_numpy(np.random.default_rng(99).integers(0, CB, (T, 16)))
    with torch.no_grad():
        wav = m(codes.T.unsqueeze(0))[0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "code2wav_tf_expected.npz"), codes=codes.numpy().astype(np.int64), wav=wav.astype(np.float32),
                        meta=np.array([H, NH, FFN, NL, WIN, DEC, T, SHIFT, CB], np.int32))
    print("wrote fixture: %d frames -> %d samples (ours: %d), std %.3f, |wav| max %.3f, clipped %.1f %%" %
          (T, wav.size, T * int(np.prod(UP + RATES)), float(wav.std()), float(np.abs(wav).max()), 100.0 * float(np.mean(np.abs(wav) >= 1.0))))


if __name__ == "__main__":
    main()
