"""Pins the oracle's transformer block math (RMSNorm, per-head q/k norm, NeoX RoPE, causal GQA attention, SwiGLU,
residuals, final norm, output matrix) against the locally installed `transformers` Qwen3 modules on random weights.
Weights are exported as F32 GGUF, so the oracle runs its float path (no activation quantisation); the only modelling
difference left is the f16 KV cache, hence the 2e-3 tolerance.  (this is the [EXT] analogue check.)"""
import os
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_oracle_matches_transformers_qwen3(tmp_path, oracle):
    from transformers import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    from gguf_writer import write_gguf
    torch.manual_seed(0)
    cfg = Qwen3Config(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=64,
                      attention_bias=False, tie_word_embeddings=False)
    cfg.rope_parameters = {"rope_type": "default", "rope_theta": 1000000.0}
    cfg._attn_implementation = "eager"
    m = Qwen3Model(cfg).eval().float()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.05 if p.ndim == 2 else 0.1) + (1.0 if p.ndim == 1 else 0.0))
    out_w = torch.randn(96, 256) * 0.05
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    t = {"output_norm.weight": sd["norm.weight"], "output.weight": out_w.numpy()}
    for l in range(2):
        p = "layers.%d." % l
        for src, dst in (("input_layernorm", "attn_norm"), ("self_attn.q_proj", "attn_q"), ("self_attn.k_proj", "attn_k"),
                         ("self_attn.v_proj", "attn_v"), ("self_attn.o_proj", "attn_output"), ("self_attn.q_norm", "attn_q_norm"),
                         ("self_attn.k_norm", "attn_k_norm"), ("post_attention_layernorm", "ffn_norm"), ("mlp.gate_proj", "ffn_gate"),
                         ("mlp.up_proj", "ffn_up"), ("mlp.down_proj", "ffn_down")):
            t["blk.%d.%s.weight" % (l, dst)] = sd[p + src + ".weight"]
    kv = {"general.architecture": "qwen3", "qwen3.embedding_length": 256, "qwen3.block_count": 2, "qwen3.attention.head_count": 2,
          "qwen3.attention.head_count_kv": 1, "qwen3.attention.key_length": 128, "qwen3.feed_forward_length": 512,
          "qwen3.attention.layer_norm_rms_epsilon": 1e-6, "qwen3.rope.freq_base": 1000000.0}
    path = str(tmp_path / "qwen3_f32.gguf")
    write_gguf(path, kv, t)
    n = 9
    x = torch.randn(1, n, 256) * 0.5
    with torch.no_grad():
        ref_h = m(inputs_embeds=x).last_hidden_state[0]
        ref_l = ref_h @ out_w.T
    om = oracle.Model(path, 64)
    for i in range(n):
        h, lg = om.eval(x[0, i].numpy(), [i, i, i, 0], 256, 0, 96)
        assert np.abs(h - ref_h[i].numpy()).max() < 2e-3 * max(1.0, np.abs(ref_h[i].numpy()).max()), i
        assert np.abs(lg - ref_l[i].numpy()).max() < 2e-3 * max(1.0, np.abs(ref_l[i].numpy()).max()), i
    om.close()
