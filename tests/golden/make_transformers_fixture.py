"""Generates the ORACLE-FREE transformer fixture: run in the build container (needs `transformers`), commit the two outputs.

    python tests/golden/make_transformers_fixture.py

Writes tests/golden/qwen3_tf_f16.gguf (+ _q8_0 and _q5_k_m copies) (a 2-layer Qwen3 decoder with f16-representable random weights, llama.cpp tensor names)
and tests/golden/qwen3_tf_expected.npz (input embeddings, final-norm hidden states and logits computed by the locally installed
`transformers` Qwen3Model in float32, eager attention).  The -m gpu test test_gpu_parity.py::test_tf_eval_vs_transformers_fixture
compares q3tts_tf_eval (the HIP path, nothing from oracle/) with these numbers; tests/test_golden_cpu.py compares the oracle with
them as well, so the two implementations are pinned to a third, independent one rather than only to each other.
Architecture = what llama.cpp's qwen3 graph computes [EXT]: pre-norm blocks, per-head q/k RMSNorm, NeoX RoPE (theta 1e6), causal
GQA attention, SwiGLU MLP, final RMSNorm, untied output matrix.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from gguf_writer import write_gguf  # noqa: E402

D, L, H, HKV, FF, V, N = 256, 2, 2, 1, 256, 96, 24


def main():
    from transformers import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    torch.manual_seed(1234)
    cfg = Qwen3Config(vocab_size=64, hidden_size=D, intermediate_size=FF, num_hidden_layers=L, num_attention_heads=H,
                      num_key_value_heads=HKV, head_dim=128, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=64,
                      attention_bias=False, tie_word_embeddings=False)
    cfg.rope_parameters = {"rope_type": "default", "rope_theta": 1000000.0}
    cfg._attn_implementation = "eager"
    m = Qwen3Model(cfg).eval().float()
    with torch.no_grad():
        for p in m.parameters():  # f16-representable values so the GGUF (F16 matrices) and the torch model hold identical weights
            w = torch.randn_like(p) * (0.06 if p.ndim == 2 else 0.1) + (1.0 if p.ndim == 1 else 0.0)
            p.copy_(w.half().float())
    out_w = (torch.randn(V, D) * 0.06).half().float()
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    t = {"output_norm.weight": sd["norm.weight"], "output.weight": out_w.numpy().astype(np.float16)}
    for l in range(L):
        p = "layers.%d." % l
        for src, dst in (("input_layernorm", "attn_norm"), ("self_attn.q_proj", "attn_q"), ("self_attn.k_proj", "attn_k"),
                         ("self_attn.v_proj", "attn_v"), ("self_attn.o_proj", "attn_output"), ("self_attn.q_norm", "attn_q_norm"),
                         ("self_attn.k_norm", "attn_k_norm"), ("post_attention_layernorm", "ffn_norm"), ("mlp.gate_proj", "ffn_gate"),
                         ("mlp.up_proj", "ffn_up"), ("mlp.down_proj", "ffn_down")):
            a = sd[p + src + ".weight"]
            t["blk.%d.%s.weight" % (l, dst)] = a.astype(np.float16) if a.ndim == 2 else a  # norms stay F32 (as llama.cpp stores them)
    kv = {"general.architecture": "qwen3", "qwen3.embedding_length": D, "qwen3.block_count": L, "qwen3.attention.head_count": H,
          "qwen3.attention.head_count_kv": HKV, "qwen3.attention.key_length": 128, "qwen3.feed_forward_length": FF,
          "qwen3.attention.layer_norm_rms_epsilon": 1e-6, "qwen3.rope.freq_base": 1000000.0}
    write_gguf(os.path.join(HERE, "qwen3_tf_f16.gguf"), kv, t)
    # the same model with every matrix quantised to ggml Q8_0 ({f16 d; i8 q[32]} blocks, d = amax/127, q = rint(w/d)): the engine's
    # int8 path (activation quantisation, integer block dots, per-block scale chain) is then checked against the SAME transformers
    # numbers, at a tolerance that covers the 8-bit weight + activation quantisation noise (measured 3e-2 of max(1, |ref|); test bar 6e-2)
    tq = {}
    for name, a in t.items():
        if getattr(a, "ndim", 0) == 2:
            w = a.astype(np.float32)
            n, k = w.shape
            wb = w.reshape(n, k // 32, 32)
            d = (np.abs(wb).max(-1) / 127).astype(np.float32)
            idv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0).astype(np.float32)
            q = np.rint(wb * idv[..., None]).astype(np.int8)
            raw = np.zeros((n, k // 32, 34), np.uint8)
            raw[..., :2] = d.astype(np.float16).view(np.uint8).reshape(n, k // 32, 2)
            raw[..., 2:] = q.view(np.uint8)
            tq[name] = (8, (n, k), raw.tobytes())
        else:
            tq[name] = a
    write_gguf(os.path.join(HERE, "qwen3_tf_q8_0.gguf"), kv, tq)
    x = torch.randn(1, N, D) * 0.5
    with torch.no_grad():
        h = m(inputs_embeds=x).last_hidden_state[0]
        lg = h @ out_w.T
    # the same model as a Q5_K_M file (Q5_K for q / k / o / gate / up, Q6_K for v / down / output -- the mixture the reference's default
    # quantisation ships, /root/reference/src/tts/engine.rs:91-95): the K-quant kernels (packed planes, Q8 activations, k_gemm_kq_mfma for the
    # 24-token prefill) against `transformers` run on the DEQUANTISED weights, so what is left between the two is the activation quantisation
    import ggml_ref as G
    t5, deq = {}, {}
    for name, a in t.items():
        if getattr(a, "ndim", 0) == 2:
            w = a.astype(np.float32)
            six = any(name.endswith(sfx) for sfx in ("attn_v.weight", "ffn_down.weight")) or name == "output.weight"
            raw = (G.enc_q6_k if six else G.enc_q5_k)(w)
            t5[name] = (14 if six else 13, w.shape, raw.tobytes())
            deq[name] = (G.deq_q6_k if six else G.deq_q5_k)(raw.reshape(-1), w.shape[1])
        else:
            t5[name] = a
    write_gguf(os.path.join(HERE, "qwen3_tf_q5_k_m.gguf"), kv, t5)
    with torch.no_grad():
        for l in range(L):
            p = "layers.%d." % l
            for src, dst in (("self_attn.q_proj", "attn_q"), ("self_attn.k_proj", "attn_k"), ("self_attn.v_proj", "attn_v"), ("self_attn.o_proj", "attn_output"),
                             ("mlp.gate_proj", "ffn_gate"), ("mlp.up_proj", "ffn_up"), ("mlp.down_proj", "ffn_down")):
                dict(m.named_parameters())[p + src + ".weight"].copy_(torch.from_numpy(deq["blk.%d.%s.weight" % (l, dst)]))
        h5 = m(inputs_embeds=x).last_hidden_state[0]
        lg5 = h5 @ torch.from_numpy(deq["output.weight"]).T
    np.savez(os.path.join(HERE, "qwen3_tf_expected.npz"), x=x[0].numpy(), hidden=h.numpy(), logits=lg.numpy(),
             hidden_q5=h5.numpy(), logits_q5=lg5.numpy(), meta=np.array([D, L, H, HKV, FF, V, N], np.int32))
    print("Q5_K_M copy: |hidden - hidden_q5| max %.3f (weight quantisation alone)" % float((h - h5).abs().max()))
    print("wrote fixture: %d tokens, |hidden| max %.3f, |logits| max %.3f" % (N, float(h.abs().max()), float(lg.abs().max())))


if __name__ == "__main__":
    main()
