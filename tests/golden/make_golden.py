"""Generates tests/golden/oracle_tiny_v1.npz from the CPU oracle on the seeded tiny model (tools/q3synth --preset tiny,
seed 1234).  The reference has no fixtures of its own for this path and cannot be built or run here,
so these vectors pin the ORACLE (against silent drift) and give the GPU tests an oracle-free expected value.
Run:  python tests/golden/make_golden.py   (needs oracle/libq3oracle.so and tools/q3synth)"""
import json
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import q3oracle as O


def main():
    out = "/tmp/q3tts_pytest_tiny"
    if not os.path.exists(os.path.join(out, "gguf_q8_0", "qwen3_tts_talker.gguf")):
        subprocess.check_call([os.path.join(ROOT, "tools", "q3synth"), "--out", out, "--preset", "tiny", "--quant", "q8_0"])
    spk = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "speakers", "vivian.json")))["spk_emb"], np.float32)
    eng = O.Engine(os.path.join(out, "gguf_q8_0"), os.path.join(out, "onnx", "q3tts_codec.gguf"), 4)
    text = np.arange(100, 108, dtype=np.int32)
    prompt = eng.assets.build_core(text, lang_id=2055, spk_emb=spk)
    g = {}
    g["prompt_checksum"] = np.array([prompt.astype(np.float64).sum(), np.abs(prompt).astype(np.float64).sum()])
    codes, pcm = eng.generate(prompt, max_steps=12, temperature=0.0, seed=42, mask_eos=True, want_pcm=True)
    g["greedy_codes"] = codes
    g["greedy_pcm_head"] = pcm[:4096]
    g["greedy_pcm_stats"] = np.array([pcm.size, float(np.sqrt(np.mean(pcm.astype(np.float64) ** 2)))])
    codes_s, _ = eng.generate(prompt, max_steps=12, temperature=0.7, top_k=40, top_p=0.9, seed=42, mask_eos=True)
    g["sampled_codes"] = codes_s
    m = O.Model(os.path.join(out, "gguf_q8_0", "qwen3_tts_talker.gguf"), 64)
    hs, ls = [], []
    for t in range(3):
        h, l = m.eval(prompt[t], [t, t, t, 0], 2048, 0, 2160)
        hs.append(h); ls.append(l)
    g["talker_hidden3"] = np.stack(hs); g["talker_logits3"] = np.stack(ls)
    m.close()
    g["project_vivian"] = eng.assets.project(spk)[:256]
    t = np.arange(6000) / 24000.0
    g["mel_chirp"] = O.mel((0.4 * np.sin(2 * np.pi * (300 + 2000 * t) * t)).astype(np.float32))
    lg = np.sin(np.arange(2160, dtype=np.float32) * 0.37) * 3
    g["sampler_kat"] = np.array([O.sample(lg, 0, 2160, temperature=0.7, top_k=40, top_p=0.9, seed=s)[0] for s in range(16)], np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_tiny_v1.npz"), **g)
    print({k: v.shape for k, v in g.items()})
    eng.close()


if __name__ == "__main__":
    main()
