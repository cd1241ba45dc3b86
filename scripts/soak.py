"""Long soak of the scheduler on the tiny model: thousands of mixed requests through 16 slots with the driver thread, random arrival
bursts, streaming reads; a sample is re-run solo afterwards and must be identical.  Not part of the test suite (minutes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import json, subprocess
import q3tts as Q
out = "/tmp/q3tts_soak_tiny"
tool = os.path.join(ROOT, "tools", "q3synth")
if not os.path.exists(tool):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
if not os.path.exists(out + "/onnx/q3tts_codec.gguf"):
    subprocess.check_call([tool, "--out", out, "--preset", "tiny", "--quant", "q8_0", "--seed", "1234"])
vivian = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "speakers", "vivian.json")))["spk_emb"], np.float32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ge = Q.Engine(out, "q8_0", max_batch=16, max_steps=16, load_codec=True)
rng = np.random.default_rng(99)
specs = []
for i in range(N):
    prompt = ge.assets.build_core(rng.integers(0, 4000, int(rng.integers(1, 60))).astype(np.int32), lang_id=2055, spk_emb=vivian)
    specs.append(dict(prompt=prompt, max_steps=int(rng.integers(0, 17)), temperature=float(rng.choice([0.0, 0.7])), top_k=int(rng.choice([0, 40])),
                      top_p=float(rng.choice([0.9, 1.0])), seed=int(rng.integers(0, 1 << 30)), mask_eos=bool(rng.integers(0, 2)), want_pcm=bool(i % 2)))
t0 = time.time()
ge.sched_start()
ids, results, inflight = [], {}, []
for i, sp in enumerate(specs):
    ids.append(ge.submit(**sp)); inflight.append(i)
    if rng.random() < 0.05:
        time.sleep(float(rng.random()) * 0.004)
    while len(inflight) > 200:                      # keep the queue bounded like a real server would
        j = inflight.pop(0)
        assert ge.wait(ids[j], 120000.0)
        results[j] = ge.result(ids[j], want_pcm=specs[j]["want_pcm"])
    if i % 37 == 0 and inflight:
        st = ge.poll(ids[inflight[0]])
        ge.fetch(ids[inflight[0]], 0, max(st["n_frames"], 1), 0, int(st["n_pcm"]))
for j in inflight:
    assert ge.wait(ids[j], 120000.0)
    results[j] = ge.result(ids[j], want_pcm=specs[j]["want_pcm"])
ge.sched_stop()
dt = time.time() - t0
frames = sum(r["codes"].shape[0] for r in results.values())
print("soak: %d requests, %d frames in %.1f s (%.0f frames/s)" % (N, frames, dt, frames / dt))
bad = 0
for j in range(0, N, 41):
    sp = specs[j]
    alone = ge.generate_batch([sp["prompt"]], max_steps=sp["max_steps"], temperature=sp["temperature"], top_k=sp["top_k"], top_p=sp["top_p"],
                              seed=sp["seed"], mask_eos=sp["mask_eos"], want_pcm=sp["want_pcm"])[0]
    ok = np.array_equal(alone["codes"], results[j]["codes"])
    if ok and sp["want_pcm"] and alone["pcm"].size:
        ok = alone["pcm"].size == results[j]["pcm"].size and float(np.sqrt(np.mean((alone["pcm"] - results[j]["pcm"]) ** 2))) < 1e-4
    bad += 0 if ok else 1
print("soak: %d sampled requests re-run solo, %d mismatches" % (len(range(0, N, 41)), bad))
ge.close()
sys.exit(1 if bad else 0)
