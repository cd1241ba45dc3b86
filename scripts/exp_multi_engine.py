"""Experiment: N engines x (64/N) slots in ONE process, each driven by its own thread, vs one 64-slot engine.
Tests whether independent AR streams overlap each other's launch/latency bubbles (a case for multi-lane AR inside one engine)."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
sys.path.insert(0, ROOT)
import q3tts as Q
import bench

def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    model = os.environ.get("Q3_BENCH_MODEL", "/tmp/q3tts_synth_full")
    bench.ensure_model(model, "q8_0")
    codec = "--no-codec" not in sys.argv
    spk = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "speakers", "vivian.json")))["spk_emb"], np.float32)
    steps = 32
    for n_eng in (1, 2, 4):
        per = total // n_eng
        engs = [Q.Engine(model, "q8_0", max_batch=per, max_prompt=1024, max_steps=4 * steps, load_codec=codec) for _ in range(n_eng)]
        prompts = [bench.build_prompt(engs[0].assets, spk, n_text=(16, 32, 64)[i % 3], seed=42 + i) for i in range(total)]
        def work(e, pl, st):
            e.generate_batch(pl, max_steps=4 * st, mask_eos=True, want_pcm=codec)
        for st in (4, steps):
            th = [threading.Thread(target=work, args=(engs[k], prompts[k * per:(k + 1) * per], st)) for k in range(n_eng)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            el = time.perf_counter() - t0
        print("engines %d x %d slots: %.3f s -> %.1f audio-s/s" % (n_eng, per, el, total * 4 * steps * 0.08 / el), flush=True)
        for e in engs: e.close()

main()
