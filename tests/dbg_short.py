"""debug helper (not a test)"""
import os, sys
import numpy as np
import pytest


@pytest.mark.gpu
def test_dbg(gpu, oracle, tiny_model, vivian):
    nslots = int(os.environ.get("DBG_SLOTS", "4"))
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=nslots, max_steps=64, load_codec=False, use_graph=os.environ.get("DBG_EAGER") != "1")
    oe = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), None, 4)
    rng = np.random.default_rng(11)
    specs = []
    for i, (n_text, steps) in enumerate(((3, 5), (30, 17), (9, 8), (1, 12), (44, 4), (12, 9), (7, 0), (20, 13), (5, 6))):
        prompt = ge.assets.build_core(rng.integers(0, 4000, n_text).astype(np.int32), lang_id=2055, spk_emb=vivian)
        specs.append(dict(prompt=prompt, max_steps=steps, temperature=0.0, top_k=30, top_p=0.85, seed=100 + i, mask_eos=(i % 2 == 0)))
    only = os.environ.get("DBG_ONLY")
    sel = [int(only)] if only else range(len(specs))
    ids = [(i, ge.submit(**specs[i])) for i in sel]
    while ge.sched_step():
        pass
    for i, rid in ids:
        sp = specs[i]
        r = ge.result(rid)
        oc, _ = oe.generate(sp["prompt"], max_steps=sp["max_steps"], temperature=0.0, top_k=30, top_p=0.85, seed=sp["seed"], mask_eos=sp["mask_eos"])
        n = min(oc.shape[0], r["codes"].shape[0])
        bad = [f for f in range(n) if not np.array_equal(oc[f], r["codes"][f])]
        print("REQ", i, "frames", oc.shape[0], r["codes"].shape[0], "first_bad", bad[0] if bad else -1, file=sys.stderr)
    ge.close(); oe.close()
