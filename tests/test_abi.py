"""C-ABI surface: every function include/*.h declares is exported by libq3tts.so, struct layouts match the reference's
#[repr(C)] mirrors, the shared object is installed as runtime/libllama.so, and compute entry points fail LOUDLY
(no fallback) when no GPU is present.  No kernels are launched here."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qwen3-tts-rust_amd")


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef\s+[^;{]*\(\s*\*\s*\w+\s*\)\s*\([^;]*\);", "", txt)   # function-pointer typedefs
    names = re.findall(r"\b((?:q3tts|llama|ggml)_\w+)\s*\(", txt)
    return sorted(set(n for n in names if not n.endswith("_cb")))


def test_all_declared_symbols_are_exported(q3):
    L = q3.lib()
    decl = _declared("q3tts.h") + _declared("q3tts_llama.h")
    assert len(decl) > 60
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, missing
    assert set(q3.SYMBOLS) <= set(decl)
    llama = [n for n in _declared("q3tts_llama.h") if n.startswith("llama_")]
    assert len(llama) == 28   # llama/mod.rs:241-292: all .expect()-resolved


def test_runtime_libllama_is_installed_and_loadable():
    p = os.path.join(PKG, "runtime", "libllama.so")   # the reference loads <cwd>/runtime/libllama.so (llama/mod.rs:152,195,216)
    assert os.path.exists(p)
    L = C.CDLL(p)
    for n in ("llama_backend_init", "llama_model_load_from_file", "llama_decode", "llama_get_logits", "llama_get_embeddings",
              "llama_memory_seq_rm", "llama_batch_init", "ggml_backend_load_all"):
        assert hasattr(L, n)


def test_struct_layouts_match_reference_repr_c(q3):
    out = (C.c_int32 * 8)()
    q3.lib().q3tts_llama_abi_sizes(out)
    # llama_model_params 72 B (n_gpu_layers@16, bools@64), llama_context_params 136 B (embeddings@112,
    # n_samplers@128), llama_batch 56 B (logits@48)
    assert list(out) == [72, 16, 64, 136, 112, 128, 56, 48]


def test_default_params_and_batch_init(q3):
    L = q3.lib()

    class Batch(C.Structure):
        _fields_ = [("n_tokens", C.c_int32), ("token", C.c_void_p), ("embd", C.POINTER(C.c_float)), ("pos", C.POINTER(C.c_int32)),
                    ("n_seq_id", C.POINTER(C.c_int32)), ("seq_id", C.POINTER(C.POINTER(C.c_int32))), ("logits", C.POINTER(C.c_int8))]
    L.llama_batch_init.restype = Batch
    L.llama_batch_init.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    b = L.llama_batch_init(32, 1024, 1)   # engine.rs:469
    b.embd[32 * 1024 - 1] = 1.0; b.pos[31] = 7; b.seq_id[31][0] = 0; b.logits[31] = 1; b.n_seq_id[31] = 1   # llama/mod.rs:556-614 writes
    assert not b.token
    L.llama_batch_free.argtypes = [Batch]
    L.llama_batch_free(b)
    sc = q3.SamplerConfig(); L.q3tts_sampler_config_default(C.byref(sc))
    assert (round(sc.temperature, 3), sc.top_k, round(sc.top_p, 3), sc.has_seed) == (0.7, 40, 0.9, 0)   # engine.rs:25-34
    ep = q3.EngineParams(); L.q3tts_engine_params_default(C.byref(ep))
    assert ep.max_steps == 512 and ep.max_prompt == 1024                                                # engine.rs:152; mod.rs:567-581


def test_compute_fails_loudly_without_gpu(q3, tiny_model):
    if q3.device_count() > 0:
        return   # on the GPU box this property cannot be observed
    try:
        q3.Engine(tiny_model, "q8_0", load_codec=False)
        raise AssertionError("engine creation must fail without a HIP device")
    except q3.Q3Error as ex:
        assert "no HIP device" in str(ex) and "no CPU fallback" in str(ex)
    import numpy as np
    try:
        q3.op_rmsnorm_quant(np.zeros((1, 256), np.float32), np.ones(256, np.float32))
        raise AssertionError("ops must fail without a HIP device")
    except q3.Q3Error as ex:
        assert "no HIP device" in str(ex)


def test_product_never_touches_the_oracle():
    bad = []
    for d, _, files in os.walk(PKG):
        if "build" in d.split(os.sep):
            continue
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".hpp", ".py", ".sh")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"q3o_|libq3oracle|q3oracle|/oracle/", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    # and the shared object has no dependency on it
    out = subprocess.run(["ldd", os.path.join(PKG, "libq3tts.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
