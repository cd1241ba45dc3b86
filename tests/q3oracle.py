"""ctypes binding of the CPU oracle (oracle/libq3oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package never does (the product fails loudly when its HIP extension is missing instead of falling back).
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_NUM_THREADS", "8")       # GPU boxes expose far more cores than their CPU share
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        path = os.path.join(ROOT, "oracle", "libq3oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.q3o_engine_create.restype = C.c_void_p
        L.q3o_engine_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.q3o_engine_free.argtypes = [C.c_void_p]
        L.q3o_engine_generate.restype = C.c_int
        L.q3o_engine_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_model_load.restype = C.c_void_p
        L.q3o_model_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        L.q3o_model_free.argtypes = [C.c_void_p]
        L.q3o_model_clear_kv.argtypes = [C.c_void_p]
        L.q3o_model_eval.restype = C.c_int
        L.q3o_model_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.q3o_assets_load.restype = C.c_void_p
        L.q3o_assets_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.q3o_assets_free.argtypes = [C.c_void_p]
        L.q3o_project.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.q3o_codec_embedding.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p]
        L.q3o_text_embedding.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.q3o_build_core.restype = C.c_int
        L.q3o_build_core.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.q3o_build_clone.restype = C.c_int
        L.q3o_build_clone.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.q3o_sampler_init.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_uint64]
        L.q3o_sample.restype = C.c_int32
        L.q3o_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.q3o_rng_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.q3o_rng_next_u32.restype = C.c_uint32
        L.q3o_rng_next_u32.argtypes = [C.c_void_p]
        L.q3o_chunker_push.restype = C.c_int
        L.q3o_chunker_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.q3o_codec_load.restype = C.c_void_p
        L.q3o_codec_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.q3o_codec_free.argtypes = [C.c_void_p]
        L.q3o_codec_reset.argtypes = [C.c_void_p]
        L.q3o_codec_samples_per_frame.restype = C.c_int
        L.q3o_codec_samples_per_frame.argtypes = [C.c_void_p]
        L.q3o_codec_decode.restype = C.c_int
        L.q3o_codec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.q3o_mel.restype = C.c_int
        L.q3o_mel.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.q3o_dequant_row.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.q3o_quant_act.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.q3o_matvec.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.q3o_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        L.q3o_headnorm128.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.q3o_sumsq_vec.restype = C.c_float
        L.q3o_sumsq_vec.argtypes = [C.c_void_p, C.c_int64]
        L.q3o_gguf_open.restype = C.c_void_p
        L.q3o_gguf_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.q3o_gguf_close.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class EngineStruct(C.Structure):
    _fields_ = [("assets", C.c_void_p), ("talker", C.c_void_p), ("predictor", C.c_void_p), ("codec", C.c_void_p),
                ("max_steps", C.c_int), ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_float),
                ("seed", C.c_uint64), ("mask_eos", C.c_int), ("n_threads", C.c_int),
                ("margins", C.c_void_p), ("margins_cap", C.c_int), ("forced", C.c_void_p), ("forced_frames", C.c_int), ("own_codes", C.c_void_p)]


class Assets:
    def __init__(self, path):
        err = C.create_string_buffer(256)
        self.h = lib().q3o_assets_load(path.encode(), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self._own = True

    @classmethod
    def borrow(cls, handle):
        a = cls.__new__(cls)
        a.h = handle
        a._own = False
        return a

    def close(self):
        if self.h and self._own:
            lib().q3o_assets_free(self.h)
        self.h = None

    def project(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n_out = 4096
        out = np.zeros(n_out, np.float32)
        lib().q3o_project(self.h, _p(x), x.size, _p(out))
        return out

    def codec_embedding(self, q, code):
        out = np.zeros(2048, np.float32)
        lib().q3o_codec_embedding(self.h, q, code, _p(out))
        return out

    def text_embedding(self, tok):
        out = np.zeros(2048, np.float32)
        lib().q3o_text_embedding(self.h, tok, _p(out))
        return out

    def build_core(self, text_ids, lang_id=2055, spk_id=None, spk_emb=None, instr_ids=None, mid=None, max_rows=4096):
        t = np.ascontiguousarray(text_ids, np.int32)
        ins = np.ascontiguousarray(instr_ids, np.int32) if instr_ids is not None else None
        se = np.ascontiguousarray(spk_emb, np.float32) if spk_emb is not None else None
        md = np.ascontiguousarray(mid, np.float32) if mid is not None else None
        out = np.zeros((max_rows, 2048), np.float32)
        n = lib().q3o_build_core(self.h, _p(t), t.size, 1 if lang_id is not None else 0, lang_id or 0,
                                 1 if spk_id is not None else 0, spk_id or 0, _p(se), _p(ins),
                                 ins.size if ins is not None else 0, _p(md), md.shape[0] if md is not None else 0,
                                 _p(out), max_rows)
        if n < 0:
            raise RuntimeError("prompt overflow")
        return out[:n].copy()

    def build_clone(self, text_ids, ref_codes, ref_text_ids, spk_emb, lang_id=2055, instr_ids=None, max_rows=4096):
        t = np.ascontiguousarray(text_ids, np.int32)
        rc = np.ascontiguousarray(ref_codes, np.int32)
        rt = np.ascontiguousarray(ref_text_ids, np.int32)
        se = np.ascontiguousarray(spk_emb, np.float32)
        ins = np.ascontiguousarray(instr_ids, np.int32) if instr_ids is not None else None
        out = np.zeros((max_rows, 2048), np.float32)
        n = lib().q3o_build_clone(self.h, _p(t), t.size, _p(rc), rc.size, _p(rt), rt.size, _p(se), lang_id, _p(ins),
                                  ins.size if ins is not None else 0, _p(out), max_rows)
        if n < 0:
            raise RuntimeError("prompt overflow")
        return out[:n].copy()


class Model:
    def __init__(self, path, n_ctx=4096):
        err = C.create_string_buffer(256)
        self.h = lib().q3o_model_load(path.encode(), n_ctx, err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def close(self):
        if self.h:
            lib().q3o_model_free(self.h)
        self.h = None

    def clear(self):
        lib().q3o_model_clear_kv(self.h)

    def eval(self, x, pos, n_embd, row0=0, row1=0):
        x = np.ascontiguousarray(x, np.float32)
        pos4 = np.ascontiguousarray(pos, np.int32)
        hid = np.zeros(n_embd, np.float32)
        logits = np.zeros(max(row1 - row0, 1), np.float32)
        rc = lib().q3o_model_eval(self.h, _p(x), _p(pos4), _p(hid), _p(logits) if row1 > row0 else None, row0, row1)
        if rc:
            raise RuntimeError("q3o_model_eval rc=%d" % rc)
        return hid, logits[: max(row1 - row0, 0)]


class Engine:
    def __init__(self, quant_dir, codec_path=None, n_threads=0):
        err = C.create_string_buffer(256)
        self.h = lib().q3o_engine_create(quant_dir.encode(), (codec_path or "").encode(), n_threads, err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.s = EngineStruct.from_address(self.h)
        self.assets = Assets.borrow(self.s.assets)

    def close(self):
        if self.h:
            lib().q3o_engine_free(self.h)
        self.h = None

    def generate(self, prompt, max_steps=8, temperature=0.0, top_k=40, top_p=0.9, seed=42, mask_eos=True, want_pcm=False):
        prompt = np.ascontiguousarray(prompt, np.float32)
        self.s.max_steps = max_steps
        self.s.temperature = temperature
        self.s.top_k = top_k
        self.s.top_p = top_p
        self.s.seed = seed
        self.s.mask_eos = 1 if mask_eos else 0
        codes = np.zeros(max_steps * 16, np.int32)
        spf = lib().q3o_codec_samples_per_frame(self.s.codec) if (want_pcm and self.s.codec) else 0
        pcm = np.zeros(max(max_steps * spf, 1), np.float32)
        npcm = C.c_int(0)
        n = lib().q3o_engine_generate(self.h, _p(prompt), prompt.shape[0], _p(codes), _p(pcm) if spf else None,
                                      pcm.size, C.byref(npcm))
        if n < 0:
            raise RuntimeError("q3o_engine_generate rc=%d" % n)
        return codes[: n * 16].reshape(n, 16).copy(), pcm[: npcm.value].copy()

    def generate_measured(self, prompt, max_steps, forced=None, mask_eos=True):
        """greedy run that also returns (own picks [n][16], margins [n][2]); `forced` [n][16] teacher-forces the trajectory (see q3o.h)"""
        margins = np.zeros((max_steps, 2), np.float32)
        own = np.zeros((max_steps, 16), np.int32)
        f = np.ascontiguousarray(forced, np.int32) if forced is not None else None
        self.s.margins, self.s.margins_cap, self.s.own_codes = margins.ctypes.data, margins.size, own.ctypes.data
        self.s.forced, self.s.forced_frames = (f.ctypes.data, f.shape[0]) if f is not None else (None, 0)
        try:
            codes, _ = self.generate(prompt, max_steps=max_steps, temperature=0.0, mask_eos=mask_eos)
        finally:
            self.s.margins, self.s.margins_cap, self.s.own_codes, self.s.forced, self.s.forced_frames = None, 0, None, None, 0
        n = codes.shape[0]
        return codes, own[:n].copy(), margins[:n].copy()


def set_arith_mode(mode):
    """0 = include/q3tts_spec.h arithmetic (default), 1 = ggml-CPU generic arithmetic (oracle/q3o_ggml.c)"""
    lib().q3o_set_arith_mode(1 if mode else 0)


class Codec:
    def __init__(self, path):
        err = C.create_string_buffer(256)
        self.h = lib().q3o_codec_load(path.encode(), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.spf = lib().q3o_codec_samples_per_frame(self.h)

    def close(self):
        if self.h:
            lib().q3o_codec_free(self.h)
        self.h = None

    def reset(self):
        lib().q3o_codec_reset(self.h)

    def decode(self, codes, is_last=False):
        codes = np.ascontiguousarray(codes, np.int64).reshape(-1, 16)
        pcm = np.zeros(codes.shape[0] * self.spf, np.float32)
        n = lib().q3o_codec_decode(self.h, _p(codes), codes.shape[0], 1 if is_last else 0, _p(pcm), pcm.size)
        if n < 0:
            raise RuntimeError("q3o_codec_decode rc=%d" % n)
        return pcm[:n]


def set_threads(n):
    """OpenMP team size of the calling thread's following oracle calls (codec, mel); Engine / Model take theirs as a constructor argument"""
    lib().q3o_set_threads(int(n))


def mel(audio):
    audio = np.ascontiguousarray(audio, np.float32)
    n = lib().q3o_mel(_p(audio), audio.size, None)
    out = np.zeros((n, 128), np.float32)
    n2 = lib().q3o_mel(_p(audio), audio.size, _p(out))
    return out[:n2]


def sample(logits, start, end, temperature=0.0, top_k=0, top_p=1.0, seed=42, state=None):
    logits = np.ascontiguousarray(logits, np.float32)
    if state is None:
        state = C.create_string_buffer(256)
        lib().q3o_sampler_init(state, temperature, top_k, top_p, seed)
    return lib().q3o_sample(state, _p(logits), logits.size, start, end), state
