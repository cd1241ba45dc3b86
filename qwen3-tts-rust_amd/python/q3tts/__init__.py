"""ctypes binding over the C ABI of libq3tts.so (include/q3tts.h).

Python here is harness plumbing for tests / bench only; the product is the shared library.  There is no
fallback: if the library is missing or no HIP device is visible, compute entry points raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.path.join(PKG_ROOT, "libq3tts.so")
_lib = None


class Q3Error(RuntimeError):
    pass


class SamplerConfig(C.Structure):  # engine.rs:13-45
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("has_seed", C.c_int32),
                ("seed", C.c_uint64)]


class EngineParams(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("quant", C.c_char_p), ("device", C.c_int32), ("max_batch", C.c_int32),
                ("max_prompt", C.c_int32), ("max_steps", C.c_int32), ("load_codec", C.c_int32), ("use_graph", C.c_int32)]


class Request(C.Structure):
    _fields_ = [("prompt", C.c_void_p), ("n_prompt", C.c_int32), ("sampler", SamplerConfig), ("max_steps", C.c_int32),
                ("mask_eos", C.c_int32), ("codes_out", C.c_void_p), ("pcm_out", C.c_void_p), ("pcm_capacity", C.c_int64),
                ("n_frames", C.c_int32), ("n_pcm", C.c_int64), ("prefill_ms", C.c_double), ("first_chunk_ms", C.c_double),
                ("total_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("frame_loop_ms", C.c_double), ("frames", C.c_int64), ("prefill_ms", C.c_double), ("gemv_ms", C.c_double),
                ("gemv_launches", C.c_int64), ("gemv_bytes", C.c_double), ("codec_ms", C.c_double), ("codec_calls", C.c_int64),
                ("talker_weight_bytes", C.c_double), ("predictor_weight_bytes", C.c_double), ("kv_bytes_per_token", C.c_double),
                ("gu_ms", C.c_double), ("gu_launches", C.c_int64), ("gu_bytes", C.c_double),
                ("sched_steps", C.c_int64), ("slot_frames", C.c_double), ("graph_frames", C.c_int64)]


class ReqStatus(C.Structure):
    _fields_ = [("state", C.c_int32), ("n_frames", C.c_int32), ("n_pcm", C.c_int64), ("queue_ms", C.c_double),
                ("prefill_ms", C.c_double), ("first_chunk_ms", C.c_double), ("total_ms", C.c_double)]


REQ_QUEUED, REQ_RUNNING, REQ_DRAINING, REQ_DONE, REQ_FAILED = 0, 1, 2, 3, -1


DECODE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32)

# every symbol include/q3tts.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "q3tts_last_error", "q3tts_version", "q3tts_device_count", "q3tts_sampler_config_default", "q3tts_engine_params_default",
    "q3tts_engine_create", "q3tts_engine_destroy", "q3tts_generate_batch", "q3tts_engine_stats", "q3tts_engine_reset_stats",
    "q3tts_engine_set_instrument", "q3tts_engine_bytes_per_step", "q3tts_assets_open", "q3tts_assets_close", "q3tts_engine_assets",
    "q3tts_assets_codec_embedding", "q3tts_assets_text_embedding", "q3tts_assets_tts_pad", "q3tts_prompt_build_core",
    "q3tts_prompt_build_clone", "q3tts_sampler_new", "q3tts_sampler_free", "q3tts_sampler_sample", "q3tts_chunker_new",
    "q3tts_chunker_free", "q3tts_chunker_push", "q3tts_decoder_create", "q3tts_decoder_destroy", "q3tts_decoder_samples_per_frame",
    "q3tts_decoder_reset", "q3tts_decoder_decode", "q3tts_mel_frames", "q3tts_mel", "q3tts_tf_open", "q3tts_tf_close",
    "q3tts_tf_dims", "q3tts_tf_clear", "q3tts_tf_eval", "q3tts_op_gemv_q8", "q3tts_op_gateup_q8", "q3tts_op_matmul_float", "q3tts_op_rmsnorm_quant", "q3tts_op_swiglu_quant",
    "q3tts_op_argmax", "q3tts_op_project", "q3tts_op_sample", "q3tts_submit", "q3tts_poll", "q3tts_fetch", "q3tts_wait",
    "q3tts_release", "q3tts_decoder_create_ex", "q3tts_decoder_decode_group", "q3tts_sched_start", "q3tts_sched_stop", "q3tts_sched_step", "q3tts_voice_register", "q3tts_submit_text",
    "q3tts_group_create", "q3tts_group_destroy", "q3tts_group_size", "q3tts_group_engine", "q3tts_group_uses_rccl", "q3tts_group_voice_register",
    "q3tts_group_submit", "q3tts_group_submit_text", "q3tts_group_device_of", "q3tts_group_poll", "q3tts_group_fetch", "q3tts_group_wait",
    "q3tts_group_release", "q3tts_group_start", "q3tts_group_stop", "q3tts_comm_available", "q3tts_comm_unique_id", "q3tts_comm_create",
    "q3tts_comm_destroy", "q3tts_comm_voice_register", "q3tts_engine_device",
    "q3tts_onnx_open", "q3tts_onnx_close", "q3tts_onnx_counts", "q3tts_onnx_summary", "q3tts_onnx_node", "q3tts_onnx_node_input", "q3tts_onnx_node_output",
    "q3tts_onnx_node_attr_ints", "q3tts_onnx_node_attr_float", "q3tts_onnx_initializer", "q3tts_onnx_op_kernel", "q3tts_onnx_decoder_contract",
    "q3tts_onnx_session_open", "q3tts_onnx_session_close", "q3tts_onnx_session_unsupported", "q3tts_onnx_session_set_input", "q3tts_onnx_session_run",
    "q3tts_onnx_session_output_info", "q3tts_onnx_session_output", "q3tts_onnx_session_launches", "q3tts_onnx_op_executable",
    "q3tts_text_nfc", "q3tts_op_gemv_kq", "q3tts_op_gateup_kq", "q3tts_onnx_decoder_open", "q3tts_onnx_decoder_close", "q3tts_onnx_decoder_reset", "q3tts_onnx_decoder_fetch", "q3tts_group_info", "q3tts_onnx_decoder_decode",
    "q3tts_decoder_state_floats", "q3tts_decoder_state_export", "q3tts_decoder_state_import", "q3tts_decoder_state_entry",
    "q3tts_tokenizer_open", "q3tts_tokenizer_close", "q3tts_tokenizer_encode", "q3tts_tokenizer_decode", "q3tts_tokenizer_vocab_size",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Q3Error("libq3tts.so not built (run qwen3-tts-rust_amd/build.sh); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.q3tts_last_error.restype = C.c_char_p
        L.q3tts_engine_create.argtypes = [C.POINTER(EngineParams), C.POINTER(C.c_void_p)]
        L.q3tts_engine_destroy.argtypes = [C.c_void_p]
        L.q3tts_generate_batch.argtypes = [C.c_void_p, C.POINTER(Request), C.c_int32, C.c_int32]
        L.q3tts_engine_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.q3tts_submit.argtypes = [C.c_void_p, C.POINTER(Request), C.c_int32, C.POINTER(C.c_int64)]
        L.q3tts_poll.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ReqStatus)]
        L.q3tts_fetch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.q3tts_wait.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        L.q3tts_release.argtypes = [C.c_void_p, C.c_int64]
        L.q3tts_sched_start.argtypes = [C.c_void_p]
        L.q3tts_sched_stop.argtypes = [C.c_void_p]
        L.q3tts_sched_step.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.q3tts_voice_register.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.q3tts_submit_text.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                        C.POINTER(SamplerConfig), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.q3tts_engine_reset_stats.argtypes = [C.c_void_p]
        L.q3tts_engine_set_instrument.argtypes = [C.c_void_p, C.c_int32]
        L.q3tts_engine_bytes_per_step.restype = C.c_double
        L.q3tts_engine_bytes_per_step.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        L.q3tts_assets_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.q3tts_assets_close.argtypes = [C.c_void_p]
        L.q3tts_engine_assets.restype = C.c_void_p
        L.q3tts_engine_assets.argtypes = [C.c_void_p]
        L.q3tts_assets_codec_embedding.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.q3tts_assets_text_embedding.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.q3tts_assets_tts_pad.argtypes = [C.c_void_p, C.c_void_p]
        L.q3tts_prompt_build_core.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.q3tts_prompt_build_clone.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                               C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.q3tts_sampler_new.restype = C.c_void_p
        L.q3tts_sampler_new.argtypes = [C.c_float, C.c_int32, C.c_float, C.c_uint64]
        L.q3tts_sampler_free.argtypes = [C.c_void_p]
        L.q3tts_sampler_sample.restype = C.c_int32
        L.q3tts_sampler_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.q3tts_chunker_new.restype = C.c_void_p
        L.q3tts_chunker_new.argtypes = [DECODE_CB, C.c_void_p]
        L.q3tts_chunker_free.argtypes = [C.c_void_p]
        L.q3tts_chunker_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.q3tts_decoder_create.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.q3tts_decoder_destroy.argtypes = [C.c_void_p]
        L.q3tts_decoder_create_ex.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.q3tts_decoder_decode_group.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.q3tts_decoder_samples_per_frame.argtypes = [C.c_void_p]
        L.q3tts_decoder_reset.argtypes = [C.c_void_p, C.c_int32]
        L.q3tts_decoder_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
        L.q3tts_mel_frames.argtypes = [C.c_int32]
        L.q3tts_mel.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.q3tts_tf_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.q3tts_tf_close.argtypes = [C.c_void_p]
        L.q3tts_tf_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
        L.q3tts_tf_clear.argtypes = [C.c_void_p]
        L.q3tts_tf_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.q3tts_op_gemv_q8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.q3tts_op_gateup_q8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.q3tts_op_matmul_float.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.q3tts_op_rmsnorm_quant.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.q3tts_op_swiglu_quant.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.q3tts_op_argmax.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.q3tts_op_sample.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p]
        L.q3tts_op_project.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc):
    if rc != 0:
        raise Q3Error(lib().q3tts_last_error().decode())


def device_count():
    return lib().q3tts_device_count()


class Assets:
    def __init__(self, path=None, handle=None):
        self._own = handle is None
        if handle is None:
            h = C.c_void_p()
            _chk(lib().q3tts_assets_open(path.encode(), C.byref(h)))
            handle = h.value
        self.h = handle

    def close(self):
        if self._own and self.h:
            lib().q3tts_assets_close(self.h)
        self.h = None

    def codec_embedding(self, q, code):
        out = np.zeros(2048, np.float32)
        _chk(lib().q3tts_assets_codec_embedding(self.h, q, code, _p(out)))
        return out

    def text_embedding(self, tok):
        out = np.zeros(2048, np.float32)
        _chk(lib().q3tts_assets_text_embedding(self.h, tok, _p(out)))
        return out

    def tts_pad(self):
        out = np.zeros(2048, np.float32)
        _chk(lib().q3tts_assets_tts_pad(self.h, _p(out)))
        return out

    def build_core(self, text_ids, lang_id=2055, spk_id=None, spk_emb=None, instr_ids=None, mid=None, max_rows=4096):
        t = np.ascontiguousarray(text_ids, np.int32)
        ins = np.ascontiguousarray(instr_ids, np.int32) if instr_ids is not None else None
        se = np.ascontiguousarray(spk_emb, np.float32) if spk_emb is not None else None
        md = np.ascontiguousarray(mid, np.float32) if mid is not None else None
        out = np.zeros((max_rows, 2048), np.float32)
        n = lib().q3tts_prompt_build_core(self.h, _p(t), t.size, -1 if lang_id is None else lang_id,
                                          -1 if spk_id is None else spk_id, _p(se), _p(ins), ins.size if ins is not None else 0,
                                          _p(md), md.shape[0] if md is not None else 0, _p(out), max_rows)
        if n < 0:
            raise Q3Error(lib().q3tts_last_error().decode())
        return out[:n].copy()

    def build_clone(self, text_ids, ref_codes, ref_text_ids, spk_emb, lang_id=2055, instr_ids=None, max_rows=4096):
        t = np.ascontiguousarray(text_ids, np.int32)
        rc = np.ascontiguousarray(ref_codes, np.int32)
        rt = np.ascontiguousarray(ref_text_ids, np.int32)
        se = np.ascontiguousarray(spk_emb, np.float32)
        ins = np.ascontiguousarray(instr_ids, np.int32) if instr_ids is not None else None
        out = np.zeros((max_rows, 2048), np.float32)
        n = lib().q3tts_prompt_build_clone(self.h, _p(t), t.size, _p(rc), rc.size, _p(rt), rt.size, _p(se), lang_id, _p(ins),
                                           ins.size if ins is not None else 0, _p(out), max_rows)
        if n < 0:
            raise Q3Error(lib().q3tts_last_error().decode())
        return out[:n].copy()


class Sampler:  # LlamaSampler (llama/mod.rs:627-776)
    def __init__(self, temperature=0.0, top_k=0, top_p=1.0, seed=42):
        self.h = lib().q3tts_sampler_new(temperature, top_k, top_p, seed)

    def sample(self, logits, start=0, end=None):
        logits = np.ascontiguousarray(logits, np.float32)
        return lib().q3tts_sampler_sample(self.h, _p(logits), logits.size, start, logits.size if end is None else end)

    def close(self):
        if self.h:
            lib().q3tts_sampler_free(self.h)
        self.h = None


class Chunker:  # engine.rs:495-543
    def __init__(self):
        self.calls = []

        def cb(user, codes, n, fin):
            self.calls.append(([codes[i] for i in range(n)], bool(fin)))

        self._cb = DECODE_CB(cb)
        self.h = lib().q3tts_chunker_new(self._cb, None)

    def push(self, codes, is_final=False):
        a = np.ascontiguousarray(codes, np.int64)
        _chk(lib().q3tts_chunker_push(self.h, _p(a) if a.size else None, a.size, 1 if is_final else 0))

    def close(self):
        if self.h:
            lib().q3tts_chunker_free(self.h)
        self.h = None


class Engine:  # TtsEngine (engine.rs:53-169) at the embedding level
    def __init__(self, model_dir, quant="q8_0", max_batch=1, max_prompt=1024, max_steps=512, load_codec=True, use_graph=True, device=0):
        p = EngineParams()
        lib().q3tts_engine_params_default(C.byref(p))
        self._md = model_dir.encode()
        self._q = quant.encode()
        p.model_dir = self._md
        p.quant = self._q
        p.device = device
        p.max_batch = max_batch
        p.max_prompt = max_prompt
        p.max_steps = max_steps
        p.load_codec = 1 if load_codec else 0
        p.use_graph = 1 if use_graph else 0
        h = C.c_void_p()
        _chk(lib().q3tts_engine_create(C.byref(p), C.byref(h)))
        self.h = h.value
        self.max_steps = max_steps
        self.assets = Assets(handle=lib().q3tts_engine_assets(self.h))

    def close(self):
        if self.h:
            lib().q3tts_engine_destroy(self.h)
        self.h = None

    @staticmethod
    def _per(v, i):
        return v[i] if isinstance(v, (list, tuple, np.ndarray)) else v

    def _fill(self, r, pr, i, max_steps, temperature, top_k, top_p, seed, mask_eos):
        r.prompt = pr.ctypes.data
        r.n_prompt = pr.shape[0]
        r.sampler.temperature = self._per(temperature, i)
        r.sampler.top_k = self._per(top_k, i)
        r.sampler.top_p = self._per(top_p, i)
        r.sampler.has_seed = 1
        r.sampler.seed = self._per(seed, i)
        r.max_steps = self._per(max_steps, i)
        r.mask_eos = 1 if self._per(mask_eos, i) else 0

    # ---- continuous-batching scheduler (q3tts_submit / poll / fetch / wait) ----
    def submit(self, prompt, max_steps=8, temperature=0.0, top_k=40, top_p=0.9, seed=42, mask_eos=True, want_pcm=False):
        pr = np.ascontiguousarray(prompt, np.float32)
        r = Request()
        self._fill(r, pr, 0, max_steps, temperature, top_k, top_p, seed, mask_eos)
        rid = C.c_int64()
        _chk(lib().q3tts_submit(self.h, C.byref(r), 1 if want_pcm else 0, C.byref(rid)))
        return rid.value

    def register_voice(self, spk_emb, ref_codes=None, ref_text_ids=None):
        spk = np.ascontiguousarray(spk_emb, np.float32)
        rc = np.ascontiguousarray(ref_codes, np.int32).reshape(-1) if ref_codes is not None else None
        rt = np.ascontiguousarray(ref_text_ids, np.int32) if ref_text_ids is not None else None
        vid = C.c_int32()
        _chk(lib().q3tts_voice_register(self.h, _p(spk), _p(rc), rc.size if rc is not None else 0, _p(rt), rt.size if rt is not None else 0,
                                        C.byref(vid)))
        return vid.value

    def submit_text(self, voice_id, text_ids, lang_id=2055, instr_ids=None, max_steps=8, temperature=0.0, top_k=40, top_p=0.9, seed=42,
                    mask_eos=True, want_pcm=False):
        t = np.ascontiguousarray(text_ids, np.int32)
        ins = np.ascontiguousarray(instr_ids, np.int32) if instr_ids is not None else None
        sc = SamplerConfig(temperature, top_k, top_p, 1, seed)
        rid = C.c_int64()
        _chk(lib().q3tts_submit_text(self.h, voice_id, _p(t), t.size, lang_id, _p(ins), ins.size if ins is not None else 0, C.byref(sc),
                                     max_steps, 1 if mask_eos else 0, 1 if want_pcm else 0, C.byref(rid)))
        return rid.value

    def poll(self, rid):
        s = ReqStatus()
        _chk(lib().q3tts_poll(self.h, rid, C.byref(s)))
        return {k: getattr(s, k) for k, _ in ReqStatus._fields_}

    def fetch(self, rid, frame_off=0, max_frames=4096, pcm_off=0, pcm_cap=0):
        codes = np.zeros(max(max_frames, 1) * 16, np.int32)
        pcm = np.zeros(max(pcm_cap, 1), np.float32)
        gf, gp = C.c_int32(), C.c_int64()
        _chk(lib().q3tts_fetch(self.h, rid, _p(codes), frame_off, max_frames, _p(pcm) if pcm_cap > 0 else None, pcm_off, pcm_cap,
                               C.byref(gf), C.byref(gp)))
        return codes[: gf.value * 16].reshape(gf.value, 16).copy(), pcm[: gp.value].copy()

    def wait(self, rid, timeout_ms=-1.0):
        rc = lib().q3tts_wait(self.h, rid, timeout_ms)
        if rc < 0:
            _chk(rc)
        return rc == 0

    def release(self, rid):
        _chk(lib().q3tts_release(self.h, rid))

    def sched_start(self):
        _chk(lib().q3tts_sched_start(self.h))

    def sched_stop(self):
        _chk(lib().q3tts_sched_stop(self.h))

    def sched_step(self):
        b = C.c_int32()
        _chk(lib().q3tts_sched_step(self.h, C.byref(b)))
        return bool(b.value)

    def result(self, rid, want_pcm=False):
        """collects a finished request (codes [n][16], pcm) and releases it"""
        st = self.poll(rid)
        codes, pcm = self.fetch(rid, 0, max(st["n_frames"], 1), 0, st["n_pcm"] if want_pcm else 0)
        self.release(rid)
        return {"codes": codes, "pcm": pcm if want_pcm else None, **st}

    def generate_batch(self, prompts, max_steps=8, temperature=0.0, top_k=40, top_p=0.9, seed=42, mask_eos=True, want_pcm=False,
                       pcm_per_frame=1920):
        """per-request values may be given as lists (max_steps, temperature, top_k, top_p, seed, mask_eos)"""
        n = len(prompts)
        reqs = (Request * n)()
        keep = []
        for i, pr in enumerate(prompts):
            pr = np.ascontiguousarray(pr, np.float32)
            ms = self._per(max_steps, i)
            codes = np.zeros(max(ms, 1) * 16, np.int32)
            pcm = np.zeros(max(ms * pcm_per_frame, 1), np.float32) if want_pcm else None
            keep.append((pr, codes, pcm))
            r = reqs[i]
            self._fill(r, pr, i, max_steps, temperature, top_k, top_p, seed, mask_eos)
            r.codes_out = codes.ctypes.data
            r.pcm_out = pcm.ctypes.data if pcm is not None else None
            r.pcm_capacity = pcm.size if pcm is not None else 0
        _chk(lib().q3tts_generate_batch(self.h, reqs, n, 1 if want_pcm else 0))
        out = []
        for i in range(n):
            nf = reqs[i].n_frames
            out.append({"codes": keep[i][1][: nf * 16].reshape(nf, 16).copy(),
                        "pcm": keep[i][2][: reqs[i].n_pcm].copy() if want_pcm else None,
                        "prefill_ms": reqs[i].prefill_ms, "first_chunk_ms": reqs[i].first_chunk_ms, "total_ms": reqs[i].total_ms})
        return out

    def stats(self):
        s = Stats()
        _chk(lib().q3tts_engine_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        lib().q3tts_engine_reset_stats(self.h)

    def set_instrument(self, on):
        lib().q3tts_engine_set_instrument(self.h, 1 if on else 0)

    def bytes_per_step(self, batch, mean_ctx):
        return lib().q3tts_engine_bytes_per_step(self.h, batch, mean_ctx)


class TfContext:  # LlamaModel + LlamaContext at the embedding level (llama/mod.rs:326-513)
    def __init__(self, path, n_ctx=4096, max_tok=64):
        h = C.c_void_p()
        _chk(lib().q3tts_tf_open(path.encode(), n_ctx, max_tok, C.byref(h)))
        self.h = h.value
        d = [C.c_int32() for _ in range(4)]
        lib().q3tts_tf_dims(self.h, *[C.byref(x) for x in d])
        self.n_embd, self.n_layer, self.n_head, self.n_vocab = [x.value for x in d]

    def close(self):
        if self.h:
            lib().q3tts_tf_close(self.h)
        self.h = None

    def clear(self):
        lib().q3tts_tf_clear(self.h)

    def eval(self, x, pos4, row0=0, row1=0):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.n_embd)
        pos4 = np.ascontiguousarray(pos4, np.int32).reshape(-1, 4)
        n = x.shape[0]
        hid = np.zeros((n, self.n_embd), np.float32)
        logits = np.zeros((n, max(row1 - row0, 1)), np.float32)
        _chk(lib().q3tts_tf_eval(self.h, _p(x), _p(pos4), n, _p(hid), _p(logits) if row1 > row0 else None, row0, row1))
        return hid, logits[:, : max(row1 - row0, 0)]


class Decoder:  # AudioDecoder (onnx.rs:324-496)
    def __init__(self, path, n_streams=1, max_frames=64, max_group=1):
        h = C.c_void_p()
        if max_group > 1 or max_frames != 64:
            _chk(lib().q3tts_decoder_create_ex(path.encode(), n_streams, max_frames, max_group, C.byref(h)))
        else:
            _chk(lib().q3tts_decoder_create(path.encode(), n_streams, C.byref(h)))
        self.h = h.value
        self.spf = lib().q3tts_decoder_samples_per_frame(self.h)

    def close(self):
        if self.h:
            lib().q3tts_decoder_destroy(self.h)
        self.h = None

    def reset(self, stream=0):
        _chk(lib().q3tts_decoder_reset(self.h, stream))

    def decode(self, codes, is_last=False, stream=0):
        codes = np.ascontiguousarray(codes, np.int64).reshape(-1, 16)
        wav = np.zeros(codes.shape[0] * self.spf, np.float32)
        valid = C.c_int64(0)
        _chk(lib().q3tts_decoder_decode(self.h, stream, _p(codes), codes.shape[0], 1 if is_last else 0, _p(wav), C.byref(valid)))
        return wav[: valid.value]


    def state_export(self, stream=0):
        L = lib()
        L.q3tts_decoder_state_floats.restype = C.c_int64
        L.q3tts_decoder_state_floats.argtypes = [C.c_void_p]
        L.q3tts_decoder_state_export.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        buf = np.zeros(L.q3tts_decoder_state_floats(self.h), np.float32)
        _chk(L.q3tts_decoder_state_export(self.h, stream, _p(buf), buf.size))
        return buf

    def state_import(self, blob, stream=0):
        lib().q3tts_decoder_state_import.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        blob = np.ascontiguousarray(blob, np.float32)
        _chk(lib().q3tts_decoder_state_import(self.h, stream, _p(blob), blob.size))

    def state_layout(self):
        L = lib()
        L.q3tts_decoder_state_entry.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        n = L.q3tts_decoder_state_entry(self.h, -1, None, None, None, None)
        out = []
        for i in range(n):
            nm, off, r, c = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
            L.q3tts_decoder_state_entry(self.h, i, C.byref(nm), C.byref(off), C.byref(r), C.byref(c))
            out.append((nm.value.decode(), off.value, r.value, c.value))
        return out

    def decode_group(self, streams, codes):
        """codes [G][n_frames][16] for G distinct streams -> wav [G][n_frames*spf] (one decode pass for the whole group)"""
        st = np.ascontiguousarray(streams, np.int32)
        codes = np.ascontiguousarray(codes, np.int64).reshape(st.size, -1, 16)
        wav = np.zeros((st.size, codes.shape[1] * self.spf), np.float32)
        _chk(lib().q3tts_decoder_decode_group(self.h, st.size, _p(st), _p(codes), codes.shape[1], _p(wav)))
        return wav


def op_gemv_q8(w_raw, n, k, xq, xd, lpr=0):
    xq = np.ascontiguousarray(xq, np.int8).reshape(-1, k)
    xd = np.ascontiguousarray(xd, np.uint16).reshape(-1, k // 32)
    w_raw = np.ascontiguousarray(w_raw, np.uint8)
    y = np.zeros((xq.shape[0], n), np.float32)
    _chk(lib().q3tts_op_gemv_q8(_p(w_raw), n, k, _p(xq), _p(xd), xq.shape[0], _p(y), lpr))
    return y


def op_gateup_q8(w_raw, ff, k, xq, xd):
    """fused gate/up + SwiGLU + quant of the batched layer path: w_raw = Q8_0 rows [2*ff][k/32][34] (gate first)"""
    xq = np.ascontiguousarray(xq, np.int8).reshape(-1, k)
    xd = np.ascontiguousarray(xd, np.uint16).reshape(-1, k // 32)
    w_raw = np.ascontiguousarray(w_raw, np.uint8)
    aq = np.zeros((xq.shape[0], ff), np.int8)
    ad = np.zeros((xq.shape[0], ff // 32), np.uint16)
    _chk(lib().q3tts_op_gateup_q8(_p(w_raw), ff, k, _p(xq), _p(xd), xq.shape[0], _p(aq), _p(ad)))
    return aq, ad


def op_matmul_float(w_raw, ggml_type, n, k, x, row0=0, nrows=None):
    """w_raw: the row-major weight bytes (f32 / f16 / bf16 per ggml_type); x [ntok][k] f32 -> y [ntok][nrows] f32"""
    nrows = n - row0 if nrows is None else nrows
    w_raw = np.ascontiguousarray(w_raw)
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros((x.shape[0], nrows), np.float32)
    _chk(lib().q3tts_op_matmul_float(_p(w_raw), ggml_type, n, k, row0, nrows, _p(x), x.shape[0], _p(y)))
    return y


def op_rmsnorm_quant(x, g, eps=1e-6):
    x = np.ascontiguousarray(x, np.float32)
    x2 = x.reshape(-1, x.shape[-1])
    d = x2.shape[1]
    g = np.ascontiguousarray(g, np.float32)
    xq = np.zeros(x2.shape, np.int8)
    xd = np.zeros((x2.shape[0], d // 32), np.uint16)
    xn = np.zeros(x2.shape, np.float32)
    _chk(lib().q3tts_op_rmsnorm_quant(_p(x2), _p(g), d, x2.shape[0], eps, _p(xq), _p(xd), _p(xn)))
    return xq, xd, xn


def op_swiglu_quant(gu, ff):
    gu = np.ascontiguousarray(gu, np.float32).reshape(-1, 2 * ff)
    aq = np.zeros((gu.shape[0], ff), np.int8)
    ad = np.zeros((gu.shape[0], ff // 32), np.uint16)
    _chk(lib().q3tts_op_swiglu_quant(_p(gu), ff, gu.shape[0], _p(aq), _p(ad)))
    return aq, ad


def op_argmax(logits, start, end, mask_idx=-1):
    logits = np.ascontiguousarray(logits, np.float32)
    out = np.zeros(1, np.int32)
    _chk(lib().q3tts_op_argmax(_p(logits), logits.size, start, end, mask_idx, _p(out)))
    return int(out[0])


def op_sample(logits, temperature, top_k, top_p, seed, n_draws=1, mask_idx=-1):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.zeros(n_draws, dtype=np.int32)
    _chk(lib().q3tts_op_sample(_p(logits), logits.size, temperature, top_k, top_p, seed, mask_idx, n_draws, _p(out)))
    return out


def op_project(x, w, b):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    y = np.zeros(b.size, np.float32)
    _chk(lib().q3tts_op_project(_p(x), _p(w), _p(b), x.size, b.size, _p(y)))
    return y


def mel(audio):
    audio = np.ascontiguousarray(audio, np.float32)
    n = lib().q3tts_mel_frames(audio.size)
    out = np.zeros((n, 128), np.float32)
    _chk(lib().q3tts_mel(_p(audio), audio.size, _p(out)))
    return out


class OnnxSession:
    """ONNX graph executed on the GPU through the C ABI (q3tts_onnx_session_*): the role `ort::Session` has in the reference's encoders"""

    def __init__(self, path, device=0):
        L = lib()
        L.q3tts_onnx_session_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.q3tts_onnx_session_close.argtypes = [C.c_void_p]
        L.q3tts_onnx_session_unsupported.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.q3tts_onnx_session_set_input.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.q3tts_onnx_session_run.argtypes = [C.c_void_p]
        L.q3tts_onnx_session_output_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
        L.q3tts_onnx_session_output.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.q3tts_onnx_session_launches.restype = C.c_int64
        L.q3tts_onnx_session_launches.argtypes = [C.c_void_p]
        h = C.c_void_p()
        _chk(L.q3tts_onnx_session_open(path.encode(), device, C.byref(h)))
        self.h = h.value

    def close(self):
        if self.h:
            lib().q3tts_onnx_session_close(self.h)
        self.h = None

    def unsupported(self):
        buf = C.create_string_buffer(4096)
        n = lib().q3tts_onnx_session_unsupported(self.h, buf, 4096)
        return [x for x in buf.value.decode().split(",") if x] if n else []

    def run(self, feeds, outputs):
        """feeds: {name: ndarray (float32 or int64)}; returns {name: ndarray} for the requested graph outputs"""
        keep = []
        for name, arr in feeds.items():
            a = np.ascontiguousarray(arr, dtype=np.int64 if np.issubdtype(np.asarray(arr).dtype, np.integer) else np.float32)
            keep.append(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _chk(lib().q3tts_onnx_session_set_input(self.h, name.encode(), 7 if a.dtype == np.int64 else 1, a.ctypes.data, shape, a.ndim))
        _chk(lib().q3tts_onnx_session_run(self.h))
        res = {}
        for name in outputs:
            dt, rk = C.c_int32(), C.c_int32()
            shape = (C.c_int64 * 8)()
            _chk(lib().q3tts_onnx_session_output_info(self.h, name.encode(), C.byref(dt), C.byref(rk), shape))
            shp = tuple(shape[i] for i in range(rk.value))
            out = np.empty(shp, dtype=np.int64 if dt.value == 7 else np.float32)
            _chk(lib().q3tts_onnx_session_output(self.h, name.encode(), out.ctypes.data, out.nbytes))
            res[name] = out.astype(bool) if dt.value == 9 else out
        return res

    def launches(self):
        return lib().q3tts_onnx_session_launches(self.h)


class OnnxDecoder:
    """The exported streaming decoder graph run through the executor with its state on the device (AudioDecoder, onnx.rs:322-458)"""

    def __init__(self, path, device=0):
        L = lib()
        L.q3tts_onnx_decoder_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.q3tts_onnx_decoder_close.argtypes = [C.c_void_p]
        L.q3tts_onnx_decoder_reset.argtypes = [C.c_void_p]
        L.q3tts_onnx_decoder_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        h = C.c_void_p()
        _chk(L.q3tts_onnx_decoder_open(path.encode(), device, C.byref(h)))
        self.h = h.value

    def close(self):
        if self.h:
            lib().q3tts_onnx_decoder_close(self.h)
        self.h = None

    def reset(self):
        _chk(lib().q3tts_onnx_decoder_reset(self.h))

    def decode(self, codes, is_final=False, max_samples_per_frame=1920):
        codes = np.ascontiguousarray(codes, np.int64).reshape(-1, 16)
        out = np.zeros(max(codes.shape[0], 1) * max_samples_per_frame, np.float32)
        n = C.c_int64()
        rc = lib().q3tts_onnx_decoder_decode(self.h, codes.ctypes.data, codes.shape[0], 1 if is_final else 0, out.ctypes.data, out.size, C.byref(n))
        if rc == 2:  # buffer too small: the chunk is kept in the handle, fetch it again into one that fits
            out = np.zeros(n.value, np.float32)
            lib().q3tts_onnx_decoder_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
            rc = lib().q3tts_onnx_decoder_fetch(self.h, out.ctypes.data, out.size, C.byref(n))
        _chk(rc)
        return out[: n.value].copy()


def op_gemv_kq(parts, k, xq, xd, lpr=0):
    """parts: [(raw bytes ndarray, ggml type, rows)] -> y [ntok, sum rows] through the K-quant GEMV / matrix-core GEMM"""
    L = lib()
    L.q3tts_op_gemv_kq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    raws = [np.ascontiguousarray(p[0]) for p in parts]
    ptrs = (C.c_void_p * len(parts))(*[r.ctypes.data for r in raws])
    types = np.asarray([p[1] for p in parts], np.int32); rows = np.asarray([p[2] for p in parts], np.int32)
    xq = np.ascontiguousarray(xq, np.int8); xd = np.ascontiguousarray(xd, np.uint16)
    ntok = xq.shape[0]
    y = np.zeros((ntok, int(rows.sum())), np.float32)
    _chk(L.q3tts_op_gemv_kq(ptrs, types.ctypes.data, rows.ctypes.data, len(parts), k, xq.ctypes.data, xd.ctypes.data, ntok, y.ctypes.data, lpr))
    return y


def op_gateup_kq(gate_raw, up_raw, qtype, ff, k, xq, xd):
    L = lib()
    L.q3tts_op_gateup_kq.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    g = np.ascontiguousarray(gate_raw); u = np.ascontiguousarray(up_raw)
    xq = np.ascontiguousarray(xq, np.int8); xd = np.ascontiguousarray(xd, np.uint16)
    ntok = xq.shape[0]
    aq = np.zeros((ntok, ff), np.int8); ad = np.zeros((ntok, ff // 32), np.uint16)
    _chk(L.q3tts_op_gateup_kq(g.ctypes.data, u.ctypes.data, qtype, ff, k, xq.ctypes.data, xd.ctypes.data, ntok, aq.ctypes.data, ad.ctypes.data))
    return aq, ad


def text_nfc(text):
    """NFC as the tokenizer's normaliser applies it (q3tts_text_nfc)"""
    L = lib()
    L.q3tts_text_nfc.restype = C.c_int64
    L.q3tts_text_nfc.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
    raw = text.encode("utf-8", "surrogatepass")
    n = L.q3tts_text_nfc(raw, None, 0)
    buf = C.create_string_buffer(int(n))
    L.q3tts_text_nfc(raw, buf, n)
    return buf.raw[: n - 1].decode("utf-8", "surrogatepass")


def onnx_op_executable(op_type):
    L = lib()
    L.q3tts_onnx_op_executable.argtypes = [C.c_char_p]
    return bool(L.q3tts_onnx_op_executable(op_type.encode()))


class OnnxModel:
    """ONNX ModelProto through the C ABI's reader (q3tts_onnx_*): nodes, attributes, initialisers, I/O contract"""

    def __init__(self, path):
        L = lib()
        L.q3tts_onnx_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.q3tts_onnx_close.argtypes = [C.c_void_p]
        L.q3tts_onnx_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
        L.q3tts_onnx_summary.restype = C.c_int64
        L.q3tts_onnx_summary.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.q3tts_onnx_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)] + [C.POINTER(C.c_int32)] * 3
        L.q3tts_onnx_node_input.restype = C.c_char_p
        L.q3tts_onnx_node_input.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.q3tts_onnx_node_output.restype = C.c_char_p
        L.q3tts_onnx_node_output.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.q3tts_onnx_node_attr_ints.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p, C.c_int32]
        L.q3tts_onnx_node_attr_float.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_float)]
        L.q3tts_onnx_initializer.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.q3tts_onnx_op_kernel.restype = C.c_char_p
        L.q3tts_onnx_op_kernel.argtypes = [C.c_char_p]
        L.q3tts_onnx_decoder_contract.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        h = C.c_void_p()
        _chk(L.q3tts_onnx_open(path.encode(), C.byref(h)))
        self.h = h.value
        c = [C.c_int32() for _ in range(4)]
        L.q3tts_onnx_counts(self.h, *[C.byref(x) for x in c])
        self.n_nodes, self.n_initializers, self.n_inputs, self.n_outputs = [x.value for x in c]

    def close(self):
        if self.h:
            lib().q3tts_onnx_close(self.h)
        self.h = None

    def summary(self):
        n = lib().q3tts_onnx_summary(self.h, None, 0)
        buf = C.create_string_buffer(n)
        lib().q3tts_onnx_summary(self.h, buf, n)
        return buf.value.decode()

    def node(self, i):
        op, nm = C.c_char_p(), C.c_char_p()
        c = [C.c_int32() for _ in range(3)]
        _chk(lib().q3tts_onnx_node(self.h, i, C.byref(op), C.byref(nm), *[C.byref(x) for x in c]))
        return {"op_type": op.value.decode(), "name": nm.value.decode(),
                "inputs": [lib().q3tts_onnx_node_input(self.h, i, j).decode() for j in range(c[0].value)],
                "outputs": [lib().q3tts_onnx_node_output(self.h, i, j).decode() for j in range(c[1].value)], "n_attr": c[2].value}

    def attr_ints(self, i, name):
        buf = (C.c_int64 * 16)()
        n = lib().q3tts_onnx_node_attr_ints(self.h, i, name.encode(), buf, 16)
        return None if n < 0 else [buf[k] for k in range(min(n, 16))]

    def attr_float(self, i, name):
        f = C.c_float()
        return f.value if lib().q3tts_onnx_node_attr_float(self.h, i, name.encode(), C.byref(f)) == 1 else None

    def initializer(self, i):
        nm, dt, nd, data, nb = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_void_p(), C.c_int64()
        dims = (C.c_int64 * 8)()
        _chk(lib().q3tts_onnx_initializer(self.h, i, C.byref(nm), C.byref(dt), dims, C.byref(nd), C.byref(data), C.byref(nb)))
        raw = C.string_at(data.value, nb.value) if data.value and nb.value else b""
        return {"name": nm.value.decode(), "dtype": dt.value, "dims": [dims[k] for k in range(nd.value)], "raw": raw}

    def decoder_contract(self):
        buf = C.create_string_buffer(4096)
        rc = lib().q3tts_onnx_decoder_contract(self.h, buf, 4096)
        return rc == 0, buf.value.decode()


def onnx_op_kernel(op_type):
    lib().q3tts_onnx_op_kernel.restype = C.c_char_p
    lib().q3tts_onnx_op_kernel.argtypes = [C.c_char_p]
    r = lib().q3tts_onnx_op_kernel(op_type.encode())
    return r.decode() if r else None


class Tokenizer:  # utils/tokenizer.rs:4-38
    def __init__(self, tokenizer_json):
        L = lib()
        L.q3tts_tokenizer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.q3tts_tokenizer_close.argtypes = [C.c_void_p]
        L.q3tts_tokenizer_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32]
        L.q3tts_tokenizer_decode.restype = C.c_int64
        L.q3tts_tokenizer_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]
        L.q3tts_tokenizer_vocab_size.argtypes = [C.c_void_p]
        h = C.c_void_p()
        _chk(L.q3tts_tokenizer_open(tokenizer_json.encode(), C.byref(h)))
        self.h = h.value

    def close(self):
        if self.h:
            lib().q3tts_tokenizer_close(self.h)
        self.h = None

    def encode(self, text):
        raw = text.encode("utf-8")
        cap = max(16, 2 * len(raw) + 8)
        ids = np.zeros(cap, np.int32)
        n = lib().q3tts_tokenizer_encode(self.h, raw, _p(ids), cap)
        if n < 0:
            raise Q3Error(lib().q3tts_last_error().decode())
        return ids[:n].tolist()

    def decode(self, ids):
        a = np.ascontiguousarray(ids, np.int32)
        n = lib().q3tts_tokenizer_decode(self.h, _p(a), a.size, None, 0)
        if n < 0:
            raise Q3Error(lib().q3tts_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        lib().q3tts_tokenizer_decode(self.h, _p(a), a.size, buf, n + 1)
        return buf.raw[:n].decode("utf-8", errors="replace")
