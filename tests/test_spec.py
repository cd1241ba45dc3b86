"""Pins include/q3tts_spec.h's scalar helpers (through the oracle's exported wrappers) against numpy."""
import ctypes as C
import numpy as np


def _L(oracle):
    L = oracle.lib()
    L.q3o_spec_f16_to_f32_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.q3o_spec_f32_to_f16_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.q3o_spec_expf_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.q3o_spec_swiglu.restype = C.c_float
    L.q3o_spec_swiglu.argtypes = [C.c_float, C.c_float]
    L.q3o_spec_quant_block32.restype = C.c_uint16
    L.q3o_spec_quant_block32.argtypes = [C.c_void_p, C.c_void_p]
    L.q3o_spec_f32_to_bf16.restype = C.c_uint16
    L.q3o_spec_f32_to_bf16.argtypes = [C.c_float]
    L.q3o_spec_mrope_stream.argtypes = [C.c_int, C.c_void_p]
    return L


def test_f16_to_f32_exhaustive(oracle):
    L = _L(oracle)
    h = np.arange(65536, dtype=np.uint16)
    out = np.zeros(65536, np.float32)
    L.q3o_spec_f16_to_f32_n(h.ctypes.data, out.ctypes.data, 65536)
    ref = h.view(np.float16).astype(np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(out[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    assert np.all(np.isnan(out[nan]))


def test_f32_to_f16_rne(oracle):
    L = _L(oracle)
    rng = np.random.default_rng(0)
    xs = [rng.standard_normal(200000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 100.0, 7e4)]
    # every representable half, the midpoints between neighbours (ties), and edge cases
    h = np.arange(0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = ((h[:-1] + h[1:]) / 2).astype(np.float32)
    edge = np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, np.inf, -np.inf, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.99e-8, 6.1e-5], np.float32)
    x = np.concatenate(xs + [mids, -mids, h.astype(np.float32), edge])
    out = np.zeros(x.size, np.uint16)
    L.q3o_spec_f32_to_f16_n(x.ctypes.data, out.ctypes.data, x.size)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(out, ref)


def test_bf16_rne(oracle):
    L = _L(oracle)
    for v, exp in [(1.0, 0x3F80), (1.00390625, 0x3F80), (1.01171875, 0x3F82), (-2.5, 0xC020)]:
        assert L.q3o_spec_f32_to_bf16(v) == exp


def test_expf_accuracy_and_edges(oracle):
    L = _L(oracle)
    x = np.concatenate([np.linspace(-87, 88.7, 400001), np.array([-1e3, -87.01, 0.0, -0.0, 88.72, 89.0, 1e3])]).astype(np.float32)
    y = np.zeros(x.size, np.float32)
    L.q3o_spec_expf_n(x.ctypes.data, y.ctypes.data, x.size)
    with np.errstate(over="ignore"):
        ref = np.exp(x.astype(np.float64))
    inner = (x >= -87) & (x <= 88.72)
    rel = np.abs(y[inner] - ref[inner]) / ref[inner]
    assert rel.max() < 3e-7
    assert np.all(y[x < -87] == 0.0)
    assert np.all(np.isinf(y[x > 88.72]))
    assert y[np.where(x == 0.0)[0][0]] == 1.0


def test_swiglu_matches_definition(oracle):
    L = _L(oracle)
    rng = np.random.default_rng(1)
    for g, u in rng.standard_normal((200, 2)) * 4:
        ref = (g / (1 + np.exp(-np.float64(g)))) * u
        assert abs(L.q3o_spec_swiglu(float(g), float(u)) - ref) < 2e-6 * max(1.0, abs(ref))
    assert L.q3o_spec_swiglu(-200.0, 3.0) == 0.0  # exp overflow branch: g/inf = -0


def test_quant_block32_follows_ggml_q8_0(oracle):
    L = _L(oracle)
    rng = np.random.default_rng(2)
    for scale in (1e-3, 1.0, 50.0):
        x = (rng.standard_normal(32) * scale).astype(np.float32)
        q = np.zeros(32, np.int8)
        d16 = L.q3o_spec_quant_block32(x.ctypes.data, q.ctypes.data)
        amax = np.abs(x).max()
        d = np.float32(amax) / np.float32(127.0)
        idv = np.float32(1.0) / d
        assert d16 == np.float32(d).astype(np.float16).view(np.uint16)
        assert np.array_equal(q, np.rint(x * idv).astype(np.int8))
    z = np.zeros(32, np.float32)
    q = np.ones(32, np.int8)
    assert L.q3o_spec_quant_block32(z.ctypes.data, q.ctypes.data) == 0 and not q.any()


def test_mrope_sector_map(oracle):
    L = _L(oracle)
    sec = np.array([24, 20, 20, 0], np.int32)
    got = [L.q3o_spec_mrope_stream(i, sec.ctypes.data) for i in range(64)]
    assert got == [0] * 24 + [1] * 20 + [2] * 20
    zero = np.zeros(4, np.int32)
    assert all(L.q3o_spec_mrope_stream(i, zero.ctypes.data) == 0 for i in range(64))
