"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libq3tts.so; the expected
values come from the CPU oracle on the same seeded inputs and from the committed golden fixtures.
Bars: codec tokens / int8 quantisation / argmax indices bit-exact; f32 intermediates bit-exact (the arithmetic spec fixes
every reduction order); PCM within 1e-4 RMS (BASELINE.json north_star)."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "oracle_tiny_v1.npz")
PCM_RMS_TOL = 1e-4


def _q8_encode(w):
    n, k = w.shape
    wb = w.reshape(n, k // 32, 32)
    d = (np.abs(wb).max(-1) / 127).astype(np.float32)
    idv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0).astype(np.float32)
    q = np.rint(wb * idv[..., None]).astype(np.int8)
    raw = np.zeros((n, k // 32, 34), np.uint8)
    raw[..., :2] = d.astype(np.float16).view(np.uint8).reshape(n, k // 32, 2)
    raw[..., 2:] = q.view(np.uint8)
    return raw


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("n,k,ntok", [(32, 256, 1), (96, 2048, 1), (4096, 2048, 1), (256, 6144, 3), (2048, 1024, 2), (160, 3072, 9), (40, 512, 5),
                                      (96, 2048, 16), (200, 6144, 33), (64, 1024, 70), (2176, 2048, 43), (33, 256, 12),
                                      # full-size batched shapes (Q3TTS-1.7B-synth): predictor down-proj (two super-segments), talker down-proj
                                      # (three super-segments = multi-slab writes), predictor gate/up-sized rows at K = 1024
                                      (1024, 3072, 33), (2048, 6144, 64), (3072, 1024, 64), (4096, 2048, 256),
                                      # the one-tile kernel (K <= 1024, one 32-token tile per workgroup) with 1, 2, 3 and 4 waves per workgroup, ragged
                                      # last tiles, and a row count that is not a multiple of 4 (falls back to the resident-tile kernel)
                                      (64, 256, 32), (128, 512, 40), (96, 768, 24), (256, 1024, 17), (2048, 1024, 128), (100, 1024, 50), (34, 512, 20)])  # ntok>=16: int8 MFMA path; 9..15: token sweep
def test_gemv_q8_bit_exact(gpu, oracle, n, k, ntok):
    rng = np.random.default_rng(n + k)
    raw = _q8_encode((rng.standard_normal((n, k)) * 0.02).astype(np.float32))
    x = (rng.standard_normal((ntok, k)) * rng.uniform(0.1, 4)).astype(np.float32)
    L = oracle.lib()
    xq = np.zeros((ntok, k), np.int8); xd = np.zeros((ntok, k // 32), np.uint16); yo = np.zeros((ntok, n), np.float32)
    for t in range(ntok):
        L.q3o_quant_act(x[t].ctypes.data, k, xq[t].ctypes.data, xd[t].ctypes.data)
        L.q3o_matvec(8, raw.ctypes.data, n, k, xq[t].ctypes.data, xd[t].ctypes.data, None, yo[t].ctypes.data)
    for lpr in (2, 4, 8, 0):
        assert np.array_equal(_bits(gpu.op_gemv_q8(raw, n, k, xq, xd, lpr)), _bits(yo)), lpr


def _rand_kq_rows(rng, qtype, n, k):
    """n rows of k weights as raw GGUF blocks with RANDOM fields (every bit pattern of quants, scales and mins is a valid block): harsher than weights
    quantised from a Gaussian -- negative / extreme Q6_K sub-block scales, all 6 bits of the Q5_K scales and mins, dense high-bit planes"""
    nsb = n * (k // 256)
    f16 = lambda lo, hi: (rng.uniform(lo, hi, nsb).astype(np.float16)).view(np.uint16)
    if qtype == 8:
        blk = np.zeros((n * (k // 32), 34), np.uint8)
        blk[:, :2] = (rng.uniform(0.001, 0.05, n * (k // 32)).astype(np.float16)).view(np.uint16).view(np.uint8).reshape(-1, 2)
        blk[:, 2:] = rng.integers(0, 256, (n * (k // 32), 32), dtype=np.uint8)
        return blk.reshape(-1)
    if qtype == 13:  # {f16 d, dmin; u8 scales[12]; u8 qh[32]; u8 qs[128]}
        blk = rng.integers(0, 256, (nsb, 176), dtype=np.uint8)
        blk[:, 0:2] = f16(0.0005, 0.01).view(np.uint8).reshape(-1, 2); blk[:, 2:4] = f16(0.0005, 0.01).view(np.uint8).reshape(-1, 2)
        return blk.reshape(-1)
    blk = rng.integers(0, 256, (nsb, 210), dtype=np.uint8)  # {u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d}
    blk[:, 208:210] = f16(0.0002, 0.004).view(np.uint8).reshape(-1, 2)
    return blk.reshape(-1)


def _oracle_rows(L, qtype, raw, n, k, xq, xd):
    y = np.zeros((xq.shape[0], n), np.float32)
    for t in range(xq.shape[0]):
        L.q3o_matvec(qtype, raw.ctypes.data, n, k, xq[t].ctypes.data, xd[t].ctypes.data, None, y[t].ctypes.data)
    return y


@pytest.mark.parametrize("k,ntok", [(256, 1), (1024, 3), (2048, 8), (3072, 12), (2048, 17), (1024, 64), (3072, 33), (6144, 64), (2048, 70)])
def test_kquant_gemv_and_matrix_core_gemm_bit_exact(gpu, oracle, k, ntok):
    """Packed K-quant planes at the op level, random block contents: Q5_K, Q6_K, Q8_0 and a fused three-tensor matrix (Q5_K + Q6_K + Q8_0 row groups, the
    per-workgroup type dispatch) through every lanes-per-row form of the GEMV and, from 16 tokens, the matrix-core GEMM (k_gemm_kq_mfma) -- against
    the oracle's block arithmetic (q3o_matvec on the raw GGUF rows), bit for bit."""
    rng = np.random.default_rng(1000 + k + ntok)
    L = oracle.lib()
    x = (rng.standard_normal((ntok, k)) * rng.uniform(0.1, 4)).astype(np.float32)
    xq = np.zeros((ntok, k), np.int8); xd = np.zeros((ntok, k // 32), np.uint16)
    for t in range(ntok):
        L.q3o_quant_act(x[t].ctypes.data, k, xq[t].ctypes.data, xd[t].ctypes.data)
    raws = {q: _rand_kq_rows(rng, q, 96, k) for q in (13, 14, 8)}
    refs = {q: _oracle_rows(L, q, raws[q], 96, k, xq, xd) for q in (13, 14, 8)}
    for lpr in (2, 4, 8, 0):
        for q in (13, 14):
            got = gpu.op_gemv_kq([(raws[q], q, 96)], k, xq, xd, lpr)
            assert np.array_equal(_bits(got), _bits(refs[q])), (q, lpr)
        got = gpu.op_gemv_kq([(raws[13], 13, 96), (raws[14], 14, 96), (raws[8], 8, 96)], k, xq, xd, lpr)   # q, k (Q5_K) + v (Q6_K) style fusion, plus Q8_0 rows
        assert np.array_equal(_bits(got), _bits(np.concatenate([refs[13], refs[14], refs[8]], 1))), ("mixed", lpr)


@pytest.mark.parametrize("qtype,ff,k,ntok", [(13, 3072, 1024, 64), (13, 6144, 2048, 33), (14, 3072, 1024, 40), (14, 512, 2048, 130), (13, 512, 2048, 16)])
def test_kquant_gateup_matrix_core_bit_exact(gpu, oracle, qtype, ff, k, ntok):
    """the fused gate / up + SwiGLU + int8 quantisation form of k_gemm_kq_mfma (two passes over the token tiles, gate sums parked in LDS) at the full
    model's shapes, random Q5_K / Q6_K blocks; oracle: matvec rows -> q3_swiglu -> quant_act"""
    import ctypes as C
    rng = np.random.default_rng(qtype + ff + k + ntok)
    L = oracle.lib()
    L.q3o_spec_swiglu_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    gate, up = _rand_kq_rows(rng, qtype, ff, k), _rand_kq_rows(rng, qtype, ff, k)
    x = (rng.standard_normal((ntok, k)) * rng.uniform(0.5, 3)).astype(np.float32)
    xq = np.zeros((ntok, k), np.int8); xd = np.zeros((ntok, k // 32), np.uint16)
    eq = np.zeros((ntok, ff), np.int8); ed = np.zeros((ntok, ff // 32), np.uint16)
    g = np.zeros(ff, np.float32); u = np.zeros(ff, np.float32)
    for t in range(ntok):
        L.q3o_quant_act(x[t].ctypes.data, k, xq[t].ctypes.data, xd[t].ctypes.data)
        L.q3o_matvec(qtype, gate.ctypes.data, ff, k, xq[t].ctypes.data, xd[t].ctypes.data, None, g.ctypes.data)
        L.q3o_matvec(qtype, up.ctypes.data, ff, k, xq[t].ctypes.data, xd[t].ctypes.data, None, u.ctypes.data)
        a = np.zeros(ff, np.float32)
        L.q3o_spec_swiglu_vec(g.ctypes.data, u.ctypes.data, ff, a.ctypes.data)
        L.q3o_quant_act(a.ctypes.data, ff, eq[t].ctypes.data, ed[t].ctypes.data)
    aq, ad = gpu.op_gateup_kq(gate, up, qtype, ff, k, xq, xd)
    assert np.array_equal(ad, ed)
    assert np.array_equal(aq, eq)


@pytest.mark.parametrize("ff,k,ntok", [(3072, 1024, 64), (6144, 2048, 64), (6144, 2048, 33), (3072, 1024, 130), (512, 2048, 16)])
def test_gateup_mfma_bit_exact(gpu, oracle, ff, k, ntok):
    """fused gate/up GEMM + SwiGLU + int8 quantisation on the matrix cores (the batched layer path of configs C3 / 256 slots) at the
    full model's shapes: predictor K = 1024 / ff = 3072, talker K = 2048 / ff = 6144; token counts with a ragged last 32-token tile and
    more than 4 tiles (the workgroup's gate-sum parking limit).  Oracle: matvec rows -> q3_swiglu -> quant_act, bit for bit."""
    import ctypes as C
    rng = np.random.default_rng(ff + k + ntok)
    raw = _q8_encode((rng.standard_normal((2 * ff, k)) * 0.03).astype(np.float32))
    x = (rng.standard_normal((ntok, k)) * rng.uniform(0.5, 3)).astype(np.float32)
    L = oracle.lib()
    L.q3o_spec_swiglu.restype = C.c_float; L.q3o_spec_swiglu.argtypes = [C.c_float, C.c_float]
    xq = np.zeros((ntok, k), np.int8); xd = np.zeros((ntok, k // 32), np.uint16)
    eq = np.zeros((ntok, ff), np.int8); ed = np.zeros((ntok, ff // 32), np.uint16)
    gu = np.zeros(2 * ff, np.float32)
    L.q3o_spec_swiglu_vec.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    for t in range(ntok):
        L.q3o_quant_act(x[t].ctypes.data, k, xq[t].ctypes.data, xd[t].ctypes.data)
        L.q3o_matvec(8, raw.ctypes.data, 2 * ff, k, xq[t].ctypes.data, xd[t].ctypes.data, None, gu.ctypes.data)
        a = np.zeros(ff, np.float32)
        L.q3o_spec_swiglu_vec(gu.ctypes.data, (gu[ff:]).ctypes.data, ff, a.ctypes.data)
        L.q3o_quant_act(a.ctypes.data, ff, eq[t].ctypes.data, ed[t].ctypes.data)
    aq, ad = gpu.op_gateup_q8(raw, ff, k, xq, xd)
    assert np.array_equal(ad, ed)
    assert np.array_equal(aq, eq)


def _float_weights(rng, ty, n, k):
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    w[rng.integers(0, n, 8), rng.integers(0, k, 8)] = 0.0
    if ty == 0:
        return w
    if ty == 1:
        return w.astype(np.float16)
    u = w.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)   # bf16, round to nearest even


@pytest.mark.gpu
@pytest.mark.parametrize("ty,n,k,ntok", [(30, 192, 2048, 32), (30, 100, 1024, 13), (30, 64, 6144, 45), (30, 130, 3072, 64), (1, 128, 2048, 33),
                                         (0, 70, 1024, 12), (30, 64, 2048, 3), (30, 256, 2048, 200),
                                         # >= 96 tokens: k_gemm_float_mfma's wide form at the shapes config C5 puts through it (32 x 133-row prefill):
                                         # K = 6144 (talker down-projection, 3 super-segments), 3072 / 1024 (predictor), N >= 2048, bf16 and f16
                                         (30, 2048, 6144, 96), (1, 2048, 6144, 133), (30, 2048, 3072, 160), (1, 2112, 1024, 97), (30, 4096, 1024, 128)])
def test_matmul_float_bit_exact(gpu, oracle, ty, n, k, ntok):
    """float-weight matmul (spec S3 float form): the matrix-core kernel (K = 1 f32 MFMA chains, ntok >= 12) and the one-wave-per-row
    GEMV (fewer tokens) against the oracle's row_dot, bit for bit; also a 64-aligned row sub-range as the predictor head uses."""
    rng = np.random.default_rng(n * 7 + k + ntok)
    w = _float_weights(rng, ty, n, k)
    x = (rng.standard_normal((ntok, k)) * rng.uniform(0.1, 4)).astype(np.float32)
    x[0, :64] = 1e-30 * rng.standard_normal(64)            # products and partial sums in the denormal range
    yo = np.zeros((ntok, n), np.float32)
    L = oracle.lib()
    for t in range(ntok):
        L.q3o_matvec(ty, w.ctypes.data, n, k, None, None, x[t].ctypes.data, yo[t].ctypes.data)
    yg = gpu.op_matmul_float(w, ty, n, k, x)
    assert np.array_equal(_bits(yg), _bits(yo))
    if n > 64:
        yg2 = gpu.op_matmul_float(w, ty, n, k, x, row0=64, nrows=n - 64)
        assert np.array_equal(_bits(yg2), _bits(yo[:, 64:]))


@pytest.mark.parametrize("d", [256, 1024, 2048])
def test_rmsnorm_quant_bit_exact(gpu, oracle, d):
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((4, d)) * 3).astype(np.float32)
    x[3] = 0.0   # all-zero row: d = 0 branch of the block quantiser
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    xq, xd, xn = gpu.op_rmsnorm_quant(x, g, 1e-6)
    L = oracle.lib()
    for t in range(4):
        y = np.zeros(d, np.float32); L.q3o_rmsnorm(x[t].ctypes.data, g.ctypes.data, d, 1e-6, y.ctypes.data)
        q = np.zeros(d, np.int8); dd = np.zeros(d // 32, np.uint16); L.q3o_quant_act(y.ctypes.data, d, q.ctypes.data, dd.ctypes.data)
        assert np.array_equal(_bits(y), _bits(xn[t])) and np.array_equal(q, xq[t]) and np.array_equal(dd, xd[t])


def test_swiglu_argmax_project(gpu, oracle, tiny_model, vivian):
    import ctypes as C
    rng = np.random.default_rng(5)
    ff = 512
    gu = (rng.standard_normal((3, 2 * ff)) * 3).astype(np.float32)
    gu[0, :4] = [-200.0, 200.0, 0.0, -0.0]
    aq, ad = gpu.op_swiglu_quant(gu, ff)
    L = oracle.lib()
    L.q3o_spec_swiglu.restype = C.c_float; L.q3o_spec_swiglu.argtypes = [C.c_float, C.c_float]
    for t in range(3):
        a = np.array([L.q3o_spec_swiglu(float(gu[t, i]), float(gu[t, ff + i])) for i in range(ff)], np.float32)
        q = np.zeros(ff, np.int8); dd = np.zeros(ff // 32, np.uint16); L.q3o_quant_act(a.ctypes.data, ff, q.ctypes.data, dd.ctypes.data)
        assert np.array_equal(q, aq[t]) and np.array_equal(dd, ad[t])
    lg = np.zeros(3072, np.float32); lg[[5, 9, 2150, 2500]] = [3.0, 3.0, 9.0, 100.0]
    assert gpu.op_argmax(lg, 0, 2160) == 2150 and gpu.op_argmax(lg, 0, 2160, mask_idx=2150) == 5 and gpu.op_argmax(lg, 6, 2160, 2150) == 9
    assert gpu.op_argmax(np.full(300, -np.inf, np.float32), 10, 290) == 10
    for _ in range(20):
        v = rng.standard_normal(2160).astype(np.float32); v[rng.integers(0, 2160, 40)] = v.max()
        assert gpu.op_argmax(v, 0, 2160) == int(np.argmax(v))
    import ggml_ref as G
    _, t = G.read_gguf(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"))
    W = np.array(t["proj.weight"][2]).view(np.float32).reshape(-1, 2048); b = np.array(t["proj.bias"][2]).view(np.float32)
    oa = oracle.Assets(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"))
    assert np.array_equal(_bits(gpu.op_project(vivian, W, b)), _bits(oa.project(vivian)[: b.size]))
    oa.close()


@pytest.mark.gpu
def test_device_sampler_matches_oracle_sampler(gpu, oracle):
    """k_sample (sort, top-k, softmax, top-p, ChaCha12 draw) vs the oracle's restatement of llama/mod.rs:666-776: same tokens"""
    rng = np.random.default_rng(77)
    cases = [(0.7, 40, 0.9), (1.0, 0, 1.0), (0.3, 5, 0.5), (1.5, 0, 0.95), (0.7, 1, 0.9), (0.05, 40, 0.9), (2.0, 3000, 0.999), (0.0, 40, 0.9)]
    for ci, (t, k, p) in enumerate(cases):
        for n in (2160, 300, 4096):
            lg = (rng.standard_normal(n) * (1.0 + ci)).astype(np.float32)
            tie = rng.integers(0, n, 12)
            lg[tie] = lg.max()          # ties at the top exercise the stable order
            lg[rng.integers(0, n, 5)] = -0.0
            mask = int(tie[0]) if ci % 2 else -1
            seed = 42 + ci
            nd = 40                     # > 16 draws crosses a ChaCha block boundary
            got = gpu.op_sample(lg, t, k, p, seed, n_draws=nd, mask_idx=mask)
            ref = lg.copy()
            if mask >= 0:
                ref[mask] = -np.inf
            exp, state = [], None
            for _ in range(nd):
                tok, state = oracle.sample(ref, 0, n, t, k, p, seed, state)
                exp.append(tok)
            exp = np.array(exp, np.int32)
            assert np.array_equal(got, exp), (t, k, p, n, got[:8], exp[:8])


def _tf_parity(gpu, oracle, path, d, n_pre, n_dec, rows):
    rng = np.random.default_rng(9)
    om = oracle.Model(path, 4096)
    gm = gpu.TfContext(path, 4096, 16)
    xs = (rng.standard_normal((n_pre + n_dec, d)) * 0.3).astype(np.float32)
    pos = np.array([[t, t, t, 0] for t in range(n_pre + n_dec)], np.int32)
    hg, lg = gm.eval(xs[:n_pre], pos[:n_pre], 0, rows)       # chunked prefill (16 tokens per launch)
    for t in range(n_pre):
        ho, lo = om.eval(xs[t], pos[t], d, 0, rows)
        assert np.array_equal(_bits(ho), _bits(hg[t])) and np.array_equal(_bits(lo), _bits(lg[t])), t
    for t in range(n_pre, n_pre + n_dec):
        h1, l1 = gm.eval(xs[t:t + 1], pos[t:t + 1], 0, rows)
        ho, lo = om.eval(xs[t], pos[t], d, 0, rows)
        assert np.array_equal(_bits(ho), _bits(h1[0])) and np.array_equal(_bits(lo), _bits(l1[0])), t
    gm.clear(); om.clear()   # llama_memory_seq_rm(-1,0,-1) semantics: positions restart
    h2, _ = gm.eval(xs[:1], pos[:1], 0, 0)
    ho, _ = om.eval(xs[0], pos[0], d, 0, 0)
    assert np.array_equal(_bits(ho), _bits(h2[0]))
    gm.close(); om.close()


def test_tf_eval_vs_transformers_fixture(gpu):
    """ORACLE-FREE: q3tts_tf_eval (HIP float-weight path: K = 1 f32 MFMA GEMMs, attention kernels, norms) against hidden states and
    logits computed by the `transformers` Qwen3Model (fixture generated in the build container by
    tests/golden/make_transformers_fixture.py).  Nothing from oracle/ or include/q3tts_spec.h's CPU side takes part, so an error
    shared by the oracle and the kernels (common header, common author) cannot hide here.  Tolerance 2e-3 relative to max(1, |ref|_inf):
    the engine keeps K/V in f16 and fixes its own summation orders.  Run as one 24-token prefill (batched kernels, attention over the
    chunk) and again token by token (decode kernels)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "qwen3_tf_expected.npz"))
    D, L, H, HKV, FF, V, N = [int(v) for v in g["meta"]]
    path = os.path.join(ROOT, "tests", "golden", "qwen3_tf_f16.gguf")
    pos = np.array([[t, t, t, 0] for t in range(N)], np.int32)

    def check(h, lg, i):
        assert np.abs(h - g["hidden"][i]).max() < 2e-3 * max(1.0, np.abs(g["hidden"][i]).max()), i
        assert np.abs(lg - g["logits"][i]).max() < 2e-3 * max(1.0, np.abs(g["logits"][i]).max()), i

    gm = gpu.TfContext(path, 64, 32)
    hg, lg = gm.eval(g["x"], pos, 0, V)
    for i in range(N):
        check(hg[i], lg[i], i)
    gm.clear()
    for i in range(N):
        h1, l1 = gm.eval(g["x"][i:i + 1], pos[i:i + 1], 0, V)
        check(h1[0], l1[0], i)
    gm.close()
    # the Q8_0 copy of the same weights through the int8 kernels (batched MFMA form for the 24-token prefill, fused 5-launch decode
    # path token by token): within the 8-bit quantisation noise of the float32 reference (measured 3e-2, bar 6e-2)
    gq = gpu.TfContext(os.path.join(ROOT, "tests", "golden", "qwen3_tf_q8_0.gguf"), 64, 32)

    def check_q(h, lg, i):
        assert np.abs(h - g["hidden"][i]).max() < 6e-2 * max(1.0, np.abs(g["hidden"][i]).max()), i
        assert np.abs(lg - g["logits"][i]).max() < 6e-2 * max(1.0, np.abs(g["logits"][i]).max()), i

    hq, lq = gq.eval(g["x"], pos, 0, V)
    for i in range(N):
        check_q(hq[i], lq[i], i)
    gq.clear()
    for i in range(N):
        h1, l1 = gq.eval(g["x"][i:i + 1], pos[i:i + 1], 0, V)
        check_q(h1[0], l1[0], i)
    gq.close()
    # the Q5_K_M copy (Q5_K + Q6_K rows, packed planes; 24-token prefill = k_gemm_kq_mfma, then the fused K-quant decode path token by token) against
    # `transformers` run on the DEQUANTISED weights (hidden_q5 / logits_q5): only the activation quantisation separates the two (bar 6e-2)
    g5 = gpu.TfContext(os.path.join(ROOT, "tests", "golden", "qwen3_tf_q5_k_m.gguf"), 64, 32)

    def check_5(h, lg, i):
        assert np.abs(h - g["hidden_q5"][i]).max() < 6e-2 * max(1.0, np.abs(g["hidden_q5"][i]).max()), i
        assert np.abs(lg - g["logits_q5"][i]).max() < 6e-2 * max(1.0, np.abs(g["logits_q5"][i]).max()), i

    h5, l5 = g5.eval(g["x"], pos, 0, V)
    for i in range(N):
        check_5(h5[i], l5[i], i)
    g5.clear()
    for i in range(N):
        h1, l1 = g5.eval(g["x"][i:i + 1], pos[i:i + 1], 0, V)
        check_5(h1[0], l1[0], i)
    g5.close()


def test_predictor_loop_vs_transformers_fixture(gpu):
    """ORACLE-FREE: the code-predictor loop (/root/reference/src/tts/engine.rs:596-640 -- two prompt rows, then 15 greedy passes, pass i reading slice i of
    the stacked output matrix and feeding its code back through codebook table i) driven through q3tts_tf_eval, against what `transformers`' own `generate()`
    emits for its public implementation of that loop (tests/golden/make_predictor_fixture.py; VERDICT r2 "missing" 2).  Every pass's logits within 2e-3 of
    max(1, |ref|) and the 15 codes equal; run with the two prompt rows as one 2-token evaluation and again one row at a time."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "predictor_tf_expected.npz"))
    D, L, H, HKV, FF, V, G, _ = [int(v) for v in g["meta"]]
    tm = gpu.TfContext(os.path.join(ROOT, "tests", "golden", "predictor_tf_f16.gguf"), 64, 32)
    for split in (False, True):
        tm.clear()
        if split:
            tm.eval(g["x"][0:1], np.array([[0, 0, 0, 0]], np.int32), 0, V)
            _, lg = tm.eval(g["x"][1:2], np.array([[1, 1, 1, 0]], np.int32), 0, V)
            lg = lg[0]
        else:
            _, lg = tm.eval(g["x"], np.array([[0, 0, 0, 0], [1, 1, 1, 0]], np.int32), 0, V)
            lg = lg[1]
        codes = []
        for i in range(G - 1):
            assert np.abs(lg - g["logits"][i]).max() < 2e-3 * max(1.0, np.abs(g["logits"][i]).max()), (split, i)
            codes.append(int(np.argmax(lg)))
            if i < G - 2:
                emb = g["tables"][i][codes[-1]].astype(np.float32)[None]
                _, lg = tm.eval(emb, np.array([[i + 2, i + 2, i + 2, 0]], np.int32), (i + 1) * V, (i + 2) * V)
                lg = lg[0]
        assert codes == g["codes"].tolist(), split
    tm.close()


def test_codec_vs_transformers_code2wav_fixture(gpu):
    """ORACLE-FREE: csrc/codec.hip (RVQ sum, sliding-window transformer, ConvNeXt up-sampling, SnakeBeta / transposed-conv / residual-unit blocks, output
    conv) on tests/golden/code2wav_tf.gguf against the waveform the `transformers` Qwen3OmniMoeCode2Wav computed for the same codes and weights
    (tests/golden/make_code2wav_fixture.py; the public analogue of qwen3_tts_decoder.onnx, /root/reference/src/models/onnx.rs:342-458).  Streamed in
    4-frame chunks like engine.rs:505-541, and once more through the grouped path; compared modulo the documented transposed-conv trim (a 45-sample
    shift, left edge skipped).  PCM bar of north_star: 1e-4 RMS."""
    from test_golden_cpu import C2W_EXP, C2W_GGUF, code2wav_compare
    g = np.load(C2W_EXP)
    codes = g["codes"]
    d = gpu.Decoder(C2W_GGUF, n_streams=3, max_frames=4, max_group=2)
    d.reset(1)
    ours = np.concatenate([d.decode(codes[o:o + 4], stream=1, is_last=o + 4 >= codes.shape[0]).copy() for o in range(0, codes.shape[0], 4)])
    assert ours.size == codes.shape[0] * d.spf
    rms, mx = code2wav_compare(ours, g)
    assert rms < 1e-5 and mx < 1e-4, (rms, mx)
    d.reset(0); d.reset(2)
    grp = np.concatenate([d.decode_group([2, 0], np.stack([codes[o:o + 4], codes[o:o + 4]])) for o in range(0, codes.shape[0], 4)], axis=1)
    for row in grp:
        rms, mx = code2wav_compare(row, g)
        assert rms < 1e-5 and mx < 1e-4, (rms, mx)
    d.close()


def test_transformer_fused_path_bit_exact(gpu, oracle, tiny_model):
    _tf_parity(gpu, oracle, os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf"), 2048, 37, 5, 2160)
    _tf_parity(gpu, oracle, os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_predictor.gguf"), 256, 2, 14, 2048)


def test_transformer_unfused_path_bit_exact(gpu, oracle, tiny_model):
    os.environ["Q3_UNFUSED"] = "1"
    try:
        _tf_parity(gpu, oracle, os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf"), 2048, 20, 3, 2160)
    finally:
        del os.environ["Q3_UNFUSED"]


def test_attention_beyond_one_chunk(gpu, oracle, tiny_model):
    """context > 256 positions exercises the chunk-merge branch of spec S7 (rule: a rare branch needs its own test)."""
    path = os.path.join(tiny_model, "gguf_q8_0", "qwen3_tts_talker.gguf")
    rng = np.random.default_rng(10)
    om = oracle.Model(path, 4096); gm = gpu.TfContext(path, 4096, 64)
    n = 300
    xs = (rng.standard_normal((n + 2, 2048)) * 0.3).astype(np.float32)
    pos = np.array([[t, t, t, 0] for t in range(n + 2)], np.int32)
    hg, _ = gm.eval(xs[:n], pos[:n])
    for t in range(n):
        ho, _ = om.eval(xs[t], pos[t], 2048)
        if t in (0, 255, 256, 257, n - 1):
            assert np.array_equal(_bits(ho), _bits(hg[t])), t
    for t in (n, n + 1):   # fused decode kernel with 2 chunks
        h1, _ = gm.eval(xs[t:t + 1], pos[t:t + 1]); ho, _ = om.eval(xs[t], pos[t], 2048)
        assert np.array_equal(_bits(ho), _bits(h1[0]))
    gm.close(); om.close()


def gpu_mod():
    import q3tts
    return q3tts


@pytest.fixture(scope="module")
def tiny_engines(gpu, oracle, tiny_model):
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=4, max_steps=64, load_codec=True)
    oe = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), os.path.join(tiny_model, "onnx", "q3tts_codec.gguf"), 4)
    yield ge, oe
    ge.close(); oe.close()


def test_engine_matches_golden_and_oracle(tiny_engines, vivian):
    ge, oe = tiny_engines
    g = np.load(GOLD)
    prompt = ge.assets.build_core(np.arange(100, 108, dtype=np.int32), lang_id=2055, spk_emb=vivian)
    r = ge.generate_batch([prompt], max_steps=12, temperature=0.0, seed=42, mask_eos=True, want_pcm=True)[0]
    assert np.array_equal(r["codes"], g["greedy_codes"])                       # bit-exact codec tokens vs committed fixture
    assert r["pcm"].size == int(g["greedy_pcm_stats"][0])
    assert np.sqrt(np.mean((r["pcm"][:4096] - g["greedy_pcm_head"]) ** 2)) < PCM_RMS_TOL
    oc, opcm = oe.generate(prompt, max_steps=12, want_pcm=True)
    assert np.array_equal(oc, r["codes"]) and np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL
    # temperature > 0: device sampler inside the frame graph (seeded ChaCha12 stream)
    rs = ge.generate_batch([prompt], max_steps=12, temperature=0.7, top_k=40, top_p=0.9, seed=42, mask_eos=True)[0]
    assert np.array_equal(rs["codes"], g["sampled_codes"])
    # determinism
    assert np.array_equal(ge.generate_batch([prompt], max_steps=12)[0]["codes"], r["codes"])


def test_engine_eos_and_ragged_lengths(tiny_engines, vivian):
    ge, oe = tiny_engines
    rng = np.random.default_rng(3)
    for n_text, steps in ((1, 5), (40, 9), (3, 10), (17, 7)):   # frame counts 5,9,10,7: leftover-flush and no-flush chunker paths
        prompt = ge.assets.build_core(rng.integers(0, 4000, n_text).astype(np.int32), lang_id=2055, spk_emb=vivian)
        r = ge.generate_batch([prompt], max_steps=steps, mask_eos=False, want_pcm=True)[0]    # natural EOS allowed
        oc, opcm = oe.generate(prompt, max_steps=steps, mask_eos=False, want_pcm=True)
        assert np.array_equal(oc, r["codes"])
        assert r["pcm"].size == opcm.size and (opcm.size == 0 or np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL)
    prompt = ge.assets.build_core(np.array([5], np.int32), lang_id=2055, spk_emb=vivian)
    assert ge.generate_batch([prompt], max_steps=0)[0]["codes"].shape == (0, 16)          # empty generation


def test_engine_batch_equals_singles(tiny_engines, vivian):
    """request-level batching: B lock-stepped sequences (different prompt lengths, one clone prompt) give the same tokens as
    one-at-a-time runs -- batch invariance of the arithmetic spec."""
    ge, oe = tiny_engines
    rng = np.random.default_rng(4)
    prompts = [ge.assets.build_core(rng.integers(0, 4000, n).astype(np.int32), lang_id=2055, spk_emb=vivian) for n in (4, 19, 33)]
    prompts.append(ge.assets.build_clone(rng.integers(0, 4000, 6).astype(np.int32), rng.integers(0, 2048, 5 * 16), rng.integers(0, 4000, 3), vivian))
    batch = ge.generate_batch(prompts, max_steps=10, mask_eos=True, want_pcm=True)
    for p, r in zip(prompts, batch):
        oc, opcm = oe.generate(p, max_steps=10, want_pcm=True)
        assert np.array_equal(oc, r["codes"])
        assert np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL
    two = ge.generate_batch(prompts[:2], max_steps=10, mask_eos=True)
    assert all(np.array_equal(a["codes"], b["codes"]) for a, b in zip(two, batch))


def test_scheduler_continuous_batching_matches_oracle(tiny_engines, vivian):
    """BASELINE config 3 mechanics: 9 requests (ragged prompts, ragged lengths, greedy and sampled, natural EOS allowed) queue into
    4 slots; sequences retire and new ones are admitted mid-flight.  Every request must equal the oracle run alone."""
    ge, oe = tiny_engines
    rng = np.random.default_rng(11)
    specs = []
    for i, (n_text, steps) in enumerate(((3, 5), (30, 17), (9, 8), (1, 12), (44, 4), (12, 9), (7, 0), (20, 13), (5, 6))):
        prompt = ge.assets.build_core(rng.integers(0, 4000, n_text).astype(np.int32), lang_id=2055, spk_emb=vivian)
        temp = 0.8 if i % 3 == 1 else 0.0
        specs.append(dict(prompt=prompt, max_steps=steps, temperature=temp, top_k=(0 if i == 4 else 30), top_p=0.85, seed=100 + i,
                          mask_eos=(i % 2 == 0), want_pcm=(i != 5)))
    ids = [ge.submit(**sp) for sp in specs[:6]]
    seen_partial = False
    busy, it = True, 0
    while busy:
        busy = ge.sched_step()
        it += 1
        if it == 2:
            ids += [ge.submit(**sp) for sp in specs[6:]]          # arrivals while the first wave is running
            busy = True
        st = ge.poll(ids[1])
        if st["state"] == gpu_mod().REQ_RUNNING and st["n_frames"] > 0:
            part, _ = ge.fetch(ids[1], 0, st["n_frames"])        # streaming read of a running request
            seen_partial = seen_partial or part.shape[0] > 0
    assert seen_partial
    stats = ge.stats()
    assert stats["sched_steps"] >= 4
    for sp, rid in zip(specs, ids):
        assert ge.poll(rid)["state"] == gpu_mod().REQ_DONE
        r = ge.result(rid, want_pcm=sp["want_pcm"])
        oc, opcm = oe.generate(sp["prompt"], max_steps=sp["max_steps"], temperature=sp["temperature"], top_k=sp["top_k"], top_p=sp["top_p"],
                               seed=sp["seed"], mask_eos=sp["mask_eos"], want_pcm=sp["want_pcm"])
        assert np.array_equal(oc, r["codes"]), (sp["max_steps"], sp["temperature"])
        if sp["want_pcm"]:
            assert r["pcm"].size == opcm.size and (opcm.size == 0 or np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL)


def test_scheduler_driver_thread_and_voices(tiny_engines, vivian):
    """background driver thread + voice registry: submit_text builds the preset / clone prompt engine-side (engine.rs:398-428)"""
    ge, oe = tiny_engines
    rng = np.random.default_rng(12)
    ref_codes = rng.integers(0, 2048, 4 * 16).astype(np.int32)
    ref_text = rng.integers(0, 4000, 3).astype(np.int32)
    v_preset = ge.register_voice(vivian)
    v_clone = ge.register_voice(vivian, ref_codes, ref_text)
    texts = [rng.integers(0, 4000, n).astype(np.int32) for n in (6, 11, 4, 9, 15)]
    ge.sched_start()
    try:
        ids = [ge.submit_text(v_clone if i % 2 else v_preset, t, lang_id=2055, max_steps=7 + i, want_pcm=True) for i, t in enumerate(texts)]
        for rid in ids:
            assert ge.wait(rid, 60000.0)
    finally:
        ge.sched_stop()
    for i, (t, rid) in enumerate(zip(texts, ids)):
        prompt = ge.assets.build_clone(t, ref_codes, ref_text, vivian) if i % 2 else ge.assets.build_core(t, lang_id=2055, spk_emb=vivian)
        r = ge.result(rid, want_pcm=True)
        assert r["first_chunk_ms"] > 0 and r["total_ms"] >= r["first_chunk_ms"]
        oc, opcm = oe.generate(prompt, max_steps=7 + i, want_pcm=True)
        assert np.array_equal(oc, r["codes"]) and np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL


def test_scheduler_soak_is_deterministic(gpu, tiny_model, vivian):
    """120 requests (random lengths 0..12, greedy and sampled, with/without PCM) through 8 slots with the background driver thread
    while the client streams partial results; every request must equal the same request run alone afterwards (batch invariance +
    no cross-talk between slots, codec streams or RNG states under continuous admission/retirement)."""
    import time
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=8, max_steps=16, load_codec=True)
    rng = np.random.default_rng(2024)
    specs = []
    for i in range(120):
        prompt = ge.assets.build_core(rng.integers(0, 4000, int(rng.integers(1, 40))).astype(np.int32), lang_id=2055, spk_emb=vivian)
        specs.append(dict(prompt=prompt, max_steps=int(rng.integers(0, 13)), temperature=float(rng.choice([0.0, 0.7, 1.1])), top_k=int(rng.choice([0, 5, 40])),
                          top_p=float(rng.choice([0.8, 1.0])), seed=int(rng.integers(0, 1 << 30)), mask_eos=bool(rng.integers(0, 2)), want_pcm=bool(i % 3)))
    ge.sched_start()
    try:
        ids = []
        for i, sp in enumerate(specs):
            ids.append(ge.submit(**sp))
            if i % 16 == 15:
                time.sleep(0.002)                                   # arrivals in bursts
                st = ge.poll(ids[i - 8])
                ge.fetch(ids[i - 8], 0, max(st["n_frames"], 1), 0, int(st["n_pcm"]))   # streaming reads race with the driver by design
        for rid in ids:
            assert ge.wait(rid, 120000.0)
    finally:
        ge.sched_stop()
    results = [ge.result(rid, want_pcm=sp["want_pcm"]) for rid, sp in zip(ids, specs)]
    for sp, r in zip(specs[::3] + specs[1::7], results[::3] + results[1::7]):
        alone = ge.generate_batch([sp["prompt"]], max_steps=sp["max_steps"], temperature=sp["temperature"], top_k=sp["top_k"], top_p=sp["top_p"],
                                  seed=sp["seed"], mask_eos=sp["mask_eos"], want_pcm=sp["want_pcm"])[0]
        assert np.array_equal(alone["codes"], r["codes"])
        if sp["want_pcm"]:
            assert alone["pcm"].size == r["pcm"].size and (r["pcm"].size == 0 or np.sqrt(np.mean((alone["pcm"] - r["pcm"]) ** 2)) < PCM_RMS_TOL)
    ge.close()


def test_cpp_host_mirror_matches_ctypes_path(gpu, tiny_model, vivian, tmp_path):
    """The C++ mirror of TtsEngine / VoiceFile / AudioSample (the drop-in surface of src/tts) drives the same C ABI: its codes must
    equal the ctypes path's for a preset voice (greedy) and a clone voice (sampled, seeded); streaming must equal blocking."""
    import struct
    pkg = os.path.join(ROOT, "qwen3-tts-rust_amd")
    exe = str(tmp_path / "host_mirror_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host", "host_mirror_main.cpp"),
                           "-L" + pkg, "-lq3tts_host", "-lq3tts", "-Wl,-rpath," + pkg])
    wav = str(tmp_path / "out.wav")
    # text in -> audio out: a tokenizer.json (written with the Python `tokenizers` package, Qwen2 pattern) in <model_dir>/tokenizer/ is picked up
    # by TtsEngine::new (engine.rs:103-104); token ids < the tiny model's text-table rows fall back deterministically otherwise
    tok_ids = None
    try:
        import test_tokenizer_cpu as TT
        tdir = os.path.join(tiny_model, "tokenizer")
        os.makedirs(tdir, exist_ok=True)
        tok, tpath = TT._build(tmp_path, vocab_size=500)
        os.replace(tpath, os.path.join(tdir, "tokenizer.json"))
        tok_ids = tok.encode("Hello, it's 42 degrees!", add_special_tokens=False).ids
    except ImportError as e:  # the text-in half is part of this test: a box without the `tokenizers` package must not pass it silently
        pytest.fail("the `tokenizers` package (the reference's own tokenizer dependency, Cargo.toml:25) is needed for the text-in half: %s" % e)
    r = subprocess.run([exe, tiny_model, os.path.join(ROOT, "tests", "golden", "speakers"), wav], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stderr[-2000:]
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.strip().splitlines()}
    assert [int(x) for x in lines["TEXTIDS"]] == tok_ids and len(lines["TEXTCODES"]) % 16 == 0 and len(lines["TEXTCODES"]) > 0
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=1, max_steps=16, load_codec=True)
    ids = np.arange(100, 108, dtype=np.int32)
    ref = ge.generate_batch([ge.assets.build_core(ids, lang_id=2055, spk_emb=vivian)], max_steps=10, temperature=0.0, seed=42, mask_eos=False, want_pcm=True)[0]
    assert np.array_equal(np.array(lines["PRESET"], np.int32).reshape(-1, 16), ref["codes"])
    clone_codes = (np.arange(48) * 37) % 2048
    refc = ge.generate_batch([ge.assets.build_clone(ids, clone_codes, np.array([7, 8, 9], np.int32), vivian)], max_steps=10, temperature=0.7, top_k=40, top_p=0.9,
                             seed=42, mask_eos=False)[0]
    assert np.array_equal(np.array(lines["CLONE"], np.int32).reshape(-1, 16), refc["codes"])
    raw = open(wav, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<I", raw[24:28])[0] == 24000
    n = (len(raw) - 44) // 2
    assert n == ref["pcm"].size == int(lines["PCM"][0])
    pcm16 = np.frombuffer(raw[44:], np.int16)
    assert np.abs(pcm16.astype(np.float32) - np.clip(ref["pcm"] * 32767.0, -32768, 32767)).max() <= 1.0     # audio.rs:36 scaling, truncation
    ge.close()


def test_engine_limits_and_errors(gpu, oracle, tiny_model, vivian):
    """maximum prompt length (1024 rows = the reference's effective cap, llama/mod.rs:567-581), over-long prompts and out-of-range
    max_steps are rejected with an error (nothing aborts across the C ABI), unknown request ids fail cleanly"""
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=2, max_prompt=1024, max_steps=8, load_codec=False)
    oe = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), None, 4)
    rng = np.random.default_rng(77)
    long_prompt = ge.assets.build_core(rng.integers(0, 4000, 1024 - 11).astype(np.int32), lang_id=2055, spk_emb=vivian)
    assert long_prompt.shape[0] == 1024
    short = ge.assets.build_core(np.array([1, 2, 3], np.int32), lang_id=2055, spk_emb=vivian)
    res = ge.generate_batch([long_prompt, short], max_steps=3, mask_eos=True)      # 1024-row prefill (4 chunks of 256) next to a 14-row one
    for p, r in zip((long_prompt, short), res):
        oc, _ = oe.generate(p, max_steps=3, mask_eos=True)
        assert np.array_equal(oc, r["codes"])
    too_long = np.concatenate([long_prompt, long_prompt[:1]])
    with pytest.raises(gpu_mod().Q3Error):
        ge.generate_batch([too_long], max_steps=2)
    with pytest.raises(gpu_mod().Q3Error):
        ge.generate_batch([short], max_steps=9)                                   # engine was built for max_steps=8
    with pytest.raises(gpu_mod().Q3Error):
        ge.poll(123456)
    assert np.array_equal(ge.generate_batch([short], max_steps=3)[0]["codes"], res[1]["codes"])   # still usable after the errors
    ge.close(); oe.close()


def test_engine_wide_batch_matches_oracle(gpu, oracle, tiny_model, vivian):
    """18 concurrent sequences: the batched-step kernels (int8-MFMA GEMM with token-tile loop, gate/up GEMM with the SwiGLU+quant
    epilogue, fused attention for many sequences, multi-token projection) and 16-stream codec groups, against oracle singles."""
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=18, max_steps=16, load_codec=True)
    oe = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), os.path.join(tiny_model, "onnx", "q3tts_codec.gguf"), 4)
    rng = np.random.default_rng(31)
    prompts = [ge.assets.build_core(rng.integers(0, 4000, 3 + (5 * i) % 23).astype(np.int32), lang_id=2055, spk_emb=vivian) for i in range(18)]
    steps = [6 + (i % 4) for i in range(18)]
    temps = [0.0 if i % 5 else 0.9 for i in range(18)]
    res = ge.generate_batch(prompts, max_steps=steps, temperature=temps, top_k=20, top_p=0.9, seed=[7 + i for i in range(18)], mask_eos=True,
                            want_pcm=True)
    for i, (p, r) in enumerate(zip(prompts, res)):
        oc, opcm = oe.generate(p, max_steps=steps[i], temperature=temps[i], top_k=20, top_p=0.9, seed=7 + i, mask_eos=True, want_pcm=True)
        assert np.array_equal(oc, r["codes"]), i
        assert r["pcm"].size == opcm.size and np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL, i
    ge.close(); oe.close()


def test_q5_k_m_engine_matches_oracle(gpu, oracle, tiny_model, vivian):
    """BASELINE.json configs[0] quantisation (Q5_K_M = Q5_K + Q6_K rows) on the GPU: K-quant rows stay packed in HBM (nibble +
    bit planes, kernels.h), are unpacked in registers and run through the fused decode kernels / the mixed-type GEMV; tokens must equal the oracle's Q5_K/Q6_K block arithmetic bit for bit."""
    qdir = os.path.join(tiny_model, "gguf_q5_k_m")
    for name, d, npre in (("qwen3_tts_talker.gguf", 2048, 21), ("qwen3_tts_predictor.gguf", 256, 2)):
        _tf_parity(gpu, oracle, os.path.join(qdir, name), d, npre, 4, 2048)
    ge = gpu.Engine(tiny_model, "q5_k_m", max_batch=2, max_steps=32, load_codec=False)
    oe = oracle.Engine(qdir, None, 4)
    prompts = [ge.assets.build_core(np.arange(100, 100 + n, dtype=np.int32), lang_id=2055, spk_emb=vivian) for n in (8, 15)]
    res = ge.generate_batch(prompts, max_steps=8, mask_eos=True)
    for p, r in zip(prompts, res):
        oc, _ = oe.generate(p, max_steps=8, mask_eos=True)
        assert np.array_equal(oc, r["codes"])
    ge.close(); oe.close()


def test_q5_k_m_wide_batch_matches_oracle(gpu, oracle, tiny_model, vivian):
    """Q5_K_M with 12 concurrent sequences: the mixed-type GEMV's token tiles (grid.z) and the batched layer path for K-quants"""
    qdir = os.path.join(tiny_model, "gguf_q5_k_m")
    ge = gpu.Engine(tiny_model, "q5_k_m", max_batch=12, max_steps=16, load_codec=False)
    oe = oracle.Engine(qdir, None, 4)
    rng = np.random.default_rng(71)
    prompts = [ge.assets.build_core(rng.integers(0, 4000, 2 + 3 * i).astype(np.int32), lang_id=2055, spk_emb=vivian) for i in range(12)]
    res = ge.generate_batch(prompts, max_steps=[4 + i % 3 for i in range(12)], temperature=[0.0 if i % 4 else 0.6 for i in range(12)], top_k=25, top_p=0.9,
                            seed=[300 + i for i in range(12)], mask_eos=True)
    for i, (p, r) in enumerate(zip(prompts, res)):
        oc, _ = oe.generate(p, max_steps=4 + i % 3, temperature=(0.0 if i % 4 else 0.6), top_k=25, top_p=0.9, seed=300 + i, mask_eos=True)
        assert np.array_equal(oc, r["codes"]), i
    ge.close(); oe.close()


def test_q5_k_m_matrix_core_batch_matches_oracle(gpu, oracle, tiny_model, vivian):
    """Q5_K_M with 20 concurrent sequences: from 16 tokens the K-quant rows go through the matrix-core GEMM (k_gemm_kq_mfma: packed planes unpacked into the int8
    B operand, Q5_K / Q6_K block chains on the accumulators, the mixed q,k,v matrix by per-workgroup type dispatch, the fused gate/up + SwiGLU form on the
    talker's K = 2048); the multi-sequence prefill packs > 16 prompt rows as well.  Tokens must equal the oracle's, sequence by sequence."""
    qdir = os.path.join(tiny_model, "gguf_q5_k_m")
    ge = gpu.Engine(tiny_model, "q5_k_m", max_batch=20, max_steps=8, load_codec=False)
    oe = oracle.Engine(qdir, None, 4)
    rng = np.random.default_rng(72)
    prompts = [ge.assets.build_core(rng.integers(0, 4000, 2 + i).astype(np.int32), lang_id=2055, spk_emb=vivian) for i in range(20)]
    res = ge.generate_batch(prompts, max_steps=[3 + i % 3 for i in range(20)], mask_eos=True)
    st = ge.stats()
    assert st["slot_frames"] / max(st["graph_frames"], 1) >= 16, "the >= 16-token graph widths were not exercised"
    for i in (0, 3, 7, 12, 19):
        oc, _ = oe.generate(prompts[i], max_steps=3 + i % 3, mask_eos=True)
        assert np.array_equal(oc, res[i]["codes"]), i
    ge.close(); oe.close()


def test_bf16_engine_matches_oracle(gpu, oracle, tiny_model, vivian):
    """BASELINE.json configs[4] weight type (bf16): f32 activations, float-weight GEMV (spec S3 float form), bit-exact tokens."""
    qdir = os.path.join(tiny_model, "gguf_bf16")
    for name, d, npre in (("qwen3_tts_talker.gguf", 2048, 19), ("qwen3_tts_predictor.gguf", 256, 2)):
        _tf_parity(gpu, oracle, os.path.join(qdir, name), d, npre, 4, 2048)
    ge = gpu.Engine(tiny_model, "bf16", max_batch=2, max_steps=32, load_codec=False)
    oe = oracle.Engine(qdir, None, 4)
    rng = np.random.default_rng(21)
    prompts = [ge.assets.build_core(np.arange(100, 109, dtype=np.int32), lang_id=2055, spk_emb=vivian),
               ge.assets.build_clone(rng.integers(0, 4000, 5).astype(np.int32), rng.integers(0, 2048, 3 * 16), rng.integers(0, 4000, 2), vivian)]
    res = ge.generate_batch(prompts, max_steps=8, mask_eos=True)
    for p, r in zip(prompts, res):
        oc, _ = oe.generate(p, max_steps=8, mask_eos=True)
        assert np.array_equal(oc, r["codes"])
    ge.close(); oe.close()


def test_bf16_mixed_voice_clone_batch_c5(gpu, oracle, tiny_model, vivian):
    """BASELINE.json configs[4] mechanics at test size: bf16 weights, 10 concurrent sequences over 3 registered voices (two clone
    voices with reference codes + reference text, one preset), prompts built engine-side, sampled and greedy mixed."""
    qdir = os.path.join(tiny_model, "gguf_bf16")
    ge = gpu.Engine(tiny_model, "bf16", max_batch=10, max_steps=16, load_codec=True)
    oe = oracle.Engine(qdir, os.path.join(tiny_model, "onnx", "q3tts_codec.gguf"), 4)
    rng = np.random.default_rng(55)
    spk2 = (vivian * 0.5 + rng.standard_normal(2048).astype(np.float32) * 0.01).astype(np.float32)
    voices = [dict(spk=vivian, codes=None, text=None),
              dict(spk=spk2, codes=rng.integers(0, 2048, 6 * 16).astype(np.int32), text=rng.integers(0, 4000, 4).astype(np.int32)),
              dict(spk=vivian, codes=rng.integers(0, 2048, 3 * 16).astype(np.int32), text=rng.integers(0, 4000, 2).astype(np.int32))]
    vids = [ge.register_voice(v["spk"], v["codes"], v["text"]) for v in voices]
    texts = [rng.integers(0, 4000, 4 + i).astype(np.int32) for i in range(10)]
    ids = [ge.submit_text(vids[i % 3], t, lang_id=2055, max_steps=5 + i % 3, temperature=(0.8 if i % 4 == 0 else 0.0), top_k=10, top_p=0.9,
                          seed=90 + i, want_pcm=True) for i, t in enumerate(texts)]
    while ge.sched_step():
        pass
    for i, (t, rid) in enumerate(zip(texts, ids)):
        v = voices[i % 3]
        prompt = (ge.assets.build_clone(t, v["codes"], v["text"], v["spk"]) if v["codes"] is not None
                  else ge.assets.build_core(t, lang_id=2055, spk_emb=v["spk"]))
        r = ge.result(rid, want_pcm=True)
        oc, opcm = oe.generate(prompt, max_steps=5 + i % 3, temperature=(0.8 if i % 4 == 0 else 0.0), top_k=10, top_p=0.9, seed=90 + i, want_pcm=True)
        assert np.array_equal(oc, r["codes"]), i
        assert np.sqrt(np.mean((opcm - r["pcm"]) ** 2)) < PCM_RMS_TOL, i
    ge.close(); oe.close()


def test_codec_decoder_chunked_vs_oracle(gpu, oracle, tiny_model):
    path = os.path.join(tiny_model, "onnx", "q3tts_codec.gguf")
    rng = np.random.default_rng(8)
    codes = rng.integers(-3, 2051, (23, 16))      # out-of-range codes are clamped like engine.rs:515-519 would
    oc = oracle.Codec(path); oc.reset(); ref = oc.decode(np.clip(codes, 0, 2047)).copy(); oc.close()
    gd = gpu.Decoder(path, 2)
    for stream, chunks in ((0, [4, 4, 4, 4, 4, 3]), (1, [1, 22])):
        gd.reset(stream)
        o, parts = 0, []
        for i, n in enumerate(chunks):
            parts.append(gd.decode(np.clip(codes[o:o + n], 0, 2047), i == len(chunks) - 1, stream)); o += n
        got = np.concatenate(parts)
        assert got.shape == ref.shape and np.sqrt(np.mean((got - ref) ** 2)) < PCM_RMS_TOL and np.abs(got - ref).max() < 1e-3
    gd.close()


def test_codec_group_decode_vs_oracle(gpu, oracle, tiny_model):
    """q3tts_decoder_decode_group: 5 streams with different histories decoded in one pass per chunk == the oracle stream by stream"""
    path = os.path.join(tiny_model, "onnx", "q3tts_codec.gguf")
    gd = gpu.Decoder(path, n_streams=6, max_frames=4, max_group=5)
    rng = np.random.default_rng(8)
    codes = rng.integers(0, 2048, (5, 12, 16))
    streams = [4, 0, 2, 5, 1]
    for s in streams:
        gd.reset(s)
    got = np.concatenate([gd.decode_group(streams, codes[:, o:o + 4]) for o in (0, 4, 8)], axis=1)
    oc = oracle.Codec(path)
    for i in range(5):
        oc.reset()
        ref = oc.decode(codes[i]).copy()
        assert np.sqrt(np.mean((ref - got[i]) ** 2)) < PCM_RMS_TOL, i
    oc.close(); gd.close()


def test_mel_kernel_vs_oracle(gpu, oracle):
    """row a16: log-mel front end (onnx.rs:167-320) on the device vs the oracle; float tolerance (f32 DFT vs double FFT)."""
    rng = np.random.default_rng(12)
    t = np.arange(24000 * 3) / 24000.0
    chirp = (0.4 * np.sin(2 * np.pi * (200 + 3000 * t) * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    for audio in (chirp, chirp[:5000], chirp[:300], np.zeros(1500, np.float32)):
        ref = oracle.mel(audio)
        got = gpu.mel(audio)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-3


def test_llama_abi_replay_matches_oracle(gpu, oracle, tiny_model, vivian, tmp_path):
    """Boundary A: the reference's loop, replayed call-for-call through runtime/libllama.so (dlopen'd from cwd/runtime)."""
    pkg = os.path.join(ROOT, "qwen3-tts-rust_amd")
    oe = oracle.Engine(os.path.join(tiny_model, "gguf_q8_0"), None, 4)
    prompt = oe.assets.build_core(np.arange(50, 61, dtype=np.int32), lang_id=2055, spk_emb=vivian)
    oc, _ = oe.generate(prompt, max_steps=6, mask_eos=True)
    oe.close()
    pf, cf = str(tmp_path / "prompt.f32"), str(tmp_path / "codes.i32")
    prompt.tofile(pf)
    r = subprocess.run([os.path.join(pkg, "ref_replay"), os.path.join(tiny_model, "gguf_q8_0"), pf, str(prompt.shape[0]), "6", cf, "1"], cwd=pkg,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    got = np.fromfile(cf, np.int32).reshape(-1, 16)
    assert np.array_equal(got, oc)
    assert "timing prefill_ms" in r.stdout   # the zero-Rust-change path reports its own plumbing cost (scripts/replay_fullsize.sh for the full model)
    # prompt longer than the batch's pos allocation / 4 (reference quirk 3): the shim must not read past the n_tokens ints the caller owns
    long_prompt = np.repeat(prompt, 100, axis=0)[:1100]
    long_prompt.tofile(pf)
    r2 = subprocess.run([os.path.join(pkg, "ref_replay"), os.path.join(tiny_model, "gguf_q8_0"), pf, "1100", "2", cf, "1"], cwd=pkg, capture_output=True, text=True)
    assert r2.returncode == 0 and np.fromfile(cf, np.int32).size == 32, r2.stderr[-1500:]


def test_two_engines_keep_their_own_assets(gpu, synth_tool, tiny_model):
    """q3tts_engine_assets returns a handle that lives inside its engine (a thread-local view made the first engine read the
    second engine's tables): two engines over different assets files, interleaved calls, one destroyed before the other is used again."""
    other = os.environ.get("Q3_TINY_MODEL2", "/tmp/q3tts_pytest_tiny_seed99")
    if not os.path.exists(os.path.join(other, ".complete")):
        subprocess.check_call([synth_tool, "--out", other, "--preset", "tiny", "--quant", "q8_0", "--seed", "99"])
        open(os.path.join(other, ".complete"), "w").write("ok")
    e1 = gpu.Engine(tiny_model, "q8_0", max_batch=1, max_steps=8, load_codec=False)
    a1 = e1.assets.text_embedding(1234).copy()
    e2 = gpu.Engine(other, "q8_0", max_batch=1, max_steps=8, load_codec=False)
    a2 = e2.assets.text_embedding(1234).copy()
    assert not np.array_equal(a1, a2)
    assert np.array_equal(e1.assets.text_embedding(1234), a1) and np.array_equal(e2.assets.text_embedding(1234), a2)
    f1 = gpu.Assets(os.path.join(tiny_model, "gguf_q8_0", "qwen3_assets.gguf"))
    assert np.array_equal(f1.text_embedding(1234), a1)
    f1.close()
    e2.close()
    assert np.array_equal(e1.assets.text_embedding(1234), a1) and np.isfinite(e1.assets.tts_pad()).all()
    e1.close()


@pytest.mark.parametrize("n_engines", [2, 0])
def test_group_api_voice_broadcast_and_round_robin(gpu, tiny_model, tmp_path, n_engines):
    """q3tts_group_* / q3tts_comm_* (multi-GPU behind the C ABI) from a plain C program: a clone voice registered through the group is
    broadcast from the first device and used by every engine; round-robin requests give identical codes on every engine and equal the
    single-engine ctypes path.  n_engines = 2 on a one-GPU box lists the device twice (two engines, peer-copy path); n_engines = 0 = one
    engine per visible device (ncclBroadcast path when the box has several)."""
    pkg = os.path.join(ROOT, "qwen3-tts-rust_amd")
    exe = str(tmp_path / "group_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host", "group_main.cpp"), "-L" + pkg, "-lq3tts",
                           "-Wl,-rpath," + pkg])
    r = subprocess.run([exe, tiny_model, str(n_engines)], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-2000:])
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.strip().splitlines()}
    spk = (0.01 * ((np.arange(2048) * 37) % 101 - 50)).astype(np.float32)
    ge = gpu.Engine(tiny_model, "q8_0", max_batch=1, max_steps=16, load_codec=False)
    prompt = ge.assets.build_clone(np.arange(100, 108, dtype=np.int32), (np.arange(48) * 37) % 2048, np.array([7, 8, 9], np.int32), spk)
    ref = ge.generate_batch([prompt], max_steps=6, temperature=0.0, seed=42, mask_eos=True)[0]
    ge.close()
    assert np.array_equal(np.array(lines["CLONE"], np.int32).reshape(-1, 16), ref["codes"])
    assert int(lines["PCM"][0]) == 6 * 1920 or int(lines["PCM"][0]) > 0


def test_decoder_state_export_import(gpu, tiny_model):
    """DecoderState (onnx.rs:461-496) as named tensors out of / into the device: a stream checkpointed after one chunk and restored into a
    DIFFERENT stream slot of another decoder continues bit-identically; the layout names the reference's state tensors."""
    path = os.path.join(tiny_model, "onnx", "q3tts_codec.gguf")
    rng = np.random.default_rng(31)
    codes = rng.integers(0, 2048, (12, 16))
    a = gpu.Decoder(path, n_streams=2)
    a.reset(0)
    head = a.decode(codes[:4], stream=0).copy()
    blob = a.state_export(0)
    tail_ref = a.decode(codes[4:], stream=0).copy()
    names = [e[0] for e in a.state_layout()]
    assert names[0] == "pre_conv_history" and "past_key_0" in names and "past_value_0" in names and any(n.startswith("conv_history.") for n in names)
    assert names[-1].startswith("counters") and blob.size == a.state_layout()[-1][1] + 2 and blob[-1] == 4.0
    b = gpu.Decoder(path, n_streams=3)
    b.reset(2)
    b.state_import(blob, stream=2)
    tail = b.decode(codes[4:], stream=2).copy()
    assert np.array_equal(tail, tail_ref) and head.size == 4 * a.spf
    a.close(); b.close()


def test_ggml_mode_engine_matches_oracle(gpu, oracle, tiny_model, vivian):
    """SURVEY 8f row f-1, GPU half: with Q3_SPEC=ggml the engine runs llama.cpp's portable arithmetic (Q8_K 256-block activations for Q5_K / Q6_K
    rows, roundf Q8_0 activations, ggml's vec_dot accumulation order, double-accumulated norms and softmax sums, glibc's expf) in csrc/ggml_mode.hip,
    and its codec tokens equal oracle/q3o_ggml.c's bit for bit -- Q8_0 and Q5_K_M, a preset and a clone prompt, two sequences stepping together.
    The mode exists so that GPU tokens can be put beside the reference's (llama.cpp b8123, /root/reference/src/models/llama/mod.rs:442-451) the day
    that binary and the real weights are at hand; it also has to DIFFER from the spec arithmetic somewhere, or the switch does nothing."""
    rng = np.random.default_rng(77)
    old = os.environ.get("Q3_SPEC")
    try:
        for quant, sub in (("q8_0", "gguf_q8_0"), ("q5_k_m", "gguf_q5_k_m")):
            os.environ["Q3_SPEC"] = "ggml"
            oracle.set_arith_mode(1)
            ge = gpu.Engine(tiny_model, quant, max_batch=2, max_steps=16, load_codec=False)
            prompts = [ge.assets.build_core(np.arange(100, 108, dtype=np.int32), lang_id=2055, spk_emb=vivian),
                       ge.assets.build_clone(rng.integers(0, 4000, 5).astype(np.int32), rng.integers(0, 2048, 3 * 16), rng.integers(0, 4000, 2), vivian)]
            res = ge.generate_batch(prompts, max_steps=[7, 5], mask_eos=True)
            ge.close()
            oe = oracle.Engine(os.path.join(tiny_model, sub), None, 4)
            gg = []
            for p, r, m in zip(prompts, res, (7, 5)):
                oc, _ = oe.generate(p, max_steps=m, mask_eos=True)
                assert np.array_equal(oc, r["codes"]), quant
                gg.append(oc)
            oe.close()
            # and the spec arithmetic gives (somewhere) different logits: same prompts through the default engine
            os.environ.pop("Q3_SPEC")
            oracle.set_arith_mode(0)
            gs = gpu.Engine(tiny_model, quant, max_batch=2, max_steps=16, load_codec=False)
            spec = gs.generate_batch(prompts, max_steps=[7, 5], mask_eos=True)
            gs.close()
            agree = np.mean([np.mean(a == b["codes"]) for a, b in zip(gg, spec)])
            assert agree < 1.0, agree   # random synthetic weights have tiny top-2 margins and the loop is autoregressive, so the runs part early; identical runs would mean the switch is inert
            print("ggml-mode vs spec-mode token agreement on the tiny %s model: %.2f" % (quant, agree))
    finally:
        oracle.set_arith_mode(0)
        if old is None:
            os.environ.pop("Q3_SPEC", None)
        else:
            os.environ["Q3_SPEC"] = old
