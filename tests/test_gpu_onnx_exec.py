"""SURVEY 8f row f-2 (second half) / row a17: the ONNX graph executor (q3tts_onnx_session_*) against numpy / torch, operator by operator and on
encoder-shaped graphs with the reference's I/O contract (/root/reference/src/models/onnx.rs:97-121 `input_values` -> `audio_codes`,
:140-163 `mels` -> `spk_emb`).  The graphs are written by tests/onnx_writer.py (no `onnx` package in the image); the real encoder files are not in
the container, so these tests pin the operator semantics, not the exported networks."""
import json
import os
import struct
import subprocess
import numpy as np
import pytest

import onnx_writer as W

pytestmark = pytest.mark.gpu
F32, I64 = W.F32, W.I64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_graph(gpu, tmp_path, nodes, inits, feeds, outs, opset=17):
    """outs: [(name, elem_type, shape)]"""
    path = os.path.join(str(tmp_path), "g%d.onnx" % np.random.randint(1 << 30))
    is_const = lambda n: W._s(4, "Constant") in n and W._s(4, "ConstantOfShape") not in n
    nodes = [n for n in nodes if is_const(n)] + [n for n in nodes if not is_const(n)]  # the tests declare constants next to their use; graphs are topological
    ins = [W.value_info(k, I64 if np.issubdtype(np.asarray(v).dtype, np.integer) else F32, list(np.asarray(v).shape)) for k, v in feeds.items()]
    open(path, "wb").write(W.model(nodes, inits, ins, [W.value_info(n, t, s) for n, t, s in outs], opset=opset))
    s = gpu.OnnxSession(path)
    assert s.unsupported() == [], s.unsupported()
    res = s.run(feeds, [n for n, _, _ in outs])
    s.close()
    return res


def erf(x):
    from scipy.special import erf as e
    return e(x)


def test_unary_ops(gpu, tmp_path):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 5, 7)).astype(np.float32) * 2
    cases = {
        "Relu": ([], lambda v: np.maximum(v, 0)), "Sigmoid": ([], lambda v: 1 / (1 + np.exp(-v))), "Tanh": ([], np.tanh), "Exp": ([], np.exp),
        "Neg": ([], lambda v: -v), "Abs": ([], np.abs), "Sin": ([], np.sin), "Cos": ([], np.cos), "Erf": ([], erf), "Floor": ([], np.floor),
        "Ceil": ([], np.ceil), "Round": ([], np.round), "Sign": ([], np.sign),
        "Elu": ([W.attr_float("alpha", 0.7)], lambda v: np.where(v > 0, v, 0.7 * (np.exp(v) - 1))),
        "LeakyRelu": ([W.attr_float("alpha", 0.1)], lambda v: np.where(v > 0, v, 0.1 * v)),
        "HardSigmoid": ([W.attr_float("alpha", 0.25), W.attr_float("beta", 0.4)], lambda v: np.clip(0.25 * v + 0.4, 0, 1)),
        "Gelu": ([], lambda v: 0.5 * v * (1 + erf(v / np.sqrt(2)))), "Softplus": ([], lambda v: np.log1p(np.exp(v))),
        "Softsign": ([], lambda v: v / (1 + np.abs(v))), "HardSwish": ([], lambda v: v * np.clip(v / 6 + 0.5, 0, 1)),
        "Selu": ([], lambda v: 1.0507009873554805 * np.where(v > 0, v, 1.6732632423543772 * (np.exp(v) - 1))),
    }
    nodes, outs = [], []
    for i, (op, (attrs, _)) in enumerate(cases.items()):
        nodes.append(W.node(op, ["x"], ["y%d" % i], attrs=attrs))
        outs.append(("y%d" % i, F32, list(x.shape)))
    nodes.append(W.node("Abs", ["x"], ["ax"]))
    nodes.append(W.node("Log", ["ax"], ["ylog"])); outs.append(("ylog", F32, list(x.shape)))
    nodes.append(W.node("Sqrt", ["ax"], ["ysqrt"])); outs.append(("ysqrt", F32, list(x.shape)))
    nodes.append(W.node("Reciprocal", ["ax"], ["yrec"])); outs.append(("yrec", F32, list(x.shape)))
    nodes.append(W.node("Gelu", ["x"], ["ygt"], attrs=[W.attr_str("approximate", "tanh")])); outs.append(("ygt", F32, list(x.shape)))
    nodes.append(W.node("Clip", ["x", "lo", "hi"], ["yclip"])); outs.append(("yclip", F32, list(x.shape)))
    inits = [W.tensor("lo", np.float32(-0.5).reshape(())), W.tensor("hi", np.float32(0.8).reshape(()))]
    r = run_graph(gpu, tmp_path, nodes, inits, {"x": x}, outs)
    for i, (op, (_, fn)) in enumerate(cases.items()):
        np.testing.assert_allclose(r["y%d" % i], fn(x.astype(np.float64)), rtol=2e-5, atol=2e-6, err_msg=op)
    ax = np.abs(x.astype(np.float64))
    np.testing.assert_allclose(r["ylog"], np.log(ax), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r["ysqrt"], np.sqrt(ax), rtol=2e-6)
    np.testing.assert_allclose(r["yrec"], 1 / ax, rtol=2e-6)
    xd = x.astype(np.float64)
    np.testing.assert_allclose(r["ygt"], 0.5 * xd * (1 + np.tanh(np.sqrt(2 / np.pi) * (xd + 0.044715 * xd ** 3))), rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(r["yclip"], np.clip(x, -0.5, 0.8))


def test_binary_broadcast_compare_where(gpu, tmp_path):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((2, 1, 4, 5)).astype(np.float32)
    b = rng.standard_normal((3, 1, 5)).astype(np.float32)
    c = rng.standard_normal((5,)).astype(np.float32)
    ops = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide, "Min": np.minimum, "Max": np.maximum,
           "PRelu": lambda x, y: np.where(x > 0, x, x * y)}
    cmps = {"Equal": np.equal, "Less": np.less, "Greater": np.greater, "LessOrEqual": np.less_equal, "GreaterOrEqual": np.greater_equal}
    nodes, outs = [], []
    shp = list(np.broadcast_shapes(a.shape, b.shape))
    for op in ops:
        nodes.append(W.node(op, ["a", "b"], ["o_" + op])); outs.append(("o_" + op, F32, shp))
    for op in cmps:
        nodes.append(W.node(op, ["a", "b"], ["o_" + op])); outs.append(("o_" + op, 9, shp))
    nodes += [W.node("Abs", ["a"], ["aa"]), W.node("Pow", ["aa", "c"], ["o_pow"]), W.node("And", ["o_Less", "o_Greater"], ["o_and"]),
              W.node("Or", ["o_Less", "o_Equal"], ["o_or"]), W.node("Not", ["o_Less"], ["o_not"]), W.node("Where", ["o_Less", "a", "b"], ["o_where"]),
              W.node("Sum", ["a", "b", "c"], ["o_sum3"]), W.node("Mean", ["a", "b", "c"], ["o_mean3"]), W.node("Mod", ["a", "b"], ["o_fmod"], attrs=[W.attr_int("fmod", 1)])]
    outs += [("o_pow", F32, list(a.shape)), ("o_and", 9, shp), ("o_or", 9, shp), ("o_not", 9, shp), ("o_where", F32, shp), ("o_sum3", F32, shp), ("o_mean3", F32, shp),
             ("o_fmod", F32, shp)]
    r = run_graph(gpu, tmp_path, nodes, [], {"a": a, "b": b, "c": c}, outs)
    for op, fn in ops.items():
        np.testing.assert_allclose(r["o_" + op], fn(a, b), rtol=1e-6, err_msg=op)
    for op, fn in cmps.items():
        np.testing.assert_array_equal(r["o_" + op], fn(a, b), err_msg=op)
    np.testing.assert_allclose(r["o_pow"], np.abs(a).astype(np.float64) ** c, rtol=3e-5)
    np.testing.assert_array_equal(r["o_and"], np.zeros(shp, bool))
    np.testing.assert_array_equal(r["o_or"], np.less_equal(a, b))
    np.testing.assert_array_equal(r["o_not"], ~np.less(a, b))
    np.testing.assert_array_equal(r["o_where"], np.where(a < b, a, b))
    np.testing.assert_allclose(r["o_sum3"], a + b + c, rtol=1e-6)
    np.testing.assert_allclose(r["o_mean3"], (a + b + c) / 3, rtol=1e-6)
    np.testing.assert_allclose(r["o_fmod"], np.fmod(a, b), rtol=1e-5, atol=1e-6)


def test_shape_arithmetic_and_data_movement(gpu, tmp_path):
    """the Shape -> Gather -> Concat -> Reshape chains every exporter emits run on the host; the data ops they parameterise run on the device"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 6, 4, 5)).astype(np.float32)
    i0 = lambda name, v: W.tensor(name, np.asarray(v, np.int64))
    inits = [i0("c0", 0), i0("c1", [1]), i0("cm1", [-1]), i0("c2", [2]), i0("ax01", [0, 1]), i0("st", [1, 4]), i0("en", [5, -9223372036854775807 - 1 + 1]),
             i0("sax", [1, 3]), i0("sst", [2, -1]), i0("idx", [[3, 0], [5, 5]]), i0("rep", [1, 2, 1, 3]), i0("splits", [1, 2, 3]), i0("c3", 3), i0("c20", 20)]
    nodes = [
        W.node("Shape", ["x"], ["shp"]),                                  # [2,6,4,5]
        W.node("Gather", ["shp", "c0"], ["n"]),                           # 2 (scalar)
        W.node("Unsqueeze", ["n", "c0u"], ["n1"]),
        W.node("Constant", [], ["c0u"], attrs=[W.attr_tensor("value", np.asarray([0], np.int64))]),
        W.node("Slice", ["shp", "c2", "c4"], ["tail"]),                   # [4,5]
        W.node("Constant", [], ["c4"], attrs=[W.attr_ints("value_ints", [4])]),
        W.node("ReduceProd", ["tail"], ["tp"], attrs=[W.attr_int("keepdims", 1)]),  # [20]
        W.node("Concat", ["n1", "cm1", "tp"], ["newshape"], attrs=[W.attr_int("axis", 0)]),
        W.node("Reshape", ["x", "newshape"], ["y_reshape"]),              # [2,6,20]
        W.node("Transpose", ["x"], ["y_tr"], attrs=[W.attr_ints("perm", [0, 2, 3, 1])]),
        W.node("Slice", ["x", "st", "en", "sax", "sst"], ["y_slice"]),    # axis1 1:5:2, axis3 4:begin:-1
        W.node("Gather", ["x", "idx"], ["y_gather"], attrs=[W.attr_int("axis", 1)]),
        W.node("Tile", ["x", "rep"], ["y_tile"]),
        W.node("Split", ["x", "splits"], ["s0", "s1", "s2"], attrs=[W.attr_int("axis", 1)]),
        W.node("Concat", ["s2", "s0", "s1"], ["y_cat"], attrs=[W.attr_int("axis", 1)]),
        W.node("Flatten", ["x"], ["y_flat"], attrs=[W.attr_int("axis", 2)]),
        W.node("Range", ["c0", "c20", "c3"], ["rng"]),                    # 0,3,...,18 (7 values, host)
        W.node("Cast", ["rng"], ["rngf"], attrs=[W.attr_int("to", 1)]),
        W.node("ConstantOfShape", ["c2b"], ["ones"], attrs=[W.attr_tensor("value", np.asarray([1.5], np.float32))]),
        W.node("Constant", [], ["c2b"], attrs=[W.attr_tensor("value", np.asarray([2, 1], np.int64))]),
        W.node("Mul", ["ones", "rngf"], ["y_outer"]),                     # [2,1] x [7] -> [2,7]
        W.node("Expand", ["rngf", "eshape"], ["y_expand"]),
        W.node("Constant", [], ["eshape"], attrs=[W.attr_tensor("value", np.asarray([3, 1, 7], np.int64))]),
        W.node("Squeeze", ["y_expand", "c1"], ["y_squeeze"]),
        W.node("Size", ["x"], ["sz"]),
        W.node("Div", ["sz", "c20"], ["y_idiv"]),                         # 240 // 20 on the host, integer semantics
        W.node("Cast", ["x"], ["xi"], attrs=[W.attr_int("to", 7)]),
        W.node("Cast", ["xi"], ["y_trunc"], attrs=[W.attr_int("to", 1)]),
    ]
    outs = [("y_reshape", F32, [2, 6, 20]), ("y_tr", F32, [2, 4, 5, 6]), ("y_slice", F32, [2, 2, 4, 5]), ("y_gather", F32, [2, 2, 2, 4, 5]), ("y_tile", F32, [2, 12, 4, 15]),
            ("y_cat", F32, [2, 6, 4, 5]), ("y_flat", F32, [12, 20]), ("y_outer", F32, [2, 7]), ("y_expand", F32, [3, 1, 7]), ("y_squeeze", F32, [3, 7]), ("y_idiv", I64, []),
            ("y_trunc", F32, list(x.shape)), ("shp", I64, [4])]
    r = run_graph(gpu, tmp_path, nodes, inits, {"x": x}, outs)
    np.testing.assert_array_equal(r["shp"], [2, 6, 4, 5])
    np.testing.assert_array_equal(r["y_reshape"], x.reshape(2, 6, 20))
    np.testing.assert_array_equal(r["y_tr"], x.transpose(0, 2, 3, 1))
    np.testing.assert_array_equal(r["y_slice"], x[:, 1:5:2, :, 4::-1])
    np.testing.assert_array_equal(r["y_gather"], np.take(x, [[3, 0], [5, 5]], axis=1))
    np.testing.assert_array_equal(r["y_tile"], np.tile(x, (1, 2, 1, 3)))
    np.testing.assert_array_equal(r["y_cat"], np.concatenate([x[:, 3:], x[:, :1], x[:, 1:3]], axis=1))
    np.testing.assert_array_equal(r["y_flat"], x.reshape(12, 20))
    rg = np.arange(0, 20, 3, dtype=np.float32)
    np.testing.assert_array_equal(r["y_outer"], np.full((2, 1), 1.5, np.float32) * rg)
    np.testing.assert_array_equal(r["y_expand"], np.broadcast_to(rg, (3, 1, 7)))
    np.testing.assert_array_equal(r["y_squeeze"], np.broadcast_to(rg, (3, 7)))
    assert int(r["y_idiv"].reshape(-1)[0]) == 12
    np.testing.assert_array_equal(r["y_trunc"], np.trunc(x))


def test_reductions_and_normalisations(gpu, tmp_path):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 4, 300)).astype(np.float32)
    xd = x.astype(np.float64)
    g = rng.standard_normal(300).astype(np.float32); b = rng.standard_normal(300).astype(np.float32)
    cg = rng.standard_normal(4).astype(np.float32); cb = rng.standard_normal(4).astype(np.float32)
    cm = rng.standard_normal(4).astype(np.float32); cv = (rng.random(4) + 0.5).astype(np.float32)
    ax = lambda name, v: W.tensor(name, np.asarray(v, np.int64))
    inits = [ax("a2", [2]), ax("a02", [0, 2]), ax("a1", [1]), ax("am1", -1), W.tensor("g", g), W.tensor("b", b), W.tensor("cg", cg), W.tensor("cb", cb),
             W.tensor("cm", cm), W.tensor("cv", cv)]
    red = {"ReduceMean": lambda v, a, k: v.mean(axis=a, keepdims=k), "ReduceMax": lambda v, a, k: v.max(axis=a, keepdims=k),
           "ReduceMin": lambda v, a, k: v.min(axis=a, keepdims=k), "ReduceL2": lambda v, a, k: np.sqrt((v * v).sum(axis=a, keepdims=k)),
           "ReduceSumSquare": lambda v, a, k: (v * v).sum(axis=a, keepdims=k), "ReduceL1": lambda v, a, k: np.abs(v).sum(axis=a, keepdims=k),
           "ReduceLogSumExp": lambda v, a, k: np.log(np.exp(v).sum(axis=a, keepdims=k))}
    nodes, outs = [], []
    for op in red:                                    # opset 18 form: axes as an input
        nodes.append(W.node(op, ["x", "a02"], ["r_" + op], attrs=[W.attr_int("keepdims", 0)])); outs.append(("r_" + op, F32, [4]))
    nodes += [W.node("ReduceSum", ["x", "a2"], ["r_sum"]), W.node("Slice", ["x", "a0s", "a2e", "a2"], ["xs"]),
              W.node("ReduceProd", ["xs", "a1"], ["r_prod"], attrs=[W.attr_int("keepdims", 0)]), W.node("Constant", [], ["a0s"], attrs=[W.attr_ints("value_ints", [0])]),
              W.node("Constant", [], ["a2e"], attrs=[W.attr_ints("value_ints", [3])]),
              W.node("ReduceMean", ["x"], ["r_all"], attrs=[W.attr_int("keepdims", 0)]),
              W.node("ArgMax", ["x"], ["r_argmax"], attrs=[W.attr_int("axis", 2), W.attr_int("keepdims", 0)]),
              W.node("ArgMin", ["x"], ["r_argmin"], attrs=[W.attr_int("axis", 1)]),
              W.node("Softmax", ["x"], ["r_sm_last"]), W.node("Softmax", ["x"], ["r_sm_mid"], attrs=[W.attr_int("axis", 1)]),
              W.node("LogSoftmax", ["x"], ["r_lsm"], attrs=[W.attr_int("axis", -1)]),
              W.node("LayerNormalization", ["x", "g", "b"], ["r_ln"], attrs=[W.attr_float("epsilon", 1e-5)]),
              W.node("InstanceNormalization", ["x", "cg", "cb"], ["r_in"], attrs=[W.attr_float("epsilon", 1e-5)]),
              W.node("BatchNormalization", ["x", "cg", "cb", "cm", "cv"], ["r_bn"], attrs=[W.attr_float("epsilon", 1e-5)]),
              W.node("CumSum", ["x", "am1"], ["r_cs"]), W.node("CumSum", ["x", "a1s"], ["r_cs_rev"], attrs=[W.attr_int("reverse", 1), W.attr_int("exclusive", 1)]),
              W.node("Constant", [], ["a1s"], attrs=[W.attr_int("value_int", 1)]),
              W.node("GlobalAveragePool", ["x"], ["r_gap"])]
    outs += [("r_sum", F32, [3, 4, 1]), ("r_prod", F32, [3, 3]), ("r_all", F32, []), ("r_argmax", I64, [3, 4]), ("r_argmin", I64, [3, 1, 300]),
             ("r_sm_last", F32, list(x.shape)), ("r_sm_mid", F32, list(x.shape)), ("r_lsm", F32, list(x.shape)), ("r_ln", F32, list(x.shape)), ("r_in", F32, list(x.shape)),
             ("r_bn", F32, list(x.shape)), ("r_cs", F32, list(x.shape)), ("r_cs_rev", F32, list(x.shape)), ("r_gap", F32, [3, 4, 1])]
    r = run_graph(gpu, tmp_path, nodes, inits, {"x": x}, outs, opset=18)
    for op, fn in red.items():
        np.testing.assert_allclose(r["r_" + op], fn(xd, (0, 2), False), rtol=3e-5, atol=1e-5, err_msg=op)
    np.testing.assert_allclose(r["r_sum"], xd.sum(axis=2, keepdims=True), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(r["r_prod"], xd[:, :, :3].prod(axis=1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["r_all"], xd.mean(), rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(r["r_argmax"], x.argmax(axis=2))
    np.testing.assert_array_equal(r["r_argmin"], x.argmin(axis=1)[:, None, :])
    sm = lambda v, a: np.exp(v - v.max(axis=a, keepdims=True)) / np.exp(v - v.max(axis=a, keepdims=True)).sum(axis=a, keepdims=True)
    np.testing.assert_allclose(r["r_sm_last"], sm(xd, 2), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r["r_sm_mid"], sm(xd, 1), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r["r_lsm"], np.log(sm(xd, 2)), rtol=2e-5, atol=2e-6)
    norm = lambda v, a: (v - v.mean(axis=a, keepdims=True)) / np.sqrt(v.var(axis=a, keepdims=True) + 1e-5)
    np.testing.assert_allclose(r["r_ln"], norm(xd, 2) * g + b, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r["r_in"], norm(xd, 2) * cg[None, :, None] + cb[None, :, None], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r["r_bn"], (xd - cm[None, :, None]) / np.sqrt(cv[None, :, None] + 1e-5) * cg[None, :, None] + cb[None, :, None], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r["r_cs"], np.cumsum(xd, axis=2), rtol=1e-5, atol=1e-4)
    rev = np.flip(np.cumsum(np.flip(xd, 1), axis=1), 1) - xd
    np.testing.assert_allclose(r["r_cs_rev"], rev, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["r_gap"], xd.mean(axis=2, keepdims=True), rtol=1e-5, atol=1e-6)


def test_matmul_gemm(gpu, tmp_path):
    rng = np.random.default_rng(5)
    a = rng.standard_normal((2, 3, 17, 33)).astype(np.float32); b = rng.standard_normal((2, 3, 33, 21)).astype(np.float32)
    w = rng.standard_normal((33, 40)).astype(np.float32); v = rng.standard_normal(33).astype(np.float32)
    bb = rng.standard_normal((1, 3, 33, 21)).astype(np.float32)
    ga = rng.standard_normal((19, 33)).astype(np.float32); gb = rng.standard_normal((40, 33)).astype(np.float32); gc = rng.standard_normal(40).astype(np.float32)
    gat = np.ascontiguousarray(ga.T); gc2 = rng.standard_normal((19, 1)).astype(np.float32)
    inits = [W.tensor("w", w), W.tensor("v", v), W.tensor("gb", gb), W.tensor("gc", gc), W.tensor("gc2", gc2)]
    nodes = [W.node("MatMul", ["a", "b"], ["m_batched"]), W.node("MatMul", ["a", "w"], ["m_weight"]), W.node("MatMul", ["a", "v"], ["m_vec"]),
             W.node("MatMul", ["a", "bb"], ["m_bcast"]),
             W.node("Gemm", ["ga", "gb", "gc"], ["g_tb"], attrs=[W.attr_int("transB", 1), W.attr_float("alpha", 0.5), W.attr_float("beta", 2.0)]),
             W.node("Gemm", ["gat", "gb", "gc2"], ["g_tab"], attrs=[W.attr_int("transA", 1), W.attr_int("transB", 1)])]
    outs = [("m_batched", F32, [2, 3, 17, 21]), ("m_weight", F32, [2, 3, 17, 40]), ("m_vec", F32, [2, 3, 17]), ("m_bcast", F32, [2, 3, 17, 21]),
            ("g_tb", F32, [19, 40]), ("g_tab", F32, [19, 40])]
    r = run_graph(gpu, tmp_path, nodes, inits, {"a": a, "b": b, "bb": bb, "ga": ga, "gat": gat}, outs)
    d = lambda z: z.astype(np.float64)
    np.testing.assert_allclose(r["m_batched"], d(a) @ d(b), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["m_weight"], d(a) @ d(w), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["m_vec"], d(a) @ d(v), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["m_bcast"], d(a) @ d(bb), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["g_tb"], 0.5 * d(ga) @ d(gb).T + 2.0 * d(gc), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["g_tab"], d(ga) @ d(gb).T + d(gc2), rtol=1e-5, atol=1e-5)


def test_convolutions_and_pad_vs_torch(gpu, tmp_path):
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(6)
    x1 = rng.standard_normal((2, 8, 50)).astype(np.float32)
    w1 = rng.standard_normal((12, 4, 5)).astype(np.float32); b1 = rng.standard_normal(12).astype(np.float32)      # groups 2, stride 2, dilation 2, pads (3, 1)
    w1s = rng.standard_normal((6, 8, 7)).astype(np.float32)                                                       # SAME_UPPER, stride 3
    wt1 = rng.standard_normal((8, 3, 4)).astype(np.float32); bt1 = rng.standard_normal(6).astype(np.float32)      # ConvTranspose: groups 2 -> 6 channels, stride 3, pads (1, 2), output_padding 1
    x2 = rng.standard_normal((1, 4, 9, 11)).astype(np.float32)
    w2 = rng.standard_normal((6, 4, 3, 2)).astype(np.float32); b2 = rng.standard_normal(6).astype(np.float32)     # 2-D: strides (2, 1), pads (1, 0, 1, 1)
    wt2 = rng.standard_normal((4, 5, 2, 3)).astype(np.float32)                                                    # 2-D transposed: strides (2, 2), dilations (1, 2)
    inits = [W.tensor(n, v) for n, v in (("w1", w1), ("b1", b1), ("w1s", w1s), ("wt1", wt1), ("bt1", bt1), ("w2", w2), ("b2", b2), ("wt2", wt2))]
    inits += [W.tensor("pads", np.asarray([0, 0, 2, 0, 0, 3], np.int64)), W.tensor("padv", np.float32(0.25).reshape(()))]
    nodes = [
        W.node("Conv", ["x1", "w1", "b1"], ["c1"], attrs=[W.attr_int("group", 2), W.attr_ints("strides", [2]), W.attr_ints("dilations", [2]), W.attr_ints("pads", [3, 1])]),
        W.node("Conv", ["x1", "w1s"], ["c1s"], attrs=[W.attr_str("auto_pad", "SAME_UPPER"), W.attr_ints("strides", [3])]),
        W.node("ConvTranspose", ["x1", "wt1", "bt1"], ["t1"], attrs=[W.attr_int("group", 2), W.attr_ints("strides", [3]), W.attr_ints("pads", [1, 2]), W.attr_ints("output_padding", [1])]),
        W.node("Conv", ["x2", "w2", "b2"], ["c2"], attrs=[W.attr_ints("strides", [2, 1]), W.attr_ints("pads", [1, 0, 1, 1])]),
        W.node("ConvTranspose", ["x2", "wt2"], ["t2"], attrs=[W.attr_ints("strides", [2, 2]), W.attr_ints("dilations", [1, 2])]),
        W.node("Pad", ["x1", "pads", "padv"], ["p_const"]),
        W.node("Pad", ["x1", "pads"], ["p_reflect"], attrs=[W.attr_str("mode", "reflect")]),
        W.node("Pad", ["x1", "pads"], ["p_edge"], attrs=[W.attr_str("mode", "edge")]),
    ]
    T = torch.from_numpy
    ref = {
        "c1": Fn.conv1d(Fn.pad(T(x1), (3, 1)), T(w1), T(b1), stride=2, dilation=2, groups=2),
        "c1s": Fn.conv1d(Fn.pad(T(x1), (2, 3)), T(w1s), None, stride=3),           # in 50, stride 3 -> 17 outputs, total pad 5 = (2, 3)
        "t1": Fn.conv_transpose1d(T(x1), T(wt1), T(bt1), stride=3, padding=0, output_padding=0, groups=2),
        "c2": Fn.conv2d(Fn.pad(T(x2), (0, 1, 1, 1)), T(w2), T(b2), stride=(2, 1)),
        "t2": Fn.conv_transpose2d(T(x2), T(wt2), None, stride=(2, 2), dilation=(1, 2)),
        "p_const": Fn.pad(T(x1), (2, 3), value=0.25), "p_reflect": Fn.pad(T(x1), (2, 3), mode="reflect"), "p_edge": Fn.pad(T(x1), (2, 3), mode="replicate"),
    }
    # pads (1, 2) + output_padding 1: output position o is position o + 1 of the unpadded result; length (50 - 1) * 3 - 1 - 2 + 4 + 1 = 149
    ref["t1"] = ref["t1"][:, :, 1:150]
    outs = [(k, F32, list(v.shape)) for k, v in ref.items()]
    r = run_graph(gpu, tmp_path, nodes, inits, {"x1": x1, "x2": x2}, outs)
    for k, v in ref.items():
        assert r[k].shape == tuple(v.shape), (k, r[k].shape, v.shape)
        np.testing.assert_allclose(r[k], v.numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


def test_matrix_core_matmul_and_conv1d_vs_torch(gpu, tmp_path):
    """MatMul / Gemm / ungrouped Conv1d / ConvTranspose1d with >= 16 rows and columns run on k_mm_mfma (exact-f32 matrix cores, implicit
    im2col; ConvTranspose = W^T x X + k_col2im1d).  Shapes here are decoder-like and deliberately ragged: several 64 x 64 tiles with partial edges, K not a
    multiple of 16, batch > 1, every stride pattern the operand loader distinguishes (row-major, transposed, broadcast batch), strides / dilations / asymmetric
    pads / output_padding.  Reference: torch in float64."""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(61)
    f = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    a, b, w, bb = f(3, 70, 45), f(3, 45, 130), f(45, 200), f(1, 45, 33)
    ga, gb, gc, gat = f(100, 77), f(90, 77), f(90), f(77, 100)
    x = f(2, 40, 333)
    w7, b7 = f(72, 40, 7), f(72)          # k = 7, dilation 3, pads (9, 9): the decoder's residual-unit convolution
    w5 = f(50, 40, 5)                     # stride 2, pads (3, 1), no bias
    w1, b1 = f(96, 40, 1), f(96)          # pointwise
    xt = f(2, 48, 57)
    wt, bt = f(48, 40, 16), f(40)         # stride 8, pads (4, 4): the decoder's upsampling block
    wt2 = f(48, 24, 5)                    # stride 3, dilation 2, pads (1, 0), output_padding 2
    inits = [W.tensor(n, v) for n, v in (("w", w), ("gb", gb), ("gc", gc), ("w7", w7), ("b7", b7), ("w5", w5), ("w1", w1), ("b1", b1), ("wt", wt), ("bt", bt), ("wt2", wt2))]
    nodes = [
        W.node("Transpose", ["bb"], ["bbT"], attrs=[W.attr_ints("perm", [0, 2, 1])]), W.node("Transpose", ["a"], ["a2"], attrs=[W.attr_ints("perm", [0, 2, 1])]),
        W.node("MatMul", ["a", "b"], ["m_batched"]), W.node("MatMul", ["a", "w"], ["m_weight"]), W.node("MatMul", ["bbT", "a2"], ["m_bcast"]),
        W.node("Gemm", ["ga", "gb", "gc"], ["g_tb"], attrs=[W.attr_int("transB", 1), W.attr_float("alpha", 0.5), W.attr_float("beta", 2.0)]),
        W.node("Gemm", ["gat", "gb"], ["g_tab"], attrs=[W.attr_int("transA", 1), W.attr_int("transB", 1)]),
        W.node("Conv", ["x", "w7", "b7"], ["c7"], attrs=[W.attr_ints("dilations", [3]), W.attr_ints("pads", [9, 9])]),
        W.node("Conv", ["x", "w5"], ["c5"], attrs=[W.attr_ints("strides", [2]), W.attr_ints("pads", [3, 1])]),
        W.node("Conv", ["x", "w1", "b1"], ["c1"]),
        W.node("ConvTranspose", ["xt", "wt", "bt"], ["t8"], attrs=[W.attr_ints("strides", [8]), W.attr_ints("pads", [4, 4])]),
        W.node("ConvTranspose", ["xt", "wt2"], ["t3"], attrs=[W.attr_ints("strides", [3]), W.attr_ints("dilations", [2]), W.attr_ints("pads", [1, 0]), W.attr_ints("output_padding", [2])]),
    ]
    D = lambda v: torch.from_numpy(v).double()
    full_t3 = Fn.conv_transpose1d(D(xt), D(wt2), None, stride=3, dilation=2)          # length (57 - 1) * 3 + 2 * 4 + 1 = 177; pads (1, 0) + output_padding 2 -> 178
    t3 = torch.zeros(2, 24, 178, dtype=torch.float64)
    t3[:, :, :176] = full_t3[:, :, 1:]
    ref = {
        "m_batched": D(a) @ D(b), "m_weight": D(a) @ D(w), "m_bcast": D(bb).transpose(1, 2) @ D(a).transpose(1, 2),
        "g_tb": 0.5 * D(ga) @ D(gb).T + 2.0 * D(gc), "g_tab": D(gat).T @ D(gb).T,
        "c7": Fn.conv1d(D(x), D(w7), D(b7), dilation=3, padding=9), "c5": Fn.conv1d(Fn.pad(D(x), (3, 1)), D(w5), None, stride=2), "c1": Fn.conv1d(D(x), D(w1), D(b1)),
        "t8": Fn.conv_transpose1d(D(xt), D(wt), D(bt), stride=8, padding=4), "t3": t3,
    }
    outs = [(k, F32, list(v.shape)) for k, v in ref.items()]
    r = run_graph(gpu, tmp_path, nodes, inits, {"a": a, "b": b, "bb": bb, "ga": ga, "gat": gat, "x": x, "xt": xt}, outs)
    for k, v in ref.items():
        assert r[k].shape == tuple(v.shape), (k, r[k].shape, v.shape)
        np.testing.assert_allclose(r[k], v.numpy(), rtol=1e-4, atol=1e-4, err_msg=k)


def test_speaker_encoder_shaped_graph(gpu, tmp_path):
    """`mels` [1, n, 128] -> `spk_emb` [1, 2048] (onnx.rs:140-163): a small TDNN / squeeze-excitation / attentive-statistics-pooling network with the
    reference's input and output names, against the same network written in torch"""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(7)
    n, C = 37, 24
    mels = rng.standard_normal((1, n, 128)).astype(np.float32)
    P = {k: (rng.standard_normal(s) * sc).astype(np.float32) for k, s, sc in (
        ("w0", (C, 128, 5), 0.05), ("b0", (C,), 0.1), ("bn_g", (C,), 1.0), ("bn_b", (C,), 0.1), ("bn_m", (C,), 0.1),
        ("wd", (C // 2, C // 2, 3), 0.2), ("bd", (C // 2,), 0.1), ("se1", (6, C, 1), 0.2), ("se1b", (6,), 0.1), ("se2", (C, 6, 1), 0.2), ("se2b", (C,), 0.1),
        ("att1", (8, C, 1), 0.2), ("att1b", (8,), 0.1), ("att2", (C, 8, 1), 0.2), ("att2b", (C,), 0.1), ("fc", (2048, 2 * C), 0.1), ("fcb", (2048,), 0.1))}
    P["bn_v"] = (rng.random(C) + 0.5).astype(np.float32)
    inits = [W.tensor(k, v) for k, v in P.items()] + [W.tensor("ax2", np.asarray([2], np.int64)), W.tensor("eps", np.float32(1e-5).reshape(())),
                                                      W.tensor("big", np.float32(1e9).reshape(())), W.tensor("half", np.asarray([C // 2, C // 2], np.int64))]
    nodes = [
        W.node("Transpose", ["mels"], ["x"], attrs=[W.attr_ints("perm", [0, 2, 1])]),
        W.node("Conv", ["x", "w0", "b0"], ["h0"], attrs=[W.attr_ints("pads", [2, 2])]),
        W.node("Relu", ["h0"], ["h0r"]),
        W.node("BatchNormalization", ["h0r", "bn_g", "bn_b", "bn_m", "bn_v"], ["h1"]),
        W.node("Split", ["h1", "half"], ["ha", "hb"], attrs=[W.attr_int("axis", 1)]),                      # Res2Net-style: second half through a dilated conv
        W.node("Conv", ["hb", "wd", "bd"], ["hbd"], attrs=[W.attr_ints("dilations", [2]), W.attr_ints("pads", [2, 2])]),
        W.node("Concat", ["ha", "hbd"], ["h2"], attrs=[W.attr_int("axis", 1)]),
        W.node("ReduceMean", ["h2"], ["s"], attrs=[W.attr_ints("axes", [2]), W.attr_int("keepdims", 1)]),   # squeeze-excitation
        W.node("Conv", ["s", "se1", "se1b"], ["s1"]), W.node("Relu", ["s1"], ["s1r"]), W.node("Conv", ["s1r", "se2", "se2b"], ["s2"]), W.node("Sigmoid", ["s2"], ["gate"]),
        W.node("Mul", ["h2", "gate"], ["h3"]), W.node("Add", ["h3", "h1"], ["h4"]),
        W.node("Conv", ["h4", "att1", "att1b"], ["a1"]), W.node("Tanh", ["a1"], ["a1t"]), W.node("Conv", ["a1t", "att2", "att2b"], ["a2"]),
        W.node("Softmax", ["a2"], ["wgt"], attrs=[W.attr_int("axis", 2)]),                                    # attentive statistics pooling
        W.node("Mul", ["h4", "wgt"], ["hw"]), W.node("ReduceSum", ["hw", "ax2"], ["mu"], attrs=[W.attr_int("keepdims", 1)]),
        W.node("Mul", ["h4", "h4"], ["hsq"]), W.node("Mul", ["hsq", "wgt"], ["hsqw"]), W.node("ReduceSum", ["hsqw", "ax2"], ["ex2"], attrs=[W.attr_int("keepdims", 1)]),
        W.node("Mul", ["mu", "mu"], ["mu2"]), W.node("Sub", ["ex2", "mu2"], ["var"]), W.node("Clip", ["var", "eps", "big"], ["varc"]), W.node("Sqrt", ["varc"], ["sd"]),
        W.node("Concat", ["mu", "sd"], ["stat"], attrs=[W.attr_int("axis", 1)]), W.node("Flatten", ["stat"], ["statf"]),
        W.node("Gemm", ["statf", "fc", "fcb"], ["spk_emb"], attrs=[W.attr_int("transB", 1)]),
    ]
    r = run_graph(gpu, tmp_path, nodes, inits, {"mels": mels}, [("spk_emb", F32, [1, 2048])], opset=13)
    T = lambda k: torch.from_numpy(P[k]).double()
    x = torch.from_numpy(mels).double().transpose(1, 2)
    h1 = Fn.batch_norm(Fn.relu(Fn.conv1d(x, T("w0"), T("b0"), padding=2)), T("bn_m"), T("bn_v"), T("bn_g"), T("bn_b"), eps=1e-5)
    ha, hb = h1[:, :C // 2], h1[:, C // 2:]
    h2 = torch.cat([ha, Fn.conv1d(hb, T("wd"), T("bd"), dilation=2, padding=2)], 1)
    gate = torch.sigmoid(Fn.conv1d(Fn.relu(Fn.conv1d(h2.mean(2, keepdim=True), T("se1"), T("se1b"))), T("se2"), T("se2b")))
    h4 = h2 * gate + h1
    wgt = torch.softmax(Fn.conv1d(torch.tanh(Fn.conv1d(h4, T("att1"), T("att1b"))), T("att2"), T("att2b")), dim=2)
    mu = (h4 * wgt).sum(2, keepdim=True)
    sd = torch.sqrt(((h4 * h4 * wgt).sum(2, keepdim=True) - mu * mu).clamp(1e-5, 1e9))
    ref = Fn.linear(torch.cat([mu, sd], 1).flatten(1), T("fc"), T("fcb")).numpy()
    assert r["spk_emb"].shape == (1, 2048)
    np.testing.assert_allclose(r["spk_emb"], ref, rtol=2e-4, atol=2e-4)


def _codec_encoder_case(seed, Tn, D, NQ, CB):
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(seed)
    wav = (rng.standard_normal((1, Tn)) * 0.3).astype(np.float32)
    P = {k: (rng.standard_normal(s) * sc).astype(np.float32) for k, s, sc in (
        ("w1", (8, 1, 7), 0.3), ("b1", (8,), 0.1), ("w2", (D, 8, 8), 0.15), ("b2", (D,), 0.1), ("w3", (D, D, 16), 0.08), ("b3", (D,), 0.1),
        ("ln_g", (D,), 1.0), ("ln_b", (D,), 0.1), ("wq", (D, D), 0.3), ("wk", (D, D), 0.3), ("wv", (D, D), 0.3), ("wo", (D, D), 0.3), ("cb", (NQ, CB, D), 1.0))}
    Tt = lambda k: torch.from_numpy(P[k]).double()
    x = torch.from_numpy(wav).double()[:, None]
    h = Fn.conv1d(Fn.elu(Fn.conv1d(Fn.elu(Fn.conv1d(x, Tt("w1"), Tt("b1"), padding=3)), Tt("w2"), Tt("b2"), stride=4, padding=2)), Tt("w3"), Tt("b3"), stride=8, padding=4)
    t = h.transpose(1, 2)
    tn = Fn.layer_norm(t, (D,), Tt("ln_g"), Tt("ln_b"), 1e-5)
    pr = torch.softmax((tn @ Tt("wq")) @ (tn @ Tt("wk")).transpose(1, 2) / np.sqrt(D), -1)
    rsd = t + (pr @ (tn @ Tt("wv"))) @ Tt("wo")
    codes, margins = [], []
    cb = torch.from_numpy(P["cb"]).double()
    for qn in range(NQ):
        dist = (cb[qn] ** 2).sum(1) - 2 * rsd @ cb[qn].T
        srt = torch.sort(dist, dim=-1).values
        margins.append((srt[..., 1] - srt[..., 0]).min().item())
        c = dist.argmin(-1)
        codes.append(c)
        rsd = rsd - cb[qn][c]
    return wav, P, torch.stack(codes, -1).numpy(), rsd.numpy(), min(margins)


def test_codec_encoder_shaped_graph(gpu, tmp_path):
    """`input_values` [1, T] -> `audio_codes` [1, F, n_q] i64 (onnx.rs:97-121): strided ELU conv stack, one self-attention block with LayerNorm, and a
    residual vector quantiser written out in ONNX ops (distances by MatMul, ArgMin, Gather, Sub), against the same network in torch (f64).  The
    seed is the first whose nearest / second-nearest codebook distances differ by more than f32 noise everywhere, so the codes are well defined."""
    Tn, D, NQ, CB = 1920, 16, 3, 32
    for seed in range(8, 40):
        wav, P, ref, ref_res, margin = _codec_encoder_case(seed, Tn, D, NQ, CB)
        if margin > 5e-3:
            break
    assert margin > 5e-3
    inits = [W.tensor(k, v) for k, v in P.items() if k != "cb"] + [W.tensor("cb%d" % q, P["cb"][q]) for q in range(NQ)]
    inits += [W.tensor("cbT%d" % q, np.ascontiguousarray(P["cb"][q].T)) for q in range(NQ)] + [W.tensor("cbn%d" % q, (P["cb"][q].astype(np.float64) ** 2).sum(1).astype(np.float32)) for q in range(NQ)]
    inits += [W.tensor("ax1", np.asarray([1], np.int64)), W.tensor("ax2", np.asarray([2], np.int64)), W.tensor("scale", np.float32(1 / np.sqrt(D)).reshape(())),
              W.tensor("two", np.float32(2).reshape(()))]
    nodes = [
        W.node("Unsqueeze", ["input_values", "ax1"], ["x"]),
        W.node("Conv", ["x", "w1", "b1"], ["h1"], attrs=[W.attr_ints("pads", [3, 3])]), W.node("Elu", ["h1"], ["h1e"]),
        W.node("Conv", ["h1e", "w2", "b2"], ["h2"], attrs=[W.attr_ints("strides", [4]), W.attr_ints("pads", [2, 2])]), W.node("Elu", ["h2"], ["h2e"]),
        W.node("Conv", ["h2e", "w3", "b3"], ["h3"], attrs=[W.attr_ints("strides", [8]), W.attr_ints("pads", [4, 4])]),
        W.node("Transpose", ["h3"], ["t"], attrs=[W.attr_ints("perm", [0, 2, 1])]),                                # [1, F, D]
        W.node("LayerNormalization", ["t", "ln_g", "ln_b"], ["tn"], attrs=[W.attr_float("epsilon", 1e-5)]),
        W.node("MatMul", ["tn", "wq"], ["q"]), W.node("MatMul", ["tn", "wk"], ["k"]), W.node("MatMul", ["tn", "wv"], ["v"]),
        W.node("Transpose", ["k"], ["kT"], attrs=[W.attr_ints("perm", [0, 2, 1])]), W.node("MatMul", ["q", "kT"], ["sc"]), W.node("Mul", ["sc", "scale"], ["scs"]),
        W.node("Softmax", ["scs"], ["pr"], attrs=[W.attr_int("axis", -1)]), W.node("MatMul", ["pr", "v"], ["ctx"]), W.node("MatMul", ["ctx", "wo"], ["att"]),
        W.node("Add", ["t", "att"], ["res0"]),
    ]
    res = "res0"
    for qn in range(NQ):  # residual VQ: argmin_c |r|^2 - 2 r.c + |c|^2  (the |r|^2 term does not change the argmin and is left out, as exporters do)
        nodes += [W.node("MatMul", [res, "cbT%d" % qn], ["dot%d" % qn]), W.node("Mul", ["dot%d" % qn, "two"], ["dot2_%d" % qn]),
                  W.node("Sub", ["cbn%d" % qn, "dot2_%d" % qn], ["dist%d" % qn]),
                  W.node("ArgMin", ["dist%d" % qn], ["code%d" % qn], attrs=[W.attr_int("axis", -1), W.attr_int("keepdims", 1)]),
                  W.node("Squeeze", ["code%d" % qn, "ax2"], ["codes%d" % qn]),
                  W.node("Gather", ["cb%d" % qn, "codes%d" % qn], ["quant%d" % qn], attrs=[W.attr_int("axis", 0)]),
                  W.node("Sub", [res, "quant%d" % qn], ["res%d" % (qn + 1)])]
        res = "res%d" % (qn + 1)
    nodes.append(W.node("Concat", ["code%d" % qn for qn in range(NQ)], ["audio_codes"], attrs=[W.attr_int("axis", 2)]))
    F = ((Tn + 4 - 8) // 4 + 1 + 8 - 16) // 8 + 1
    assert ref.shape == (1, F, NQ)
    r = run_graph(gpu, tmp_path, nodes, inits, {"input_values": wav}, [("audio_codes", I64, [1, F, NQ]), (res, F32, [1, F, D])])
    assert r["audio_codes"].shape == (1, F, NQ) and r["audio_codes"].dtype == np.int64
    np.testing.assert_array_equal(r["audio_codes"], ref)
    np.testing.assert_allclose(r[res], ref_res, rtol=1e-3, atol=2e-4)


def test_unsupported_op_is_reported(gpu, tmp_path):
    path = os.path.join(str(tmp_path), "u.onnx")
    open(path, "wb").write(W.model([W.node("LSTM", ["x"], ["y"])], [], [W.value_info("x", F32, [1, 2])], [W.value_info("y", F32, [1, 2])]))
    s = gpu.OnnxSession(path)
    assert s.unsupported() == ["LSTM"] and not gpu.onnx_op_executable("LSTM") and gpu.onnx_op_executable("Conv")
    with pytest.raises(RuntimeError, match="LSTM"):
        s.run({"x": np.zeros((1, 2), np.float32)}, ["y"])
    s.close()


def _write_wav(path, samples_i16, rate, channels=1):
    data = np.asarray(samples_i16, "<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * 2 * channels, 2 * channels, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def test_create_voice_file_through_the_encoder_graphs(gpu, tiny_model, tmp_path):
    """row a17 end to end: a model directory with `onnx/qwen3_tts_codec_encoder.onnx` + `onnx/qwen3_tts_speaker_encoder.onnx` (test-written graphs with the
    reference's tensor names) makes TtsEngine::create_voice_file work (engine.rs:324-387: 24 kHz WAV -> audio_codes + 2048-d speaker embedding) and lets
    `generate` start from raw reference audio, writing the `.cache` file (engine.rs:275-301).  Expected values: the same graphs through ctypes."""
    rng = np.random.default_rng(11)
    # --- codec encoder: input_values [1, T] -> audio_codes [1, F, 16]: two strided convs, then 16 "quantisers" = argmax over 16 slices of a projection
    D, NQ, CB = 12, 16, 8
    Pc = {"w1": (rng.standard_normal((8, 1, 9)) * 0.3).astype(np.float32), "b1": (rng.standard_normal(8) * 0.1).astype(np.float32),
          "w2": (rng.standard_normal((D, 8, 16)) * 0.1).astype(np.float32), "b2": (rng.standard_normal(D) * 0.1).astype(np.float32),
          "proj": (rng.standard_normal((D, NQ * CB)) * 0.5).astype(np.float32)}
    cn = [W.node("Unsqueeze", ["input_values", "ax1"], ["x"]),
          W.node("Conv", ["x", "w1", "b1"], ["h1"], attrs=[W.attr_ints("strides", [8]), W.attr_ints("pads", [4, 4])]), W.node("Elu", ["h1"], ["h1e"]),
          W.node("Conv", ["h1e", "w2", "b2"], ["h2"], attrs=[W.attr_ints("strides", [16]), W.attr_ints("pads", [8, 7])]), W.node("Tanh", ["h2"], ["h2t"]),
          W.node("Transpose", ["h2t"], ["t"], attrs=[W.attr_ints("perm", [0, 2, 1])]), W.node("MatMul", ["t", "proj"], ["lg"]),
          W.node("Shape", ["lg"], ["lgs"]), W.node("Slice", ["lgs", "c0", "c2"], ["bf"]), W.node("Concat", ["bf", "qc"], ["ns"], attrs=[W.attr_int("axis", 0)]),
          W.node("Reshape", ["lg", "ns"], ["lg4"]), W.node("ArgMax", ["lg4"], ["audio_codes"], attrs=[W.attr_int("axis", 3), W.attr_int("keepdims", 0)])]
    ci = [W.tensor(k, v) for k, v in Pc.items()] + [W.tensor("ax1", np.asarray([1], np.int64)), W.tensor("c0", np.asarray([0], np.int64)),
                                                   W.tensor("c2", np.asarray([2], np.int64)), W.tensor("qc", np.asarray([NQ, CB], np.int64))]
    # --- speaker encoder: mels [1, n, 128] -> spk_emb [1, 2048]: conv + relu + mean / std pooling + linear
    C = 16
    Ps = {"w0": (rng.standard_normal((C, 128, 3)) * 0.05).astype(np.float32), "b0": (rng.standard_normal(C) * 0.1).astype(np.float32),
          "fc": (rng.standard_normal((2048, 2 * C)) * 0.1).astype(np.float32), "fcb": (rng.standard_normal(2048) * 0.1).astype(np.float32)}
    sn = [W.node("Transpose", ["mels"], ["x"], attrs=[W.attr_ints("perm", [0, 2, 1])]), W.node("Conv", ["x", "w0", "b0"], ["h"], attrs=[W.attr_ints("pads", [1, 1])]),
          W.node("Relu", ["h"], ["hr"]), W.node("ReduceMean", ["hr"], ["mu"], attrs=[W.attr_ints("axes", [2]), W.attr_int("keepdims", 0)]),
          W.node("Mul", ["hr", "hr"], ["h2"]), W.node("ReduceMean", ["h2"], ["m2"], attrs=[W.attr_ints("axes", [2]), W.attr_int("keepdims", 0)]),
          W.node("Mul", ["mu", "mu"], ["mumu"]), W.node("Sub", ["m2", "mumu"], ["var"]), W.node("Relu", ["var"], ["varp"]), W.node("Sqrt", ["varp"], ["sd"]),
          W.node("Concat", ["mu", "sd"], ["st"], attrs=[W.attr_int("axis", 1)]), W.node("Gemm", ["st", "fc", "fcb"], ["spk_emb"], attrs=[W.attr_int("transB", 1)])]
    si = [W.tensor(k, v) for k, v in Ps.items()]
    mdir = tmp_path / "model"
    (mdir / "onnx").mkdir(parents=True)
    for e in os.listdir(tiny_model):
        if e != "onnx":
            os.symlink(os.path.join(tiny_model, e), str(mdir / e))
    for e in os.listdir(os.path.join(tiny_model, "onnx")):
        os.symlink(os.path.join(tiny_model, "onnx", e), str(mdir / "onnx" / e))
    enc_path, spk_path = str(mdir / "onnx" / "qwen3_tts_codec_encoder.onnx"), str(mdir / "onnx" / "qwen3_tts_speaker_encoder.onnx")
    open(enc_path, "wb").write(W.model(cn, ci, [W.value_info("input_values", F32, [1, "T"])], [W.value_info("audio_codes", I64, [1, "F", 16])], opset=13))
    open(spk_path, "wb").write(W.model(sn, si, [W.value_info("mels", F32, [1, "n", 128])], [W.value_info("spk_emb", F32, [1, 2048])], opset=13))
    t = np.arange(24000 * 1) / 24000.0
    pcm = (np.sin(2 * np.pi * 220 * t) * 0.4 + rng.standard_normal(t.size) * 0.05)
    i16 = np.clip(np.round(pcm * 32767), -32768, 32767).astype(np.int16)
    ref_wav, ref2_wav = str(tmp_path / "ref.wav"), str(tmp_path / "ref2.wav")
    _write_wav(ref_wav, np.stack([i16, -i16], 1).reshape(-1), 24000, channels=2)   # stereo: create_voice_file keeps channel 0
    _write_wav(ref2_wav, i16[:12000], 24000)
    _write_wav(ref_wav + ".16k.wav", i16[:1000], 16000)
    pkg = os.path.join(ROOT, "qwen3-tts-rust_amd")
    exe = str(tmp_path / "voice_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host", "voice_main.cpp"), "-L" + pkg, "-lq3tts_host", "-lq3tts",
                           "-Wl,-rpath," + pkg])
    vjson = str(tmp_path / "voice.json")
    r = subprocess.run([exe, str(mdir), ref_wav, vjson, ref2_wav], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-2000:])
    v = json.load(open(vjson))
    audio = (i16.astype(np.float32) / 32768.0)
    s1 = gpu.OnnxSession(enc_path)
    codes = s1.run({"input_values": audio[None, :]}, ["audio_codes"])["audio_codes"]
    s1.close()
    assert codes.shape[0] == 1 and codes.shape[2] == 16 and codes.min() >= 0 and codes.max() < CB and len(np.unique(codes)) > 2
    assert v["audio_codes"] == codes.reshape(-1).tolist() and v["ref_text"] == "reference text"
    s2 = gpu.OnnxSession(spk_path)
    emb = s2.run({"mels": gpu.mel(audio)[None]}, ["spk_emb"])["spk_emb"]
    s2.close()
    got = np.asarray(v.get("speaker_embedding", v.get("spk_emb")), np.float32)
    assert got.shape == (2048,) and np.isfinite(got).all() and np.abs(got).max() > 0
    np.testing.assert_allclose(got, emb.reshape(-1), rtol=1e-6, atol=1e-7)
    assert os.path.exists(str(tmp_path / "ref2.cache"))


def test_streaming_decoder_contract_over_the_executor(gpu, tmp_path):
    """The exported decoder's I/O contract (onnx.rs:341-458): `audio_codes`, `is_last` and zero-length-initialised state tensors in, `final_wav`,
    `valid_samples` and `next_*` out, state carried on the device between chunks.  The graph here is a small causal network written for the test
    (embedding sum, causal conv with `pre_conv_history`, a running mean through `past_key_0`, a second causal conv with `conv_history`, transposed-conv
    upsampling; its pads, slices and ranges are computed from Shape nodes as exporters do).  Chunked decoding must equal a numpy model of the whole
    sequence, trimmed per chunk by `valid_samples`."""
    rng = np.random.default_rng(21)
    D, V = 8, 32
    table = rng.standard_normal((V, D)).astype(np.float32) * 0.3
    wc = rng.standard_normal((D, D, 3)).astype(np.float32) * 0.2
    w2 = rng.standard_normal((D, D, 2)).astype(np.float32) * 0.2
    wt = rng.standard_normal((D, 1, 4)).astype(np.float32) * 0.3
    I = lambda name, v: W.tensor(name, np.asarray(v, np.int64))
    inits = [W.tensor("table", table), W.tensor("wc", wc), W.tensor("w2", w2), W.tensor("wt", wt), I("ax2", [2]), I("i0", 0), I("i1", 1), I("i2", 2), I("i4", 4), I("z1", [0]),
             I("zz", [0, 0]), I("zzz", [0, 0, 0]), I("m2", [-2]), I("m1", [-1]), I("m3", [-3]), I("big", [2 ** 62]), I("axk", [2]), I("shp_k", [1, -1, 2, 4]), I("shp_m", [1, -1, D]),
             I("shp_c", [1, 1, -1, 1]), W.tensor("halff", np.float32(0.5).reshape(())), W.tensor("twof", np.float32(2).reshape(()))]
    n = W.node
    nodes = [
        n("Gather", ["table", "audio_codes"], ["emb"]), n("ReduceSum", ["emb", "ax2"], ["x0"], attrs=[W.attr_int("keepdims", 0)]),
        n("Transpose", ["x0"], ["x"], attrs=[W.attr_ints("perm", [0, 2, 1])]),
        # causal conv (k = 3) with history
        n("Concat", ["pre_conv_history", "x"], ["cat1"], attrs=[W.attr_int("axis", 2)]),
        n("Shape", ["pre_conv_history"], ["s1"]), n("Gather", ["s1", "i2"], ["h1"]), n("Sub", ["i2", "h1"], ["p1"]), n("Unsqueeze", ["p1", "z1"], ["p1u"]),
        n("Concat", ["zz", "p1u", "zzz"], ["pads1"], attrs=[W.attr_int("axis", 0)]), n("Pad", ["cat1", "pads1"], ["pad1"]),
        n("Conv", ["pad1", "wc"], ["y"]), n("Slice", ["pad1", "m2", "big", "ax2"], ["next_pre_conv_history"]),
        # running mean over everything seen so far, through the key cache of layer 0
        n("Transpose", ["y"], ["yt"], attrs=[W.attr_ints("perm", [0, 2, 1])]), n("Reshape", ["yt", "shp_k"], ["k4"]),
        n("Transpose", ["k4"], ["knew"], attrs=[W.attr_ints("perm", [0, 2, 1, 3])]),
        n("Concat", ["past_key_0", "knew"], ["next_key_0"], attrs=[W.attr_int("axis", 2)]),
        n("Mul", ["knew", "twof"], ["vnew"]), n("Concat", ["past_value_0", "vnew"], ["next_value_0"], attrs=[W.attr_int("axis", 2)]),
        n("CumSum", ["next_key_0", "i2"], ["cs"]),
        n("Shape", ["audio_codes"], ["sc"]), n("Gather", ["sc", "i1"], ["N"]), n("Neg", ["N"], ["negN"]), n("Unsqueeze", ["negN", "z1"], ["negNu"]),
        n("Slice", ["cs", "negNu", "big", "axk"], ["csn"]),
        n("Shape", ["next_key_0"], ["sk"]), n("Gather", ["sk", "i2"], ["total"]), n("Sub", ["total", "N"], ["t0"]), n("Add", ["t0", "i1"], ["r0"]), n("Add", ["total", "i1"], ["r1"]),
        n("Range", ["r0", "r1", "i1"], ["cnt"]), n("Cast", ["cnt"], ["cntf"], attrs=[W.attr_int("to", 1)]), n("Reshape", ["cntf", "shp_c"], ["cnt4"]),
        n("Div", ["csn", "cnt4"], ["mean4"]), n("Transpose", ["mean4"], ["mean4t"], attrs=[W.attr_ints("perm", [0, 2, 1, 3])]), n("Reshape", ["mean4t", "shp_m"], ["mean3"]),
        n("Transpose", ["mean3"], ["m"], attrs=[W.attr_ints("perm", [0, 2, 1])]), n("Add", ["y", "m"], ["z"]),
        # latent buffer: the last three frames, carried only
        n("Concat", ["latent_buffer", "z"], ["lat"], attrs=[W.attr_int("axis", 2)]), n("Slice", ["lat", "m3", "big", "ax2"], ["next_latent_buffer"]),
        # second causal conv (k = 2) with its own history
        n("Concat", ["conv_history", "z"], ["cat2"], attrs=[W.attr_int("axis", 2)]),
        n("Shape", ["conv_history"], ["s2"]), n("Gather", ["s2", "i2"], ["h2"]), n("Sub", ["i1", "h2"], ["p2"]), n("Unsqueeze", ["p2", "z1"], ["p2u"]),
        n("Concat", ["zz", "p2u", "zzz"], ["pads2"], attrs=[W.attr_int("axis", 0)]), n("Pad", ["cat2", "pads2"], ["pad2"]),
        n("Conv", ["pad2", "w2"], ["u"]), n("Slice", ["pad2", "m1", "big", "ax2"], ["next_conv_history"]),
        n("ConvTranspose", ["u", "wt"], ["up"], attrs=[W.attr_ints("strides", [4])]), n("Tanh", ["up"], ["final_wav"]),
        # valid_samples: everything on the last chunk, two samples held back otherwise
        n("Mul", ["N", "i4"], ["full"]), n("Sub", ["full", "i2"], ["held"]), n("Unsqueeze", ["full", "z1"], ["fullu"]), n("Unsqueeze", ["held", "z1"], ["heldu"]),
        n("Greater", ["is_last", "halff"], ["lastb"]), n("Where", ["lastb", "fullu", "heldu"], ["valid_samples"]),
    ]
    ins = [W.value_info("audio_codes", I64, [1, "N", 16]), W.value_info("is_last", F32, [1]), W.value_info("pre_conv_history", F32, [1, D, "h"]),
           W.value_info("latent_buffer", F32, [1, D, "l"]), W.value_info("conv_history", F32, [1, D, "c"])]
    outs = [W.value_info("final_wav", F32, [1, 1, "S"]), W.value_info("valid_samples", I64, [1]), W.value_info("next_pre_conv_history", F32, [1, D, 2]),
            W.value_info("next_latent_buffer", F32, [1, D, "l2"]), W.value_info("next_conv_history", F32, [1, D, 1])]
    for i in range(8):
        ins += [W.value_info("past_key_%d" % i, F32, [1, 2, "t", 4]), W.value_info("past_value_%d" % i, F32, [1, 2, "t", 4])]
        outs += [W.value_info("next_key_%d" % i, F32, [1, 2, "t2", 4]), W.value_info("next_value_%d" % i, F32, [1, 2, "t2", 4])]
        if i > 0:
            nodes += [n("Identity", ["past_key_%d" % i], ["next_key_%d" % i]), n("Identity", ["past_value_%d" % i], ["next_value_%d" % i])]
    path = str(tmp_path / "dec.onnx")
    open(path, "wb").write(W.model(nodes, inits, ins, outs, opset=13))
    m = gpu.OnnxModel(path)
    assert m.decoder_contract() == (True, "") or m.decoder_contract()[0] is True
    m.close()
    T = 11
    codes = rng.integers(0, V, (T, 16))
    # numpy model of the whole sequence
    x = table[codes].sum(1).T.astype(np.float64)                                    # [D, T]
    xp = np.concatenate([np.zeros((D, 2)), x], 1)
    y = np.stack([sum(wc[:, :, k].astype(np.float64) @ xp[:, t + k] for k in range(3)) for t in range(T)], 1)
    mean = np.cumsum(y, 1) / np.arange(1, T + 1)
    z = y + mean
    zp = np.concatenate([np.zeros((D, 1)), z], 1)
    u = np.stack([sum(w2[:, :, k].astype(np.float64) @ zp[:, t + k] for k in range(2)) for t in range(T)], 1)
    wav = np.tanh(np.einsum("dt,dj->tj", u, wt[:, 0, :].astype(np.float64)).reshape(-1))   # [4 T]
    dec = gpu.OnnxDecoder(path)
    for chunks in ([3, 4, 1, 3], [11], [1] * 11):
        dec.reset()
        got, exp, at = [], [], 0
        for ci, c in enumerate(chunks):
            last = ci == len(chunks) - 1
            pcm = dec.decode(codes[at:at + c], is_final=last, max_samples_per_frame=4)
            assert pcm.size == (4 * c if last else 4 * c - 2)
            got.append(pcm); exp.append(wav[4 * at: 4 * at + pcm.size]); at += c
        np.testing.assert_allclose(np.concatenate(got), np.concatenate(exp), rtol=2e-4, atol=2e-5, err_msg=str(chunks))
    assert dec.decode(np.zeros((0, 16), np.int64)).size == 0                         # n_frames == 0 -> empty (onnx.rs:350-353)
    dec.close()


def test_more_ops_trilu_scatter_resize_groupnorm(gpu, tmp_path):
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 6, 5, 7)).astype(np.float32)
    kv = rng.standard_normal((1, 2, 9, 4)).astype(np.float32)
    upd = rng.standard_normal((3, 4)).astype(np.float32)
    gidx = rng.integers(0, 5, (2, 6, 3, 7))
    gg = rng.standard_normal(6).astype(np.float32); gb = rng.standard_normal(6).astype(np.float32)
    I = lambda name, v: W.tensor(name, np.asarray(v, np.int64))
    inits = [I("km1", -1), I("sidx", [[0, 1, 2], [0, 0, 8], [0, 1, 0]]), I("gidx", gidx), W.tensor("upd", upd), W.tensor("gg", gg), W.tensor("gb", gb),
             W.tensor("scales", np.asarray([1, 1, 2, 3], np.float32)), W.tensor("roi", np.zeros(0, np.float32)), I("sizes", [2, 6, 10, 7])]
    n = W.node
    nodes = [n("Trilu", ["x"], ["tu"]), n("Trilu", ["x", "km1"], ["tl"], attrs=[W.attr_int("upper", 0)]),
             n("GatherElements", ["x", "gidx"], ["ge"], attrs=[W.attr_int("axis", 2)]),
             n("ScatterND", ["kv", "sidx", "upd"], ["sc"]),
             n("Resize", ["x", "roi", "scales"], ["rs"], attrs=[W.attr_str("mode", "nearest"), W.attr_str("coordinate_transformation_mode", "asymmetric"), W.attr_str("nearest_mode", "floor")]),
             n("Resize", ["x", "roi", "", "sizes"], ["rz"], attrs=[W.attr_str("mode", "nearest"), W.attr_str("coordinate_transformation_mode", "asymmetric"), W.attr_str("nearest_mode", "floor")]),
             n("GroupNormalization", ["x", "gg", "gb"], ["gn"], attrs=[W.attr_int("num_groups", 3), W.attr_float("epsilon", 1e-5)]),
             n("LpNormalization", ["x"], ["lp"], attrs=[W.attr_int("axis", 1), W.attr_int("p", 2)]),
             n("Mish", ["x"], ["mish"]), n("Celu", ["x"], ["celu"], attrs=[W.attr_float("alpha", 1.5)]), n("ThresholdedRelu", ["x"], ["thr"], attrs=[W.attr_float("alpha", 0.3)])]
    outs = [("tu", F32, list(x.shape)), ("tl", F32, list(x.shape)), ("ge", F32, list(gidx.shape)), ("sc", F32, list(kv.shape)), ("rs", F32, [2, 6, 10, 21]), ("rz", F32, [2, 6, 10, 7]),
            ("gn", F32, list(x.shape)), ("lp", F32, list(x.shape)), ("mish", F32, list(x.shape)), ("celu", F32, list(x.shape)), ("thr", F32, list(x.shape))]
    r = run_graph(gpu, tmp_path, nodes, inits, {"x": x, "kv": kv}, outs, opset=21)
    np.testing.assert_array_equal(r["tu"], np.triu(x))
    np.testing.assert_array_equal(r["tl"], np.tril(x, -1))
    np.testing.assert_array_equal(r["ge"], np.take_along_axis(x, gidx, axis=2))
    ref = kv.copy(); ref[0, 1, 2] = upd[0]; ref[0, 0, 8] = upd[1]; ref[0, 1, 0] = upd[2]
    np.testing.assert_array_equal(r["sc"], ref)
    np.testing.assert_array_equal(r["rs"], np.repeat(np.repeat(x, 2, axis=2), 3, axis=3))
    np.testing.assert_array_equal(r["rz"], np.repeat(x, 2, axis=2))
    np.testing.assert_allclose(r["gn"], Fn.group_norm(torch.from_numpy(x).double(), 3, torch.from_numpy(gg).double(), torch.from_numpy(gb).double(), 1e-5).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r["lp"], x / np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True)), rtol=2e-5, atol=1e-6)
    xd = x.astype(np.float64)
    np.testing.assert_allclose(r["mish"], xd * np.tanh(np.log1p(np.exp(xd))), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r["celu"], np.maximum(0, xd) + np.minimum(0, 1.5 * (np.exp(xd / 1.5) - 1)), rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(r["thr"], np.where(x > 0.3, x, 0))


def _raw_tensor(name, dims, dtype, payload, field=9):
    """TensorProto with dims and payload chosen independently (for malformed-file cases)"""
    return b"".join(W._vi(1, d) for d in dims) + W._vi(2, dtype) + W._ld(field, payload) + W._s(8, name)


def test_malformed_graphs_are_errors_not_faults(gpu, tmp_path):
    """A truncated or crafted .onnx must become a clean error naming the tensor / node: payloads are checked against the declared dims
    (initialisers, Constant, ConstantOfShape), element counts against overflow, and every interpreter launch against its operand shapes
    (Conv groups / weight / bias extents, normalisation parameter lengths, Transpose perm, Concat / Split / Gather axes, Tile repeats)."""
    x = np.arange(2 * 4 * 6, dtype=np.float32).reshape(2, 4, 6)
    ok_w = np.ones((4, 4, 3), np.float32)

    def expect_error(nodes, inits, feeds, outs, needle, opset=17):
        with pytest.raises(Exception) as ei:
            run_graph(gpu, tmp_path, nodes, inits, feeds, outs, opset=opset)
        assert needle in str(ei.value), str(ei.value)

    add = [W.node("Add", ["x", "w"], ["y"])]
    yo = [("y", F32, [2, 4, 6])]
    # initialiser payloads
    expect_error(add, [_raw_tensor("w", [2, 4, 6], F32, b"\0" * (4 * 47))], {"x": x}, yo, "raw_data holds")                       # truncated raw_data
    expect_error(add, [_raw_tensor("w", [2, 4, 6], F32, b"\0" * (4 * 10), field=4)], {"x": x}, yo, "typed data holds")            # short float_data
    expect_error(add, [_raw_tensor("w", [1 << 40, 1 << 40], F32, b"\0" * 16)], {"x": x}, yo, "overflows")                          # element count overflow
    expect_error(add, [_raw_tensor("w", [6], 11, b"")], {"x": x}, yo, "raw_data holds")                                            # double without data
    expect_error(add, [_raw_tensor("w", [6], 16, b"\0" * 12)], {"x": x}, yo, "element type 16")                                    # bf16: no decoder
    # Constant / ConstantOfShape attribute tensors
    expect_error([W.node("Constant", [], ["w"], "c0", [W._s(1, "value") + W._ld(5, _raw_tensor("", [2, 4, 6], F32, b"\0" * 8)) + W._vi(20, 4)])] + add,
                 [], {"x": x}, yo, "raw_data holds")
    expect_error([W.node("ConstantOfShape", ["shp"], ["w"], "c1", [W._s(1, "value") + W._ld(5, _raw_tensor("", [1], F32, b"")) + W._vi(20, 4)])] + add,
                 [W.tensor("shp", np.array([2, 4, 6], np.int64), typed=True)], {"x": x}, yo, "holds")
    # operand shapes the kernels trust
    conv = lambda attrs, w=ok_w, b=None: ([W.node("Conv", ["x", "w"] + (["b"] if b is not None else []), ["y"], "cv", attrs)],
                                          [W.tensor("w", w)] + ([W.tensor("b", b)] if b is not None else []))
    for attrs, w, b, needle in (([W.attr_int("group", 0)], ok_w, None, "group"), ([W.attr_int("group", 3)], ok_w, None, "group"),
                                ([W.attr_ints("strides", [0])], ok_w, None, "positive"), ([], np.ones((4, 3, 3), np.float32), None, "weight shape"),
                                ([], ok_w, np.ones(5, np.float32), "bias length"), ([W.attr_ints("pads", [1, 1, 1])], ok_w, None, "spatial rank")):
        n, i = conv(attrs, w, b)
        expect_error(n, i, {"x": x}, [("y", F32, [2, 4, 4])], needle)
    expect_error([W.node("ConvTranspose", ["x", "w"], ["y"], "ct")], [W.tensor("w", np.ones((3, 2, 3), np.float32))], {"x": x}, [("y", F32, [2, 2, 8])], "weight shape")
    one4, one3 = np.ones(4, np.float32), np.ones(3, np.float32)
    expect_error([W.node("BatchNormalization", ["x", "s", "b", "m", "v"], ["y"], "bn")],
                 [W.tensor("s", one4), W.tensor("b", one4), W.tensor("m", one3), W.tensor("v", one4)], {"x": x}, yo, "BatchNormalization")
    expect_error([W.node("InstanceNormalization", ["x", "s", "b"], ["y"], "in")], [W.tensor("s", one3), W.tensor("b", one4)], {"x": x}, yo, "InstanceNormalization")
    expect_error([W.node("LayerNormalization", ["x", "s"], ["y"], "ln", [W.attr_int("axis", -1)])], [W.tensor("s", one4)], {"x": x}, yo, "LayerNormalization")
    expect_error([W.node("Transpose", ["x"], ["y"], "tp", [W.attr_ints("perm", [0, 0, 2])])], [], {"x": x}, yo, "permutation")
    expect_error([W.node("Transpose", ["x"], ["y"], "tp", [W.attr_ints("perm", [0, 1, 2, 3])])], [], {"x": x}, yo, "perm has")
    expect_error([W.node("Concat", ["x", "x"], ["y"], "cc", [W.attr_int("axis", 5)])], [], {"x": x}, yo, "axis out of range")
    expect_error([W.node("Split", ["x"], ["y", "z"], "sp", [W.attr_int("axis", 1), W.attr_ints("split", [1, 2])])], [], {"x": x}, yo, "add up")
    expect_error([W.node("Gather", ["x", "i"], ["y"], "ga", [W.attr_int("axis", 3)])], [W.tensor("i", np.array([0], np.int64), typed=True)], {"x": x}, yo, "axis out of range")
    expect_error([W.node("Tile", ["x", "r"], ["y"], "ti")], [W.tensor("r", np.array([1, 2], np.int64), typed=True)], {"x": x}, yo, "repeats")
    expect_error([W.node("Cast", ["x"], ["y"], "ca", [W.attr_int("to", 10)])], [], {"x": x}, yo, "Cast to element type 10")
    # Resize: ONNX's default half_pixel + round_prefer_floor equals the implemented sampling for whole-number up-scaling only
    sc2 = W.tensor("sc", np.array([1, 1, 2], np.float32)); sc15 = W.tensor("sc", np.array([1, 1, 1.5], np.float32))
    r = run_graph(gpu, tmp_path, [W.node("Resize", ["x", "", "sc"], ["y"], "rs")], [sc2], {"x": x}, [("y", F32, [2, 4, 12])])
    assert np.array_equal(r["y"], np.repeat(x, 2, axis=2))
    expect_error([W.node("Resize", ["x", "", "sc"], ["y"], "rs")], [sc15], {"x": x}, [("y", F32, [2, 4, 9])], "whole-number")
    r = run_graph(gpu, tmp_path, [W.node("Resize", ["x", "", "sc"], ["y"], "rs", [W.attr_str("coordinate_transformation_mode", "asymmetric"), W.attr_str("nearest_mode", "floor")])],
                  [sc15], {"x": x}, [("y", F32, [2, 4, 9])])
    assert np.array_equal(r["y"], x[:, :, (np.arange(9) / 1.5).astype(int)])
    r = run_graph(gpu, tmp_path, [W.node("Resize", ["x", "sc"], ["y"], "rs10")], [sc2], {"x": x}, [("y", F32, [2, 4, 12])], opset=10)   # opset-10 input layout
    assert np.array_equal(r["y"], np.repeat(x, 2, axis=2))


def test_decoder_state_import_rejects_bad_blobs(gpu, tiny_model):
    """q3tts_decoder_state_import validates the caller's length and the blob's trailer before committing: a blob of another size, a cached-position
    count outside the attention window, NaN or negative counters are errors -- not out-of-range device indexing on the next decode."""
    path = os.path.join(tiny_model, "onnx", "q3tts_codec.gguf")
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 2048, (8, 16))
    d = gpu.Decoder(path, n_streams=2)
    d.reset(0)
    d.decode(codes[:4], stream=0)
    blob = d.state_export(0)
    ref = d.decode(codes[4:], stream=0).copy()
    for mutate in (lambda b: b[:-1], lambda b: np.concatenate([b, [0.0]]).astype(np.float32)):
        with pytest.raises(Exception, match="floats"):
            d.state_import(mutate(blob.copy()), stream=1)
    for pos, bad in ((-2, 1e6), (-2, -1.0), (-2, float("nan")), (-2, 2.5), (-1, -4.0), (-1, float("inf")), (-1, float("nan"))):
        b = blob.copy(); b[pos] = bad
        with pytest.raises(Exception, match="state_import"):
            d.state_import(b, stream=1)
    d.reset(1)
    d.state_import(blob, stream=1)                     # the untouched blob still restores, and nothing above was committed half-way
    assert np.array_equal(d.decode(codes[4:], stream=1), ref)
    d.close()
