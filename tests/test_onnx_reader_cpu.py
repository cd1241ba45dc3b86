"""the engine's ONNX reader (csrc/onnx_reader.cpp behind q3tts_onnx_*) on graphs this test writes itself with a
hand-rolled protobuf writer (tests/onnx_writer.py; the `onnx` package is not installed and the reference's .onnx files are not in the
image).  The graph mimics the streaming decoder of /root/reference/src/models/onnx.rs:355-455: same input / output names, dtypes and
state tensors, weights taken from the synthetic codec GGUF."""
import os
import subprocess
import numpy as np
import pytest

import onnx_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _decoder_like_model(tiny_model, with_contract=True):
    import ggml_ref as G
    _, t = G.read_gguf(os.path.join(tiny_model, "onnx", "q3tts_codec.gguf"))
    cb0 = np.array(t["codec.codebook.0"][2]).view(np.float32).reshape(2048, 32)
    pre_w = np.array(t["codec.pre_conv.weight"][2]).view(np.float32).reshape(64, 32, 3)
    pre_b = np.array(t["codec.pre_conv.bias"][2]).view(np.float32)
    inits = [W.tensor("codebook.0", cb0), W.tensor("pre_conv.weight", pre_w), W.tensor("pre_conv.bias", pre_b, typed=True),
             W.tensor("axes0", np.array([0, 2], np.int64), typed=True), W.tensor("alpha", np.array([0.5], np.float32))]
    nodes = [
        W.node("Gather", ["codebook.0", "audio_codes"], ["emb"], "rvq0", [W.attr_int("axis", 0)]),
        W.node("Concat", ["pre_conv_history", "emb"], ["ext"], "hist_cat", [W.attr_int("axis", 2)]),
        W.node("Conv", ["ext", "pre_conv.weight", "pre_conv.bias"], ["c1"], "pre_conv",
               [W.attr_ints("kernel_shape", [3]), W.attr_ints("dilations", [1]), W.attr_ints("strides", [1]), W.attr_ints("pads", [0, 0]), W.attr_int("group", 1)]),
        W.node("Mul", ["c1", "alpha"], ["ax"], "snake_ax"), W.node("Sin", ["ax"], ["sn"], "snake_sin"), W.node("Mul", ["sn", "sn"], ["sn2"], "snake_sq"),
        W.node("Add", ["c1", "sn2"], ["y"], "snake_add"),
        W.node("ConvTranspose", ["y", "pre_conv.weight"], ["up"], "up0", [W.attr_ints("kernel_shape", [4]), W.attr_ints("strides", [2])]),
        W.node("LeakyRelu", ["up"], ["lr"], "act", [W.attr_float("alpha", 0.1)]),
        W.node("LSTM", ["lr"], ["final_wav"], "not_supported_here"),
        W.node("Slice", ["ext", "axes0"], ["next_pre_conv_history"], "hist_out"),
    ]
    ins = [W.value_info("audio_codes", W.I64, [1, "N", 16]), W.value_info("is_last", W.F32, [1])]
    outs = [W.value_info("final_wav", W.F32, [1, "T"]), W.value_info("valid_samples", W.I64, [1])]
    if with_contract:
        ins += [W.value_info("pre_conv_history", W.F32, [1, 512, "T0"]), W.value_info("latent_buffer", W.F32, [1, 1024, "T1"]),
                W.value_info("conv_history", W.F32, [1, 1024, "T2"])]
        outs += [W.value_info("next_pre_conv_history", W.F32, [1, 512, "T0n"]), W.value_info("next_latent_buffer", W.F32, [1, 1024, "T1n"]),
                 W.value_info("next_conv_history", W.F32, [1, 1024, "T2n"])]
        for i in range(8):
            ins += [W.value_info("past_key_%d" % i, W.F32, [1, 16, "Tk", 64]), W.value_info("past_value_%d" % i, W.F32, [1, 16, "Tk", 64])]
            outs += [W.value_info("next_key_%d" % i, W.F32, [1, 16, "Tkn", 64]), W.value_info("next_value_%d" % i, W.F32, [1, 16, "Tkn", 64])]
    return W.model(nodes, inits, ins, outs, graph_name="qwen3_tts_decoder_like"), cb0, pre_w, pre_b


def test_reader_parses_nodes_attributes_initializers_and_contract(q3, tiny_model, tmp_path):
    blob, cb0, pre_w, pre_b = _decoder_like_model(tiny_model)
    path = str(tmp_path / "decoder_like.onnx")
    open(path, "wb").write(blob)
    m = q3.OnnxModel(path)
    assert (m.n_nodes, m.n_initializers) == (11, 5) and m.n_inputs == 2 + 3 + 16 and m.n_outputs == 2 + 3 + 16
    n0, n2, n7, n8 = m.node(0), m.node(2), m.node(7), m.node(8)
    assert n0["op_type"] == "Gather" and n0["inputs"] == ["codebook.0", "audio_codes"] and n0["outputs"] == ["emb"] and m.attr_ints(0, "axis") == [0]
    assert n2["op_type"] == "Conv" and n2["name"] == "pre_conv" and n2["n_attr"] == 5
    assert m.attr_ints(2, "kernel_shape") == [3] and m.attr_ints(2, "pads") == [0, 0] and m.attr_ints(2, "group") == [1] and m.attr_ints(2, "nope") is None
    assert n7["op_type"] == "ConvTranspose" and m.attr_ints(7, "strides") == [2]
    assert n8["op_type"] == "LeakyRelu" and abs(m.attr_float(8, "alpha") - 0.1) < 1e-7
    i0, i1, i2, i3 = m.initializer(0), m.initializer(1), m.initializer(2), m.initializer(3)
    assert i0["name"] == "codebook.0" and i0["dtype"] == 1 and i0["dims"] == [2048, 32] and np.array_equal(np.frombuffer(i0["raw"], np.float32), cb0.reshape(-1))
    assert i1["dims"] == [64, 32, 3] and np.array_equal(np.frombuffer(i1["raw"], np.float32), pre_w.reshape(-1))
    assert i2["name"] == "pre_conv.bias" and np.array_equal(np.frombuffer(i2["raw"], np.float32), pre_b)            # typed float_data field
    assert i3["dtype"] == 7 and np.frombuffer(i3["raw"], np.int64).tolist() == [0, 2]                                  # typed int64_data field
    ok, missing = m.decoder_contract()
    assert ok and missing == ""
    text = m.summary()
    assert "input  audio_codes i64 [1,N,16]" in text and "past_key_7 f32 [1,16,Tk,64]" in text and "streaming-decoder I/O contract (onnx.rs:355-455): satisfied" in text
    assert "LSTM x1 -> NO KERNEL YET" in text and "Conv x1 -> k_conv_gemm" in text and "nodes served by existing kernels: 9 / 11" in text
    m.close()
    assert q3.onnx_op_kernel("ConvTranspose") and q3.onnx_op_kernel("LSTM") is None and q3.onnx_op_kernel("NoSuchOp") is None
    tool = os.path.join(ROOT, "tools", "q3onnx_dump")
    if os.path.exists(tool):
        r = subprocess.run([tool, path, "--nodes"], capture_output=True, text=True)
        assert r.returncode == 0 and "Gather" in r.stdout and "hist_out" in r.stdout and "satisfied" in r.stdout


def test_reader_reports_missing_contract_and_rejects_garbage(q3, tiny_model, tmp_path):
    blob, *_ = _decoder_like_model(tiny_model, with_contract=False)
    path = str(tmp_path / "no_state.onnx")
    open(path, "wb").write(blob)
    m = q3.OnnxModel(path)
    ok, missing = m.decoder_contract()
    assert not ok and "input:pre_conv_history" in missing and "output:next_value_7" in missing
    m.close()
    for name, data in (("empty.onnx", b""), ("trunc.onnx", blob[: len(blob) // 2]), ("noise.onnx", bytes(range(256)) * 4), ("nograph.onnx", b"\x08\x08")):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        with pytest.raises(q3.Q3Error):
            q3.OnnxModel(p)


def test_executor_operator_table(q3):
    """the executor's operator table is a host-side lookup (no device needed): what an exported encoder / decoder may use, and what it may not"""
    Q = q3
    for op in ("Conv", "ConvTranspose", "MatMul", "Gemm", "LayerNormalization", "Softmax", "Gather", "Slice", "Concat", "Reshape", "Shape", "Where", "ArgMin", "Pad", "Elu", "Erf",
               "CumSum", "Range", "ConstantOfShape", "Expand", "Tile", "ReduceMean", "InstanceNormalization", "BatchNormalization", "Split", "Cast", "Clip", "Pow"):
        assert Q.onnx_op_executable(op), op
    for op in ("LSTM", "GRU", "Einsum", "TopK", "NonZero", "NotAnOp"):
        assert not Q.onnx_op_executable(op), op
