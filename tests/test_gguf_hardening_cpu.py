"""Model files are external input: malformed GGUF headers must fail with an error through the C ABI (q3tts_assets_open runs the product's
GGUF reader on the host), never crash (alignment 0 -> SIGFPE, unchecked string lengths, partial blocks, wrapping offsets)."""
import os
import struct
import numpy as np
import pytest

from gguf_writer import write_gguf


def _s(x):
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def _header(n_tensors, kvs, tensors_meta):
    out = bytearray(b"GGUF" + struct.pack("<IQQ", 3, n_tensors, len(kvs)))
    for k, (ty, payload) in kvs.items():
        out += _s(k) + struct.pack("<I", ty) + payload
    out += tensors_meta
    return bytes(out)


def _tensor_meta(name, dims, ty, off):
    return _s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", ty, off)


@pytest.mark.parametrize("case", ["alignment_zero", "alignment_not_pow2", "huge_string", "partial_block", "zero_dim", "offset_wraps", "truncated", "not_gguf"])
def test_malformed_gguf_is_rejected_not_crashing(q3, tmp_path, case):
    p = str(tmp_path / (case + ".gguf"))
    data = b"\0" * 4096
    if case == "alignment_zero":
        blob = _header(1, {"general.alignment": (4, struct.pack("<I", 0))}, _tensor_meta("proj.weight", [32, 4], 0, 0)) + data
    elif case == "alignment_not_pow2":
        blob = _header(1, {"general.alignment": (4, struct.pack("<I", 48))}, _tensor_meta("proj.weight", [32, 4], 0, 0)) + data
    elif case == "huge_string":
        blob = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + struct.pack("<Q", 0xFFFFFFFFFFFFFFF0) + b"abc"
    elif case == "partial_block":   # Q8_0 row of 40 elements: not a multiple of the 32-element block
        blob = _header(1, {}, _tensor_meta("w", [40, 4], 8, 0)) + data
    elif case == "zero_dim":
        blob = _header(1, {}, _tensor_meta("w", [0, 4], 0, 0)) + data
    elif case == "offset_wraps":
        blob = _header(1, {}, _tensor_meta("w", [32, 4], 0, 0xFFFFFFFFFFFFFF00)) + data
    elif case == "truncated":
        good = str(tmp_path / "good.gguf")
        write_gguf(good, {"general.architecture": "x"}, {"proj.weight": np.zeros((4, 32), np.float32)})
        blob = open(good, "rb").read()[:60]
    else:
        blob = b"GGML" + b"\0" * 64
    open(p, "wb").write(blob)
    with pytest.raises(q3.Q3Error):
        q3.Assets(p)
