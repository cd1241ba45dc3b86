"""Hand-rolled ONNX (protobuf) writer for test fixtures: enough of onnx.proto3 to build ModelProto files without the `onnx` package.
Field numbers: ModelProto {ir_version 1, producer_name 2, graph 7, opset_import 8}; GraphProto {node 1, name 2, initializer 5, input 11,
output 12}; NodeProto {input 1, output 2, name 3, op_type 4, attribute 5}; AttributeProto {name 1, f 2, i 3, s 4, t 5, floats 7, ints 8,
type 20}; TensorProto {dims 1, data_type 2, float_data 4, int64_data 7, name 8, raw_data 9}; ValueInfoProto {name 1, type 2};
TypeProto {tensor_type 1 {elem_type 1, shape 2 {dim 1 {dim_value 1, dim_param 2}}}}."""
import struct
import numpy as np

F32, I64 = 1, 7


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _s(field, text):
    return _ld(field, text.encode())


def _vi(field, value):
    return _key(field, 0) + _varint(value)


def tensor(name, arr, typed=False):
    """TensorProto; typed=True stores float_data / int64_data (packed) instead of raw_data"""
    arr = np.asarray(arr)
    if arr.ndim and not arr.flags.c_contiguous:   # (np.ascontiguousarray would turn a 0-d scalar into a 1-element vector)
        arr = np.ascontiguousarray(arr)
    dt = F32 if arr.dtype == np.float32 else I64
    out = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, dt)
    if typed and dt == F32:
        out += _ld(4, arr.astype("<f4").tobytes())
    elif typed:
        out += _ld(7, b"".join(_varint(int(v)) for v in arr.reshape(-1)))
    else:
        out += _ld(9, arr.tobytes())
    return out + _s(8, name)


def attr_ints(name, vals):
    return _s(1, name) + _ld(8, b"".join(_varint(int(v)) for v in vals)) + _vi(20, 7)


def attr_int(name, v):
    return _s(1, name) + _vi(3, v) + _vi(20, 2)


def attr_float(name, v):
    return _s(1, name) + _key(2, 5) + struct.pack("<f", v) + _vi(20, 1)


def attr_str(name, text):
    return _s(1, name) + _ld(4, text.encode()) + _vi(20, 3)


def attr_tensor(name, arr):
    return _s(1, name) + _ld(5, tensor("", arr)) + _vi(20, 4)


def attr_floats(name, vals):
    return _s(1, name) + _ld(7, b"".join(struct.pack("<f", float(v)) for v in vals)) + _vi(20, 6)


def node(op_type, inputs, outputs, name="", attrs=()):
    out = b"".join(_s(1, i) for i in inputs) + b"".join(_s(2, o) for o in outputs) + _s(3, name) + _s(4, op_type)
    return out + b"".join(_ld(5, a) for a in attrs)


def value_info(name, elem_type, shape):
    dims = b""
    for d in shape:
        dims += _ld(1, _s(2, d) if isinstance(d, str) else _vi(1, d))
    return _s(1, name) + _ld(2, _ld(1, _vi(1, elem_type) + _ld(2, dims)))


def model(nodes, initializers, inputs, outputs, graph_name="g", producer="q3tts-test", opset=17, ir_version=8):
    g = b"".join(_ld(1, n) for n in nodes) + _s(2, graph_name) + b"".join(_ld(5, t) for t in initializers)
    g += b"".join(_ld(11, v) for v in inputs) + b"".join(_ld(12, v) for v in outputs)
    return _vi(1, ir_version) + _s(2, producer) + _ld(7, g) + _ld(8, _s(1, "") + _vi(2, opset))
