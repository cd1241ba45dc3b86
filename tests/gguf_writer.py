"""Tiny GGUF v3 writer (python) for test fixtures: F32 / F16 tensors + u32/f32/str/i32-array metadata."""
import struct
import numpy as np


def write_gguf(path, kv, tensors):
    """kv: {key: int|float|str|list[int]}; tensors: {name: np.float32 or np.float16 array} (row-major; ne = reversed shape)."""
    def s(x):
        b = x.encode()
        return struct.pack("<Q", len(b)) + b
    out = bytearray(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(kv)))
    for k, v in kv.items():
        out += s(k)
        if isinstance(v, str):
            out += struct.pack("<I", 8) + s(v)
        elif isinstance(v, float):
            out += struct.pack("<If", 6, v)
        elif isinstance(v, list):
            out += struct.pack("<IIQ", 9, 5, len(v)) + b"".join(struct.pack("<i", x) for x in v)
        else:
            out += struct.pack("<II", 4, int(v))
    off = 0
    blobs = []
    for name, a in tensors.items():
        if isinstance(a, tuple):   # (ggml type id, logical shape, raw bytes): pre-encoded block formats (Q8_0 ...)
            ty, shape, raw = a
            out += s(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in reversed(shape))
            out += struct.pack("<IQ", ty, off)
            blobs.append(bytes(raw))
            off += (len(blobs[-1]) + 31) // 32 * 32
            continue
        f16 = getattr(a, "dtype", None) == np.float16
        a = np.ascontiguousarray(a, np.float16 if f16 else np.float32)
        out += s(name) + struct.pack("<I", a.ndim) + b"".join(struct.pack("<Q", d) for d in reversed(a.shape))
        out += struct.pack("<IQ", 1 if f16 else 0, off)
        blobs.append(a.tobytes())
        off += (len(blobs[-1]) + 31) // 32 * 32
    out += b"\0" * ((32 - len(out) % 32) % 32)
    with open(path, "wb") as f:
        f.write(out)
        for b in blobs:
            f.write(b + b"\0" * ((32 - len(b) % 32) % 32))
