"""numpy restatement of the PUBLIC ggml block formats (dequantisation only) used to cross-check the oracle's C code.
Q8_0 {f16 d; i8 qs[32]}; Q5_K {f16 d,dmin; u8 scales[12]; u8 qh[32]; u8 qs[128]}; Q6_K {u8 ql[128]; u8 qh[64]; i8 sc[16]; f16 d}."""
import numpy as np


def deq_q8_0(raw, k):
    b = raw.reshape(-1, 34)
    d = b[:, :2].copy().view(np.float16).astype(np.float32)
    q = b[:, 2:].copy().view(np.int8).astype(np.float32)
    return (d * q).reshape(-1, k)


def _scale_min_k4(j, q):
    if j < 4:
        return q[j] & 63, q[j + 4] & 63
    return (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4), (q[j + 4] >> 4) | ((q[j] >> 6) << 4)


def deq_q5_k(raw, k):
    out = []
    for blk in raw.reshape(-1, 176):
        d = blk[0:2].copy().view(np.float16).astype(np.float32)[0]
        dmin = blk[2:4].copy().view(np.float16).astype(np.float32)[0]
        scales, qh, qs = blk[4:16].astype(np.int32), blk[16:48].astype(np.int32), blk[48:176].astype(np.int32)
        y = np.zeros(256, np.float32)
        is_, u1, u2, ql = 0, 1, 2, 0
        for j in range(0, 256, 64):
            sc, m = _scale_min_k4(is_, scales)
            d1, m1 = d * np.float32(sc), dmin * np.float32(m)
            sc, m = _scale_min_k4(is_ + 1, scales)
            d2, m2 = d * np.float32(sc), dmin * np.float32(m)
            lo = (qs[ql:ql + 32] & 0xF) + np.where(qh & u1, 16, 0)
            hi = (qs[ql:ql + 32] >> 4) + np.where(qh & u2, 16, 0)
            y[j:j + 32] = d1 * lo.astype(np.float32) - m1
            y[j + 32:j + 64] = d2 * hi.astype(np.float32) - m2
            ql += 32; is_ += 2; u1 <<= 2; u2 <<= 2
        out.append(y)
    return np.concatenate(out).reshape(-1, k)


def deq_q6_k(raw, k):
    out = []
    for blk in raw.reshape(-1, 210):
        ql, qh = blk[0:128].astype(np.int32), blk[128:192].astype(np.int32)
        sc = blk[192:208].copy().view(np.int8).astype(np.float32)
        d = blk[208:210].copy().view(np.float16).astype(np.float32)[0]
        y = np.zeros(256, np.float32)
        for n in range(2):
            L, H, S = ql[64 * n:], qh[32 * n:], sc[8 * n:]
            for l in range(32):
                is_ = l // 16
                q1 = ((L[l] & 0xF) | (((H[l] >> 0) & 3) << 4)) - 32
                q2 = ((L[l + 32] & 0xF) | (((H[l] >> 2) & 3) << 4)) - 32
                q3 = ((L[l] >> 4) | (((H[l] >> 4) & 3) << 4)) - 32
                q4 = ((L[l + 32] >> 4) | (((H[l] >> 6) & 3) << 4)) - 32
                y[128 * n + l] = d * S[is_] * q1
                y[128 * n + l + 32] = d * S[is_ + 2] * q2
                y[128 * n + l + 64] = d * S[is_ + 4] * q3
                y[128 * n + l + 96] = d * S[is_ + 6] * q4
        out.append(y)
    return np.concatenate(out).reshape(-1, k)


ROW_BYTES = {0: lambda k: 4 * k, 1: lambda k: 2 * k, 30: lambda k: 2 * k, 8: lambda k: k // 32 * 34, 13: lambda k: k // 256 * 176,
             14: lambda k: k // 256 * 210}


def read_gguf(path):
    """Minimal independent GGUF parser (python): returns (kv dict, {name: (type, shape ne, raw bytes)})."""
    import struct
    data = np.memmap(path, dtype=np.uint8, mode="r")
    buf = memoryview(data)
    pos = [0]

    def rd(fmt):
        v = struct.unpack_from("<" + fmt, buf, pos[0])
        pos[0] += struct.calcsize("<" + fmt)
        return v[0]

    def rstr():
        n = rd("Q")
        s = bytes(buf[pos[0]:pos[0] + n]).decode()
        pos[0] += n
        return s
    assert bytes(buf[0:4]) == b"GGUF"
    pos[0] = 4
    ver = rd("I"); nt = rd("Q"); nkv = rd("Q")
    assert ver >= 2
    scal = {0: "B", 1: "b", 2: "H", 3: "h", 4: "I", 5: "i", 6: "f", 7: "B", 10: "Q", 11: "q", 12: "d"}
    kv = {}
    for _ in range(nkv):
        key = rstr(); t = rd("I")
        if t == 8:
            kv[key] = rstr()
        elif t == 9:
            at = rd("I"); n = rd("Q")
            kv[key] = [rstr() if at == 8 else rd(scal[at]) for _ in range(n)]
        else:
            kv[key] = rd(scal[t])
    infos = []
    for _ in range(nt):
        name = rstr(); nd = rd("I")
        ne = [rd("Q") for _ in range(nd)]
        ty = rd("I"); off = rd("Q")
        infos.append((name, ty, ne, off))
    align = kv.get("general.alignment", 32)
    start = pos[0] + (align - pos[0] % align) % align
    tensors = {}
    for name, ty, ne, off in infos:
        rows = int(np.prod(ne[1:])) if len(ne) > 1 else 1
        nb = ROW_BYTES[ty](ne[0]) * rows
        tensors[name] = (ty, ne, data[start + off:start + off + nb])
    return kv, tensors


# ---------------------------------------------------------------------------------------------------------------------------------
# Encoders (test fixtures only): any VALID encoding of the public block formats will do -- the readers above define the format; these are
# plain min/max quantisers, not ggml's search -- so a fixture can hold K-quant copies of weights that came from somewhere else (transformers).
def enc_q5_k(w):
    """w [n][k] f32 (k % 256 == 0) -> raw bytes [n][k/256][176] of Q5_K super-blocks: x ~ d*sc_j*q - dmin*m_j, q in 0..31, sc/m 6 bit"""
    w = np.ascontiguousarray(w, np.float32)
    n, k = w.shape
    x = w.reshape(-1, 8, 32)                                   # [super-block][sub-block][32]
    lo, hi = np.minimum(x.min(-1), 0.0), x.max(-1)
    step = np.maximum(hi - lo, 1e-12) / 31.0                   # per sub-block step and (non-negative) offset
    off = -lo
    d = (step.max(-1) / 63.0).astype(np.float16).astype(np.float32)
    dmin = (off.max(-1) / 63.0).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(step / np.maximum(d, 1e-30)[:, None]), 1, 63).astype(np.int32)
    mq = np.clip(np.rint(off / np.maximum(dmin, 1e-30)[:, None]), 0, 63).astype(np.int32)
    q = np.clip(np.rint((x + (dmin[:, None] * mq)[..., None]) / np.maximum(d[:, None] * sc, 1e-30)[..., None]), 0, 31).astype(np.int32)
    out = np.zeros((x.shape[0], 176), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = dmin.astype(np.float16).view(np.uint8).reshape(-1, 2)
    s8 = np.zeros((x.shape[0], 12), np.int32)
    for j in range(4):
        s8[:, j] = sc[:, j] | ((sc[:, j + 4] >> 4) << 6)
        s8[:, j + 4] = mq[:, j] | ((mq[:, j + 4] >> 4) << 6)
        s8[:, j + 8] = (sc[:, j + 4] & 0xF) | ((mq[:, j + 4] & 0xF) << 4)
    out[:, 4:16] = s8.astype(np.uint8)
    qh = np.zeros((x.shape[0], 32), np.int32)
    qs = np.zeros((x.shape[0], 128), np.int32)
    for j in range(8):
        jj, hib = j >> 1, j & 1
        qs[:, 32 * jj:32 * jj + 32] |= (q[:, j] & 0xF) << (4 * hib)
        qh |= (q[:, j] >> 4) << (2 * jj + hib)
    out[:, 16:48] = qh.astype(np.uint8)
    out[:, 48:176] = qs.astype(np.uint8)
    return out.reshape(n, k // 256 * 176)


def enc_q6_k(w):
    """w [n][k] f32 -> raw bytes [n][k/256][210] of Q6_K super-blocks: x ~ d*sc_s*q, q in -32..31, sc int8 per 16 elements"""
    w = np.ascontiguousarray(w, np.float32)
    n, k = w.shape
    x = w.reshape(-1, 16, 16)                                  # [super-block][16 sub-blocks][16]
    amax = np.abs(x).max(-1)
    step = np.maximum(amax, 1e-12) / 31.0
    d = (step.max(-1) / 127.0).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(step / np.maximum(d, 1e-30)[:, None]), 1, 127).astype(np.int32)
    q = np.clip(np.rint(x / np.maximum(d[:, None] * sc, 1e-30)[..., None]), -32, 31).astype(np.int32) + 32   # stored 0..63
    q = q.reshape(-1, 256)
    out = np.zeros((q.shape[0], 210), np.uint8)
    ql = np.zeros((q.shape[0], 128), np.int32)
    qh = np.zeros((q.shape[0], 64), np.int32)
    for half in range(2):
        base = 128 * half
        for grp in range(4):                                   # elements base + 32*grp + l
            v = q[:, base + 32 * grp:base + 32 * grp + 32]
            col = 64 * half + (32 if grp & 1 else 0)
            ql[:, col:col + 32] |= (v & 0xF) << (4 if grp >= 2 else 0)
            qh[:, 32 * half:32 * half + 32] |= (v >> 4) << (2 * grp)
    out[:, 0:128] = ql.astype(np.uint8)
    out[:, 128:192] = qh.astype(np.uint8)
    out[:, 192:208] = sc.astype(np.int8).view(np.uint8)
    out[:, 208:210] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    return out.reshape(n, k // 256 * 210)
