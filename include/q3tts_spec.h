/*
 * q3tts_spec.h -- NORMATIVE arithmetic specification ("Q3TTS arithmetic spec v1")
 *
 * The reference (IuvenisSapiens/Qwen3-TTS-Rust) delegates all transformer arithmetic to llama.cpp
 * b8123, which is not in the reference tree.  Token parity "oracle <-> HIP" therefore
 * needs ONE written-down arithmetic; this header is it.  Every float reduction has a fixed order,
 * every transcendental is a fixed polynomial in IEEE fma/mul/add, so the C oracle (gcc, x86-64) and
 * the HIP kernels (gfx950) produce BIT-IDENTICAL logits and hence bit-identical codec tokens.
 *
 * The arithmetic mirrors ggml's CPU path in structure (activations quantised to int8 blocks of 32
 * with an f16 scale, exact integer block dot products, f32 accumulation of block terms), differing
 * only in the (explicitly fixed) order of the f32 accumulation.
 *
 * Compiles as C11, C++17 and HIP device code.  Both sides MUST be built with -ffp-contract=off and
 * without fast-math; fused multiply-adds appear only where q3_fmaf is written.
 *
 * Protocol constants are source-pinned by the reference: prompt.rs:5-16, engine.rs:505-518,555-558,587-596.
 */
#ifndef Q3TTS_SPEC_H
#define Q3TTS_SPEC_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define Q3_HD __host__ __device__ static inline
#else
#define Q3_HD static inline
#endif

/* ---------------- protocol constants (reference file:line in comments) ---------------- */
#define Q3_CODEC_PAD        2148   /* prompt.rs:5  */
#define Q3_CODEC_BOS        2149   /* prompt.rs:6  */
#define Q3_CODEC_EOS        2150   /* prompt.rs:7, engine.rs:558 */
#define Q3_TEXT_BOS         151672 /* prompt.rs:8  */
#define Q3_TEXT_EOS         151673 /* prompt.rs:9  */
#define Q3_CODEC_THINK      2154   /* prompt.rs:10 */
#define Q3_CODEC_NOTHINK    2155   /* prompt.rs:11 */
#define Q3_CODEC_THINK_BOS  2156   /* prompt.rs:12 */
#define Q3_CODEC_THINK_EOS  2157   /* prompt.rs:13 */
#define Q3_TEXT_AUDIO_MARKER 151671 /* prompt.rs:16, assets_manager.rs:244 */
#define Q3_CODEC_AUDIO_START 2160  /* prompt.rs:68 */
#define Q3_LANG_CHINESE     2055   /* engine.rs:267,407,425 */
#define Q3_SAMPLE_END       2160   /* engine.rs:555: talker sampling range [0,2160) */
#define Q3_N_CODEBOOKS      16     /* engine.rs:587 */
#define Q3_CODEBOOK_SIZE    2048   /* engine.rs:588-589,518 */
#define Q3_CHUNK_CODES      64     /* engine.rs:510 */
#define Q3_EMBD             2048   /* assets_manager.rs:244-247,423-425 (hard-coded row width) */
#define Q3_SAMPLE_RATE      24000  /* engine.rs:653 */
#define Q3_TALKER_NCTX      4096   /* engine.rs:133 */
#define Q3_PRED_NCTX        512    /* engine.rs:136 */
#define Q3_DEFAULT_MAX_STEPS 512   /* engine.rs:152 */

/* ---------------- arithmetic-spec structural constants ---------------- */
#define Q3_QBLK      32    /* activation / weight quantisation block (elements)            */
#define Q3_SEG_BLKS  8     /* blocks per segment: one f32 fma chain (256 elements)           */
#define Q3_SEG       256
#define Q3_SSEG_SEGS 8     /* segments per super-segment (2048 elements)                     */
#define Q3_SSEG      2048
#define Q3_HEAD_DIM  128   /* talker & predictor head dim (spec lane maps assume 128)        */
#define Q3_ATT_CHUNK 256   /* attention positions per softmax chunk                          */

/* ---------------- f16 / bf16 <-> f32, IEEE round-to-nearest-even ---------------- */
Q3_HD uint32_t q3_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
Q3_HD float q3_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

Q3_HD float q3_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            out = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        out = sign | 0x7F800000u | (man << 13);
    } else {
        out = sign | ((exp + 112u) << 23) | (man << 13);
    }
    return q3_bits_f32(out);
}

Q3_HD uint16_t q3_f32_to_f16(float f) {
    uint32_t x = q3_f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
    }
    if (ax >= 0x477FF000u) { /* >= 65520 -> inf after rounding */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25, ties to even -> 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
    /* for normal numbers half_man carries the implicit bit at 0x400: adding (hexp-1)<<10 folds it in,
       and a mantissa overflow correctly bumps the exponent */
    uint32_t out;
    if (hexp == 0) out = half_man;
    else out = ((hexp - 1u) << 10) + half_man;
    return (uint16_t)(sign | out);
}

Q3_HD float q3_bf16_to_f32(uint16_t h) { return q3_bits_f32((uint32_t)h << 16); }
Q3_HD uint16_t q3_f32_to_bf16(float f) {
    uint32_t x = q3_f32_bits(f);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x40u);
    return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
}

/* ---------------- explicit fused multiply-add ---------------- */
#if defined(__HIP_DEVICE_COMPILE__)
#define q3_fmaf(a, b, c) __builtin_fmaf((a), (b), (c))
#define q3_rintf(x) __builtin_rintf(x)
#define q3_sqrtf(x) __builtin_sqrtf(x)
#define q3_fabsf(x) __builtin_fabsf(x)
#else
#define q3_fmaf(a, b, c) fmaf((a), (b), (c))
#define q3_rintf(x) rintf(x)
#define q3_sqrtf(x) sqrtf(x)
#define q3_fabsf(x) fabsf(x)
#endif

/* ---------------- expf: Cephes-style, only fma/mul/add/rint + exponent bit add ----------------
 * |rel err| ~ 1e-7.  Returns 0 for x < -87, +inf for x > 88.72. */
Q3_HD float q3_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.72f) return q3_bits_f32(0x7F800000u);
    float n = q3_rintf(x * 1.44269504088896341f);
    float r = q3_fmaf(n, -0.693359375f, x);
    r = q3_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = q3_fmaf(p, r, 1.3981999507E-3f);
    p = q3_fmaf(p, r, 8.3334519073E-3f);
    p = q3_fmaf(p, r, 4.1665795894E-2f);
    p = q3_fmaf(p, r, 1.6666665459E-1f);
    p = q3_fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float y = q3_fmaf(p, r2, r);
    y = y + 1.0f;
    int32_t ni = (int32_t)n;
    if (ni > 127) { /* x in (88.02, 88.72]: split the scale to stay finite */
        y = y * 2.0f;
        ni -= 1;
    }
    uint32_t yb = q3_f32_bits(y) + ((uint32_t)ni << 23);
    return q3_bits_f32(yb);
}

/* silu(g)*u, SwiGLU (spec S8) */
Q3_HD float q3_swiglu(float g, float u) {
    float e = q3_expf(-g);
    float den = 1.0f + e;
    float sg = g / den;
    return sg * u;
}

/* ---------------- activation quantisation, one block of 32 (spec S2; ggml quantize_row_q8_0 shape)
 * amax exact; d = amax/127; id = 1/d; q = rint(x*id); stored scale = f16(d). */
Q3_HD uint16_t q3_quant_block32(const float* x, int8_t* q) {
    float amax = 0.0f;
    for (int i = 0; i < 32; i++) {
        float a = q3_fabsf(x[i]);
        if (a > amax) amax = a;
    }
    float d = amax / 127.0f;
    float id = (d != 0.0f) ? (1.0f / d) : 0.0f;
    for (int i = 0; i < 32; i++) q[i] = (int8_t)(int)q3_rintf(x[i] * id);
    return q3_f32_to_f16(d);
}

/* RoPE rotation of one NeoX pair (x1 = x[i], x2 = x[i+64]) (spec S5) */
Q3_HD void q3_rope_pair(float x1, float x2, float c, float s, float* o1, float* o2) {
    float t = x2 * s;
    *o1 = q3_fmaf(x1, c, -t);
    float u = x1 * s;
    *o2 = q3_fmaf(x2, c, u);
}

/* which M-RoPE position stream (0..3) rotates pair index i; sections sum may be 0 => stream 0.
 * llama.cpp-style sector assignment [EXT]. */
Q3_HD int q3_mrope_stream(int i, const int32_t sec[4]) {
    int tot = sec[0] + sec[1] + sec[2] + sec[3];
    if (tot <= 0) return 0;
    int s = i % tot;
    if (s < sec[0]) return 0;
    if (s < sec[0] + sec[1]) return 1;
    if (s < sec[0] + sec[1] + sec[2]) return 2;
    return 3;
}

/* ---------------- "ggml-CPU" arithmetic mode (opt-in, Q3_SPEC=ggml) ----------------
 * expf as glibc's generic (non-FMA) float routine computes it [EXT: sysdeps/ieee754/flt-32/e_expf.c + exp2f_data, N = 32]: a 32-entry table of
 * 2^(i/32) and a degree-3 polynomial, all in double, one rounding to float at the end.  Restated so that the oracle's ggml mode and the HIP
 * ggml-mode kernels share ONE definition (libm's expf differs between hosts: FMA builds contract differently); on the build container's glibc the
 * two agree on every one of 200 000 random arguments (table recomputed with exact integer arithmetic, tests/test_ggml_mode_cpu.py keeps checking). */
Q3_HD float q3_expf_ggml(float x) {
    static const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
        0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
        0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    if (x != x) return x;
    if (x > 88.72283172607421875f) return q3_bits_f32(0x7F800000u);
    if (x < -103.97208404541015625f) return 0.0f;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double z = InvLn2N * (double)x;
    double kd = z + Shift;
    uint64_t ki; memcpy(&ki, &kd, 8);
    kd = kd - Shift;
    const double r = z - kd;
    uint64_t t = T[ki % 32] + (ki << (52 - 5));
    double s; memcpy(&s, &t, 8);
    const double zz = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * s;
    return (float)y;
}
/* ggml-quants.c nearest_int: magic-number rounding (half to even), |v| <= 4194303 */
Q3_HD int q3_nearest_int_ggml(float fval) {
    float val = fval + 12582912.f;
    int32_t i; memcpy(&i, &val, 4);
    return (i & 0x007fffff) - 0x00400000;
}

/* ggml tensor types used by this engine (public GGUF spec [EXT]) */
enum q3_ggml_type {
    Q3_T_F32 = 0, Q3_T_F16 = 1, Q3_T_Q8_0 = 8, Q3_T_Q5_K = 13, Q3_T_Q6_K = 14, Q3_T_BF16 = 30
};

#endif /* Q3TTS_SPEC_H */
