// Shared device/host helpers for the gfx950 kernels (internal; the public
// C ABI is include/openeat_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#define OE_WAVE 64

// ---- error plumbing -------------------------------------------------------
extern "C" void oe_set_error(const char* fmt, ...);

#define OE_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            oe_set_error(__VA_ARGS__);                          \
            return -1;                                          \
        }                                                       \
    } while (0)

#define OE_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            oe_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return (int)e_;                                                          \
        }                                                                            \
    } while (0)

static inline int oe_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave-level reductions (64 lanes) --------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// (max, sum-of-exp) pair merge for online log-sum-exp
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { s = 0.f; m = mn; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}
__device__ __forceinline__ void wave_lse(float& m, float& s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float m2 = __shfl_xor(m, o, 64);
        float s2 = __shfl_xor(s, o, 64);
        lse_merge(m, s, m2, s2);
    }
}

// ---- counter-based RNG (Philox-4x32-10): feature-dither noise (augment.hip) ----
// One call yields 4 uniform 32-bit words for (seed, counter).
__device__ __forceinline__ uint4 philox4(uint64_t seed, uint64_t ctr) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// Dropout masks.  Element idx of the tensor a mask applies to takes the 16-bit field (idx & 3) of the 64-bit hash of
// (seed, idx >> 2); it is kept iff field >= thr with thr = round(p * 65536), and kept values are scaled by
// 65536 / (65536 - thr) - the exact inverse of the realised keep probability (p = 0.1 -> 0.100006).  The hash is the
// final() mix of Bob Jenkins' lookup3 (public domain): add / xor / rotate only, 21 full-rate instructions for four
// elements.  (The first version drew Philox-4x32-10: 40 quarter-rate 32-bit multiplies per call - in-kernel stamps put
// the generator at 20 % of the attention forward and ~5 us of every GEMM with a dropout epilogue.  Philox stays for the
// feature-dither noise, augment.hip.)  Masks are regenerated in backward from the same (seed, element index), never
// stored.  drop_words8(seed, i) packs the two hashes of the aligned block of eight elements 8i..8i+7 as
// {fields 0,1 | 2,3 | 4,5 | 6,7} for consumers that share an eight-element block between neighbouring lanes.
__device__ __forceinline__ unsigned oe_rotl32(unsigned x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ uint2 drop_hash4(uint64_t seed, uint64_t quad) {
    unsigned a = (unsigned)quad ^ (unsigned)seed;
    unsigned b = (unsigned)(quad >> 32) ^ (unsigned)(seed >> 32);
    unsigned c = 0xdeadbeefu + (unsigned)seed;
    c ^= b; c -= oe_rotl32(b, 14);
    a ^= c; a -= oe_rotl32(c, 11);
    b ^= a; b -= oe_rotl32(a, 25);
    c ^= b; c -= oe_rotl32(b, 16);
    a ^= c; a -= oe_rotl32(c, 4);
    b ^= a; b -= oe_rotl32(a, 14);
    c ^= b; c -= oe_rotl32(b, 24);
    return make_uint2(b, c);                     // fields 0, 1 = low / high half of .x; fields 2, 3 = of .y
}
__device__ __forceinline__ uint4 drop_words8(uint64_t seed, uint64_t idx8) {
    const uint2 h0 = drop_hash4(seed, 2 * idx8), h1 = drop_hash4(seed, 2 * idx8 + 1);
    return make_uint4(h0.x, h0.y, h1.x, h1.y);
}
struct DropParams { unsigned thr; float inv_keep; };
__device__ __forceinline__ DropParams drop_params(float p) {
    DropParams d;
    d.thr = (unsigned)(p * 65536.f + 0.5f);
    d.inv_keep = 65536.f / (65536.f - (float)d.thr);
    return d;
}
__device__ __forceinline__ float drop_field(unsigned w, int half, const DropParams& d) {
    return ((w >> (16 * half)) & 0xFFFFu) >= d.thr ? d.inv_keep : 0.f;
}
// one element on its own (ragged edges / unaligned rows): a whole hash for one field
__device__ __forceinline__ float drop_elem(unsigned long long seed, unsigned long long idx, const DropParams& d) {
    const uint2 h = drop_hash4(seed, idx >> 2);
    return drop_field((idx & 2) ? h.y : h.x, (int)(idx & 1), d);
}
// the eight elements of the aligned block idx8
__device__ __forceinline__ void drop_block8(unsigned long long seed, unsigned long long idx8, const DropParams& d, float (&m)[8]) {
    const uint4 r = drop_words8(seed, idx8);
    m[0] = drop_field(r.x, 0, d); m[1] = drop_field(r.x, 1, d); m[2] = drop_field(r.y, 0, d); m[3] = drop_field(r.y, 1, d);
    m[4] = drop_field(r.z, 0, d); m[5] = drop_field(r.z, 1, d); m[6] = drop_field(r.w, 0, d); m[7] = drop_field(r.w, 1, d);
}

// ---- fp32 -> bf16 pieces for the split matrix products ---------------------------------------------------------
// A product of two fp32 values on the bf16 matrix cores is a sum of TERMS bf16 x bf16 products (each exact in the fp32
// accumulator) of round-to-nearest pieces p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1):
//   TERMS 1: a0 b0                                             (plain bf16 inputs, ~2^-9 per product)
//   TERMS 3: a0 b0 + a0 b1 + a1 b0                             (two pieces, ~2^-17 per product)
//   TERMS 6: a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0     (three pieces: x = p0 + p1 + p2 EXACTLY - 8 + 8 + 8
//            significant bits cover the 24 of an fp32 - and what is dropped, a1 b2 + a2 b1 + a2 b2, is below
//            2^-24 |a||b|, one fp32 rounding of the product: the reference's fp32 arithmetic at 6/16 of the fp32-MFMA cost)
template <int TERMS> struct oe_npl { static constexpr int N = (TERMS == 6) ? 3 : (TERMS == 3) ? 2 : 1; };
template <int NPL>
__device__ __forceinline__ void oe_split_bf16(float x, __bf16 (&p)[NPL]) {
    p[0] = (__bf16)x;
    if constexpr (NPL > 1) {
        const float r1 = x - (float)p[0];            // exact
        p[1] = (__bf16)r1;
        if constexpr (NPL > 2) p[2] = (__bf16)(r1 - (float)p[1]);   // exact, and representable: at most 8 significant bits are left
    }
}
// Eight values -> their NPL pieces as MFMA fragments (bf16x8 each), two values at a time on the packed conversions
// (v_cvt_pk_bf16_f32, v_pk_add_f32): the same round-to-nearest pieces as oe_split_bf16, bit for bit, without the per-element
// inserts the scalar form leaves the compiler to assemble into registers (ring kernel: 11.7 vector instructions per MFMA).
typedef float oe_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 oe_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 oe_bf16x8 __attribute__((ext_vector_type(8)));
template <int NPL>
__device__ __forceinline__ void oe_split8(const float (&x)[8], oe_bf16x8 (&p)[NPL]) {
    oe_bf16x2 a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oe_f32x2 v;
        v[0] = x[2 * i]; v[1] = x[2 * i + 1];
        a[i] = __builtin_convertvector(v, oe_bf16x2);
        if constexpr (NPL > 1) {
            const oe_f32x2 r1 = v - __builtin_convertvector(a[i], oe_f32x2);           // exact
            b[i] = __builtin_convertvector(r1, oe_bf16x2);
            if constexpr (NPL > 2) c[i] = __builtin_convertvector(r1 - __builtin_convertvector(b[i], oe_f32x2), oe_bf16x2);
        }
    }
    auto join = [](const oe_bf16x2 (&q)[4]) {
        const auto lo = __builtin_shufflevector(q[0], q[1], 0, 1, 2, 3), hi = __builtin_shufflevector(q[2], q[3], 0, 1, 2, 3);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    p[0] = join(a);
    if constexpr (NPL > 1) p[1] = join(b);
    if constexpr (NPL > 2) p[2] = join(c);
}
// ... and four values -> bf16x4 per piece (staging stores of the kernels that keep bf16 planes in LDS / HBM)
typedef __bf16 oe_bf16x4v __attribute__((ext_vector_type(4)));
template <int NPL>
__device__ __forceinline__ void oe_split4(const float (&x)[4], oe_bf16x4v (&p)[NPL]) {
    oe_bf16x2 a[2], b[2], c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        oe_f32x2 v;
        v[0] = x[2 * i]; v[1] = x[2 * i + 1];
        a[i] = __builtin_convertvector(v, oe_bf16x2);
        if constexpr (NPL > 1) {
            const oe_f32x2 r1 = v - __builtin_convertvector(a[i], oe_f32x2);
            b[i] = __builtin_convertvector(r1, oe_bf16x2);
            if constexpr (NPL > 2) c[i] = __builtin_convertvector(r1 - __builtin_convertvector(b[i], oe_f32x2), oe_bf16x2);
        }
    }
    p[0] = __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3);
    if constexpr (NPL > 1) p[1] = __builtin_shufflevector(b[0], b[1], 0, 1, 2, 3);
    if constexpr (NPL > 2) p[2] = __builtin_shufflevector(c[0], c[1], 0, 1, 2, 3);
}
// the TERMS products of one fragment pair, smallest first; F holds NPL fragments f.p[0..NPL)
template <int TERMS, class F, class ACC>
__device__ __forceinline__ ACC oe_mma_terms(const F& a, const F& b, ACC c) {
    if constexpr (TERMS == 6) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    }
    if constexpr (TERMS >= 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
}

typedef __bf16 oe_bf16x4 __attribute__((ext_vector_type(4)));
// four consecutive values -> their three bf16 pieces, one 8-byte store per plane (planes `pstride` elements apart)
__device__ __forceinline__ void store_planes4(__bf16* dst, long pstride, const float4& v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    oe_bf16x4 pl[3];
    oe_split4<3>(x, pl);
#pragma unroll
    for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4*>(dst + n * pstride) = pl[n];
}
__device__ __forceinline__ void store_planes1(__bf16* dst, long pstride, float v) {
    __bf16 q[3];
    oe_split_bf16<3>(v, q);
    dst[0] = q[0]; dst[pstride] = q[1]; dst[2 * pstride] = q[2];
}

// ---- activations ------------------------------------------------------------
// ids follow the reference's table (utils/common.py:160-173): relu, swish, tanh, hardtanh, selu, gelu (erf form)
enum { OE_ACT_NONE = 0, OE_ACT_RELU = 1, OE_ACT_SWISH = 2, OE_ACT_TANH = 3, OE_ACT_HARDTANH = 4, OE_ACT_SELU = 5, OE_ACT_GELU = 6 };
#define OE_SELU_ALPHA 1.6732632423543772848170429916717f
#define OE_SELU_SCALE 1.0507009873554804934193349852946f

// 1 / (1 + e^-x) on the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (~10 instructions: scale, Newton steps,
// fix-up): the sigmoid runs on every element of the feed-forward / GLU / swish epilogues with the matrix pipe idle (tools/epi_cost.py)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float act_fwd(int act, float x) {
    if (act == OE_ACT_RELU) return fmaxf(x, 0.f);
    if (act == OE_ACT_SWISH) return x * sigmoidf_(x);
    if (act == OE_ACT_TANH) return tanhf(x);
    if (act == OE_ACT_HARDTANH) return fminf(fmaxf(x, -1.f), 1.f);
    if (act == OE_ACT_SELU) return OE_SELU_SCALE * (x > 0.f ? x : OE_SELU_ALPHA * (expf(x) - 1.f));
    if (act == OE_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    return x;
}
__device__ __forceinline__ float act_bwd(int act, float x) {  // d act / dx at pre-activation x
    if (act == OE_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (act == OE_ACT_SWISH) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }
    if (act == OE_ACT_TANH) { float t = tanhf(x); return 1.f - t * t; }
    if (act == OE_ACT_HARDTANH) return (x > -1.f && x < 1.f) ? 1.f : 0.f;
    if (act == OE_ACT_SELU) return OE_SELU_SCALE * (x > 0.f ? 1.f : OE_SELU_ALPHA * expf(x));
    if (act == OE_ACT_GELU) return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
    return 1.f;
}
