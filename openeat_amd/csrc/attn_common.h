// Shared by the attention kernels (attention.hip: fp32-MFMA kernels and the small-problem bf16 kernels;
// attention_bf16.hip: the LDS-plane bf16 kernels): launch parameters, dropout indexing, the accumulator row map.
#pragma once
#include "oe_common.h"
#include "../../include/openeat_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NEG_INF (-INFINITY)

struct AttnParams {
    const float* q; long q_bs, q_rs;     // batch stride, row (time) stride; head h at +h*D
    const float* k; long k_bs, k_rs;
    const float* v; long v_bs, v_rs;
    float* o; long o_bs, o_rs;
    const float* d_o;                    // same strides as o
    const float* o_in;
    float* dq; float* dk; float* dv;     // same strides as q / k / v
    float* lse;                          // (B,H,T1)
    float* delta;                        // (B,H,T1)
    const unsigned char* mask; long m_bs, m_rs;   // (B, 1|T1, T2) bytes; m_rs = 0 for a key-only mask
    const float* keybias;                // (B,H,T2) or null (already divided by sqrt(dk))
    float* dkeybias;                     // (B,H,T2) or null
    int B, H, T1, T2, D;
    float scale;
    float drop_p; unsigned long long seed; const unsigned long long* seed_dev;
    int causal;                          // hint: mask[b, i, j] = 0 for every j > i, so key chunks past a block's last query can be skipped
};
__device__ __forceinline__ unsigned long long eff_seed(unsigned long long seed, const unsigned long long* dev) {
    return seed + (dev ? *dev * 0x9E3779B97F4A7C15ull : 0ull);
}

__device__ __forceinline__ int acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

// max / sum of a value with its partner lane (lane ^ 32): one v_permlane32_swap (VALU rate) instead of an LDS round trip
// (ds_bpermute).  After the swap of (x, x): first result = {own | lower half's}, second = {upper half's | own}.
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- dropout of the attention weights -------------------------------------------------------------------------
// Mask definition (shared by the three kernels of attention.hip and the three of attention_bf16.hip, so that a forward of
// one family and a backward of the other agree): element (row, key j), row = (b*H + h)*T1 + i, takes the 16-bit field
// (j & 3) of the 64-bit hash of (seed, row, j >> 2); it is kept iff field >= round(p * 65536) and scaled by the exact
// inverse of the realised keep probability (DropParams, oe_common.h).  The hash is the final() mix of Bob Jenkins'
// lookup3 (public domain; add / xor / rotate only, 21 full-rate instructions for four elements - the Philox of the GEMM
// epilogues costs 40 quarter-rate 32-bit multiplies for eight).  One hash serves the four consecutive keys a lane of the
// forward / dQ kernels holds per accumulator group; in dK/dV (lane = key, registers = queries) the four lanes of a key
// quad compute four of the sixteen query rows each and pass the words round with DPP quad broadcasts.
__device__ __forceinline__ uint2 attn_drop_hash(unsigned long long seed, unsigned long long row, unsigned quad) {
    return drop_hash4(seed ^ ((unsigned long long)quad << 32), row);          // oe_common.h: the lookup3 final() mix
}
__device__ __forceinline__ float attn_drop_scale(const uint2& h, int field, const DropParams& d) {
    const unsigned w = (field & 2) ? h.y : h.x;
    return drop_field(w, field & 1, d);
}
// one element on its own (reference form of the definition; ragged / unaligned uses)
__device__ __forceinline__ float attn_drop_elem(unsigned long long seed, unsigned long long row, int j, const DropParams& d) {
    return attn_drop_scale(attn_drop_hash(seed, row, (unsigned)(j >> 2)), j & 3, d);
}
// forward / dQ: lane = query `row`, registers r = keys j0 + acc_row(r, lk) (j0 a multiple of 32): four hashes
__device__ __forceinline__ void attn_drop_qlane(unsigned long long seed, unsigned long long row, int j0, int lk, const DropParams& d,
                                                float (&m)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 h = attn_drop_hash(seed, row, (unsigned)((j0 + 8 * g + 4 * lk) >> 2));
        m[4 * g + 0] = drop_field(h.x, 0, d); m[4 * g + 1] = drop_field(h.x, 1, d);
        m[4 * g + 2] = drop_field(h.y, 0, d); m[4 * g + 3] = drop_field(h.y, 1, d);
    }
}
// forward only: zero the dropped weights, leave the kept ones unscaled (the 1 / keep-probability factor is constant: the
// forward applies it once to the finished output row)
__device__ __forceinline__ void attn_drop_qlane_keep(unsigned long long seed, unsigned long long row, int j0, int lk, const DropParams& d,
                                                     float (&pr)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 h = attn_drop_hash(seed, row, (unsigned)((j0 + 8 * g + 4 * lk) >> 2));
        pr[4 * g + 0] = (h.x & 0xFFFFu) >= d.thr ? pr[4 * g + 0] : 0.f; pr[4 * g + 1] = (h.x >> 16) >= d.thr ? pr[4 * g + 1] : 0.f;
        pr[4 * g + 2] = (h.y & 0xFFFFu) >= d.thr ? pr[4 * g + 2] : 0.f; pr[4 * g + 3] = (h.y >> 16) >= d.thr ? pr[4 * g + 3] : 0.f;
    }
}
template <int K> __device__ __forceinline__ unsigned quad_bcast(unsigned x) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)x, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
// dK/dV: lane = key kj (lanes 4m..4m+3 hold the keys of one quad: the key tile starts at a multiple of 32), registers
// r = queries i0 + acc_row(r, lk) (i0 a multiple of 32).  rowbase = (b*H + h)*T1.  Lane c of a quad hashes the rows with
// (r & 3) == c; the words travel by quad broadcast.  Rows past T1 / keys past T2 get some scale: their weights are zero.
__device__ __forceinline__ void attn_drop_klane(unsigned long long seed, unsigned long long rowbase, int i0, int kj, int lk,
                                                const DropParams& d, float (&m)[16]) {
    const int c = kj & 3;
    const unsigned quad = (unsigned)(kj >> 2);
    uint2 own[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) own[g] = attn_drop_hash(seed, rowbase + (unsigned long long)(i0 + 8 * g + 4 * lk + c), quad);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // register 4g + K is row i0 + 8g + 4lk + K, hashed by lane K of the quad; this lane's field sits in word (c >> 1)
        const unsigned x0 = quad_bcast<0>(own[g].x), y0 = quad_bcast<0>(own[g].y), x1 = quad_bcast<1>(own[g].x), y1 = quad_bcast<1>(own[g].y);
        const unsigned x2 = quad_bcast<2>(own[g].x), y2 = quad_bcast<2>(own[g].y), x3 = quad_bcast<3>(own[g].x), y3 = quad_bcast<3>(own[g].y);
        m[4 * g + 0] = drop_field((c & 2) ? y0 : x0, c & 1, d); m[4 * g + 1] = drop_field((c & 2) ? y1 : x1, c & 1, d);
        m[4 * g + 2] = drop_field((c & 2) ? y2 : x2, c & 1, d); m[4 * g + 3] = drop_field((c & 2) ? y3 : x3, c & 1, d);
    }
}

// attention_bf16.hip: return 1 when the problem does not qualify (the caller then launches the kernels of attention.hip)
int oe_attn_planes_fwd_try(const AttnParams& p, int terms, hipStream_t st);
int oe_attn_planes_dq_try(const AttnParams& p, int terms, hipStream_t st);
int oe_attn_planes_dkdv_try(const AttnParams& p, int terms, hipStream_t st);
