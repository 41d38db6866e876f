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
};
__device__ __forceinline__ unsigned long long eff_seed(unsigned long long seed, const unsigned long long* dev) {
    return seed + (dev ? *dev * 0x9E3779B97F4A7C15ull : 0ull);
}

__device__ __forceinline__ int acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

// ---- dropout of the attention weights -------------------------------------------------------------------------
// Element (b, h, query i, key j) has index idx = ((b*H + h)*T1 + i)*T2 + j; mask definition in oe_common.h
// (call idx >> 3, 16-bit field idx & 7).  RNG was ~half of the forward and most of the dK/dV kernel when every lane generated the numbers for its own
// registers; the lanes of a wave now share calls (T2 % 8 == 0: a block never straddles two query rows):
//   * forward / dQ (lane = query, registers = keys): the lanes lk = 0 / 1 of a query hold the two halves of each
//     8-key block - each computes two of the four blocks of a 32-key tile and swaps halves with its partner;
//   * dK/dV (lane = key, registers = 16 queries): the eight lanes of a key block need the same 16 calls (one per
//     query row) - each computes two and the fields are fetched with lane shuffles.
// Any other T2 takes the per-element path (one call per element; same mask by definition).
// forward / dQ: scales for this lane's 16 registers (keys j0 + acc_row(r, lk)) of query row `rowbase / T2`
__device__ __forceinline__ void drop_tile_qlane(unsigned long long seed, unsigned long long rowbase, int j0, int lk, bool aligned,
                                                const DropParams& d, float (&m)[16]) {
    if (aligned) {
        const unsigned long long blk = (rowbase + j0) >> 3;
        const uint4 ca = philox4(seed, blk + lk), cb = philox4(seed, blk + 2 + lk);     // blocks g = lk and g = 2 + lk
        const unsigned r0 = __shfl_xor(lk ? ca.x : ca.z, 32, 64), r1 = __shfl_xor(lk ? ca.y : ca.w, 32, 64);
        const unsigned r2 = __shfl_xor(lk ? cb.x : cb.z, 32, 64), r3 = __shfl_xor(lk ? cb.y : cb.w, 32, 64);
        unsigned w[4][2];                                                                 // [block g][word of this lane's half]
        w[0][0] = lk ? r0 : ca.x; w[0][1] = lk ? r1 : ca.y;
        w[1][0] = lk ? ca.z : r0; w[1][1] = lk ? ca.w : r1;
        w[2][0] = lk ? r2 : cb.x; w[2][1] = lk ? r3 : cb.y;
        w[3][0] = lk ? cb.z : r2; w[3][1] = lk ? cb.w : r3;
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = drop_field(w[r >> 2][(r & 3) >> 1], r & 1, d);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = drop_elem(seed, rowbase + j0 + acc_row(r, lk), d);
    }
}


// attention_bf16.hip: return 1 when the problem does not qualify (the caller then launches the kernels of attention.hip)
int oe_attn_planes_fwd_try(const AttnParams& p, int terms, hipStream_t st);
int oe_attn_planes_dq_try(const AttnParams& p, int terms, hipStream_t st);
int oe_attn_planes_dkdv_try(const AttnParams& p, int terms, hipStream_t st);
