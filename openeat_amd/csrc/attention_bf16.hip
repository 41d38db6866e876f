// Fused multi-head attention on the bf16 matrix cores with the streamed operand kept in LDS as bf16 PLANES
// (attention.py:65-97,112-117,189-209 and their autograd; precision 1 = hi plane only, precision 3 = hi + lo planes and
// the three-term product hi*hi + hi*lo + lo*hi, as oe_gemm_args.precision).
//
// What changed against the first generation of bf16 kernels (attention.hip, still used for small problems and for the
// exact-fp32 mode): those kept fp32 tiles in LDS and split every fragment to bf16 on EVERY use (each of the waves that
// consumed a tile redid the conversion), staged 32-key tiles behind two barriers each, and re-staged the whole key range
// once per 64 queries.  Here
//   * a block is 4 waves = 4 x 32 rows of the resident axis (queries for forward / dQ, keys for dK/dV): the streamed
//     tensors are staged once per 128 resident rows;
//   * the streamed axis moves in chunks of 64 rows: global fp32 -> registers (issued before the previous chunk's
//     MFMAs) -> bf16 hi / lo written ONCE into a double-buffered LDS image, one barrier per chunk;
//   * row fragments (products over features) are one ds_read_b128 per plane; column fragments (products over the 32
//     keys / queries of a tile, whose k-slots follow the accumulator row map so that the score tile never leaves
//     registers) are two ds_read_b64_tr_b16 per plane on the SAME image - gfx950's transposing LDS read - instead of
//     eight scalar reads and a conversion;
//   * outputs leave through a wave-private LDS patch as whole rows (256 contiguous bytes per store instruction)
//     instead of one 4-byte store per lane and row.
// Image: [64 rows][DPAD + 8] bf16, i.e. a 144-byte pitch at DPAD = 64: conflict-free for the b128 row reads (16 rows
// land on 16 different 16-byte slots of the 256-byte bank row), 2-way on the transposed reads (they are a few per cent
// of the LDS traffic).  LDS per block at DPAD = 64, precision 3: 2 buffers x 2 tensors x 2 planes x 9216 B = 72 KiB.
//
// Index conventions, dropout mask definition, lse / delta exchange between the kernels: attn_common.h / attention.hip.
#include <stdlib.h>
#include "attn_common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

#define PL_ROWS 64                     // rows of the streamed axis per chunk
#define PL_THREADS 256

template <int TERMS> struct PFrag { bf16x8 hi, lo; };

template <int TERMS>
__device__ __forceinline__ f32x16 pmma(const PFrag<TERMS>& a, const PFrag<TERMS>& b, f32x16 c) {
    if (TERMS == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}
template <int TERMS>
__device__ __forceinline__ void psplit(const float (&x)[8], PFrag<TERMS>& f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f.hi[e] = (__bf16)x[e];
        if (TERMS == 3) f.lo[e] = (__bf16)(x[e] - (float)f.hi[e]);
    }
}

// One streamed tensor's image in LDS: `planes` consecutive [PL_ROWS][PITCH] bf16 arrays (hi, then lo).
template <int DPAD, int TERMS>
struct Plane {
    static constexpr int PITCH = DPAD + 8;
    static constexpr int PLANE_ELEMS = PL_ROWS * PITCH;
    static constexpr int ELEMS = PLANE_ELEMS * (TERMS == 3 ? 2 : 1);
    // row fragment: A[row][16 s + 8 g + e], e = 0..7
    static __device__ __forceinline__ void row_frag(const __bf16* img, int row, int s, int g, PFrag<TERMS>& f) {
        const __bf16* p = img + row * PITCH + 16 * s + 8 * g;
        f.hi = *reinterpret_cast<const bf16x8*>(p);
        if (TERMS == 3) f.lo = *reinterpret_cast<const bf16x8*>(p + PLANE_ELEMS);
    }
    // column fragment: A[row0 + acc_row(8 s + e, g)][col], e = 0..7 - two transposing reads of 4 rows x 16 columns per
    // 16-lane group: lane 4q + pp of a group supplies the address of row q, columns 4pp..4pp+3 of the group's block and
    // receives column (lane & 15) of the four rows.  Groups 0/1 (g = 0) and 2/3 (g = 1) take their own row blocks.
    static __device__ __forceinline__ void col_frag(const __bf16* img, int row0, int col32, int s, int lane, PFrag<TERMS>& f) {
        const int i = lane & 15, grp = lane >> 4;
        const int g = grp >> 1;
        const __bf16* p = img + (row0 + 16 * s + 4 * g + (i >> 2)) * PITCH + col32 + 16 * (grp & 1) + 4 * (i & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * PITCH));
        union { s16x4 h[2]; bf16x8 v; } u;
        u.h[0] = a0; u.h[1] = a1;
        f.hi = u.v;
        if (TERMS == 3) {
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + PLANE_ELEMS));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + PLANE_ELEMS + 8 * PITCH));
            u.h[0] = b0; u.h[1] = b1;
            f.lo = u.v;
        }
    }
};

// A chunk of PL_ROWS rows x DPAD features of an fp32 tensor on its way global -> registers -> LDS planes.
template <int DPAD>
struct ChunkRegs {
    static constexpr int N = PL_ROWS * (DPAD / 4) / PL_THREADS;       // float4 per thread: 4 (DPAD 64) or 2 (DPAD 32)
    float4 v[N];
};
template <int DPAD>
__device__ __forceinline__ void chunk_load(ChunkRegs<DPAD>& t, const float* src, long rs, int r0, int nrows_total, int D, bool vec) {
#pragma unroll
    for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
        const int e = threadIdx.x + i * PL_THREADS;
        const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gr = r0 + row;
        if (gr < nrows_total && c4 < D) {
            const float* p = src + (long)gr * rs + c4;
            if (vec) val = *reinterpret_cast<const float4*>(p);
            else { val.x = p[0]; if (c4 + 1 < D) val.y = p[1]; if (c4 + 2 < D) val.z = p[2]; if (c4 + 3 < D) val.w = p[3]; }
        }
        t.v[i] = val;
    }
}
template <int DPAD, int TERMS>
__device__ __forceinline__ void chunk_store(const ChunkRegs<DPAD>& t, __bf16* img) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    using P = Plane<DPAD, TERMS>;
#pragma unroll
    for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
        const int e = threadIdx.x + i * PL_THREADS;
        const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
        const float x[4] = {t.v[i].x, t.v[i].y, t.v[i].z, t.v[i].w};
        bf16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hi[k] = (__bf16)x[k];
            if (TERMS == 3) lo[k] = (__bf16)(x[k] - (float)hi[k]);
        }
        __bf16* d = img + row * P::PITCH + c4;
        *reinterpret_cast<bf16x4*>(d) = hi;
        if (TERMS == 3) *reinterpret_cast<bf16x4*>(d + P::PLANE_ELEMS) = lo;
    }
}

// eight consecutive features of one row -> registers (two float4 when aligned); zeros past D or when !ok
__device__ __forceinline__ void load8f(const float* row, int d0, int D, bool vec, bool ok, float (&out)[8]) {
    if (vec && ok && d0 + 8 <= D) {
        const float4 a = *reinterpret_cast<const float4*>(row + d0), c = *reinterpret_cast<const float4*>(row + d0 + 4);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = c.x; out[5] = c.y; out[6] = c.z; out[7] = c.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = (ok && d0 + e < D) ? row[d0 + e] : 0.f;
    }
}

// An accumulator set acc[DT] holding X^T (rows = features t*32 + acc_row(r, lk), column = this lane's row lq of the
// resident axis) leaves as rows of X: through a wave-private [32][DPAD + 1] fp32 patch, then one 4-byte store per lane
// with the 64 lanes on consecutive features of one row (256 contiguous bytes per instruction at DPAD 64).
template <int DPAD>
__device__ __forceinline__ void store_rows(float* patch, const f32x16 (&acc)[DPAD / 32], float mul, float* dst, long rs, int row0,
                                           int nrows_total, int D, int lane) {
    constexpr int PP = DPAD + 1;
    const int lq = lane & 31, lk = lane >> 5;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < DPAD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[lq * PP + t * 32 + acc_row(r, lk)] = acc[t][r] * mul;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (DPAD == 64) {
        for (int rr = 0; rr < 32; ++rr) {
            const int gr = row0 + rr;
            if (gr < nrows_total && lane < D) dst[(long)gr * rs + lane] = patch[rr * PP + lane];
        }
    } else {
        for (int rr = 0; rr < 32; rr += 2) {
            const int gr = row0 + rr + lk;
            if (gr < nrows_total && lq < D) dst[(long)gr * rs + lq] = patch[(rr + lk) * PP + lq];
        }
    }
}

// ------------------------------------------------------------------ forward (MODE 0) and dQ (MODE 1) --
// wave w of block x owns queries (4x + w) * 32 .. + 31 of (b, h); keys stream through the planes.
template <int DPAD, int TERMS, int MODE>
__global__ __launch_bounds__(PL_THREADS) void attn_planes_q_kernel(AttnParams p) {
    using P = Plane<DPAD, TERMS>;
    constexpr int DT = DPAD / 32, KS = DPAD / 16;
    constexpr int IMG = P::ELEMS;                                   // bf16 elements of one tensor's image
    constexpr int PATCH_BYTES = 4 * 32 * (DPAD + 1) * 4;
    constexpr int IMG_BYTES = 2 * 2 * IMG * 2;                      // [buffer][K | V]
    __shared__ __attribute__((aligned(16))) char lds_raw[(IMG_BYTES > PATCH_BYTES ? IMG_BYTES : PATCH_BYTES)];
    __shared__ float kb_s[2][PL_ROWS];                              // per-key bias of the chunk; -inf = key masked for every query
    __bf16* imgs = reinterpret_cast<__bf16*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bool wave_live = q0 < p.T1;                               // wave-uniform
    const int qi = q0 + lq;
    const bool q_ok = qi < p.T1;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* kbp = p.k + (long)b * p.k_bs + h * p.D;
    const float* vbp = p.v + (long)b * p.v_bs + h * p.D;
    const long bh = (long)b * p.H + h;
    const bool vk = (p.D % 4 == 0) && (p.k_rs % 4 == 0) && ((((uintptr_t)kbp) & 15) == 0);
    const bool vv = (p.D % 4 == 0) && (p.v_rs % 4 == 0) && ((((uintptr_t)vbp) & 15) == 0);

    // Q^T fragments (B operand of S^T = K Q^T), pre-scaled; dQ also dO^T fragments, delta and lse of this lane's query
    PFrag<TERMS> qfr[KS], dofr[MODE == 1 ? KS : 1];
    float dpart = 0.f;
    const long orow = (long)b * p.o_bs + (long)(q_ok ? qi : 0) * p.o_rs + h * p.D;
    {
        const bool vq = (p.D % 8 == 0) && (p.q_rs % 4 == 0) && ((((uintptr_t)qb) & 15) == 0);
        const bool vo = (p.D % 8 == 0) && (p.o_rs % 4 == 0) && (p.o_bs % 4 == 0) &&
                        (MODE != 1 || (((((uintptr_t)p.d_o) | ((uintptr_t)p.o_in)) & 15) == 0));
        const float* qrow = qb + (long)(q_ok ? qi : 0) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d0 = 16 * s + 8 * lk;
            float x[8];
            load8f(qrow, d0, p.D, vq, q_ok, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= p.scale;
            psplit<TERMS>(x, qfr[s]);
            if (MODE == 1) {
                float y[8], ov[8];
                load8f(p.d_o + orow, d0, p.D, vo, q_ok, y);
                load8f(p.o_in + orow, d0, p.D, vo, q_ok, ov);
#pragma unroll
                for (int e = 0; e < 8; ++e) dpart += y[e] * ov[e];
                psplit<TERMS>(y, dofr[s]);
            }
        }
    }
    float m_run = NEG_INF, l_run = 0.f, lse_i = 0.f, delta_i = 0.f;
    if (MODE == 1) {
        delta_i = dpart + __shfl_xor(dpart, 32, 64);
        if (q_ok) {
            lse_i = p.lse[bh * p.T1 + qi];
            if (lk == 0) p.delta[bh * p.T1 + qi] = delta_i;
        }
    }
    f32x16 oacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    const DropParams dpar = drop_params(p.drop_p);
    const bool drop_aligned = (p.T2 % 8) == 0;
    const unsigned long long drop_rowbase = ((unsigned long long)(bh * p.T1 + (q_ok ? qi : 0))) * p.T2;
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);
    const bool key_mask = p.mask && p.m_rs == 0;
    const unsigned char* mrow = (p.mask && !key_mask) ? p.mask + (long)b * p.m_bs + (long)(q_ok ? qi : 0) * p.m_rs : nullptr;

    ChunkRegs<DPAD> kreg, vreg;
    float kb_next = 0.f;
    auto prefetch = [&](int j0) {
        chunk_load<DPAD>(kreg, kbp, p.k_rs, j0, p.T2, p.D, vk);
        chunk_load<DPAD>(vreg, vbp, p.v_rs, j0, p.T2, p.D, vv);
        if (threadIdx.x < PL_ROWS) {
            const int kj = j0 + threadIdx.x;
            float v = NEG_INF;
            if (kj < p.T2 && (!key_mask || p.mask[(long)b * p.m_bs + kj] != 0)) v = p.keybias ? p.keybias[bh * p.T2 + kj] : 0.f;
            kb_next = v;
        }
    };
    auto commit = [&](int buf) {
        chunk_store<DPAD, TERMS>(kreg, imgs + (buf * 2 + 0) * IMG);
        chunk_store<DPAD, TERMS>(vreg, imgs + (buf * 2 + 1) * IMG);
        if (threadIdx.x < PL_ROWS) kb_s[buf][threadIdx.x] = kb_next;
    };
    const int nchunks = (p.T2 + PL_ROWS - 1) / PL_ROWS;
    prefetch(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) prefetch((c + 1) * PL_ROWS);
        const __bf16* Ki = imgs + (buf * 2 + 0) * IMG;
        const __bf16* Vi = imgs + (buf * 2 + 1) * IMG;
        const int ntile = min(PL_ROWS / 32, (p.T2 - c * PL_ROWS + 31) >> 5);
        if (wave_live) {
            for (int jt = 0; jt < ntile; ++jt) {
                const int j0 = c * PL_ROWS + jt * 32;
                // S^T[key, query]
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    PFrag<TERMS> a;
                    P::row_frag(Ki, jt * 32 + lq, s, lk, a);
                    sacc = pmma<TERMS>(a, qfr[s], sacc);
                }
                float pr[16];
                float tmax = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kr = acc_row(r, lk);
                    float sv = sacc[r] + kb_s[buf][jt * 32 + kr];      // -inf for keys past T2 / masked keys
                    if (mrow && j0 + kr < p.T2 && mrow[j0 + kr] == 0) sv = NEG_INF;
                    pr[r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
                if (MODE == 0) {
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                    const float m_new = fmaxf(m_run, tmax);
                    const float corr = (m_new == NEG_INF) ? 1.f : __expf(m_run - m_new);
                    float psum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = (pr[r] == NEG_INF) ? 0.f : __expf(pr[r] - m_new);
                        psum += e;
                        pr[r] = e;
                    }
                    psum += __shfl_xor(psum, 32, 64);
                    l_run = l_run * corr + psum;
                    m_run = m_new;
#pragma unroll
                    for (int t = 0; t < DT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[t][r] *= corr;
                    if (p.drop_p > 0.f) {
                        float dm[16];
                        drop_tile_qlane(seed_eff, drop_rowbase, j0, lk, drop_aligned, dpar, dm);
#pragma unroll
                        for (int r = 0; r < 16; ++r) pr[r] *= dm[r];
                    }
                    // O^T[dv, query] += V^T[dv, key] P^T[key, query]
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                        PFrag<TERMS> pf;
                        psplit<TERMS>(x, pf);
#pragma unroll
                        for (int t = 0; t < DT; ++t) {
                            PFrag<TERMS> a;
                            P::col_frag(Vi, jt * 32, t * 32, s, lane, a);
                            oacc[t] = pmma<TERMS>(a, pf, oacc[t]);
                        }
                    }
                } else {
                    // P^T = exp(S^T - lse); dP^T[key, query] = V[key,:] . dO[query,:]
                    f32x16 dpacc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dpacc[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        PFrag<TERMS> a;
                        P::row_frag(Vi, jt * 32 + lq, s, lk, a);
                        dpacc = pmma<TERMS>(a, dofr[s], dpacc);
                    }
                    float dmask[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dmask[r] = 1.f;
                    if (p.drop_p > 0.f) drop_tile_qlane(seed_eff, drop_rowbase, j0, lk, drop_aligned, dpar, dmask);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = (pr[r] == NEG_INF) ? 0.f : __expf(pr[r] - lse_i);
                        pr[r] = pv * (dpacc[r] * dmask[r] - delta_i);     // dS^T
                    }
                    // dQ^T[d, query] += K^T[d, key] dS^T[key, query]
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                        PFrag<TERMS> df;
                        psplit<TERMS>(x, df);
#pragma unroll
                        for (int t = 0; t < DT; ++t) {
                            PFrag<TERMS> a;
                            P::col_frag(Ki, jt * 32, t * 32, s, lane, a);
                            oacc[t] = pmma<TERMS>(a, df, oacc[t]);
                        }
                    }
                }
            }
        }
        if (c + 1 < nchunks) commit(buf ^ 1);
        __syncthreads();
    }
    // ---- write back through the wave's patch (the images are dead: every wave is past the last barrier)
    float* patch = reinterpret_cast<float*>(lds_raw) + wave * (32 * (DPAD + 1));
    if (!wave_live) return;
    if (MODE == 0) {
        const float mul = (l_run > 0.f) ? 1.f / l_run : 0.f;
        if (q_ok && lk == 0) p.lse[bh * p.T1 + qi] = (l_run > 0.f) ? m_run + __logf(l_run) : NEG_INF;
        store_rows<DPAD>(patch, oacc, mul, p.o + (long)b * p.o_bs + h * p.D, p.o_rs, q0, p.T1, p.D, lane);
    } else {
        store_rows<DPAD>(patch, oacc, p.scale, p.dq + (long)b * p.q_bs + h * p.D, p.q_rs, q0, p.T1, p.D, lane);
    }
}

// ------------------------------------------------------------- dK / dV -------
// wave w of block x owns keys (4x + w) * 32 .. + 31 of (b, h); queries (Q and dO rows, lse, delta) stream through the planes.
template <int DPAD, int TERMS>
__global__ __launch_bounds__(PL_THREADS) void attn_planes_k_kernel(AttnParams p) {
    using P = Plane<DPAD, TERMS>;
    constexpr int DT = DPAD / 32, KS = DPAD / 16;
    constexpr int IMG = P::ELEMS;
    constexpr int PATCH_BYTES = 4 * 32 * (DPAD + 1) * 4;
    constexpr int IMG_BYTES = 2 * 2 * IMG * 2;                      // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) char lds_raw[(IMG_BYTES > PATCH_BYTES ? IMG_BYTES : PATCH_BYTES)];
    __shared__ float ld_s[2][2][PL_ROWS];                           // [buffer][lse | delta]
    __bf16* imgs = reinterpret_cast<__bf16*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lj = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    const bool wave_live = k0 < p.T2;
    const int kj = k0 + lj;
    const bool k_ok = kj < p.T2;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* dob = p.d_o + (long)b * p.o_bs + h * p.D;
    const long bh = (long)b * p.H + h;
    const bool vq = (p.D % 4 == 0) && (p.q_rs % 4 == 0) && ((((uintptr_t)qb) & 15) == 0);
    const bool vo = (p.D % 4 == 0) && (p.o_rs % 4 == 0) && ((((uintptr_t)dob) & 15) == 0);
    PFrag<TERMS> kfr[KS], vfr[KS];
    {
        const float* krow = p.k + (long)b * p.k_bs + (long)(k_ok ? kj : 0) * p.k_rs + h * p.D;
        const float* vrow = p.v + (long)b * p.v_bs + (long)(k_ok ? kj : 0) * p.v_rs + h * p.D;
        const bool vk8 = (p.D % 8 == 0) && (p.k_rs % 4 == 0) && (p.k_bs % 4 == 0) && ((((uintptr_t)p.k) & 15) == 0);
        const bool vv8 = (p.D % 8 == 0) && (p.v_rs % 4 == 0) && (p.v_bs % 4 == 0) && ((((uintptr_t)p.v) & 15) == 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float x[8], y[8];
            load8f(krow, 16 * s + 8 * lk, p.D, vk8, k_ok, x);
            load8f(vrow, 16 * s + 8 * lk, p.D, vv8, k_ok, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= p.scale;
            psplit<TERMS>(x, kfr[s]);
            psplit<TERMS>(y, vfr[s]);
        }
    }
    const float kbias = (p.keybias && k_ok) ? p.keybias[bh * p.T2 + kj] : 0.f;
    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[t][r] = 0.f; dvacc[t][r] = 0.f; }
    float dbias = 0.f;
    const DropParams dpar = drop_params(p.drop_p);
    const bool drop_aligned = (p.T2 % 8) == 0;
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);

    ChunkRegs<DPAD> qreg, oreg;
    float lse_next = 0.f, delta_next = 0.f;
    auto prefetch = [&](int i0) {
        chunk_load<DPAD>(qreg, qb, p.q_rs, i0, p.T1, p.D, vq);
        chunk_load<DPAD>(oreg, dob, p.o_rs, i0, p.T1, p.D, vo);
        if (threadIdx.x < PL_ROWS) {
            const int qi = i0 + threadIdx.x;
            lse_next = qi < p.T1 ? p.lse[bh * p.T1 + qi] : 0.f;
            delta_next = qi < p.T1 ? p.delta[bh * p.T1 + qi] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        chunk_store<DPAD, TERMS>(qreg, imgs + (buf * 2 + 0) * IMG);
        chunk_store<DPAD, TERMS>(oreg, imgs + (buf * 2 + 1) * IMG);
        if (threadIdx.x < PL_ROWS) { ld_s[buf][0][threadIdx.x] = lse_next; ld_s[buf][1][threadIdx.x] = delta_next; }
    };
    const int nchunks = (p.T1 + PL_ROWS - 1) / PL_ROWS;
    prefetch(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) prefetch((c + 1) * PL_ROWS);
        const __bf16* Qi = imgs + (buf * 2 + 0) * IMG;
        const __bf16* Oi = imgs + (buf * 2 + 1) * IMG;
        const int ntile = min(PL_ROWS / 32, (p.T1 - c * PL_ROWS + 31) >> 5);
        if (wave_live) {
            for (int it = 0; it < ntile; ++it) {
                const int i0 = c * PL_ROWS + it * 32;
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    PFrag<TERMS> a;
                    P::row_frag(Qi, it * 32 + lj, s, lk, a);
                    sacc = pmma<TERMS>(a, kfr[s], sacc);
                    P::row_frag(Oi, it * 32 + lj, s, lk, a);
                    dpacc = pmma<TERMS>(a, vfr[s], dpacc);
                }
                float pd[16], ds[16], dsc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
                if (p.drop_p > 0.f) {
                    if (drop_aligned) {
                        // the 8 lanes of a key block share 16 calls (one per query row of the tile): lane c computes rows c, c + 8
                        const int cc = lj & 7;
                        const unsigned long long kblk = (unsigned long long)(k0 + (lj & ~7));
                        const unsigned long long rowa = (unsigned long long)(bh * p.T1 + min(i0 + acc_row(cc, lk), p.T1 - 1)) * p.T2;
                        const unsigned long long rowb = (unsigned long long)(bh * p.T1 + min(i0 + acc_row(cc + 8, lk), p.T1 - 1)) * p.T2;
                        const uint4 wa = philox4(seed_eff, (rowa + kblk) >> 3), wb = philox4(seed_eff, (rowb + kblk) >> 3);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int src = (lane & ~7) | (r & 7);
                            const uint4 ws = (r < 8) ? wa : wb;
                            const unsigned x0 = __shfl(ws.x, src, 64), x1 = __shfl(ws.y, src, 64), x2 = __shfl(ws.z, src, 64), x3 = __shfl(ws.w, src, 64);
                            const unsigned w = (cc >> 1) == 0 ? x0 : (cc >> 1) == 1 ? x1 : (cc >> 1) == 2 ? x2 : x3;
                            dsc[r] = drop_field(w, cc & 1, dpar);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dsc[r] = drop_elem(seed_eff, ((unsigned long long)(bh * p.T1 + min(i0 + acc_row(r, lk), p.T1 - 1))) * p.T2 + min(kj, p.T2 - 1), dpar);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = acc_row(r, lk);
                    const int qi = i0 + qr;
                    bool ok = k_ok && qi < p.T1;
                    if (ok && p.mask) ok = p.mask[(long)b * p.m_bs + (long)qi * p.m_rs + kj] != 0;
                    const float pv = ok ? __expf(sacc[r] + kbias - ld_s[buf][0][it * 32 + qr]) : 0.f;
                    pd[r] = pv * dsc[r];                                                     // dropped attention weights
                    ds[r] = pv * (dpacc[r] * dsc[r] - ld_s[buf][1][it * 32 + qr]);           // dS
                    dbias += ds[r];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8], y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { x[e] = pd[8 * s + e]; y[e] = ds[8 * s + e]; }
                    PFrag<TERMS> pf, df;
                    psplit<TERMS>(x, pf);
                    psplit<TERMS>(y, df);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        PFrag<TERMS> a;
                        P::col_frag(Oi, it * 32, t * 32, s, lane, a);
                        dvacc[t] = pmma<TERMS>(a, pf, dvacc[t]);
                        P::col_frag(Qi, it * 32, t * 32, s, lane, a);
                        dkacc[t] = pmma<TERMS>(a, df, dkacc[t]);
                    }
                }
            }
        }
        if (c + 1 < nchunks) commit(buf ^ 1);
        __syncthreads();
    }
    if (!wave_live) return;
    dbias += __shfl_xor(dbias, 32, 64);
    if (k_ok && p.dkeybias && lk == 0) p.dkeybias[bh * p.T2 + kj] = dbias;
    float* patch = reinterpret_cast<float*>(lds_raw) + wave * (32 * (DPAD + 1));
    store_rows<DPAD>(patch, dkacc, p.scale, p.dk + (long)b * p.k_bs + h * p.D, p.k_rs, k0, p.T2, p.D, lane);
    store_rows<DPAD>(patch, dvacc, 1.f, p.dv + (long)b * p.v_bs + h * p.D, p.v_rs, k0, p.T2, p.D, lane);
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
// OE_ATTN_PLANES: 0 = never, 1 = where the structure fits (default): at least three 32-row tiles on the resident axis
// (a block is four waves of one tile each; smaller problems - the decoders' 31-token queries - keep the split-key kernels
// of attention.hip) and D <= 64.
static int planes_mode() {
    static const int mode = getenv("OE_ATTN_PLANES") ? atoi(getenv("OE_ATTN_PLANES")) : 1;
    return mode;
}
static bool planes_fit(int resident_rows) { return planes_mode() != 0 && resident_rows > 64; }

int oe_attn_planes_fwd_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T1) || (terms != 1 && terms != 3)) return 1;
    dim3 grid(oe_cdiv(p.T1, 128), p.H, p.B);
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<32, 3, 0>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<32, 1, 0>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<64, 3, 0>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 1, 0>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
int oe_attn_planes_dq_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T1) || (terms != 1 && terms != 3)) return 1;
    dim3 grid(oe_cdiv(p.T1, 128), p.H, p.B);
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<32, 3, 1>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<32, 1, 1>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<64, 3, 1>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 1, 1>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
int oe_attn_planes_dkdv_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T2) || (terms != 1 && terms != 3)) return 1;
    dim3 grid(oe_cdiv(p.T2, 128), p.H, p.B);
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_k_kernel<32, 3>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_k_kernel<32, 1>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_k_kernel<64, 3>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_k_kernel<64, 1>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
