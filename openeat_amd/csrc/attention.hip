// Fused multi-head attention for the encoder (relative-position form) and the
// decoder (plain form): scores -> mask -> softmax -> dropout -> .V without ever
// materialising the (B,H,T1,T2) score tensor, forward and backward.
//
// Replaces /root/reference/openeat/modules/attention.py:65-97 (forward_attention),
// :112-117 and :189-209 (score computation) and their autograd.
//
// Relative-position scores without rel_shift (attention.py:202-204) factor as
//     ((q+u) K^T + (q+v) P^T)/sqrt(dk) = q (K+P)^T / sqrt(dk) + (u.K_j + v.P_j)/sqrt(dk)
// i.e. ordinary attention on keys K' = K+P plus a per-key bias; the host side
// prepares K' and the bias (oe_relpos_prepare), so one kernel serves both forms.
//
// Matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, args->precision 0) or v_mfma_f32_32x32x16_bf16 with the
// operands split to bf16 in registers (precision 1: plain bf16 products, 3: three-term split, 2^-17 per product, 6: six terms on three exact pieces; same
// meaning as oe_gemm_args.precision).  Layout trick: the forward
// and dQ kernels compute the TRANSPOSED score tile S^T = K Q^T, so the query
// sits on the lane (softmax statistics are per-lane scalars) and the key on
// the accumulator rows, which is exactly the B-operand layout of the following
// product over keys (O^T = V^T P^T, dQ^T = K^T dS^T): P never leaves registers.
// The dK/dV kernel uses the untransposed tile for the products over queries.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s); algorithmic flops
// fwd = 4*B*H*T1*T2*D, bwd = 14*B*H*T1*T2*D (S recomputed in both kernels).
#include <stdlib.h>
#include "attn_common.h"

// Diagnostic build only (-DOE_GEMM_STAMPS, tools/attn_stamps.py): s_memtime sums per phase of the forward kernel.
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* oe_attn_stamp_buf = nullptr;
extern "C" int oe_debug_set_attn_stamp_buffer(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(oe_attn_stamp_buf), &p, sizeof(p));
}
#define ATT_NOW(var)                                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define ATT_ACC(slot, t0)                                                                    \
    do { unsigned long long t1_; ATT_NOW(t1_); att_acc[slot] += t1_ - t0; t0 = t1_; } while (0)
#else
#define ATT_NOW(var) do { } while (0)
#define ATT_ACC(slot, t0) do { } while (0)
#endif

// ---- bf16 matrix-core variants (TERMS = 1: bf16 products; TERMS = 3: hi*hi + hi*lo + lo*hi, 2^-17 per product) ----
// v_mfma_f32_32x32x16_bf16: lane (i = lane&31, g = lane>>5) supplies 8 consecutive k-slots 8g..8g+7 of row i (A) /
// column i (B).  The k-slot -> (feature | key | query) assignment is free as long as A and B agree, so
//   * products over features take slots = 8 consecutive features (row fragment, A from an LDS tile row);
//   * products over the 32 keys / queries of a tile take slot e of step s = accumulator row acc_row(8s+e, g): the
//     B operand is then simply registers 8s..8s+7 of the score tile this lane already holds (P never leaves
//     registers, as in the fp32 kernels), and A is read from LDS at those rows (column fragment).
// The tiles stay fp32 in LDS; fragments are split to bf16 hi (+ lo) in registers.
// (TERMS = 6: three exact pieces and six products, oe_common.h - the arithmetic of precision 6 where the planes kernels of
// attention_bf16.hip do not fit: the decoders' 31-query self- and source attention)
template <int TERMS> struct BFrag { bf16x8 p[oe_npl<TERMS>::N]; };

template <int TERMS>
__device__ __forceinline__ void bsplit(const float (&x)[8], BFrag<TERMS>& f) {
    oe_split8<oe_npl<TERMS>::N>(x, f.p);
}
template <int TERMS>
__device__ __forceinline__ f32x16 bmma(const BFrag<TERMS>& a, const BFrag<TERMS>& b, f32x16 c) {
    return oe_mma_terms<TERMS>(a, b, c);
}
// A[row][16s + 8g + e] of an LDS tile with row stride LD
template <int TERMS, int LD>
__device__ __forceinline__ void row_frag(const float* tile, int row, int s, int g, BFrag<TERMS>& f) {
    float x[8];
    const float* p = tile + row * LD + 16 * s + 8 * g;
    if constexpr (LD % 4 == 0) {
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = c.x; x[5] = c.y; x[6] = c.z; x[7] = c.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e];
    }
    bsplit<TERMS>(x, f);
}
// A[acc_row(8s + e, g)][col] of an LDS tile: the rows this lane's accumulator registers 8s..8s+7 stand for
template <int TERMS, int LD>
__device__ __forceinline__ void col_frag(const float* tile, int col, int s, int g, BFrag<TERMS>& f) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[((e & 3) + 8 * (2 * s + (e >> 2)) + 4 * g) * LD + col];
    bsplit<TERMS>(x, f);
}
#define ATT_WAVES 2                       // 32-row tiles (queries, or keys in the dK/dV kernel) per block
#define ATT_GROUP (64 * ATT_WAVES)        // threads of one wave group: they stage and consume the same LDS tiles
#define ATT_MAX_SPLIT 2                   // wave groups that share the reduction loop of a tile (template parameter SPLIT)

// Work split.  B*H*ceil(T/32) tiles are only ~1000 waves at the path's sizes - one per SIMD, each a serial chain
// over the other sequence axis.  A tile's chain can be cut in SPLIT parts run by different wave groups of the same
// block (keys for forward / dQ, queries for dK/dV) and merged through LDS at the end (online-softmax merge of
// (m, l, O) for the forward, plain sums for the gradients); thread t belongs to group t / ATT_GROUP.  Measured at
// config 2: the forward gains 22 % from SPLIT = 2, the two backward kernels (VALU-bound on the bf16 fragment
// splits, so a second wave per SIMD only competes for the same pipe) lose 7-14 % and stay at SPLIT = 1.
//
// A [32][DPAD] tile travels global -> registers (issued early) -> LDS (row stride LD, written late):
// the global latency of tile j+1 hides under the MFMA/softmax work of tile j.
template <int DPAD>
struct TileRegs {
    static constexpr int N = 32 * (DPAD / 4) / ATT_GROUP;
    float4 v[N];
};

template <int DPAD>
__device__ __forceinline__ void tile_load(TileRegs<DPAD>& t, const float* src, long rs, int r0, int nrows_total, int D) {
    const bool vec = (rs % 4 == 0) && ((((uintptr_t)src) & 15) == 0) && (D % 4 == 0);
#pragma unroll
    for (int i = 0; i < TileRegs<DPAD>::N; ++i) {
        const int e = (threadIdx.x % ATT_GROUP) + i * ATT_GROUP;
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

template <int DPAD, int LD>
__device__ __forceinline__ void tile_store(const TileRegs<DPAD>& t, float* dst) {
#pragma unroll
    for (int i = 0; i < TileRegs<DPAD>::N; ++i) {
        const int e = (threadIdx.x % ATT_GROUP) + i * ATT_GROUP;
        const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
        float* d = dst + row * LD + c4;
        if constexpr (LD % 4 == 0) *reinterpret_cast<float4*>(d) = t.v[i];
        else { d[0] = t.v[i].x; d[1] = t.v[i].y; d[2] = t.v[i].z; d[3] = t.v[i].w; }
    }
}

// Stage a [32][DPAD] tile synchronously (used where no pipelining is needed).
template <int DPAD, int LD>
__device__ __forceinline__ void stage_tile(float* dst, const float* src, long rs, int r0, int nrows_total, int D,
                                           float mul) {
    TileRegs<DPAD> t;
    tile_load<DPAD>(t, src, rs, r0, nrows_total, D);
    tile_store<DPAD, LD>(t, dst);
}

// ------------------------------------------------------------------ forward --
// MODE 0: forward (writes O, LSE).  MODE 1: dQ (reads dO, LSE, delta; writes dQ).
// (second launch bound = waves per SIMD the register allocation must leave room for: the split forward needs two
// of its four-wave blocks per CU to have all 512 blocks of config 2 resident at once)
template <int DPAD, int MODE, int TERMS, int SPLIT>
__global__ __launch_bounds__(ATT_GROUP * SPLIT, (MODE == 1 && TERMS == 6) ? 1 : SPLIT) void attn_qtile_kernel(AttnParams p) {
    constexpr int LD = DPAD + 4;             // pitch: rows stay 16-byte aligned (b128 LDS accesses) and 4 banks apart
    constexpr int DT = DPAD / 32;
    constexpr int KS = DPAD / 16;            // bf16 k-steps over the features
    __shared__ __attribute__((aligned(16))) float KVs[SPLIT][2][32 * LD];         // [group][K | V] tiles; reused as the merge buffer at the end
    __shared__ float kbs[SPLIT][32];                 // per-key bias of the tile; -inf marks a key masked for every query
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % ATT_WAVES, grp = threadIdx.x / ATT_GROUP;
    const int gtid = threadIdx.x % ATT_GROUP;
    float* Ks = KVs[grp][0];
    float* Vs = KVs[grp][1];
    float* kb_s = kbs[grp];
    const int lq = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * ATT_WAVES + wave) * 32;
    const int qi = q0 + lq;                       // this lane's query
    const bool q_ok = qi < p.T1;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* kb = p.k + (long)b * p.k_bs + h * p.D;
    const float* vb = p.v + (long)b * p.v_bs + h * p.D;
    const long bh = (long)b * p.H + h;

    // Q^T fragments (B operand of S^T = K Q^T), pre-scaled; for dQ also dO^T fragments
    float qf[TERMS == 0 ? DPAD / 2 : 1];
    float dof[(TERMS == 0 && MODE == 1) ? DPAD / 2 : 1];
    BFrag<TERMS == 0 ? 1 : TERMS> qfr[TERMS == 0 ? 1 : KS], dofr[(TERMS != 0 && MODE == 1) ? KS : 1];
    // dQ pass: delta_i = sum_d dO[i,d] * O[i,d] is formed here from the dO values this lane loads anyway (the two lanes of a
    // query hold complementary halves of the features) and published for the dK/dV kernel that follows on the stream
    float dpart = 0.f;
    const long orow = (long)b * p.o_bs + (long)qi * p.o_rs + h * p.D;
    if constexpr (TERMS == 0) {
#pragma unroll
        for (int s = 0; s < DPAD / 2; ++s) {
            const int d = 2 * s + lk;
            qf[s] = (q_ok && d < p.D) ? qb[(long)qi * p.q_rs + d] * p.scale : 0.f;
            if (MODE == 1) {
                dof[s] = (q_ok && d < p.D) ? p.d_o[orow + d] : 0.f;
                dpart += (q_ok && d < p.D) ? dof[s] * p.o_in[orow + d] : 0.f;
            }
        }
    } else {
        // eight consecutive features per step: two float4 loads when rows are 16-byte aligned and D is a multiple of 8
        const bool vq = (p.D % 8 == 0) && (p.q_rs % 4 == 0) && ((((uintptr_t)qb) & 15) == 0);
        const bool vo = (p.D % 8 == 0) && (p.o_rs % 4 == 0) && (p.o_bs % 4 == 0) &&
                        (MODE != 1 || (((((uintptr_t)p.d_o) | ((uintptr_t)p.o_in)) & 15) == 0));
        auto load8 = [&](const float* row, int d0, bool vec, float (&out)[8]) {
            if (vec && q_ok && d0 + 8 <= p.D) {
                const float4 a = *reinterpret_cast<const float4*>(row + d0), c = *reinterpret_cast<const float4*>(row + d0 + 4);
                out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = c.x; out[5] = c.y; out[6] = c.z; out[7] = c.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (q_ok && d0 + e < p.D) ? row[d0 + e] : 0.f;
            }
        };
        const float* qrow = qb + (long)(q_ok ? qi : 0) * p.q_rs;
        const long orow_c = q_ok ? orow : (long)b * p.o_bs + h * p.D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d0 = 16 * s + 8 * lk;
            float x[8], y[8];
            load8(qrow, d0, vq, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= p.scale;
            if (MODE == 1) {
                float ov[8];
                load8(p.d_o + orow_c, d0, vo, y);
                load8(p.o_in + orow_c, d0, vo, ov);
#pragma unroll
                for (int e = 0; e < 8; ++e) dpart += y[e] * ov[e];
            }
            bsplit<TERMS>(x, qfr[s]);
            if (MODE == 1) bsplit<TERMS>(y, dofr[s]);
        }
    }
    float m_run = NEG_INF, l_run = 0.f;
    float lse_i = 0.f, delta_i = 0.f;
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
    const unsigned long long drop_row = (unsigned long long)(bh * p.T1 + (q_ok ? qi : 0));        // attn_common.h: mask definition
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);
    // a (B,1,T2) key mask is the same for every query: fold it into the staged per-key bias;
    // only a full (B,T1,T2) mask (decoder self-attention) is read per (query, key)
    const bool key_mask = p.mask && p.m_rs == 0;
    const unsigned char* mrow = (p.mask && !key_mask) ? p.mask + (long)b * p.m_bs + (long)(q_ok ? qi : 0) * p.m_rs : nullptr;

#ifdef OE_GEMM_STAMPS
    unsigned long long att_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, att_t = 0, att_t0 = 0;
    ATT_NOW(att_t0);
    att_t = att_t0;
#endif
    TileRegs<DPAD> kreg, vreg;
    float kb_next = 0.f;
    auto prefetch = [&](int j0) {
        tile_load<DPAD>(kreg, kb, p.k_rs, j0, p.T2, p.D);
        tile_load<DPAD>(vreg, vb, p.v_rs, j0, p.T2, p.D);
        if (gtid < 32) {
            const int kj = j0 + gtid;
            float v = NEG_INF;
            if (kj < p.T2 && (!key_mask || p.mask[(long)b * p.m_bs + kj] != 0)) v = p.keybias ? p.keybias[bh * p.T2 + kj] : 0.f;
            kb_next = v;
        }
    };
    // this group's share of the key tiles: [grp * per, (grp + 1) * per); tiles past T2 are all-masked (bias -inf) and
    // every group runs the same number of iterations (block-wide barriers inside)
    const int per = (((p.T2 + 31) >> 5) + SPLIT - 1) / SPLIT;
    const int j_begin = grp * per * 32;
    prefetch(j_begin);
    ATT_ACC(0, att_t);                                  // prologue: Q fragments, first prefetch issue
    for (int it = 0; it < per; ++it) {
        const int j0 = j_begin + it * 32;
        __syncthreads();
        ATT_ACC(1, att_t);                              // wait at the first barrier
        tile_store<DPAD, LD>(kreg, Ks);
        tile_store<DPAD, LD>(vreg, Vs);
        if (gtid < 32) kb_s[gtid] = kb_next;
        __syncthreads();
        ATT_ACC(2, att_t);                              // tile store (waits for the prefetched data) + second barrier
        if (it + 1 < per) prefetch(j0 + 32);
        ATT_ACC(3, att_t);                              // prefetch issue
        // S^T[key, query]
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        if constexpr (TERMS == 0) {
#pragma unroll
            for (int s = 0; s < DPAD / 2; ++s)
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[lq * LD + 2 * s + lk], qf[s], sacc, 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                BFrag<TERMS> a;
                row_frag<TERMS, LD>(Ks, lq, s, lk, a);
                sacc = bmma<TERMS>(a, qfr[s], sacc);
            }
        }
        ATT_ACC(4, att_t);                              // S fragments + MFMAs (issue)
        float pr[16];
        float tmax = NEG_INF;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = acc_row(r, lk);
            float sv = sacc[r] + kb_s[kr];                 // -inf for keys past T2 / masked keys
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
                attn_drop_qlane(seed_eff, drop_row, j0, lk, dpar, dm);
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] *= dm[r];
            }
            ATT_ACC(5, att_t);                          // softmax (waits for the S MFMAs)
            // O^T[dv, query] += V^T[dv, key] P^T[key, query]
            if constexpr (TERMS == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = acc_row(r, lk);
#pragma unroll
                    for (int t = 0; t < DT; ++t)
                        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[krow * LD + t * 32 + lq], pr[r], oacc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                    BFrag<TERMS> pf;
                    bsplit<TERMS>(x, pf);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        BFrag<TERMS> a;
                        col_frag<TERMS, LD>(Vs, t * 32 + lq, s, lk, a);
                        oacc[t] = bmma<TERMS>(a, pf, oacc[t]);
                    }
                }
            }
        } else {
            // P^T = exp(S^T - lse); dP^T[key, query] = V[key,:] . dO[query,:]
            f32x16 dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dpacc[r] = 0.f;
            if constexpr (TERMS == 0) {
#pragma unroll
                for (int s = 0; s < DPAD / 2; ++s)
                    dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[lq * LD + 2 * s + lk], dof[s], dpacc, 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    BFrag<TERMS> a;
                    row_frag<TERMS, LD>(Vs, lq, s, lk, a);
                    dpacc = bmma<TERMS>(a, dofr[s], dpacc);
                }
            }
            float dmask[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dmask[r] = 1.f;
            if (p.drop_p > 0.f) attn_drop_qlane(seed_eff, drop_row, j0, lk, dpar, dmask);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = (pr[r] == NEG_INF) ? 0.f : __expf(pr[r] - lse_i);
                pr[r] = pv * (dpacc[r] * dmask[r] - delta_i);     // dS^T
            }
            // dQ^T[d, query] += K^T[d, key] dS^T[key, query]
            if constexpr (TERMS == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = acc_row(r, lk);
#pragma unroll
                    for (int t = 0; t < DT; ++t)
                        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[krow * LD + t * 32 + lq], pr[r], oacc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                    BFrag<TERMS> df;
                    bsplit<TERMS>(x, df);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        BFrag<TERMS> a;
                        col_frag<TERMS, LD>(Ks, t * 32 + lq, s, lk, a);
                        oacc[t] = bmma<TERMS>(a, df, oacc[t]);
                    }
                }
            }
        }
    }
    ATT_ACC(6, att_t);                                  // last PV issue -> here (loop exit)
#ifdef OE_GEMM_STAMPS
    if (oe_attn_stamp_buf && MODE == 0 && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y < 4 && blockIdx.z < 32) {
        unsigned long long* o = oe_attn_stamp_buf + (blockIdx.z * 4 + blockIdx.y) * 8;
        for (int i = 0; i < 7; ++i) o[i] = att_acc[i];
        o[7] = att_t - att_t0;
    }
#endif
    if constexpr (SPLIT == 2) {
        // ---- merge the key halves: group 1 parks (m, l, O^T) / dQ^T in LDS, group 0 combines and writes
        __syncthreads();                               // every wave is done with the K/V tiles the buffer overlays
        float* mb = &KVs[0][0][0] + wave * (DPAD * 32 + 64);      // [DPAD rows][32 queries] + m[32] + l[32] per q-tile
        if (grp == 1) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mb[(t * 32 + acc_row(r, lk)) * 32 + lq] = oacc[t][r];
            if (MODE == 0 && lk == 0) { mb[DPAD * 32 + lq] = m_run; mb[DPAD * 32 + 32 + lq] = l_run; }
        }
        __syncthreads();
        if (grp == 1) return;
        float a1 = 1.f, a2 = 1.f;
        if (MODE == 0) {
            const float m2 = mb[DPAD * 32 + lq], l2 = mb[DPAD * 32 + 32 + lq];
            const float m = fmaxf(m_run, m2);
            a1 = (m_run == NEG_INF) ? 0.f : __expf(m_run - m);
            a2 = (m2 == NEG_INF) ? 0.f : __expf(m2 - m);
            l_run = l_run * a1 + l2 * a2;
            m_run = m;
        }
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] = oacc[t][r] * a1 + mb[(t * 32 + acc_row(r, lk)) * 32 + lq] * a2;
    }
    if (!q_ok) return;
    // ---- write back: lane = query, accumulator rows = feature index
    float mul;
    float* dst;
    if (MODE == 0) {
        mul = (l_run > 0.f) ? 1.f / l_run : 0.f;
        dst = p.o + (long)b * p.o_bs + (long)qi * p.o_rs + h * p.D;
        if (lk == 0) p.lse[bh * p.T1 + qi] = (l_run > 0.f) ? m_run + __logf(l_run) : NEG_INF;
    } else {
        mul = p.scale;
        dst = p.dq + (long)b * p.q_bs + (long)qi * p.q_rs + h * p.D;
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + acc_row(r, lk);
            if (d < p.D) dst[d] = oacc[t][r] * mul;
        }
}

// ------------------------------------------------------------- dK / dV -------
template <int DPAD, int TERMS, int SPLIT>
__global__ __launch_bounds__(ATT_GROUP * SPLIT) void attn_ktile_bwd_kernel(AttnParams p) {
    constexpr int LD = DPAD + 4;             // pitch: rows stay 16-byte aligned (b128 LDS accesses) and 4 banks apart
    constexpr int DT = DPAD / 32;
    constexpr int KS = DPAD / 16;
    __shared__ __attribute__((aligned(16))) float QOs[SPLIT][2][32 * LD];         // [group][Q | dO] tiles; reused as the merge buffer at the end
    __shared__ float lds_s[SPLIT][2][32];            // [group][lse | delta]
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % ATT_WAVES, grp = threadIdx.x / ATT_GROUP;
    const int gtid = threadIdx.x % ATT_GROUP;
    float* Qs = QOs[grp][0];
    float* Os = QOs[grp][1];
    float* lse_s = lds_s[grp][0];
    float* delta_s = lds_s[grp][1];
    const int lj = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int k0 = (blockIdx.x * ATT_WAVES + wave) * 32;
    const int kj = k0 + lj;                      // this lane's key
    const bool k_ok = kj < p.T2;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* dob = p.d_o + (long)b * p.o_bs + h * p.D;
    const long bh = (long)b * p.H + h;
    float kf[TERMS == 0 ? DPAD / 2 : 1], vf[TERMS == 0 ? DPAD / 2 : 1];
    BFrag<TERMS == 0 ? 1 : TERMS> kfr[TERMS == 0 ? 1 : KS], vfr[TERMS == 0 ? 1 : KS];
    if constexpr (TERMS == 0) {
#pragma unroll
        for (int s = 0; s < DPAD / 2; ++s) {
            const int d = 2 * s + lk;
            const bool ok = k_ok && d < p.D;
            kf[s] = ok ? p.k[(long)b * p.k_bs + (long)kj * p.k_rs + h * p.D + d] * p.scale : 0.f;
            vf[s] = ok ? p.v[(long)b * p.v_bs + (long)kj * p.v_rs + h * p.D + d] : 0.f;
        }
    } else {
        // eight consecutive features per step: two float4 loads when the rows are 16-byte aligned and D is a multiple of 8
        const float* krow = p.k + (long)b * p.k_bs + (long)(k_ok ? kj : 0) * p.k_rs + h * p.D;
        const float* vrow = p.v + (long)b * p.v_bs + (long)(k_ok ? kj : 0) * p.v_rs + h * p.D;
        const bool vk = (p.D % 8 == 0) && (p.k_rs % 4 == 0) && (p.k_bs % 4 == 0) && ((((uintptr_t)p.k) & 15) == 0);
        const bool vv = (p.D % 8 == 0) && (p.v_rs % 4 == 0) && (p.v_bs % 4 == 0) && ((((uintptr_t)p.v) & 15) == 0);
        auto load8 = [&](const float* row, int d0, bool vec, float (&out)[8]) {
            if (vec && k_ok && d0 + 8 <= p.D) {
                const float4 a = *reinterpret_cast<const float4*>(row + d0), c = *reinterpret_cast<const float4*>(row + d0 + 4);
                out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = c.x; out[5] = c.y; out[6] = c.z; out[7] = c.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (k_ok && d0 + e < p.D) ? row[d0 + e] : 0.f;
            }
        };
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float x[8], y[8];
            load8(krow, 16 * s + 8 * lk, vk, x);
            load8(vrow, 16 * s + 8 * lk, vv, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= p.scale;
            bsplit<TERMS>(x, kfr[s]);
            bsplit<TERMS>(y, vfr[s]);
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
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);

    // this group's share of the query tiles; tiles past T1 contribute nothing (ok = false for every element)
    const int per = (((p.T1 + 31) >> 5) + SPLIT - 1) / SPLIT;
    TileRegs<DPAD> qreg, oreg;
    float lse_next = 0.f, delta_next = 0.f;
    auto prefetch = [&](int i0) {
        tile_load<DPAD>(qreg, qb, p.q_rs, i0, p.T1, p.D);
        tile_load<DPAD>(oreg, dob, p.o_rs, i0, p.T1, p.D);
        if (gtid < 32) {
            const int qi = i0 + gtid;
            lse_next = qi < p.T1 ? p.lse[bh * p.T1 + qi] : 0.f;
            delta_next = qi < p.T1 ? p.delta[bh * p.T1 + qi] : 0.f;
        }
    };
    prefetch(grp * per * 32);
    for (int it = 0; it < per; ++it) {
        const int i0 = (grp * per + it) * 32;
        __syncthreads();
        tile_store<DPAD, LD>(qreg, Qs);
        tile_store<DPAD, LD>(oreg, Os);
        if (gtid < 32) { lse_s[gtid] = lse_next; delta_s[gtid] = delta_next; }
        __syncthreads();
        if (it + 1 < per) prefetch(i0 + 32);
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        if constexpr (TERMS == 0) {
#pragma unroll
            for (int s = 0; s < DPAD / 2; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[lj * LD + 2 * s + lk], kf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[lj * LD + 2 * s + lk], vf[s], dpacc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                BFrag<TERMS> a;
                row_frag<TERMS, LD>(Qs, lj, s, lk, a);
                sacc = bmma<TERMS>(a, kfr[s], sacc);
                row_frag<TERMS, LD>(Os, lj, s, lk, a);
                dpacc = bmma<TERMS>(a, vfr[s], dpacc);
            }
        }
        float pd[16], ds[16], dsc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
        if (p.drop_p > 0.f) attn_drop_klane(seed_eff, (unsigned long long)(bh * p.T1), i0, kj, lk, dpar, dsc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = acc_row(r, lk);
            const int qi = i0 + qr;
            bool ok = k_ok && qi < p.T1;
            if (ok && p.mask) ok = p.mask[(long)b * p.m_bs + (long)qi * p.m_rs + kj] != 0;
            const float pv = ok ? __expf(sacc[r] + kbias - lse_s[qr]) : 0.f;
            const float dscale = dsc[r];
            pd[r] = pv * dscale;                                   // dropped attention weights
            ds[r] = pv * (dpacc[r] * dscale - delta_s[qr]);        // dS
            dbias += ds[r];
        }
        if constexpr (TERMS == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = acc_row(r, lk);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    dvacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[qr * LD + t * 32 + lj], pd[r], dvacc[t], 0, 0, 0);
                    dkacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qr * LD + t * 32 + lj], ds[r], dkacc[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[8], y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] = pd[8 * s + e]; y[e] = ds[8 * s + e]; }
                BFrag<TERMS> pf, df;
                bsplit<TERMS>(x, pf);
                bsplit<TERMS>(y, df);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    BFrag<TERMS> a;
                    col_frag<TERMS, LD>(Os, t * 32 + lj, s, lk, a);
                    dvacc[t] = bmma<TERMS>(a, pf, dvacc[t]);
                    col_frag<TERMS, LD>(Qs, t * 32 + lj, s, lk, a);
                    dkacc[t] = bmma<TERMS>(a, df, dkacc[t]);
                }
            }
        }
    }
    dbias += __shfl_xor(dbias, 32, 64);
    if constexpr (SPLIT == 2) {
        // ---- merge the query halves: group 1 parks dK^T, dV^T and the bias gradient in LDS, group 0 adds
        __syncthreads();
        float* mb = &QOs[0][0][0] + wave * (2 * DPAD * 32);        // [dK | dV][DPAD rows][32 keys] per key tile
        float* mbias = &lds_s[0][0][0] + wave * 32;
        if (grp == 1) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    mb[(t * 32 + acc_row(r, lk)) * 32 + lj] = dkacc[t][r];
                    mb[DPAD * 32 + (t * 32 + acc_row(r, lk)) * 32 + lj] = dvacc[t][r];
                }
            if (lk == 0) mbias[lj] = dbias;
        }
        __syncthreads();
        if (grp == 1) return;
        dbias += mbias[lj];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[t][r] += mb[(t * 32 + acc_row(r, lk)) * 32 + lj];
                dvacc[t][r] += mb[DPAD * 32 + (t * 32 + acc_row(r, lk)) * 32 + lj];
            }
    }
    if (!k_ok) return;
    if (p.dkeybias && lk == 0) p.dkeybias[bh * p.T2 + kj] = dbias;
    float* dkd = p.dk + (long)b * p.k_bs + (long)kj * p.k_rs + h * p.D;
    float* dvd = p.dv + (long)b * p.v_bs + (long)kj * p.v_rs + h * p.D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + acc_row(r, lk);
            if (d < p.D) { dkd[d] = dkacc[t][r] * p.scale; dvd[d] = dvacc[t][r]; }
        }
}

static int fill_params(AttnParams& p, const oe_attn_args* a, const char* who) {
    if (!(a && a->q && a->k && a->v)) { oe_set_error("%s: null q/k/v", who); return -1; }
    if (!(a->B > 0 && a->H > 0 && a->T1 > 0 && a->T2 > 0 && a->D > 0 && a->D <= 64)) {
        oe_set_error("%s: bad shape B=%d H=%d T1=%d T2=%d D=%d (D must be <= 64)", who, a->B, a->H, a->T1, a->T2, a->D);
        return -1;
    }
    if (!(a->drop_p >= 0.f && a->drop_p < 1.f)) { oe_set_error("%s: drop_p out of range", who); return -1; }
    if (!(a->precision == 0 || a->precision == 1 || a->precision == 3 || a->precision == 6)) { oe_set_error("%s: precision must be 0, 1, 3 or 6", who); return -1; }
    p.q = a->q; p.q_bs = a->q_bstride; p.q_rs = a->q_rstride;
    p.k = a->k; p.k_bs = a->k_bstride; p.k_rs = a->k_rstride;
    p.v = a->v; p.v_bs = a->v_bstride; p.v_rs = a->v_rstride;
    p.o = a->out; p.o_in = a->out; p.o_bs = a->o_bstride; p.o_rs = a->o_rstride;
    p.d_o = a->d_out; p.dq = a->dq; p.dk = a->dk; p.dv = a->dv;
    p.lse = a->lse; p.delta = a->delta;
    p.mask = a->mask; p.m_bs = a->mask_bstride; p.m_rs = a->mask_rstride;
    p.keybias = a->keybias; p.dkeybias = a->dkeybias;
    p.B = a->B; p.H = a->H; p.T1 = a->T1; p.T2 = a->T2; p.D = a->D;
    p.scale = a->scale; p.drop_p = a->drop_p; p.seed = a->seed; p.seed_dev = a->seed_dev;
    p.causal = a->causal && a->mask && a->mask_rstride != 0 && a->T1 == a->T2;       // only with a full (B, T1, T2) mask that says so too
    return 0;
}

extern "C" int oe_attention_fwd(const oe_attn_args* a, void* stream) {
    AttnParams p{};
    if (int rc = fill_params(p, a, "oe_attention_fwd")) return rc;
    OE_REQUIRE(a->out && a->lse, "oe_attention_fwd: null out/lse");
    dim3 grid(oe_cdiv(p.T1, 32 * ATT_WAVES), p.H, p.B);
    hipStream_t st = (hipStream_t)stream;
    if (a->precision != 0 && oe_attn_planes_fwd_try(p, a->precision, st) == 0) {      // attention_bf16.hip
        OE_LAUNCH_CHECK("oe_attention_fwd (bf16 planes)");
        return 0;
    }
#define ATT_FWD(TT)                                                                                       \
    do {                                                                                                  \
        if (p.D <= 32) hipLaunchKernelGGL((attn_qtile_kernel<32, 0, TT, 2>), grid, dim3(ATT_GROUP * 2), 0, st, p); \
        else hipLaunchKernelGGL((attn_qtile_kernel<64, 0, TT, 2>), grid, dim3(ATT_GROUP * 2), 0, st, p);        \
    } while (0)
    // precision 6 where the planes kernel does not fit (short query axes: the decoders): the six-term product on fragments split in registers
    if (a->precision == 3) ATT_FWD(3); else if (a->precision == 1) ATT_FWD(1); else if (a->precision == 6) ATT_FWD(6); else ATT_FWD(0);
#undef ATT_FWD
    OE_LAUNCH_CHECK("oe_attention_fwd");
    return 0;
}

extern "C" int oe_attention_bwd(const oe_attn_args* a, void* stream) {
    AttnParams p{};
    if (int rc = fill_params(p, a, "oe_attention_bwd")) return rc;
    OE_REQUIRE(a->out && a->lse && a->d_out && a->dq && a->dk && a->dv && a->delta, "oe_attention_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    // (delta = rowsum(dO * O) is computed by the dQ kernel and read by the dK/dV kernel: no separate launch)
    dim3 gq(oe_cdiv(p.T1, 32 * ATT_WAVES), p.H, p.B), gk(oe_cdiv(p.T2, 32 * ATT_WAVES), p.H, p.B);
    // each of the two kernels takes the LDS-plane form (attention_bf16.hip) where its resident axis is long enough; the
    // dQ kernel always runs first: it publishes delta for the dK/dV kernel
    const bool q_done = a->precision != 0 && oe_attn_planes_dq_try(p, a->precision, st) == 0;
#define ATT_BWD(TT)                                                                                       \
    do {                                                                                                  \
        if (!q_done) {                                                                                    \
            if (p.D <= 32) hipLaunchKernelGGL((attn_qtile_kernel<32, 1, TT, 1>), gq, dim3(ATT_GROUP), 0, st, p);  \
            else hipLaunchKernelGGL((attn_qtile_kernel<64, 1, TT, 1>), gq, dim3(ATT_GROUP), 0, st, p);    \
        }                                                                                                 \
        if (!(TT != 0 && oe_attn_planes_dkdv_try(p, TT, st) == 0)) {                                      \
            if (p.D <= 32) hipLaunchKernelGGL((attn_ktile_bwd_kernel<32, TT, 1>), gk, dim3(ATT_GROUP), 0, st, p); \
            else hipLaunchKernelGGL((attn_ktile_bwd_kernel<64, TT, 1>), gk, dim3(ATT_GROUP), 0, st, p);   \
        }                                                                                                 \
    } while (0)
    // precision 6: dQ and dK/dV on three planes where they fit (attention_bf16.hip); what does not fit on exact fp32 products
    static const bool dkdv6 = !(getenv("OE_ATTN_DKDV6") && atoi(getenv("OE_ATTN_DKDV6")) == 0);      // 0: A/B against the fp32 kernel
    if (a->precision == 6) {
        if (!q_done) {
            // few query tiles against a long key axis (the decoders' source attention: 31 queries, 248 keys, one block per (b, h)):
            // two wave groups split the keys, as the forward does
            const bool split = (long)gq.x * gq.y * gq.z <= 512 && p.T2 >= 128;
            if (split) {
                if (p.D <= 32) hipLaunchKernelGGL((attn_qtile_kernel<32, 1, 6, 2>), gq, dim3(ATT_GROUP * 2), 0, st, p);
                else hipLaunchKernelGGL((attn_qtile_kernel<64, 1, 6, 2>), gq, dim3(ATT_GROUP * 2), 0, st, p);
            } else {
                if (p.D <= 32) hipLaunchKernelGGL((attn_qtile_kernel<32, 1, 6, 1>), gq, dim3(ATT_GROUP), 0, st, p);
                else hipLaunchKernelGGL((attn_qtile_kernel<64, 1, 6, 1>), gq, dim3(ATT_GROUP), 0, st, p);
            }
        }
        if (!(dkdv6 && oe_attn_planes_dkdv_try(p, 6, st) == 0)) {
            if (p.D <= 32) hipLaunchKernelGGL((attn_ktile_bwd_kernel<32, 6, 1>), gk, dim3(ATT_GROUP), 0, st, p);
            else hipLaunchKernelGGL((attn_ktile_bwd_kernel<64, 6, 1>), gk, dim3(ATT_GROUP), 0, st, p);
        }
    } else if (a->precision == 3) ATT_BWD(3); else if (a->precision == 1) ATT_BWD(1); else ATT_BWD(0);
#undef ATT_BWD
    OE_LAUNCH_CHECK("oe_attention_bwd");
    return 0;
}
