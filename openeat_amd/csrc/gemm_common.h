// Shared pieces of the GEMM kernels (fp32-MFMA gemm.hip, bf16-MFMA gemm_bf16.hip):
// operand addressing (plain / conv2 im2col gather), bounded vector loads, the epilogue.
#pragma once
#include "oe_common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N).  Used where the body is too large
// for `#pragma unroll` to be honoured but must index register arrays statically.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


// Diagnostic build only (-DOE_GEMM_STAMPS, tools/gemm_stamps.py): per-workgroup s_memtime stamps at the
// phase boundaries, written to a buffer nothing else reads.  No stamp exists in the shipped library.
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* oe_stamp_buf = nullptr;   // per translation unit; only gemm_dma.hip sets its copy
#define OE_STAMP(slot)                                                                                     \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        unsigned long long t_;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (oe_stamp_buf && threadIdx.x == 0 && blockIdx.x < 4096) oe_stamp_buf[blockIdx.x * 8 + (slot)] = t_; \
    } while (0)
#else
#define OE_STAMP(slot) do { } while (0)
#endif

struct OperandDesc {
    const float* p;
    long ld;
    int vec_ok;     // 16-byte vector loads are legal (pointer/ld alignment)
    // conv im2col gather (NHWC input (B,T1,F1,C), KS x KS kernel, stride S, output (B,T2,F2)):
    int T1, F1, T2, F2, C, KS, S;
};

template <bool GATHER>
__device__ __forceinline__ long addr_row(const OperandDesc& d, long r) {
    if (!GATHER) return r * d.ld;
    int f = (int)(r % d.F2);
    long q = r / d.F2;
    int t = (int)(q % d.T2);
    long b = q / d.T2;
    return ((b * d.T1 + d.S * t) * (long)d.F1 + d.S * f) * d.C;
}
template <bool GATHER>
__device__ __forceinline__ long addr_col(const OperandDesc& d, long c) {
    if (!GATHER) return c;
    int seg = d.KS * d.C;
    int kh = (int)(c / seg);
    return (long)kh * d.F1 * d.C + (c - (long)kh * seg);
}

// Load 4 consecutive logical elements (along the contiguous direction) with
// bounds: n_valid in [0,4] elements are in range.
__device__ __forceinline__ float4 load4(const float* p, int n_valid, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_valid >= 4 && vec_ok) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (n_valid > 0) v.x = p[0];
        if (n_valid > 1) v.y = p[1];
        if (n_valid > 2) v.z = p[2];
        if (n_valid > 3) v.w = p[3];
    }
    return v;
}

struct EpiParams {
    float alpha;
    const float* alpha_dev;
    const float* bias;
    int act;
    float* preact_out;
    const float* actgrad_in;
    long ld_aux;
    float drop_p;
    unsigned long long seed;
    const unsigned long long* seed_dev;
    const unsigned char* rowmask;
    const float* residual;
    long ldr;
    int res_row_mod;
    float beta;
    int accumulate;
    int atomic;
    float* a_colsum;   // optional: out[m] += alpha * sum_k A(m,k) for a k-major A (bias gradient fused into wgrad)
    int scatter, sc_t1, sc_f1, sc_t2, sc_f2, sc_s;   // output row scatter (oe_gemm_args.out_scatter)
    int actgrad_bf16;  // actgrad_in points at bf16 values (plane 0 of the activation: its sign is the activation's), ld_aux in elements
    __bf16* c_planes;  // optional: the output ALSO as three bf16 planes (oe_common.h), for a consumer on gemm_pl.hip
    long c_pstride;    // elements between planes
    long ld_cp;        // row stride of a plane
};

// physical row of logical row r = (b, t, f) over (sc_t2, sc_f2) when the output (and the act-grad source) is a strided
// sub-grid of a (B, sc_t1, sc_f1) tensor
__device__ __forceinline__ long scatter_row(const EpiParams& ep, long r) {
    const int f = (int)(r % ep.sc_f2);
    const long q = r / ep.sc_f2;
    const int t = (int)(q % ep.sc_t2);
    const long b = q / ep.sc_t2;
    return (b * ep.sc_t1 + (long)ep.sc_s * t) * ep.sc_f1 + (long)ep.sc_s * f;
}
// the four rows row0, row0 + 8, row0 + 16, row0 + 24 of an epilogue pass: one division pair, then carries
__device__ __forceinline__ void pass_rows(const EpiParams& ep, long row0, long (&ro)[4]) {
    if (!ep.scatter) {
#pragma unroll
        for (int p = 0; p < 4; ++p) ro[p] = row0 + 8 * p;
        return;
    }
    int f = (int)(row0 % ep.sc_f2);
    long q = row0 / ep.sc_f2;
    int t = (int)(q % ep.sc_t2);
    long b = q / ep.sc_t2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ro[p] = (b * ep.sc_t1 + (long)ep.sc_s * t) * ep.sc_f1 + (long)ep.sc_s * f;
        f += 8;
        while (f >= ep.sc_f2) {
            f -= ep.sc_f2;
            if (++t == ep.sc_t2) { t = 0; ++b; }
        }
    }
}


int oe_gemm_bf16_dispatch(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk,
                          const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, int terms, hipStream_t st);

// LDS-DMA variant (gemm_dma.hip): returns 1 when the problem does not qualify, else the launch status
int oe_gemm_tn_planes_try(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk, const EpiParams& ep,
                          int terms, hipStream_t st);
int oe_gemm_dma_try(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk, const EpiParams& ep,
                    bool a_kmajor, bool b_kmajor, bool gather_b, int terms, int tile, hipStream_t st);
// pre-split operands (gemm_pl.hip): returns 1 when the problem does not qualify
int oe_gemm_hyb_try(const OperandDesc& A, const OperandDesc& B, const void* Bp, long b_pstride, float* C, long ldc, int M, int N, int K, int sk,
                    const EpiParams& ep, bool b_kmajor, hipStream_t st);
int oe_gemm_pl_try(const OperandDesc& A, const OperandDesc& B, const void* Ap, long a_pstride, const void* Bp, long b_pstride, float* C, long ldc,
                   int M, int N, int K, int sk, const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, hipStream_t st, int korder = 0,
                   bool k_padded = false);

// One output value: x = acc*alpha + bias -> (pre-activation kept) -> act fwd, or times act'(aux) in a
// backward GEMM -> dropout mask -> dead-row zeroing -> res + beta*x.
__device__ __forceinline__ float epi_value(const EpiParams& ep, float v, float alpha, float bias, float aux, float dm, bool row_dead,
                                           float res, float& pre) {
    float x = v * alpha + bias;
    pre = x;
    // only relu and swish are fused into GEMM epilogues (oe_gemm_f32 rejects the others; the host applies those
    // with oe_act_fwd / oe_act_grad) - the transcendental-heavy ones would cost every GEMM kernel registers
    if (ep.actgrad_in) x *= (ep.act == OE_ACT_RELU) ? (aux > 0.f ? 1.f : 0.f) : (ep.act == OE_ACT_SWISH) ? act_bwd(OE_ACT_SWISH, aux) : 1.f;
    else x = (ep.act == OE_ACT_RELU) ? fmaxf(x, 0.f) : (ep.act == OE_ACT_SWISH) ? x * sigmoidf_(x) : x;
    x *= dm;
    if (row_dead) x = 0.f;
    return res + ep.beta * x;
}

// Epilogue shared by the GEMM kernels.  `acc` holds TM x TN 32x32 accumulator tiles in the MFMA C/D layout
// (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); `lds` must provide 4*32*36 floats and all
// waves of the block must call this together (it synchronises).
//
// vmcnt counts loads and stores together, in order: a wait for a load issued after a store also waits for
// that store to be acknowledged, and hipcc waits vmcnt(0) at every use of a conditionally loaded value
// once a branch separates load and use.  So the paths below keep their loads out of branches that also
// hold stores: blocks that are interior and 16-byte aligned take straight-line code with every auxiliary
// load of a 32x32 tile issued ahead of the tile's stores; only ragged-edge blocks take the bounds-checked path.
// four consecutive values of the act-grad source at element offset `off`: fp32, or bf16 (ep.actgrad_bf16: one 8-byte load)
__device__ __forceinline__ float4 load_aux4(const EpiParams& ep, long off) {
    if (!ep.actgrad_bf16) return *reinterpret_cast<const float4*>(ep.actgrad_in + off);
    const oe_bf16x4 h = *reinterpret_cast<const oe_bf16x4*>(reinterpret_cast<const __bf16*>(ep.actgrad_in) + off);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ float load_aux1(const EpiParams& ep, long off) {
    return ep.actgrad_bf16 ? (float)reinterpret_cast<const __bf16*>(ep.actgrad_in)[off] : ep.actgrad_in[off];
}
// WM = waves along M (2: the 256-thread kernels; 4: gemm_pl.hip's 512-thread blocks, whose lds must then hold 8 patches);
// two waves along N always: the block tile is (32 TM WM) x (64 TN).
template <int TM, int TN, int WM = 2>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], float* lds, float* __restrict__ C, long ldc, int M, int N,
                                              long m0, long n0, const EpiParams& ep, int tile_z) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lk = lane >> 5;
    float alpha = ep.alpha;
    if (ep.alpha_dev) alpha *= *ep.alpha_dev;
    const DropParams dpar = drop_params(ep.drop_p);
    const unsigned long long seed = ep.seed + (ep.seed_dev ? *ep.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const bool first_split = (tile_z == 0);
    const bool interior = (m0 + 32 * TM * WM <= M) && (n0 + 64 * TN <= N);     // block-uniform
    constexpr int EP_LD = 36;
    if (ep.atomic) {
        // split-K accumulation: atomics straight from the accumulators.  Register r of a 32x32 tile is two
        // full 128-byte row segments per wave-instruction (lanes 0-31 / 32-63) - the shape float atomics
        // run at full rate with; going through the row-major patch would issue 32-byte fragments.
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long col = n0 + wn * (32 * TN) + j * 32 + lrow;
                float bias = (col < N && ep.bias && first_split) ? ep.bias[col] : 0.f;
                asm volatile("" : "+v"(bias));      // resolve the load here, not in front of every atomic
                float* base = C + (m0 + wm * (32 * TM) + i * 32 + 4 * lk) * ldc + col;
                if (interior) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) atomicAdd(base + ((r & 3) + 8 * (r >> 2)) * ldc, acc[i][j][r] * alpha + bias);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = m0 + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (row < M && col < N) atomicAdd(C + row * ldc + col, acc[i][j][r] * alpha + bias);
                    }
                }
            }
        return;
    }
    // Plain outputs (alpha, bias, activation, optional pre-activation copy; nothing to load, no dropout): store straight
    // from the accumulators.  Register r of a 32x32 tile is two full 128-byte row segments per wave-instruction, so the
    // LDS round trip of the general path buys nothing here.
    if (interior && !ep.scatter && !ep.accumulate && !ep.actgrad_in && !ep.residual && !ep.rowmask && ep.drop_p <= 0.f && ep.beta == 1.f && !ep.c_planes) {
        static_for<0, TM * TN>([&](auto tile_idx) {
            constexpr int i = decltype(tile_idx)::value / TN, j = decltype(tile_idx)::value % TN;
            const long col = n0 + wn * (32 * TN) + j * 32 + lrow;
            const float bias = (ep.bias && first_split) ? ep.bias[col] : 0.f;
            const long rbase = m0 + wm * (32 * TM) + i * 32 + 4 * lk;
            float* cbase = C + rbase * ldc + col;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * alpha + bias;
            if (ep.preact_out) {
                float* pbase = ep.preact_out + rbase * ep.ld_aux + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) pbase[((r & 3) + 8 * (r >> 2)) * ep.ld_aux] = v[r];
            }
            if (ep.act == OE_ACT_SWISH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= sigmoidf_(v[r]);
            } else if (ep.act == OE_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
        });
        return;
    }

    // Each wave parks one 32x32 accumulator tile at a time in its own LDS patch and re-reads it row-major,
    // so that every global access of the epilogue (C, residual, pre-activation, act-grad input) is a
    // coalesced float4 row segment.
    float* patch = lds + wave * (32 * EP_LD);
    __syncthreads();   // every wave is done with the operand tiles that the patches overlay
    OE_STAMP(5);
    const bool c_vec = (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0);
    const bool aux_vec = (ep.ld_aux % 4 == 0) && (((uintptr_t)ep.preact_out & 15) == 0) && (((uintptr_t)ep.actgrad_in & 15) == 0);
    const bool res_vec = (ep.ldr % 4 == 0) && (((uintptr_t)ep.residual & 15) == 0);
    const int prow = lane >> 3, pcol = (lane & 7) * 4;       // this lane's (row within a pass, first column) of the patch

    if (interior && c_vec && aux_vec && res_vec && !ep.accumulate) {
        // auxiliary inputs (act-grad source, residual, row mask) are double-buffered across the wave's tiles: tile t+1's
        // loads are issued while tile t is still in its LDS round trip, so their HBM latency hides behind tile t's
        // math and stores instead of standing in front of tile t+1's
        float4 auxb[2][4], resb[2][4];
        bool deadb[2][4];
        auto issue_loads = [&](int i, int j, float4 (&aux)[4], float4 (&res)[4], bool (&dead)[4]) {
            const long row0 = m0 + wm * (32 * TM) + i * 32 + prow;
            const long col = n0 + wn * (32 * TN) + j * 32 + pcol;
#pragma unroll
            for (int p = 0; p < 4; ++p) { aux[p] = make_float4(0.f, 0.f, 0.f, 0.f); res[p] = aux[p]; dead[p] = false; }
            if (ep.actgrad_in) {
                long ro[4];
                pass_rows(ep, row0, ro);
#pragma unroll
                for (int p = 0; p < 4; ++p) aux[p] = load_aux4(ep, ro[p] * ep.ld_aux + col);
            }
            if (ep.residual) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const long rr = ep.res_row_mod > 0 ? (row0 + 8 * p) % ep.res_row_mod : row0 + 8 * p;
                    res[p] = *reinterpret_cast<const float4*>(ep.residual + rr * ep.ldr + col);
                }
            }
            if (ep.rowmask) {
#pragma unroll
                for (int p = 0; p < 4; ++p) dead[p] = !ep.rowmask[row0 + 8 * p];
            }
        };
        issue_loads(0, 0, auxb[0], resb[0], deadb[0]);
        static_for<0, TM * TN>([&](auto tile_idx) {
            {
                constexpr int tix = decltype(tile_idx)::value;
                constexpr int i = tix / TN, j = tix % TN;
                const long row0 = m0 + wm * (32 * TM) + i * 32 + prow;
                const long col = n0 + wn * (32 * TN) + j * 32 + pcol;
                float4 (&aux)[4] = auxb[tix & 1];
                float4 (&res)[4] = resb[tix & 1];
                bool (&dead)[4] = deadb[tix & 1];
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ep.bias && first_split) { b4.x = ep.bias[col]; b4.y = ep.bias[col + 1]; b4.z = ep.bias[col + 2]; b4.w = ep.bias[col + 3]; }
                if constexpr (tix + 1 < TM * TN) issue_loads((tix + 1) / TN, (tix + 1) % TN, auxb[(tix + 1) & 1], resb[(tix + 1) & 1], deadb[(tix + 1) & 1]);
                // ---- accumulators -> patch -> row-major float4 (wave-private: a wave-level fence is enough)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * EP_LD + lrow] = acc[i][j][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                float4 t[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) t[p] = *reinterpret_cast<const float4*>(patch + (p * 8 + prow) * EP_LD + pcol);
                if (i == 0 && j == 0) OE_STAMP(6);
                // ---- values: every feature test is one wave-uniform branch per tile, never per element
                float4 x[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    x[p] = make_float4(t[p].x * alpha + b4.x, t[p].y * alpha + b4.y, t[p].z * alpha + b4.z, t[p].w * alpha + b4.w);
                float* cdst = C + row0 * ldc + col;
                if (ep.preact_out) {
                    float* pdst = ep.preact_out + row0 * ep.ld_aux + col;
#pragma unroll
                    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(pdst + 8 * p * ep.ld_aux) = x[p];
                }
                if (ep.actgrad_in) {
                    if (ep.act == OE_ACT_SWISH) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            x[p].x *= act_bwd(OE_ACT_SWISH, aux[p].x); x[p].y *= act_bwd(OE_ACT_SWISH, aux[p].y);
                            x[p].z *= act_bwd(OE_ACT_SWISH, aux[p].z); x[p].w *= act_bwd(OE_ACT_SWISH, aux[p].w);
                        }
                    } else if (ep.act == OE_ACT_RELU) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            x[p].x = aux[p].x > 0.f ? x[p].x : 0.f; x[p].y = aux[p].y > 0.f ? x[p].y : 0.f;
                            x[p].z = aux[p].z > 0.f ? x[p].z : 0.f; x[p].w = aux[p].w > 0.f ? x[p].w : 0.f;
                        }
                    }
                } else if (ep.act == OE_ACT_SWISH) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        x[p].x *= sigmoidf_(x[p].x); x[p].y *= sigmoidf_(x[p].y); x[p].z *= sigmoidf_(x[p].z); x[p].w *= sigmoidf_(x[p].w);
                    }
                } else if (ep.act == OE_ACT_RELU) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        x[p].x = fmaxf(x[p].x, 0.f); x[p].y = fmaxf(x[p].y, 0.f); x[p].z = fmaxf(x[p].z, 0.f); x[p].w = fmaxf(x[p].w, 0.f);
                    }
                }
                if (ep.drop_p > 0.f) {
                    const unsigned long long e0 = (unsigned long long)(row0 * N + col);
                    if ((N & 7) == 0) {
                        // lanes 2i / 2i+1 hold the two halves of an 8-element block in every pass: the even lane draws the
                        // calls of passes 0 and 2, the odd lane those of passes 1 and 3, and they swap the other half
                        const int odd = lane & 1;
                        const unsigned long long b0 = (e0 - 4 * odd) + (unsigned long long)(8 * odd) * N;     // block of pass `odd`
                        const uint4 ca = drop_words8(seed, b0 >> 3), cb = drop_words8(seed, (b0 + (unsigned long long)16 * N) >> 3);
                        const unsigned r0 = __shfl_xor(odd ? ca.x : ca.z, 1, 64), r1 = __shfl_xor(odd ? ca.y : ca.w, 1, 64);
                        const unsigned r2 = __shfl_xor(odd ? cb.x : cb.z, 1, 64), r3 = __shfl_xor(odd ? cb.y : cb.w, 1, 64);
                        unsigned w[4][2];                                   // [pass][word of this lane's half]
                        w[0][0] = odd ? r0 : ca.x; w[0][1] = odd ? r1 : ca.y;
                        w[1][0] = odd ? ca.z : r0; w[1][1] = odd ? ca.w : r1;
                        w[2][0] = odd ? r2 : cb.x; w[2][1] = odd ? r3 : cb.y;
                        w[3][0] = odd ? cb.z : r2; w[3][1] = odd ? cb.w : r3;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            x[p].x *= drop_field(w[p][0], 0, dpar); x[p].y *= drop_field(w[p][0], 1, dpar);
                            x[p].z *= drop_field(w[p][1], 0, dpar); x[p].w *= drop_field(w[p][1], 1, dpar);
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const unsigned long long e = e0 + (unsigned long long)(8 * p) * N;
                            x[p].x *= drop_elem(seed, e, dpar); x[p].y *= drop_elem(seed, e + 1, dpar);
                            x[p].z *= drop_elem(seed, e + 2, dpar); x[p].w *= drop_elem(seed, e + 3, dpar);
                        }
                    }
                }
                if (ep.rowmask) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) if (dead[p]) x[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (ep.residual || ep.beta != 1.f) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        x[p] = make_float4(res[p].x + ep.beta * x[p].x, res[p].y + ep.beta * x[p].y, res[p].z + ep.beta * x[p].z, res[p].w + ep.beta * x[p].w);
                }
                if (ep.scatter) {
                    long ro[4];
                    pass_rows(ep, row0, ro);
#pragma unroll
                    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(C + ro[p] * ldc + col) = x[p];
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(cdst + 8 * p * ldc) = x[p];
                }
                if (ep.c_planes) {
                    __bf16* pd = ep.c_planes + row0 * ep.ld_cp + col;
#pragma unroll
                    for (int p = 0; p < 4; ++p) store_planes4(pd + 8 * p * ep.ld_cp, ep.c_pstride, x[p]);
                }
                if (i == 0 && j == 0) OE_STAMP(7);
            }
        });
        return;
    }

    // ragged-edge / unaligned blocks: bounds-checked
    static_for<0, TM * TN>([&](auto tile_idx) {
        {
            constexpr int i = decltype(tile_idx)::value / TN, j = decltype(tile_idx)::value % TN;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * EP_LD + lrow] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const long row_base = m0 + wm * (32 * TM) + i * 32;
            const long col = n0 + wn * (32 * TN) + j * 32 + pcol;
            const int ncol = (int)max(0L, min(4L, (long)N - col));
            float bias4[4] = {0.f, 0.f, 0.f, 0.f};
            if (ep.bias && first_split) for (int e = 0; e < ncol; ++e) bias4[e] = ep.bias[col + e];
            for (int pass = 0; pass < 4; ++pass) {
                const int lr = pass * 8 + prow;
                const long row = row_base + lr;
                if (row >= M || ncol == 0) continue;
                const float4 t4 = *reinterpret_cast<const float4*>(patch + lr * EP_LD + pcol);
                float v[4] = {t4.x, t4.y, t4.z, t4.w};
                const bool full = (ncol == 4);
                float aux[4] = {0.f, 0.f, 0.f, 0.f}, res[4] = {0.f, 0.f, 0.f, 0.f};
                const long orow = ep.scatter ? scatter_row(ep, row) : row;      // where this row lives in C / the act-grad source
                if (ep.actgrad_in) {
                    const long aoff = orow * ep.ld_aux + col;
                    if (full && aux_vec) { float4 a4 = load_aux4(ep, aoff); aux[0] = a4.x; aux[1] = a4.y; aux[2] = a4.z; aux[3] = a4.w; }
                    else for (int e = 0; e < ncol; ++e) aux[e] = load_aux1(ep, aoff + e);
                }
                if (ep.residual) {
                    const float* rp = ep.residual + (ep.res_row_mod > 0 ? row % ep.res_row_mod : row) * ep.ldr + col;
                    if (full && res_vec) { float4 r4 = *reinterpret_cast<const float4*>(rp); res[0] = r4.x; res[1] = r4.y; res[2] = r4.z; res[3] = r4.w; }
                    else for (int e = 0; e < ncol; ++e) res[e] = rp[e];
                }
                const bool row_dead = ep.rowmask && !ep.rowmask[row];
                float pre[4];
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (ep.drop_p > 0.f) {
                    const unsigned long long e0 = (unsigned long long)(row * N + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dm[e] = drop_elem(seed, e0 + e, dpar);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = epi_value(ep, v[e], alpha, bias4[e], aux[e], dm[e], row_dead, res[e], pre[e]);
                if (ep.preact_out) {
                    float* pp = ep.preact_out + row * ep.ld_aux + col;
                    if (full && aux_vec) *reinterpret_cast<float4*>(pp) = make_float4(pre[0], pre[1], pre[2], pre[3]);
                    else for (int e = 0; e < ncol; ++e) pp[e] = pre[e];
                }
                float* dst = C + orow * ldc + col;
                if (full && c_vec) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (ep.accumulate) { const float4 c4 = *reinterpret_cast<const float4*>(dst); o.x += c4.x; o.y += c4.y; o.z += c4.z; o.w += c4.w; }
                    *reinterpret_cast<float4*>(dst) = o;
                } else {
                    for (int e = 0; e < ncol; ++e) dst[e] = ep.accumulate ? dst[e] + v[e] : v[e];
                }
                if (ep.c_planes) {
                    for (int e = 0; e < ncol; ++e) store_planes1(ep.c_planes + row * ep.ld_cp + col + e, ep.c_pstride, v[e]);
                }
            }
        }
    });
}
