// Convolution kernels of the path that are not GEMM-shaped:
//   conv1   Conv2d(1, C, 3, stride 2) + ReLU        (subsampling.py:77-78), NHWC output
//   col2im  gather form of the Conv2d(C,C,3,2) input gradient + ReLU mask
//   dwconv  GLU + depthwise Conv1d(K) of the Conformer conv module (convolution.py:104-107)
// All HBM-bound (a handful of MACs per byte); windows are staged in LDS.
#include <stdlib.h>
#include "oe_common.h"
#include "../../include/openeat_hip.h"

// ------------------------------------------------------------------ conv1 ----
// y[b,t,f,c] = relu(bias[c] + sum_{kh,kw} w[c][kh][kw] * x[b, 2t+kh, 2f+kw])
// thread = 4 consecutive channels of one output position; weights live in registers.
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int B, int T, int F, int T1, int F1,
                                                         int C, int pos_per_block, float* __restrict__ y,
                                                         __bf16* __restrict__ y_planes, long plane_stride) {
    const int cpt = C >> 2;                        // threads per position
    const int grp = threadIdx.x / cpt;             // position slot inside the block
    const int c4 = (threadIdx.x % cpt) * 4;
    const int ngrp = blockDim.x / cpt;
    float wr[4][9], br[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        br[e] = bias[c4 + e];
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[e][k] = w[(c4 + e) * 9 + k];
    }
    const long npos = (long)B * T1 * F1;
    const long p0 = (long)blockIdx.x * pos_per_block;
    for (long pos = p0 + grp; pos < min(npos, p0 + pos_per_block); pos += ngrp) {
        const int f = (int)(pos % F1);
        const int t = (int)((pos / F1) % T1);
        const long b = pos / ((long)F1 * T1);
        const float* xp = x + (b * T + 2 * t) * F + 2 * f;
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = xp[kh * F + kw];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = br[e];
#pragma unroll
            for (int k = 0; k < 9; ++k) s += wr[e][k] * xv[k];
            o[e] = fmaxf(s, 0.f);
        }
        const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
        if (y) *reinterpret_cast<float4*>(y + pos * C + c4) = o4;
        // precision 6: the activation also leaves as three bf16 planes (oe_common.h) - conv2's gather reads those, and a split
        // pass of its own would read these 636 MB back (config 2) to write them
        if (y_planes) store_planes4(y_planes + pos * C + c4, plane_stride, o4);
    }
}

// dw[c][k] += sum_pos dy[pos,c] * x[.., 2t+kh, 2f+kw] ; db[c] += sum_pos dy[pos,c]
// (dy is the gradient w.r.t. the pre-ReLU output, i.e. already masked by y>0.)
// The kernel streams dy once (636 MB at config 2) against 40 FMAs per float4: it is bound by how many loads a wave keeps in
// flight.  UB positions go through the loop together - their UB float4 loads of dy and 9 UB window taps are issued before
// the first FMA - and, with UNIFORM (C == 256: a wave is exactly one position group), the taps are wave-uniform scalar
// loads instead of nine vector loads of one address each (they cost as much issue time as the FMAs did).
template <int UB, bool UNIFORM>
__global__ __launch_bounds__(UNIFORM ? 512 : 256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int T,
                                                           int F, int T1, int F1, int C, int pos_per_block,
                                                           float* __restrict__ dw, float* __restrict__ db) {
    const int cpt = C >> 2;
    const int grp = UNIFORM ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : threadIdx.x / cpt;
    const int c4 = (threadIdx.x % cpt) * 4;
    const int ngrp = blockDim.x / cpt;
    float acc[4][10];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[e][k] = 0.f;
    const long npos = (long)B * T1 * F1;
    const long p0 = (long)blockIdx.x * pos_per_block;
    // (f, t, b) of this group's position advance by carries: two 64-bit divisions per position cost as much as its 40 FMAs
    const long pstart = p0 + grp, pend = min(npos, p0 + pos_per_block);
    int f = (int)(pstart % F1), t = (int)((pstart / F1) % T1);
    long b = pstart / ((long)F1 * T1);
    for (long pos = pstart; pos < pend; pos += (long)UB * ngrp) {
        float xv[UB][9];
        float4 g4[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long pu = pos + (long)u * ngrp;
            const bool ok = pu < pend;                                  // (wave-uniform with UNIFORM)
            const float* xp = x + (b * T + 2 * t) * F + 2 * f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) xv[u][kh * 3 + kw] = ok ? xp[kh * F + kw] : 0.f;
            g4[u] = ok ? *reinterpret_cast<const float4*>(dy + pu * C + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            f += ngrp;
            while (f >= F1) { f -= F1; if (++t == T1) { t = 0; ++b; } }
            if (b >= B) { b = B - 1; t = 0; f = 0; }                    // past the end: any valid address (the values are not used)
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float g[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[e][k] += g[e] * xv[u][k];
                acc[e][9] += g[e];
            }
        }
    }
    // reduce the position slots of this block through LDS, then one atomic per output per block
    extern __shared__ float sh[];                  // [ngrp][C][10]; UNIFORM: [C][10], the waves add into it (ds_add_f32)
    if (UNIFORM) {
        for (int i = threadIdx.x; i < C * 10; i += blockDim.x) sh[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 10; ++k) atomicAdd(&sh[(c4 + e) * 10 + k], acc[e][k]);
        __syncthreads();
        for (int i = threadIdx.x; i < C * 10; i += blockDim.x) {
            const int c = i / 10, k = i % 10;
            if (k < 9) atomicAdd(dw + c * 9 + k, sh[i]); else atomicAdd(db + c, sh[i]);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 10; ++k) sh[(grp * C + c4 + e) * 10 + k] = acc[e][k];
    __syncthreads();
    for (int i = threadIdx.x; i < C * 10; i += blockDim.x) {
        float s = 0.f;
        for (int gq = 0; gq < ngrp; ++gq) s += sh[gq * C * 10 + i];
        const int c = i / 10, k = i % 10;
        if (k < 9) atomicAdd(dw + c * 9 + k, s); else atomicAdd(db + c, s);
    }
}

static int conv1_block(int C) {
    const int cpt = C / 4;
    return cpt * (256 / cpt > 0 ? 256 / cpt : 1);
}

extern "C" int oe_conv1_fwd_pl(const float* x, const float* w, const float* bias, int B, int T, int F, int C, float* y,
                               void* y_planes, long plane_stride, void* stream);
extern "C" int oe_conv1_fwd(const float* x, const float* w, const float* bias, int B, int T, int F, int C, float* y,
                            void* stream) {
    return oe_conv1_fwd_pl(x, w, bias, B, T, F, C, y, nullptr, 0, stream);
}
extern "C" int oe_conv1_fwd_pl(const float* x, const float* w, const float* bias, int B, int T, int F, int C, float* y,
                               void* y_planes, long plane_stride, void* stream) {
    OE_REQUIRE(x && w && bias && (y || y_planes), "oe_conv1_fwd: null pointer");     // y may be NULL when the planes are the only copy wanted
    OE_REQUIRE(!y_planes || ((((uintptr_t)y_planes) & 7) == 0 && plane_stride % 4 == 0), "oe_conv1_fwd_pl: planes must be 8-byte aligned");
    OE_REQUIRE(B > 0 && T >= 3 && F >= 3 && C > 0 && C % 4 == 0 && C <= 1024, "oe_conv1_fwd: bad shape (C %% 4 == 0, C <= 1024)");
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const long npos = (long)B * T1 * F1;
    const int ppb = 64;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(oe_cdiv(npos, ppb)), dim3(conv1_block(C)), 0, (hipStream_t)stream, x, w, bias, B, T,
                       F, T1, F1, C, ppb, y, (__bf16*)y_planes, plane_stride);
    OE_LAUNCH_CHECK("conv1_fwd");
    return 0;
}

extern "C" int oe_conv1_wgrad(const float* x, const float* dy, int B, int T, int F, int C, float* dw, float* db, void* stream) {
    OE_REQUIRE(x && dy && dw && db, "oe_conv1_wgrad: null pointer");
    OE_REQUIRE(B > 0 && T >= 3 && F >= 3 && C > 0 && C % 4 == 0 && C <= 1024, "oe_conv1_wgrad: bad shape");
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const long npos = (long)B * T1 * F1;
    // C == 256: blocks of eight waves (a wave = one position group; the taps are scalar loads), 1024 positions per block - half
    // the block-end reductions and atomics of 256-thread blocks at the same number of waves (cold operands at config 2:
    // 252 us at 1024 positions per 256-thread block, 282 at 512, 423 at 128)
    static const int ppb_env = getenv("OE_CONV1_WGRAD_PPB") ? atoi(getenv("OE_CONV1_WGRAD_PPB")) : 0;     // tuning
    static const int variant = getenv("OE_CONV1_WGRAD") ? atoi(getenv("OE_CONV1_WGRAD")) : 0;             // tuning: 1 = the plain form
    const bool uni = (C == 256) && variant != 1;
    // ... and exactly two blocks per CU (512 blocks: 240 us; 607 blocks of 1024 positions: 257 us; the plain form: 300 us)
    const int ppb = ppb_env > 0 ? ppb_env : (uni ? (int)max(256L, (npos + 511) / 512) : 512);
    const int threads = uni ? 512 : conv1_block(C);
    const int ngrp = threads / (C / 4);
    const size_t lds = uni ? (size_t)C * 10 * sizeof(float) : (size_t)ngrp * C * 10 * sizeof(float);
#define C1W(UBB, UNI) hipLaunchKernelGGL((conv1_wgrad_kernel<UBB, UNI>), dim3(oe_cdiv(npos, ppb)), dim3(threads), lds, (hipStream_t)stream, x, dy, B, T, F, T1,  \
                                         F1, C, ppb, dw, db)
    if (uni && variant == 8) C1W(8, true);
    else if (uni) C1W(4, true);
    else if (variant == 1) C1W(1, false);
    else C1W(4, false);
#undef C1W
    OE_LAUNCH_CHECK("conv1_wgrad");
    return 0;
}

// ----------------------------------------------------------------- col2im ----
// dx[b,t1,f1,c] = (y1[b,t1,f1,c] > 0) * sum_{kh,kw : t=(t1-kh)/2, f=(f1-kw)/2 integral, in range}
//                 dcol[(b,t,f)][(kh*3+kw)*C + c]
__global__ __launch_bounds__(256) void col2im_relu_kernel(const float* __restrict__ dcol, const float* __restrict__ y1, int B,
                                                           int T1, int F1, int T2, int F2, int C, int KS, int S,
                                                           float* __restrict__ dx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
    const int cv = C >> 2;
    const long total = (long)B * T1 * F1 * cv;
    if (idx >= total) return;
    const int c = (int)(idx % cv) * 4;
    const long pos = idx / cv;
    const int f1 = (int)(pos % F1);
    const int t1 = (int)((pos / F1) % T1);
    const long b = pos / ((long)F1 * T1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KS; ++kh) {
        const int tt = t1 - kh;
        if (tt < 0 || (tt % S) || (tt / S) >= T2) continue;
        for (int kw = 0; kw < KS; ++kw) {
            const int ff = f1 - kw;
            if (ff < 0 || (ff % S) || (ff / S) >= F2) continue;
            const long m = (b * T2 + (tt / S)) * F2 + (ff / S);
            const float4 v = *reinterpret_cast<const float4*>(dcol + m * ((long)KS * KS * C) + (kh * KS + kw) * C + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    const float4 y = *reinterpret_cast<const float4*>(y1 + pos * C + c);
    s.x = y.x > 0.f ? s.x : 0.f; s.y = y.y > 0.f ? s.y : 0.f; s.z = y.z > 0.f ? s.z : 0.f; s.w = y.w > 0.f ? s.w : 0.f;
    *reinterpret_cast<float4*>(dx + pos * C + c) = s;
}

extern "C" int oe_col2im_relu_ks(const float* dcol, const float* y1, int B, int T1, int F1, int C, int KS, int S, float* dx,
                                 void* stream) {
    OE_REQUIRE(dcol && y1 && dx && B > 0 && KS >= 1 && S >= 1 && T1 >= KS && F1 >= KS && C > 0 && C % 4 == 0, "oe_col2im_relu: bad arguments");
    const int T2 = (T1 - KS) / S + 1, F2 = (F1 - KS) / S + 1;
    const long total = (long)B * T1 * F1 * (C / 4);
    hipLaunchKernelGGL(col2im_relu_kernel, dim3(oe_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dcol, y1, B, T1, F1, T2,
                       F2, C, KS, S, dx);
    OE_LAUNCH_CHECK("col2im_relu");
    return 0;
}
extern "C" int oe_col2im_relu(const float* dcol, const float* y1, int B, int T1, int F1, int C, float* dx, void* stream) {
    return oe_col2im_relu_ks(dcol, y1, B, T1, F1, C, 3, 2, dx, stream);
}

// Diagnostic build only (-DOE_GEMM_STAMPS): phase stamps of the depthwise-conv backward (tools/dwconv_stamps.py)
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* oe_dw_stamp_buf = nullptr;
extern "C" int oe_debug_set_dw_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(oe_dw_stamp_buf), &p, sizeof(p)); }
#define DW_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (oe_dw_stamp_buf && threadIdx.x == 0 && blockIdx.y < 8) oe_dw_stamp_buf[(blockIdx.y * 16 + (blockIdx.x & 15)) * 8 + (slot)] = t_; \
    } while (0)
#else
#define DW_STAMP(slot) do { } while (0)
#endif

// -------------------------------------------------------- GLU + depthwise ----
#define DW_TT 16
#define DW_MAXK 31      // largest kernel size; the kernels are instantiated for K <= 7, 15, 31 (per-tap loops are unrolled)
// y[b,t,c] = bias[c] + sum_k w[c][k] * g[b, t - pad_left + k, c],  g = a[:, :d] * sigmoid(a[:, d:])
// gpad (optional, [d]): value of g on the virtual frames t < 0.  The causal variant of the reference pads
// its input BEFORE the pointwise conv (convolution.py:92-93), so those frames carry GLU(pointwise bias).
// KT = compile-time bound of the kernel size; EXACT: K == KT, so the per-tap loops carry no runtime test (with a test
// every tap became a scalar branch and every LDS read its own wait)
// LN (z != nullptr): the LayerNorm + activation that follows the depthwise convolution (convolution.py:107-111) in the same
// launch - a block owns DW_TT whole rows of y, so it parks them in LDS and each wave normalises four of them exactly as
// layernorm_fwd_kernel does (same per-lane float4 order, same two-pass statistics: the same bits), writing z = act(LN(y)) and
// the (mean, rstd) pairs its backward reads.
template <int KT, bool EXACT>
__global__ __launch_bounds__(256) void dwconv_glu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ gpad,
                                                              int T, int d, int K, int pad_left, float* __restrict__ y,
                                                              const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps,
                                                              int ln_act, float* __restrict__ z, float* __restrict__ ln_stats) {
    extern __shared__ __attribute__((aligned(16))) float win[];       // [(DW_TT + K - 1)][d] (+ [DW_TT][d] with the fused LayerNorm)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * DW_TT;
    const int rows = DW_TT + K - 1;
    const int dv = d >> 2;
    // staging: four window elements per thread at a time, their global loads issued together (one at a time, each load's
    // HBM latency would be exposed in turn)
    for (int e0 = threadIdx.x; e0 < rows * dv; e0 += 256 * 4) {
        float4 xv[4], gv[4];
        int kind[4];                                   // 0: zero, 1: from a, 2: pad value
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            kind[u] = 0;
            xv[u] = gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < rows * dv) {
                const int r = e / dv, c = (e % dv) * 4;
                const int t = t0 - pad_left + r;
                if (t >= 0 && t < T) {
                    const float* ap = a + ((long)b * T + t) * 2 * d;
                    xv[u] = *reinterpret_cast<const float4*>(ap + c);
                    gv[u] = *reinterpret_cast<const float4*>(ap + d + c);
                    kind[u] = 1;
                } else if (t < 0 && gpad) {
                    xv[u] = *reinterpret_cast<const float4*>(gpad + c);
                    kind[u] = 2;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            if (e < rows * dv) {
                const int r = e / dv, c = (e % dv) * 4;
                float4 g = xv[u];
                if (kind[u] == 1)
                    g = make_float4(xv[u].x * sigmoidf_(gv[u].x), xv[u].y * sigmoidf_(gv[u].y), xv[u].z * sigmoidf_(gv[u].z), xv[u].w * sigmoidf_(gv[u].w));
                *reinterpret_cast<float4*>(win + r * d + c) = g;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        float wk[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) wk[k] = (EXACT || k < K) ? w[c * K + k] : 0.f;
        const float bc = bias[c];
#pragma unroll
        for (int tt = 0; tt < DW_TT; ++tt) {
            const int t = t0 + tt;
            float s = bc;
#pragma unroll
            for (int k = 0; k < KT; ++k) if (EXACT || k < K) s += wk[k] * win[(tt + k) * d + c];
            if (t < T) y[((long)b * T + t) * d + c] = s;
            if (z) win[(rows + tt) * d + c] = s;
        }
    }
    if (z == nullptr) return;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* g4 = reinterpret_cast<const float4*>(ln_gamma);
    const float4* b4 = reinterpret_cast<const float4*>(ln_beta);
    for (int tt = wave; tt < DW_TT; tt += 4) {
        const int t = t0 + tt;
        if (t >= T) break;                                           // wave-uniform
        const float4* yr = reinterpret_cast<const float4*>(win + (rows + tt) * d);
        float sum = 0.f;
        for (int i = lane; i < dv; i += 64) { const float4 v = yr[i]; sum += v.x + v.y + v.z + v.w; }
        const float mean = wave_sum(sum) / d;
        float q = 0.f;
        for (int i = lane; i < dv; i += 64) {
            const float4 v = yr[i];
            const float p0 = v.x - mean, p1 = v.y - mean, p2 = v.z - mean, p3 = v.w - mean;
            q += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
        }
        const float rstd = rsqrtf(wave_sum(q) / d + ln_eps);
        const long row = (long)b * T + t;
        if (lane == 0) { ln_stats[row * 2] = mean; ln_stats[row * 2 + 1] = rstd; }
        float4* zr = reinterpret_cast<float4*>(z + row * d);
        for (int i = lane; i < dv; i += 64) {
            const float4 v = yr[i], g = g4[i], bb = b4[i];
            zr[i] = make_float4(act_fwd(ln_act, (v.x - mean) * rstd * g.x + bb.x), act_fwd(ln_act, (v.y - mean) * rstd * g.y + bb.y),
                                act_fwd(ln_act, (v.z - mean) * rstd * g.z + bb.z), act_fwd(ln_act, (v.w - mean) * rstd * g.w + bb.w));
        }
    }
}

// da (B*T, 2d) = GLU'(a, dg),  dg[t,c] = sum_k w[c][k] * dy[t + pad_left - k, c]
// dw[c][k] += sum_t dy[t,c] * g[t - pad_left + k, c] ; db[c] += sum_t dy[t,c]
template <int KT, bool EXACT>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dy,
                                                              const float* __restrict__ w, const float* __restrict__ gpad,
                                                              int T, int d, int K, int pad_left,
                                                              float* __restrict__ da, float* __restrict__ partial,
                                                              float* __restrict__ dgpad) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int rows = DW_TT + K - 1;
    float* gwin = sh;                   // g rows  t0 - pad_left ..            (for dw)
    float* dwin = sh + rows * d;        // dy rows t0 + pad_left - (K-1) ..    (for dg)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * DW_TT;
    const int dv = d >> 2;
    DW_STAMP(0);
    // staging: eight window elements per thread at a time (the whole window at d = 256, K = 15), all their global
    // loads issued together
    for (int e0 = threadIdx.x; e0 < rows * dv; e0 += 256 * 8) {
        float4 xv[8], gv[8], qv[8];
        int kind[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            kind[u] = 0;
            xv[u] = gv[u] = qv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < rows * dv) {
                const int r = e / dv, c = (e % dv) * 4;
                const int tg = t0 - pad_left + r;
                if (tg >= 0 && tg < T) {
                    const float* ap = a + ((long)b * T + tg) * 2 * d;
                    xv[u] = *reinterpret_cast<const float4*>(ap + c);
                    gv[u] = *reinterpret_cast<const float4*>(ap + d + c);
                    kind[u] = 1;
                } else if (tg < 0 && gpad) {
                    xv[u] = *reinterpret_cast<const float4*>(gpad + c);
                    kind[u] = 2;
                }
                const int td = t0 + pad_left - (K - 1) + r;
                if (td >= 0 && td < T) qv[u] = *reinterpret_cast<const float4*>(dy + ((long)b * T + td) * d + c);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            if (e < rows * dv) {
                const int r = e / dv, c = (e % dv) * 4;
                float4 g = xv[u];
                if (kind[u] == 1)
                    g = make_float4(xv[u].x * sigmoidf_(gv[u].x), xv[u].y * sigmoidf_(gv[u].y), xv[u].z * sigmoidf_(gv[u].z), xv[u].w * sigmoidf_(gv[u].w));
                *reinterpret_cast<float4*>(gwin + r * d + c) = g;
                *reinterpret_cast<float4*>(dwin + r * d + c) = qv[u];
            }
        }
    }
    DW_STAMP(1);
    __syncthreads();
    DW_STAMP(2);
    for (int c = threadIdx.x; c < d; c += 256) {
        float wk[KT], dwk[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) { wk[k] = (EXACT || k < K) ? w[c * K + k] : 0.f; dwk[k] = 0.f; }
        float dbs = 0.f;
        // this channel's (x, gate) of the tile's frames, all loads in flight at once: inside the frame loop they would be
        // one dependent global round trip per frame (the loop body also stores)
        float xs[DW_TT], gs[DW_TT];
#pragma unroll
        for (int tt = 0; tt < DW_TT; ++tt) {
            const int t = min(t0 + tt, T - 1);
            const float* ap = a + ((long)b * T + t) * 2 * d;
            xs[tt] = ap[c];
            gs[tt] = ap[d + c];
        }
        DW_STAMP(3);
#pragma unroll
        for (int tt = 0; tt < DW_TT; ++tt) {
            const int t = t0 + tt;
            if (t < T) {                  // (a predicate, not a break: the loop must unroll so that xs / gs stay registers)
            // dy[t] sits at window row tt + (K-1) - pad_left ... relative to dwin start t0 + pad_left - (K-1):
            //   row(td) = td - (t0 + pad_left - (K-1));  td = t + pad_left - k  ->  row = tt + (K-1) - k
            float dg = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) if (EXACT || k < K) dg += wk[k] * dwin[(tt + (K - 1) - k) * d + c];
            // GLU backward at (t, c)
            const float xv = xs[tt], gv = gs[tt];
            const float sg = sigmoidf_(gv);
            float* dap = da + ((long)b * T + t) * 2 * d;
            dap[c] = dg * sg;
            dap[d + c] = dg * xv * sg * (1.f - sg);
            // parameter gradients: dy[t] is dwin row tt + (K-1) - pad_left ; g[t - pad_left + k] is gwin row tt + k
            const float dyt = dwin[(tt + (K - 1) - pad_left) * d + c];
            dbs += dyt;
#pragma unroll
            for (int k = 0; k < KT; ++k) if (EXACT || k < K) dwk[k] += dyt * gwin[(tt + k) * d + c];
            }
        }
        DW_STAMP(4);
        // per-block partials [blk][K+1][d] (coalesced over c); reduced in fixed order afterwards
        float* pp = partial + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * (K + 1)) * d + c;
#pragma unroll
        for (int k = 0; k < KT; ++k) if (EXACT || k < K) pp[(long)k * d] = dwk[k];
        pp[(long)K * d] = dbs;
        DW_STAMP(5);
        // gradient of the pad value: virtual frames tau in [-pad_left, -1] (first time tile only; its dy
        // window starts at frame pad_left-(K-1) <= 0):  dg[tau] = sum_k w[k] dy[tau + pad_left - k]
        if (dgpad && blockIdx.x == 0) {
            float acc = 0.f;
            const int td0 = pad_left - (K - 1);
            for (int tau = -pad_left; tau < 0; ++tau)
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int td = tau + pad_left - k;
                    if ((EXACT || k < K) && td >= 0 && td - td0 < rows) acc += wk[k] * dwin[(td - td0) * d + c];
                }
            atomicAdd(dgpad + c, acc);
        }
    }
}

// DW_PR_ROWS partial rows per block: 64 -> each thread has 16 loads in flight and an address sees B*T/16/64 adders
// (with 16 rows per block the launch was 2048 blocks of four loads per thread and was bound by its scattered atomics)
#define DW_PR_ROWS 64
__global__ __launch_bounds__(256) void dwconv_param_reduce_kernel(const float* __restrict__ partial, int nblocks, int d, int K,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cx;                     // (k, c), c fastest
    const int n = (K + 1) * d;
    const int b0 = blockIdx.y * DW_PR_ROWS, b1 = min(nblocks, b0 + DW_PR_ROWS);
    float s = 0.f;
    if (idx < n) {
#pragma unroll 4
        for (int b = b0 + ry; b < b1; b += 4) s += partial[(long)b * n + idx];
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && idx < n) {
        const float v = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
        const int k = idx / d, c = idx % d;
        if (k < K) atomicAdd(dw + c * K + k, v); else atomicAdd(db + c, v);
    }
}

extern "C" size_t oe_dwconv_glu_bwd_workspace_floats(int B, int T, int d, int K) {
    return (size_t)B * oe_cdiv(T, DW_TT) * (K + 1) * d;
}

extern "C" int oe_dwconv_glu_ln_fwd(const float* a, const float* w, const float* bias, const float* gpad, int B, int T, int d,
                                    int K, int causal, float* y, const float* ln_gamma, const float* ln_beta, float ln_eps, int ln_act,
                                    float* z, float* ln_stats, void* stream);
extern "C" int oe_dwconv_glu_fwd(const float* a, const float* w, const float* bias, const float* gpad, int B, int T, int d,
                                 int K, int causal, float* y, void* stream) {
    return oe_dwconv_glu_ln_fwd(a, w, bias, gpad, B, T, d, K, causal, y, nullptr, nullptr, 0.f, 0, nullptr, nullptr, stream);
}
// z (optional) = act(LayerNorm(y; ln_gamma, ln_beta, ln_eps)) and ln_stats = (mean, rstd) per row, from the same launch
extern "C" int oe_dwconv_glu_ln_fwd(const float* a, const float* w, const float* bias, const float* gpad, int B, int T, int d,
                                    int K, int causal, float* y, const float* ln_gamma, const float* ln_beta, float ln_eps, int ln_act,
                                    float* z, float* ln_stats, void* stream) {
    OE_REQUIRE(a && w && bias && y, "oe_dwconv_glu_fwd: null pointer");
    OE_REQUIRE(!z || (ln_gamma && ln_beta && ln_stats), "oe_dwconv_glu_ln_fwd: the fused LayerNorm needs gamma, beta and a statistics buffer");
    OE_REQUIRE(B > 0 && T > 0 && d > 0 && d % 4 == 0 && K >= 1 && K <= DW_MAXK, "oe_dwconv_glu_fwd: bad shape (d %% 4, K <= %d)", DW_MAXK);
    OE_REQUIRE(causal || (K % 2 == 1), "oe_dwconv_glu_fwd: kernel size must be odd for the symmetric convolution");
    const int pad_left = causal ? K - 1 : (K - 1) / 2;
    const size_t lds = (size_t)(DW_TT + K - 1 + (z ? DW_TT : 0)) * d * sizeof(float);
    OE_REQUIRE(lds <= 160 * 1024, "oe_dwconv_glu_fwd: window does not fit LDS (d=%d)", d);
#define DW_FWD(KT, EX)                                                                                                           \
    do {                                                                                                                     \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dwconv_glu_fwd_kernel<KT, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((dwconv_glu_fwd_kernel<KT, EX>), dim3(oe_cdiv(T, DW_TT), B), dim3(256), lds, (hipStream_t)stream, a, w, bias, gpad, T, d, \
                           K, pad_left, y, ln_gamma, ln_beta, ln_eps, ln_act, z, ln_stats);                                  \
    } while (0)
    if (K == 15) DW_FWD(15, true); else if (K == 7) DW_FWD(7, true); else if (K == 31) DW_FWD(31, true);
    else if (K < 7) DW_FWD(7, false); else if (K < 15) DW_FWD(15, false); else DW_FWD(31, false);
#undef DW_FWD
    OE_LAUNCH_CHECK("dwconv_glu_fwd");
    return 0;
}

extern "C" int oe_dwconv_glu_bwd(const float* a, const float* dy, const float* w, const float* gpad, int B, int T, int d, int K,
                                 int causal, float* da, float* dw, float* db, float* dgpad, float* workspace, void* stream) {
    OE_REQUIRE(a && dy && w && da && dw && db && workspace, "oe_dwconv_glu_bwd: null pointer");
    OE_REQUIRE(!dgpad || (K - 1 <= DW_TT + K - 1), "oe_dwconv_glu_bwd: pad window");
    OE_REQUIRE(B > 0 && T > 0 && d > 0 && d % 4 == 0 && K >= 1 && K <= DW_MAXK, "oe_dwconv_glu_bwd: bad shape");
    const int pad_left = causal ? K - 1 : (K - 1) / 2;
    const size_t lds = (size_t)2 * (DW_TT + K - 1) * d * sizeof(float);
    OE_REQUIRE(lds <= 160 * 1024, "oe_dwconv_glu_bwd: window does not fit LDS (d=%d)", d);
#define DW_BWD(KT, EX)                                                                                                           \
    do {                                                                                                                     \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dwconv_glu_bwd_kernel<KT, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((dwconv_glu_bwd_kernel<KT, EX>), dim3(oe_cdiv(T, DW_TT), B), dim3(256), lds, (hipStream_t)stream, a, dy, w, gpad, T, d, \
                           K, pad_left, da, workspace, dgpad);                                                               \
    } while (0)
    if (K == 15) DW_BWD(15, true); else if (K == 7) DW_BWD(7, true); else if (K == 31) DW_BWD(31, true);
    else if (K < 7) DW_BWD(7, false); else if (K < 15) DW_BWD(15, false); else DW_BWD(31, false);
#undef DW_BWD
    OE_LAUNCH_CHECK("dwconv_glu_bwd");
    hipLaunchKernelGGL(dwconv_param_reduce_kernel, dim3(oe_cdiv((K + 1) * d, 64), oe_cdiv(B * oe_cdiv(T, DW_TT), DW_PR_ROWS)), dim3(256), 0, (hipStream_t)stream, workspace,
                       B * oe_cdiv(T, DW_TT), d, K, dw, db);
    OE_LAUNCH_CHECK("dwconv_param_reduce");
    return 0;
}


// ---- helpers of the stride-2 3x3 input gradient (ops._conv_dgrad_k3s2) --------------------------------------------------
// (1) dy (B, To, Fo, C) -> dyp (B, To + 2, Fo + 2, C) with a border of zeros, one pass (was torch.zeros + a strided copy:
//     two passes over 170 MB at config 2).  C % 4 == 0.
__global__ __launch_bounds__(256) void pad1_nhwc_kernel(const float4* __restrict__ dy, int B, int To, int Fo, int C4, float4* __restrict__ out) {
    const long n = (long)B * (To + 2) * (Fo + 2) * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long q = i / C4;
        const int f = (int)(q % (Fo + 2)) - 1;
        q /= (Fo + 2);
        const int t = (int)(q % (To + 2)) - 1;
        const long b = q / (To + 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < To && f >= 0 && f < Fo) v = dy[((b * To + t) * Fo + f) * C4 + c];
        out[i] = v;
    }
}
extern "C" int oe_pad1_nhwc(const float* dy, int B, int To, int Fo, int C, float* out, void* stream) {
    OE_REQUIRE(dy && out && B > 0 && To > 0 && Fo > 0 && C > 0 && C % 4 == 0, "oe_pad1_nhwc: bad arguments (C must be a multiple of 4)");
    OE_REQUIRE((((uintptr_t)dy | (uintptr_t)out) & 15) == 0, "oe_pad1_nhwc: pointers must be 16-byte aligned");
    const long n = (long)B * (To + 2) * (Fo + 2) * (C / 4);
    hipLaunchKernelGGL(pad1_nhwc_kernel, dim3((unsigned)min((long)oe_cdiv(n, 256), 65536L)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, B, To, Fo, C / 4, (float4*)out);
    OE_LAUNCH_CHECK("pad1_nhwc");
    return 0;
}
// (1') the same padding straight into three bf16 planes (oe_common.h: p0 + p1 + p2 = x), `out` optional: with the four
//      parity-class GEMMs on pre-split operands nothing reads the padded fp32 tensor, and one pass (read dy, write planes)
//      replaces the pad pass and the split pass over it (config 2: 63 + 68 us, 172 MB less written and read back)
__global__ __launch_bounds__(256) void pad1_nhwc_planes_kernel(const float4* __restrict__ dy, int B, int To, int Fo, int C4, float4* __restrict__ out,
                                                                __bf16* __restrict__ planes, long pstride) {
    const long n = (long)B * (To + 2) * (Fo + 2) * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long q = i / C4;
        const int f = (int)(q % (Fo + 2)) - 1;
        q /= (Fo + 2);
        const int t = (int)(q % (To + 2)) - 1;
        const long b = q / (To + 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < To && f >= 0 && f < Fo) v = dy[((b * To + t) * Fo + f) * C4 + c];
        if (out) out[i] = v;
        store_planes4(planes + 4 * i, pstride, v);
    }
}
extern "C" int oe_pad1_nhwc_planes(const float* dy, int B, int To, int Fo, int C, float* out, void* planes, long plane_stride, void* stream) {
    OE_REQUIRE(dy && planes && B > 0 && To > 0 && Fo > 0 && C > 0 && C % 4 == 0, "oe_pad1_nhwc_planes: bad arguments (C must be a multiple of 4)");
    OE_REQUIRE((((uintptr_t)dy | (uintptr_t)out) & 15) == 0 && (((uintptr_t)planes) & 7) == 0 && plane_stride % 4 == 0,
               "oe_pad1_nhwc_planes: dy / out must be 16-byte, the planes 8-byte aligned");
    const long n = (long)B * (To + 2) * (Fo + 2) * (C / 4);
    hipLaunchKernelGGL(pad1_nhwc_planes_kernel, dim3((unsigned)min((long)oe_cdiv(n, 256), 65536L)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, B, To, Fo, C / 4, (float4*)out, (__bf16*)planes, plane_stride);
    OE_LAUNCH_CHECK("pad1_nhwc_planes");
    return 0;
}
// (2) the B operands of the four parity classes (t1 % 2, f1 % 2) from the OIHW weight w[co][ci][3][3], back to back:
//     class (pt, pf) uses taps khs = {2, 0} (pt = 0) or {1} (pt = 1), kws likewise; its operand is [ci][(window row, window
//     col, co)].  Offsets (floats): (0,0) 0, (0,1) 4 C^2, (1,0) 6 C^2, (1,1) 8 C^2; 9 C^2 in all.  (Was 10 stacks + 4 copies.)
__global__ __launch_bounds__(256) void conv_dgrad_k3s2_weights_kernel(const float* __restrict__ w, int C, float* __restrict__ out) {
    const long n = 9L * C * C;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long cc = (long)C * C;
    int cls, KH, KW;
    long base;
    if (i < 4 * cc) { cls = 0; KH = 2; KW = 2; base = 0; }
    else if (i < 6 * cc) { cls = 1; KH = 2; KW = 1; base = 4 * cc; }
    else if (i < 8 * cc) { cls = 2; KH = 1; KW = 2; base = 6 * cc; }
    else { cls = 3; KH = 1; KW = 1; base = 8 * cc; }
    const long r = i - base;                       // [ci][khi][kwi][co]
    const int co = (int)(r % C);
    long q = r / C;
    const int kwi = (int)(q % KW);
    q /= KW;
    const int khi = (int)(q % KH);
    const int ci = (int)(q / KH);
    const int kh = (cls >> 1) ? 1 : (khi == 0 ? 2 : 0);      // pt = cls >> 1, pf = cls & 1
    const int kw = (cls & 1) ? 1 : (kwi == 0 ? 2 : 0);
    out[i] = w[(((long)co * C + ci) * 3 + kh) * 3 + kw];
}
extern "C" int oe_conv_dgrad_k3s2_weights(const float* w, int C, float* out, void* stream) {
    OE_REQUIRE(w && out && C > 0, "oe_conv_dgrad_k3s2_weights: bad arguments");
    hipLaunchKernelGGL(conv_dgrad_k3s2_weights_kernel, dim3(oe_cdiv(9L * C * C, 256)), dim3(256), 0, (hipStream_t)stream, w, C, out);
    OE_LAUNCH_CHECK("conv_dgrad_k3s2_weights");
    return 0;
}
