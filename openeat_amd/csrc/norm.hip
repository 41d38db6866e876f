// LayerNorm forward / backward (wave per row) and column sums.
// HBM-bound: forward reads x once and writes y once (2*rows*d*4 bytes);
// backward reads dy and x, writes dx (3*rows*d*4 bytes).
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define LN_MAXV 8   // float4 per lane -> d <= 2048

// PAIR: a second LayerNorm (gamma2, beta2, eps2) applied to the first one's output in the same pass - the back-to-back
// norms at an encoder layer boundary (encoder_layer.py:109-110 `x = norm_final(x)`, then the next layer's :79-80
// `x = norm_ff_macaron(x)`, or encoder.py's after_norm behind the last layer): y2 = LN2(LN1(x)), stats2 = LN2's
// (mean, rstd); y (the first norm's output, the next block's residual) is optional then.
template <int NV, bool PAIR>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int rows, int d,
                                                             const unsigned char* __restrict__ rowmask, int act,
                                                             float* __restrict__ y, float* __restrict__ stats,
                                                             __bf16* __restrict__ ypl, long pstride,
                                                             const float* __restrict__ gamma2, const float* __restrict__ beta2, float eps2,
                                                             float* __restrict__ y2, float* __restrict__ stats2) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = d >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * d);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) { v[j] = xr[i]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
    }
    const float mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
            q += a * a + b * b + c * c + e * e;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / d + eps);
    if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    const bool dead = rowmask && !rowmask[row];
    float4* yr = reinterpret_cast<float4*>(y + row * d);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!dead) {
                const float4 g = g4[i], bb = b4[i];
                o.x = act_fwd(act, (v[j].x - mean) * rstd * g.x + bb.x);
                o.y = act_fwd(act, (v[j].y - mean) * rstd * g.y + bb.y);
                o.z = act_fwd(act, (v[j].z - mean) * rstd * g.z + bb.z);
                o.w = act_fwd(act, (v[j].w - mean) * rstd * g.w + bb.w);
            }
            if (!PAIR || y) yr[i] = o;
            if (ypl) store_planes4(ypl + row * d + 4 * i, pstride, o);       // the output as bf16 planes too (gemm_pl.hip operand)
            if (PAIR) v[j] = o;
        }
    }
    if (PAIR) {
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) s2 += v[j].x + v[j].y + v[j].z + v[j].w;
        }
        const float mean2 = wave_sum(s2) / d;
        float q2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                float a = v[j].x - mean2, b = v[j].y - mean2, c = v[j].z - mean2, e = v[j].w - mean2;
                q2 += a * a + b * b + c * c + e * e;
            }
        }
        const float rstd2 = rsqrtf(wave_sum(q2) / d + eps2);
        if (lane == 0) { stats2[row * 2] = mean2; stats2[row * 2 + 1] = rstd2; }
        float4* y2r = reinterpret_cast<float4*>(y2 + row * d);
        const float4* g24 = reinterpret_cast<const float4*>(gamma2);
        const float4* b24 = reinterpret_cast<const float4*>(beta2);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const float4 g = g24[i], bb = b24[i];
                y2r[i] = make_float4((v[j].x - mean2) * rstd2 * g.x + bb.x, (v[j].y - mean2) * rstd2 * g.y + bb.y,
                                     (v[j].z - mean2) * rstd2 * g.z + bb.z, (v[j].w - mean2) * rstd2 * g.w + bb.w);
            }
        }
    }
}

#define LNB_ROWS 16   // rows per block in backward (4 per wave, all loaded before the first is reduced)

// Backward is a latency chain per row (load dy/x -> two wave reductions -> store); with one row in flight per
// wave the 3*rows*d*4 bytes move at a fraction of the HBM rate.  Each wave therefore issues the loads of its
// RB rows back to back and only then reduces them one by one.
// PAIR: backward of y2 = LN2(u), u = LN1(x) in one pass (the forward's PAIR): dy is the gradient of y2, `add` the gradient
// that reaches u on its other path (the next block's residual), u is recomputed from x and LN1's statistics; the row first
// goes through LN2's backward (partials of its parameter gradients to partial2), the result through LN1's (act = 0, no mask).
template <int NV, int RB, bool PAIR>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                             const float* __restrict__ stats,
                                                             int rows, int d, const unsigned char* __restrict__ rowmask,
                                                             const float* add, float* dx,
                                                             float* __restrict__ partial, float* __restrict__ gout, float g_alpha,
                                                             float g_p, unsigned long long g_seed,
                                                             const unsigned long long* __restrict__ g_seed_dev,
                                                             const unsigned char* __restrict__ g_rowmask,
                                                             __bf16* __restrict__ opl, long opl_stride,
                                                             const float* __restrict__ gamma2, const float* __restrict__ stats2,
                                                             float* __restrict__ partial2) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [4 waves][2][d]
    // optional second output gout = g_alpha * dropout_mask(g_seed) * dx (oe_dropout_scale's definition: element idx belongs
    // to Philox call idx >> 3): the gradient the PREVIOUS block's backward starts from, i.e. its `residual + dropout(.)`
    // output dropout applied to this kernel's dx - saves that block a separate elementwise launch
    const DropParams g_dpar = drop_params(g_p);
    const unsigned long long g_seed_eff = g_seed + (g_seed_dev ? *g_seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 gam[NV], bet[NV], dg[NV], db[NV];
    float4 gam2[PAIR ? NV : 1], dg2[PAIR ? NV : 1], db2[PAIR ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        gam[j] = (i < nv) ? g4[i] : make_float4(0, 0, 0, 0);
        bet[j] = (i < nv && (act || PAIR)) ? b4[i] : make_float4(0, 0, 0, 0);
        dg[j] = make_float4(0, 0, 0, 0);
        db[j] = make_float4(0, 0, 0, 0);
        if (PAIR) {
            gam2[j] = (i < nv) ? reinterpret_cast<const float4*>(gamma2)[i] : make_float4(0, 0, 0, 0);
            dg2[j] = make_float4(0, 0, 0, 0);
            db2[j] = make_float4(0, 0, 0, 0);
        }
    }
    const long w0 = (long)blockIdx.x * LNB_ROWS + wave * (LNB_ROWS / 4);
    for (int rb = 0; rb < LNB_ROWS / 4; rb += RB) {
        // ---- loads of RB rows (rows past the end re-read the last row; nothing of them is used or stored)
        float4 dyv[RB][NV], xv[RB][NV], av[RB][NV];
        float mean[RB], rstd[RB];
        bool live[RB], valid[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const long row = w0 + rb + k;
            valid[k] = row < rows;
            const long rc = valid[k] ? row : rows - 1;
            live[k] = valid[k] && !(rowmask && !rowmask[rc]);
            mean[k] = stats[rc * 2];
            rstd[k] = stats[rc * 2 + 1];
            const float4* dyr = reinterpret_cast<const float4*>(dy + rc * d);
            const float4* xr = reinterpret_cast<const float4*>(x + rc * d);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + 64 * j;
                const int ic = i < nv ? i : 0;
                dyv[k][j] = dyr[ic];
                xv[k][j] = xr[ic];
                av[k][j] = add ? reinterpret_cast<const float4*>(add + rc * d)[ic] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // ---- one row at a time: reductions, dx, parameter-gradient partials
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            if (PAIR) {
                // LN2's backward on this row first: u = LN1(x) recomputed, dyv <- add + LN2'(dy); `add` is used up
                const long rc = min(w0 + rb + k, (long)rows - 1);
                const float mean2 = stats2[rc * 2], rstd2 = stats2[rc * 2 + 1];
                float4 g2[NV], xh2[NV];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + 64 * j;
                    const bool on = live[k] && i < nv;
                    const float4 v = xv[k][j];
                    const float4 u = make_float4((v.x - mean[k]) * rstd[k] * gam[j].x + bet[j].x, (v.y - mean[k]) * rstd[k] * gam[j].y + bet[j].y,
                                                 (v.z - mean[k]) * rstd[k] * gam[j].z + bet[j].z, (v.w - mean[k]) * rstd[k] * gam[j].w + bet[j].w);
                    xh2[j] = make_float4((u.x - mean2) * rstd2, (u.y - mean2) * rstd2, (u.z - mean2) * rstd2, (u.w - mean2) * rstd2);
                    float4 t = dyv[k][j];
                    if (!on) { t = make_float4(0.f, 0.f, 0.f, 0.f); xh2[j] = t; }
                    g2[j] = make_float4(t.x * gam2[j].x, t.y * gam2[j].y, t.z * gam2[j].z, t.w * gam2[j].w);
                    s1 += g2[j].x + g2[j].y + g2[j].z + g2[j].w;
                    s2 += g2[j].x * xh2[j].x + g2[j].y * xh2[j].y + g2[j].z * xh2[j].z + g2[j].w * xh2[j].w;
                    dg2[j].x += t.x * xh2[j].x; dg2[j].y += t.y * xh2[j].y; dg2[j].z += t.z * xh2[j].z; dg2[j].w += t.w * xh2[j].w;
                    db2[j].x += t.x; db2[j].y += t.y; db2[j].z += t.z; db2[j].w += t.w;
                }
                const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float4 o = av[k][j];
                    o.x += rstd2 * (g2[j].x - c1 - xh2[j].x * c2);
                    o.y += rstd2 * (g2[j].y - c1 - xh2[j].y * c2);
                    o.z += rstd2 * (g2[j].z - c1 - xh2[j].z * c2);
                    o.w += rstd2 * (g2[j].w - c1 - xh2[j].w * c2);
                    dyv[k][j] = o;
                    av[k][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            float4* dxr = reinterpret_cast<float4*>(dx + (w0 + rb + k) * d);
            float4 g[NV], xh[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + 64 * j;
                const bool on = live[k] && i < nv;
                float4 t = dyv[k][j];
                const float4 v = xv[k][j];
                xh[j] = make_float4((v.x - mean[k]) * rstd[k], (v.y - mean[k]) * rstd[k], (v.z - mean[k]) * rstd[k], (v.w - mean[k]) * rstd[k]);
                if (act) {   // y = act(ln): chain through the activation at the recomputed pre-activation
                    t.x *= act_bwd(act, xh[j].x * gam[j].x + bet[j].x); t.y *= act_bwd(act, xh[j].y * gam[j].y + bet[j].y);
                    t.z *= act_bwd(act, xh[j].z * gam[j].z + bet[j].z); t.w *= act_bwd(act, xh[j].w * gam[j].w + bet[j].w);
                }
                if (!on) { t = make_float4(0.f, 0.f, 0.f, 0.f); xh[j] = t; }      // masked rows may hold anything
                g[j] = make_float4(t.x * gam[j].x, t.y * gam[j].y, t.z * gam[j].z, t.w * gam[j].w);
                s1 += g[j].x + g[j].y + g[j].z + g[j].w;
                s2 += g[j].x * xh[j].x + g[j].y * xh[j].y + g[j].z * xh[j].z + g[j].w * xh[j].w;
                dg[j].x += t.x * xh[j].x; dg[j].y += t.y * xh[j].y; dg[j].z += t.z * xh[j].z; dg[j].w += t.w * xh[j].w;
                db[j].x += t.x; db[j].y += t.y; db[j].z += t.z; db[j].w += t.w;
            }
            const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
            if (valid[k]) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + 64 * j;
                    if (i < nv) {
                        float4 o = av[k][j];
                        if (live[k]) {
                            o.x += rstd[k] * (g[j].x - c1 - xh[j].x * c2);
                            o.y += rstd[k] * (g[j].y - c1 - xh[j].y * c2);
                            o.z += rstd[k] * (g[j].z - c1 - xh[j].z * c2);
                            o.w += rstd[k] * (g[j].w - c1 - xh[j].w * c2);
                        }
                        dxr[i] = o;
                        // bf16 planes of what the previous block's GEMMs will read: the dropped copy if there is one, else dx
                        if (opl && !gout) store_planes4(opl + (w0 + rb + k) * d + 4 * i, opl_stride, o);
                        if (gout) {
                            const long grow = w0 + rb + k;
                            const unsigned long long e0 = (unsigned long long)grow * d + 4 * i;
                            float4 gq = make_float4(o.x * g_alpha, o.y * g_alpha, o.z * g_alpha, o.w * g_alpha);
                            if (g_p > 0.f) {
                                const uint4 r = drop_words8(g_seed_eff, e0 >> 3);
                                const bool hi = (e0 >> 2) & 1;                 // second half of the call's eight fields
                                const unsigned wa = hi ? r.z : r.x, wb = hi ? r.w : r.y;
                                gq.x *= drop_field(wa, 0, g_dpar); gq.y *= drop_field(wa, 1, g_dpar);
                                gq.z *= drop_field(wb, 0, g_dpar); gq.w *= drop_field(wb, 1, g_dpar);
                            }
                            if (g_rowmask && !g_rowmask[grow]) gq = make_float4(0.f, 0.f, 0.f, 0.f);
                            reinterpret_cast<float4*>(gout + grow * d)[i] = gq;
                            if (opl) store_planes4(opl + grow * d + 4 * i, opl_stride, gq);
                        }
                    }
                }
            }
        }
    }
    // cross-wave reduction of the parameter gradients, then one atomic per column per block
    float4* shg = reinterpret_cast<float4*>(sh) + wave * 2 * nv;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) { shg[i] = dg[j]; shg[nv + i] = db[j]; }
    }
    __syncthreads();
    // one partial row [dgamma | dbeta] per block; summed in fixed order by ln_param_reduce_kernel
    for (int c = threadIdx.x; c < 2 * d; c += 256)
        partial[(long)blockIdx.x * 2 * d + c] = sh[c] + sh[2 * d + c] + sh[4 * d + c] + sh[6 * d + c];
    if (PAIR) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) { shg[i] = dg2[j]; shg[nv + i] = db2[j]; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * d; c += 256)
            partial2[(long)blockIdx.x * 2 * d + c] = sh[c] + sh[2 * d + c] + sh[4 * d + c] + sh[6 * d + c];
    }
}

// second stage: 64 columns x 16 partial rows per block, LDS reduce, one atomic per column per block
// (<= nblocks/16 adders per address)
#define PR_ROWS 16
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, int nblocks, int d,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int b0 = blockIdx.y * PR_ROWS, b1 = min(nblocks, b0 + PR_ROWS);
    float s = 0.f;
    if (c < 2 * d) for (int b = b0 + ry; b < b1; b += 4) s += partial[(long)b * 2 * d + c];
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < 2 * d) {
        const float v = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
        if (c < d) atomicAdd(dgamma + c, v); else atomicAdd(dbeta + (c - d), v);
    }
}

extern "C" int oe_layernorm_fwd_pl(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                                   const unsigned char* rowmask, int act, float* y, float* stats, void* y_planes, long plane_stride, void* stream);
extern "C" int oe_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                                const unsigned char* rowmask, int act, float* y, float* stats, void* stream) {
    return oe_layernorm_fwd_pl(x, gamma, beta, eps, rows, d, rowmask, act, y, stats, nullptr, 0, stream);
}
// y_planes (optional): y also as three bf16 planes (rows, d) each, plane_stride elements apart
extern "C" int oe_layernorm_fwd_pl(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                                   const unsigned char* rowmask, int act, float* y, float* stats, void* y_planes, long plane_stride, void* stream) {
    OE_REQUIRE(x && gamma && beta && y, "oe_layernorm_fwd: null pointer");
    OE_REQUIRE(!y_planes || ((((uintptr_t)y_planes) & 7) == 0 && plane_stride % 4 == 0), "oe_layernorm_fwd_pl: planes must be 8-byte aligned");
    OE_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 256 * LN_MAXV, "oe_layernorm_fwd: d=%d must be a multiple of 4 and <= %d", d, 256 * LN_MAXV);
#define LN_FWD(NVV) hipLaunchKernelGGL((layernorm_fwd_kernel<NVV, false>), dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, \
                                       beta, eps, rows, d, rowmask, act, y, stats, (__bf16*)y_planes, plane_stride, nullptr, nullptr, 0.f, nullptr, nullptr)
    if (d <= 256) LN_FWD(1); else if (d <= 512) LN_FWD(2); else if (d <= 1024) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
    OE_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" size_t oe_layernorm_bwd_workspace_floats(int rows, int d) { return (size_t)oe_cdiv(rows, LNB_ROWS) * 2 * d; }

// dx and the per-block partial sums of the parameter gradients (into the workspace).  oe_layernorm_bwd = this + the
// reduction of those partials; a caller that keeps the workspaces of many calls alive can reduce them all with ONE launch
// of oe_layernorm_param_reduce_table instead (93 reductions of 4.6 us each per step at config 2).
extern "C" int oe_layernorm_bwd_dx_drop_pl(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                           const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                                           float* dx, float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                                           const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace,
                                           void* out_planes, long plane_stride, void* stream);
extern "C" int oe_layernorm_bwd_dx_drop(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                        const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                                        float* dx, float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                                        const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace,
                                        void* stream) {
    return oe_layernorm_bwd_dx_drop_pl(dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, gout, g_alpha, g_p, g_seed, g_seed_dev,
                                       g_rowmask, workspace, nullptr, 0, stream);
}
// out_planes (optional): bf16 planes of gout when gout is given, else of dx - the tensor the previous block's GEMMs consume
extern "C" int oe_layernorm_bwd_dx_drop_pl(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                           const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                                           float* dx, float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                                           const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace,
                                           void* out_planes, long plane_stride, void* stream) {
    OE_REQUIRE(!out_planes || ((((uintptr_t)out_planes) & 7) == 0 && plane_stride % 4 == 0), "oe_layernorm_bwd_dx_drop_pl: planes must be 8-byte aligned");
    OE_REQUIRE(dy && x && gamma && stats && dx && (beta || !act), "oe_layernorm_bwd: null pointer");
    OE_REQUIRE(!gout || (d % 8 == 0 && g_p >= 0.f && g_p < 1.f && gout != dx), "oe_layernorm_bwd_dx_drop: the dropped output needs d %% 8 == 0, 0 <= p < 1");
    OE_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 256 * LN_MAXV, "oe_layernorm_bwd: d=%d must be a multiple of 4 and <= %d", d, 256 * LN_MAXV);
    OE_REQUIRE(workspace, "oe_layernorm_bwd: null workspace");
    const int nb = oe_cdiv(rows, LNB_ROWS);
#define LN_BWD(NVV, RBB) hipLaunchKernelGGL((layernorm_bwd_kernel<NVV, RBB, false>), dim3(nb), dim3(256), (size_t)8 * d * sizeof(float), (hipStream_t)stream, \
                                            dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, workspace, gout, g_alpha, g_p,  \
                                            g_seed, g_seed_dev, g_rowmask, (__bf16*)out_planes, plane_stride, nullptr, nullptr, nullptr)
    if (d <= 256) LN_BWD(1, 4); else if (d <= 512) LN_BWD(2, 2); else if (d <= 1024) LN_BWD(4, 1); else LN_BWD(8, 1);
#undef LN_BWD
    OE_LAUNCH_CHECK("layernorm_bwd");
    return 0;
}

// y2 = LN2(LN1(x)) in one pass (layernorm_fwd_kernel<.., PAIR>): y1 (LN1's output, optional), stats1 / stats2 as oe_layernorm_fwd's.
extern "C" int oe_layernorm_pair_fwd(const float* x, const float* gamma1, const float* beta1, float eps1, const float* gamma2,
                                     const float* beta2, float eps2, int rows, int d, float* y1, float* stats1, float* y2, float* stats2,
                                     void* stream) {
    OE_REQUIRE(x && gamma1 && beta1 && gamma2 && beta2 && stats1 && y2 && stats2, "oe_layernorm_pair_fwd: null pointer");
    OE_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 256 * LN_MAXV, "oe_layernorm_pair_fwd: d=%d must be a multiple of 4 and <= %d", d, 256 * LN_MAXV);
#define LN_PFWD(NVV) hipLaunchKernelGGL((layernorm_fwd_kernel<NVV, true>), dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma1, \
                                        beta1, eps1, rows, d, nullptr, 0, y1, stats1, nullptr, 0L, gamma2, beta2, eps2, y2, stats2)
    if (d <= 256) LN_PFWD(1); else if (d <= 512) LN_PFWD(2); else if (d <= 1024) LN_PFWD(4); else LN_PFWD(8);
#undef LN_PFWD
    OE_LAUNCH_CHECK("layernorm_pair_fwd");
    return 0;
}

// Backward of the pair: dy2 = gradient of y2, add = the gradient reaching y1 on its other path (NULL: none), dx as
// oe_layernorm_bwd_dx_drop's (with its optional dropped copy gout); workspace1 / workspace2 (oe_layernorm_bwd_workspace_floats each)
// receive the partials of LN1's / LN2's parameter gradients (oe_layernorm_param_reduce[_table] sums them).
extern "C" int oe_layernorm_pair_bwd_dx_drop(const float* dy2, const float* x, const float* gamma1, const float* beta1, const float* stats1,
                                             const float* gamma2, const float* stats2, int rows, int d, const float* add, float* dx,
                                             float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                                             const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace1,
                                             float* workspace2, void* stream) {
    OE_REQUIRE(dy2 && x && gamma1 && beta1 && stats1 && gamma2 && stats2 && dx && workspace1 && workspace2, "oe_layernorm_pair_bwd: null pointer");
    OE_REQUIRE(!gout || (d % 8 == 0 && g_p >= 0.f && g_p < 1.f && gout != dx), "oe_layernorm_pair_bwd: the dropped output needs d %% 8 == 0, 0 <= p < 1");
    OE_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 256 * LN_MAXV, "oe_layernorm_pair_bwd: d=%d must be a multiple of 4 and <= %d", d, 256 * LN_MAXV);
    const int nb = oe_cdiv(rows, LNB_ROWS);
#define LN_PBWD(NVV, RBB) hipLaunchKernelGGL((layernorm_bwd_kernel<NVV, RBB, true>), dim3(nb), dim3(256), (size_t)8 * d * sizeof(float), (hipStream_t)stream, \
                                             dy2, x, gamma1, beta1, 0, stats1, rows, d, nullptr, add, dx, workspace1, gout, g_alpha, g_p,  \
                                             g_seed, g_seed_dev, g_rowmask, nullptr, 0L, gamma2, stats2, workspace2)
    if (d <= 256) LN_PBWD(1, 4); else if (d <= 512) LN_PBWD(2, 2); else if (d <= 1024) LN_PBWD(4, 1); else LN_PBWD(8, 1);
#undef LN_PBWD
    OE_LAUNCH_CHECK("layernorm_pair_bwd");
    return 0;
}

extern "C" int oe_layernorm_bwd_dx(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                   const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                                   float* dx, float* workspace, void* stream) {
    return oe_layernorm_bwd_dx_drop(dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, nullptr, 1.f, 0.f, 0ull, nullptr,
                                    nullptr, workspace, stream);
}

extern "C" int oe_layernorm_param_reduce(const float* workspace, int rows, int d, float* dgamma, float* dbeta, void* stream) {
    OE_REQUIRE(workspace && dgamma && dbeta && rows > 0 && d > 0, "oe_layernorm_param_reduce: bad arguments");
    const int nb = oe_cdiv(rows, LNB_ROWS);
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(oe_cdiv(2 * d, 64), oe_cdiv(nb, PR_ROWS)), dim3(256), 0, (hipStream_t)stream, workspace, nb, d,
                       dgamma, dbeta);
    OE_LAUNCH_CHECK("ln_param_reduce");
    return 0;
}

extern "C" int oe_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                                float* dx, float* dgamma, float* dbeta, float* workspace, void* stream) {
    OE_REQUIRE(dgamma && dbeta, "oe_layernorm_bwd: null pointer");
    const int rc = oe_layernorm_bwd_dx(dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, workspace, stream);
    return rc ? rc : oe_layernorm_param_reduce(workspace, rows, d, dgamma, dbeta, stream);
}

// Table-driven reduction of the partials of `n` LayerNorm backward calls: entry e = 5 int64 words
// { workspace pointer, rows, d, dgamma pointer, dbeta pointer }.  blockIdx.y = entry, blockIdx.x = (column group of 64,
// slice of 64 partial rows); entries with fewer column groups / slices leave the surplus blocks idle.  The table lives in
// device memory, so a captured graph can hold the launch while the host fills the table after the capture.
#define PRT_ROWS 64
__global__ __launch_bounds__(256) void ln_param_reduce_table_kernel(const long long* __restrict__ table, int max_cgroups) {
    __shared__ float sh[4][64];
    const long long* e = table + (long)blockIdx.y * 5;
    const float* partial = reinterpret_cast<const float*>(e[0]);
    const int rows = (int)e[1], d = (int)e[2];
    float* dgamma = reinterpret_cast<float*>(e[3]);
    float* dbeta = reinterpret_cast<float*>(e[4]);
    if (partial == nullptr) return;
    const int nb = (rows + LNB_ROWS - 1) / LNB_ROWS;
    const int cg = blockIdx.x % max_cgroups, slice = blockIdx.x / max_cgroups;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = cg * 64 + cx;
    const int b0 = slice * PRT_ROWS, b1 = min(nb, b0 + PRT_ROWS);
    if (cg * 64 >= 2 * d || b0 >= nb) return;                  // block-uniform
    float s = 0.f;
    if (c < 2 * d) {
#pragma unroll 4
        for (int b = b0 + ry; b < b1; b += 4) s += partial[(long)b * 2 * d + c];
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < 2 * d) {
        const float v = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
        if (c < d) atomicAdd(dgamma + c, v); else atomicAdd(dbeta + (c - d), v);
    }
}

extern "C" int oe_layernorm_param_reduce_table(const long long* table, int n, int max_rows, int max_d, void* stream) {
    OE_REQUIRE(table && n > 0 && max_rows > 0 && max_d > 0, "oe_layernorm_param_reduce_table: bad arguments");
    const int max_cgroups = oe_cdiv(2 * max_d, 64), max_slices = oe_cdiv(oe_cdiv(max_rows, LNB_ROWS), PRT_ROWS);
    hipLaunchKernelGGL(ln_param_reduce_table_kernel, dim3(max_cgroups * max_slices, n), dim3(256), 0, (hipStream_t)stream, table, max_cgroups);
    OE_LAUNCH_CHECK("ln_param_reduce_table");
    return 0;
}

// ---- column sums (bias gradients) -------------------------------------------
// block = 64 float4 column groups x 4 row groups over CS_ROWS rows; LDS reduce over the row groups,
// then one atomic per column per block (few hundred adders per address: cheap next to the read).
#define CS_ROWS 32
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ldx, int m, int n, float alpha,
                                                      const float* __restrict__ alpha_dev, float* __restrict__ out, int vec) {
    __shared__ float4 sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * 4;
    const long r0 = (long)blockIdx.y * CS_ROWS;
    const long r1 = min((long)m, r0 + CS_ROWS);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n) {
        if (vec && col + 3 < n) {
            for (long r = r0 + ry; r < r1; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + col);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else {
            for (long r = r0 + ry; r < r1; r += 4) {
                const float* p = x + r * ldx + col;
                s.x += p[0];
                if (col + 1 < n) s.y += p[1];
                if (col + 2 < n) s.z += p[2];
                if (col + 3 < n) s.w += p[3];
            }
        }
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < n) {
        float a = alpha;
        if (alpha_dev) a *= *alpha_dev;
        const float4 t0 = sh[0][cx], t1 = sh[1][cx], t2 = sh[2][cx], t3 = sh[3][cx];
        atomicAdd(out + col, (t0.x + t1.x + t2.x + t3.x) * a);
        if (col + 1 < n) atomicAdd(out + col + 1, (t0.y + t1.y + t2.y + t3.y) * a);
        if (col + 2 < n) atomicAdd(out + col + 2, (t0.z + t1.z + t2.z + t3.z) * a);
        if (col + 3 < n) atomicAdd(out + col + 3, (t0.w + t1.w + t2.w + t3.w) * a);
    }
}

extern "C" int oe_colsum_f32(const float* x, long ldx, int m, int n, float alpha, const float* alpha_dev, float* out,
                             int accumulate, void* stream) {
    OE_REQUIRE(x && out && m > 0 && n > 0 && ldx >= n, "oe_colsum_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st);
        if (e != hipSuccess) { oe_set_error("oe_colsum_f32: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    const int vec = (ldx % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(oe_cdiv(n, 256), oe_cdiv(m, CS_ROWS)), dim3(256), 0, st, x, ldx, m, n, alpha,
                       alpha_dev, out, vec);
    OE_LAUNCH_CHECK("colsum");
    return 0;
}
