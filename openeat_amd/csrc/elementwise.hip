// Small memory-bound kernels of the path: relative-position key preparation,
// GLU, dropout scaling, token embedding, weight layout swaps, axpby.
// All HBM-bound: bytes moved = inputs read once + outputs written once.
#include <stdlib.h>
#include "oe_common.h"
#include "../../include/openeat_hip.h"

// ---------------------------------------------------------------- rel-pos ----
// kp[b,t,h,:] = k[b,t,h,:] + p[t,h,:] ; keybias[b,h,t] = scale*(u_h.k + v_h.p)
// one wave per (b,t) row of d = H*D floats: coalesced float4 traffic; the per-head dot products are
// reduced with shuffles over the D/4 lanes of a head when D/4 is a power of two (generic fallback below).
__global__ __launch_bounds__(256) void relpos_prepare_rows_kernel(const float* __restrict__ k, long k_bs, long k_rs,
                                                                   const float* __restrict__ p, long p_rs,
                                                                   const float* __restrict__ u, const float* __restrict__ v, int B,
                                                                   int T, int H, int D, float scale, float* __restrict__ kp,
                                                                   float* __restrict__ keybias) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int t = (int)(row % T);
    const long b = row / T;
    const int d = H * D, lph = D >> 2;                     // lanes per head
    const float* kr = k + b * k_bs + (long)t * k_rs;
    const float* pr = p + (long)t * p_rs;
    float* out = kp + row * d;
    for (int c0 = 0; c0 < d; c0 += 256) {
        const int c = c0 + lane * 4;
        float s = 0.f;
        if (c < d) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + c), pv = *reinterpret_cast<const float4*>(pr + c);
            const float4 uv = *reinterpret_cast<const float4*>(u + c), vv = *reinterpret_cast<const float4*>(v + c);
            *reinterpret_cast<float4*>(out + c) = make_float4(kv.x + pv.x, kv.y + pv.y, kv.z + pv.z, kv.w + pv.w);
            s = uv.x * kv.x + uv.y * kv.y + uv.z * kv.z + uv.w * kv.w + vv.x * pv.x + vv.y * pv.y + vv.z * pv.z + vv.w * pv.w;
        }
        for (int o = 1; o < lph; o <<= 1) s += __shfl_xor(s, o, 64);
        if (c < d && (lane % lph) == 0) keybias[(b * H + c / D) * T + t] = s * scale;
    }
}
__global__ void relpos_prepare_kernel(const float* __restrict__ k, long k_bs, long k_rs, const float* __restrict__ p, long p_rs,
                                      const float* __restrict__ u, const float* __restrict__ v, int B, int T, int H, int D,
                                      float scale, float* __restrict__ kp, float* __restrict__ keybias) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * T * H) return;
    const int h = (int)(idx % H);
    const int t = (int)((idx / H) % T);
    const int b = (int)(idx / ((long)H * T));
    const float* kr = k + (long)b * k_bs + (long)t * k_rs + h * D;
    const float* pr = p + (long)t * p_rs + h * D;
    float* out = kp + ((long)b * T + t) * H * D + h * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
        const float kv = kr[d], pv = pr[d];
        out[d] = kv + pv;
        s += u[h * D + d] * kv + v[h * D + d] * pv;
    }
    keybias[((long)b * H + h) * T + t] = s * scale;
}

// dk[b,t,h,:] = dkp + scale*dkb*u_h (written with k's strides);
// dp[t,h,:]   = sum_b (dkp + scale*dkb*v_h) ; du_h += sum scale*dkb*k ; dv_h += sum scale*dkb*p
__global__ void relpos_backward_kernel(const float* __restrict__ dkp, const float* __restrict__ dkeybias,
                                       const float* __restrict__ k, long k_bs, long k_rs, const float* __restrict__ p, long p_rs,
                                       const float* __restrict__ u, const float* __restrict__ v, int B, int T, int H, int D,
                                       float scale, float* __restrict__ dk, float* __restrict__ dp, long dp_rs,
                                       float* __restrict__ du, float* __restrict__ dv) {
    // one thread per (t, h, d): loops over the batch -> deterministic dp
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)T * H * D) return;
    const int d = (int)(idx % D);
    const int h = (int)((idx / D) % H);
    const int t = (int)(idx / ((long)D * H));
    const float uu = u[h * D + d], vv = v[h * D + d], pv = p[(long)t * p_rs + h * D + d];
    float dps = 0.f, dus = 0.f, dvs = 0.f;
    // eight utterances at a time: all their loads are issued before the first store (one utterance at a time the 32
    // iterations were 32 dependent round trips, 21 us per call)
    for (int b0 = 0; b0 < B; b0 += 8) {
        float g[8], gb[8], kv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = min(b0 + i, B - 1);
            g[i] = dkp[((long)b * T + t) * H * D + h * D + d];
            gb[i] = dkeybias[((long)b * H + h) * T + t] * scale;
            kv[i] = k[(long)b * k_bs + (long)t * k_rs + h * D + d];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (b0 + i < B) {
                dus += gb[i] * kv[i];
                dk[(long)(b0 + i) * k_bs + (long)t * k_rs + h * D + d] = g[i] + gb[i] * uu;
                dps += g[i] + gb[i] * vv;
                dvs += gb[i] * pv;
            }
        }
    }
    dp[(long)t * dp_rs + h * D + d] = dps;
    atomicAdd(du + h * D + d, dus);
    atomicAdd(dv + h * D + d, dvs);
}

// The same, one block per time step: lane = four consecutive features of the (H, D) row, wave w = a quarter of the batch - every
// wave's loads (eight utterances' dkp / k float4s and key-bias gradients) go out in ONE round trip (the kernel above makes
// ceil(B / 8) dependent ones per thread: 16 us per call at config 2, 12 calls per step), the four partial sums of dp / du / dv meet in
// LDS and are added in wave order (dp stays deterministic; du / dv are atomics across the time steps as before).
__global__ __launch_bounds__(256) void relpos_backward_rows_kernel(const float* __restrict__ dkp, const float* __restrict__ dkeybias,
                                                                    const float* __restrict__ k, long k_bs, long k_rs, const float* __restrict__ p,
                                                                    long p_rs, const float* __restrict__ u, const float* __restrict__ v, int B, int T,
                                                                    int H, int D, float scale, float* __restrict__ dk, float* __restrict__ dp, long dp_rs,
                                                                    float* __restrict__ du, float* __restrict__ dv) {
    extern __shared__ float4 rp_sh[];                    // [3 (dp, du, dv)][4 waves][dq]
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dq = (H * D) >> 2;
    const int bpw = (B + 3) >> 2;
    const int b_lo = wave * bpw, b_hi = min(B, b_lo + bpw);
    for (int c = lane; c < dq; c += 64) {
        const int h = (4 * c) / D;
        const float4 uu = reinterpret_cast<const float4*>(u)[c], vv = reinterpret_cast<const float4*>(v)[c];
        const float4 pv = *reinterpret_cast<const float4*>(p + (long)t * p_rs + 4 * c);
        float4 dps = make_float4(0.f, 0.f, 0.f, 0.f), dus = dps, dvs = dps;
        for (int b0 = b_lo; b0 < b_hi; b0 += 8) {
            float4 g[8], kv[8];
            float gb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = min(b0 + i, b_hi - 1);
                g[i] = *reinterpret_cast<const float4*>(dkp + ((long)b * T + t) * H * D + 4 * c);
                gb[i] = dkeybias[((long)b * H + h) * T + t] * scale;
                kv[i] = *reinterpret_cast<const float4*>(k + (long)b * k_bs + (long)t * k_rs + 4 * c);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (b0 + i < b_hi) {
                    dus.x += gb[i] * kv[i].x; dus.y += gb[i] * kv[i].y; dus.z += gb[i] * kv[i].z; dus.w += gb[i] * kv[i].w;
                    *reinterpret_cast<float4*>(dk + (long)(b0 + i) * k_bs + (long)t * k_rs + 4 * c) =
                        make_float4(g[i].x + gb[i] * uu.x, g[i].y + gb[i] * uu.y, g[i].z + gb[i] * uu.z, g[i].w + gb[i] * uu.w);
                    dps.x += g[i].x + gb[i] * vv.x; dps.y += g[i].y + gb[i] * vv.y; dps.z += g[i].z + gb[i] * vv.z; dps.w += g[i].w + gb[i] * vv.w;
                    dvs.x += gb[i] * pv.x; dvs.y += gb[i] * pv.y; dvs.z += gb[i] * pv.z; dvs.w += gb[i] * pv.w;
                }
            }
        }
        rp_sh[(0 * 4 + wave) * dq + c] = dps;
        rp_sh[(1 * 4 + wave) * dq + c] = dus;
        rp_sh[(2 * 4 + wave) * dq + c] = dvs;
    }
    __syncthreads();
    // final sums in SCALAR layout, all four waves, consecutive lanes on consecutive floats: an atomic instruction then touches two
    // cache lines of du / dv, not eight (every block adds into the same 2 KB: with float4-strided lanes this tail alone cost 15 us)
    const float* shf = reinterpret_cast<const float*>(rp_sh);
    const int dfl = H * D;
    for (int i = threadIdx.x; i < dfl; i += 256) {
        float s3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            s3[q] = ((shf[(q * 4 + 0) * dfl + i] + shf[(q * 4 + 1) * dfl + i]) + shf[(q * 4 + 2) * dfl + i]) + shf[(q * 4 + 3) * dfl + i];
        dp[(long)t * dp_rs + i] = s3[0];
        atomicAdd(du + i, s3[1]);
        atomicAdd(dv + i, s3[2]);
    }
}

extern "C" int oe_relpos_prepare(const float* k, long k_bstride, long k_rstride, const float* p, long p_rstride,
                                 const float* u, const float* v, int B, int T, int H, int D, float scale, float* kp,
                                 float* keybias, void* stream) {
    OE_REQUIRE(k && p && u && v && kp && keybias && B > 0 && T > 0 && H > 0 && D > 0, "oe_relpos_prepare: bad arguments");
    const int lph = D / 4;
    const bool rows_ok = (D % 4 == 0) && lph >= 1 && lph <= 64 && (lph & (lph - 1)) == 0 && (256 % (4 * lph) == 0 || 4 * lph >= 256) &&
                         (k_rstride % 4 == 0) && (k_bstride % 4 == 0) && (p_rstride % 4 == 0) &&
                         (((uintptr_t)k | (uintptr_t)p | (uintptr_t)u | (uintptr_t)v | (uintptr_t)kp) & 15) == 0;
    if (rows_ok) {
        hipLaunchKernelGGL(relpos_prepare_rows_kernel, dim3(oe_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, k,
                           k_bstride, k_rstride, p, p_rstride, u, v, B, T, H, D, scale, kp, keybias);
    } else {
        const long n = (long)B * T * H;
        hipLaunchKernelGGL(relpos_prepare_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, k, k_bstride, k_rstride,
                           p, p_rstride, u, v, B, T, H, D, scale, kp, keybias);
    }
    OE_LAUNCH_CHECK("relpos_prepare");
    return 0;
}

extern "C" int oe_relpos_backward(const float* dkp, const float* dkeybias, const float* k, long k_bstride, long k_rstride,
                                  const float* p, long p_rstride, const float* u, const float* v, int B, int T, int H, int D,
                                  float scale, float* dk, float* dp, long dp_rstride, float* du, float* dv, void* stream) {
    OE_REQUIRE(dkp && dkeybias && k && p && u && v && dk && dp && du && dv, "oe_relpos_backward: null pointer");
    const long n = (long)T * H * D;
    const bool rows_ok = (D % 4 == 0) && (k_bstride % 4 == 0) && (k_rstride % 4 == 0) && (p_rstride % 4 == 0) && (dp_rstride % 4 == 0) &&
                         (((uintptr_t)dkp | (uintptr_t)k | (uintptr_t)p | (uintptr_t)u | (uintptr_t)v | (uintptr_t)dk | (uintptr_t)dp) & 15) == 0 &&
                         (size_t)3 * 4 * (H * D / 4) * sizeof(float4) <= 64 * 1024;
    static const bool rows_on = !(getenv("OE_RELPOS_BWD_ROWS") && atoi(getenv("OE_RELPOS_BWD_ROWS")) == 0);      // 0: the per-element form (A/B)
    if (rows_ok && rows_on) {
        hipLaunchKernelGGL(relpos_backward_rows_kernel, dim3(T), dim3(256), (size_t)3 * 4 * (H * D / 4) * sizeof(float4), (hipStream_t)stream, dkp,
                           dkeybias, k, k_bstride, k_rstride, p, p_rstride, u, v, B, T, H, D, scale, dk, dp, dp_rstride, du, dv);
        OE_LAUNCH_CHECK("relpos_backward");
        return 0;
    }
    hipLaunchKernelGGL(relpos_backward_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dkp, dkeybias, k,
                       k_bstride, k_rstride, p, p_rstride, u, v, B, T, H, D, scale, dk, dp, dp_rstride, du, dv);
    OE_LAUNCH_CHECK("relpos_backward");
    return 0;
}

// -------------------------------------------------------------------- GLU ----
// y[m,c] = a[m,c] * sigmoid(a[m,c+d])   (torch.nn.functional.glu over channels)
__global__ void glu_fwd_kernel(const float* __restrict__ a, long rows, int d, float* __restrict__ y) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of y
    const int dv = d >> 2;
    if (idx >= rows * dv) return;
    const long r = idx / dv;
    const int c = (int)(idx % dv) * 4;
    const float4 x = *reinterpret_cast<const float4*>(a + r * 2 * d + c);
    const float4 g = *reinterpret_cast<const float4*>(a + r * 2 * d + d + c);
    float4 o = make_float4(x.x * sigmoidf_(g.x), x.y * sigmoidf_(g.y), x.z * sigmoidf_(g.z), x.w * sigmoidf_(g.w));
    *reinterpret_cast<float4*>(y + r * d + c) = o;
}
__global__ void glu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dy, long rows, int d, float* __restrict__ da) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int dv = d >> 2;
    if (idx >= rows * dv) return;
    const long r = idx / dv;
    const int c = (int)(idx % dv) * 4;
    const float4 x = *reinterpret_cast<const float4*>(a + r * 2 * d + c);
    const float4 g = *reinterpret_cast<const float4*>(a + r * 2 * d + d + c);
    const float4 gy = *reinterpret_cast<const float4*>(dy + r * d + c);
    const float sx = sigmoidf_(g.x), sy = sigmoidf_(g.y), sz = sigmoidf_(g.z), sw = sigmoidf_(g.w);
    *reinterpret_cast<float4*>(da + r * 2 * d + c) = make_float4(gy.x * sx, gy.y * sy, gy.z * sz, gy.w * sw);
    *reinterpret_cast<float4*>(da + r * 2 * d + d + c) =
        make_float4(gy.x * x.x * sx * (1.f - sx), gy.y * x.y * sy * (1.f - sy), gy.z * x.z * sz * (1.f - sz), gy.w * x.w * sw * (1.f - sw));
}
extern "C" int oe_glu_fwd(const float* a, long rows, int d, float* y, void* stream) {
    OE_REQUIRE(a && y && rows > 0 && d > 0 && d % 4 == 0, "oe_glu_fwd: bad arguments (d %% 4 == 0 required)");
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(oe_cdiv(rows * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, a, rows, d, y);
    OE_LAUNCH_CHECK("glu_fwd");
    return 0;
}
extern "C" int oe_glu_bwd(const float* a, const float* dy, long rows, int d, float* da, void* stream) {
    OE_REQUIRE(a && dy && da && rows > 0 && d > 0 && d % 4 == 0, "oe_glu_bwd: bad arguments (d %% 4 == 0 required)");
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(oe_cdiv(rows * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, a, dy, rows, d, da);
    OE_LAUNCH_CHECK("glu_bwd");
    return 0;
}

// ---------------------------------------------------------------- dropout ----
// out[i] = alpha * x[i] * keep(seed, i)/(1-p) ; rowmask (optional, per row of `cols`) zeroes rows.
__global__ void dropout_scale_kernel(const float* x, long n, int cols, float alpha, float p, unsigned long long seed,
                                     const unsigned long long* __restrict__ seed_dev,
                                     const unsigned char* __restrict__ rowmask, float* out, int vec) {
    const long i8 = (long)blockIdx.x * blockDim.x + threadIdx.x;       // block of 8 consecutive elements = one Philox call
    const long i = i8 * 8;
    if (i >= n) return;
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    float mm[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (p > 0.f) drop_block8(seed, (unsigned long long)i8, drop_params(p), mm);
    if (vec && i + 7 < n) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 v = *reinterpret_cast<const float4*>(x + i + 4 * h);
            float o[4] = {v.x * alpha * mm[4 * h], v.y * alpha * mm[4 * h + 1], v.z * alpha * mm[4 * h + 2], v.w * alpha * mm[4 * h + 3]};
            if (rowmask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!rowmask[(i + 4 * h + e) / cols]) o[e] = 0.f;
            }
            *reinterpret_cast<float4*>(out + i + 4 * h) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (int e = 0; e < 8 && i + e < n; ++e) {
            float v = x[i + e] * alpha * mm[e];
            if (rowmask && !rowmask[(i + e) / cols]) v = 0.f;
            out[i + e] = v;
        }
    }
}
extern "C" int oe_dropout_scale(const float* x, long n, int cols, float alpha, float p, unsigned long long seed,
                                const unsigned long long* seed_dev, const unsigned char* rowmask, float* out, void* stream) {
    OE_REQUIRE(x && out && n > 0 && cols > 0 && p >= 0.f && p < 1.f, "oe_dropout_scale: bad arguments");
    const int vec = ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0;
    hipLaunchKernelGGL(dropout_scale_kernel, dim3(oe_cdiv((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, n, cols, alpha,
                       p, seed, seed_dev, rowmask, out, vec);
    OE_LAUNCH_CHECK("dropout_scale");
    return 0;
}

// ------------------------------------------------------- fp32 -> bf16 planes ----
// x (rows, cols) fp32 -> three bf16 planes p0 + p1 + p2 = x exactly (oe_common.h): the operand format of gemm_pl.hip.
// One thread per 8 consecutive elements: two float4 in, one 16-byte store per plane.
typedef __bf16 oe_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long ld, long rows, int cols8, __bf16* __restrict__ pl, long ldp,
                                                           long pstride) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols8) return;
    const long r = idx / cols8;
    const int c = (int)(idx - r * cols8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(x + r * ld + c), b = *reinterpret_cast<const float4*>(x + r * ld + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    oe_bf16x8 o[3];
    oe_split8<3>(v, o);
#pragma unroll
    for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x8*>(pl + n * pstride + r * ldp + c) = o[n];
}
extern "C" int oe_split_planes(const float* x, long ld, long rows, long cols, void* planes, long ldp, long plane_stride, void* stream) {
    OE_REQUIRE(x && planes && rows > 0 && cols > 0, "oe_split_planes: bad arguments");
    OE_REQUIRE(cols % 8 == 0 && ld % 4 == 0 && ldp % 8 == 0 && plane_stride % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)planes)) & 15) == 0,
               "oe_split_planes: cols must be a multiple of 8, rows and planes 16-byte aligned");
    const long n = rows * (cols / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, (int)(cols / 8), (__bf16*)planes, ldp,
                       plane_stride);
    OE_LAUNCH_CHECK("split_planes");
    return 0;
}

// -------------------------------------------------------------- embedding ----
// out[r,:] = table[tok[r],:]*xscale + pe[r % L,:]   (decoder.py:144-147,186 ; embedding.py:59)
__global__ void embed_fwd_kernel(const long long* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pe,
                                 long rows, int L, int d, int V, float xscale, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * d) return;
    const long r = idx / d;
    const int c = (int)(idx % d);
    long long t = tok[r];
    if (t < 0 || t >= V) t = 0;
    out[idx] = table[t * d + c] * xscale + pe[(r % L) * d + c];
}
__global__ void embed_bwd_kernel(const long long* __restrict__ tok, const float* __restrict__ dout, long rows, int d, int V,
                                 float xscale, float* __restrict__ dtable) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * d) return;
    const long r = idx / d;
    const int c = (int)(idx % d);
    const long long t = tok[r];
    if (t < 0 || t >= V) return;
    atomicAdd(dtable + t * d + c, dout[idx] * xscale);
}
extern "C" int oe_embed_fwd(const long long* tokens, const float* table, const float* pe, long rows, int L, int d, int V,
                            float xscale, float* out, void* stream) {
    OE_REQUIRE(tokens && table && pe && out && rows > 0 && L > 0 && d > 0 && V > 0, "oe_embed_fwd: bad arguments");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(oe_cdiv(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, tokens, table, pe, rows,
                       L, d, V, xscale, out);
    OE_LAUNCH_CHECK("embed_fwd");
    return 0;
}
extern "C" int oe_embed_bwd(const long long* tokens, const float* dout, long rows, int d, int V, float xscale, float* dtable,
                            void* stream) {
    OE_REQUIRE(tokens && dout && dtable && rows > 0 && d > 0 && V > 0, "oe_embed_bwd: bad arguments");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(oe_cdiv(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, tokens, dout, rows, d, V,
                       xscale, dtable);
    OE_LAUNCH_CHECK("embed_bwd");
    return 0;
}

// ------------------------------------------------------------ layout swap ----
// in [A][B][C] -> out [A][C][B]  (conv2 weight OIHW <-> O(HW)I, Linear(19d->d) column order)
__global__ void swap_last2_kernel(const float* __restrict__ in, long A, int Bd, int Cd, float* __restrict__ out, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // index into out
    if (idx >= A * Bd * Cd) return;
    const int b = (int)(idx % Bd);
    const int c = (int)((idx / Bd) % Cd);
    const long a = idx / ((long)Bd * Cd);
    const float v = in[(a * Bd + b) * Cd + c];
    out[idx] = accumulate ? out[idx] + v : v;
}
extern "C" int oe_swap_last2(const float* in, long A, int Bd, int Cd, float* out, int accumulate, void* stream) {
    OE_REQUIRE(in && out && A > 0 && Bd > 0 && Cd > 0, "oe_swap_last2: bad arguments");
    hipLaunchKernelGGL(swap_last2_kernel, dim3(oe_cdiv(A * Bd * Cd, 256)), dim3(256), 0, (hipStream_t)stream, in, A, Bd, Cd, out,
                       accumulate);
    OE_LAUNCH_CHECK("swap_last2");
    return 0;
}

// ------------------------------------------------------------------ axpby ----
__global__ void axpby_kernel(const float* x, const float* y, long n, float a, float b, const float* a_dev, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a_dev) a *= *a_dev;
    if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
extern "C" int oe_axpby(const float* x, const float* y, long n, float a, float b, const float* a_dev, float* out, void* stream) {
    OE_REQUIRE(x && out && n > 0, "oe_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, a, b, a_dev, out);
    OE_LAUNCH_CHECK("axpby");
    return 0;
}

// Joint loss of asr_model.py:150-157 / :196-198 in ONE launch (was ~8 scalar launches forward, ~7 backward):
//   att = la * (1 - r) + lr * r ;  loss = wc * lc + (1 - wc) * att      (lr, lc may be NULL: term absent)
// Every product / sum is rounded on its own, in the reference's order, so the value is the one torch computes.
__global__ void loss_combine_kernel(const float* __restrict__ lc, const float* __restrict__ la, const float* __restrict__ lr, float wc, float r,
                                    float inv_accum, float* __restrict__ out) {
    float att = la[0];
    if (lr) att = __fadd_rn(__fmul_rn(att, 1.f - r), __fmul_rn(lr[0], r));
    float loss = att;
    if (lc) loss = __fadd_rn(__fmul_rn(wc, lc[0]), __fmul_rn(1.f - wc, att));
    if (inv_accum != 1.f) loss = __fmul_rn(loss, inv_accum);
    out[0] = loss;
}
// gradients of the three losses from the incoming scalar g: d_lc = g wc, d_att = g (1 - wc), d_la = d_att (1 - r), d_lr = d_att r
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, float wc, float r, float inv_accum, int has_lc, int has_lr,
                                        float* __restrict__ d_lc, float* __restrict__ d_la, float* __restrict__ d_lr) {
    float gg = g[0];
    if (inv_accum != 1.f) gg = __fmul_rn(gg, inv_accum);
    const float d_att = has_lc ? __fmul_rn(gg, 1.f - wc) : gg;
    if (has_lc) d_lc[0] = __fmul_rn(gg, wc);
    d_la[0] = has_lr ? __fmul_rn(d_att, 1.f - r) : d_att;
    if (has_lr) d_lr[0] = __fmul_rn(d_att, r);
}
extern "C" int oe_loss_combine(const float* loss_ctc, const float* loss_att, const float* loss_att_r, float ctc_weight, float reverse_weight,
                               float inv_accum, float* out, void* stream) {
    OE_REQUIRE(loss_att && out, "oe_loss_combine: null pointer");
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_ctc, loss_att, loss_att_r, ctc_weight, reverse_weight, inv_accum, out);
    OE_LAUNCH_CHECK("loss_combine");
    return 0;
}
extern "C" int oe_loss_combine_bwd(const float* g, float ctc_weight, float reverse_weight, float inv_accum, float* d_ctc, float* d_att,
                                   float* d_att_r, void* stream) {
    OE_REQUIRE(g && d_att, "oe_loss_combine_bwd: null pointer");
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, ctc_weight, reverse_weight, inv_accum, d_ctc != nullptr,
                       d_att_r != nullptr, d_ctc, d_att, d_att_r);
    OE_LAUNCH_CHECK("loss_combine_bwd");
    return 0;
}

// global CMVN (modules/cmvn.py:43-45): y = (x - mean[f]) * istd[f]
__global__ void cmvn_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ istd, long n, int F,
                            float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int f = (int)(i % F); y[i] = (x[i] - mean[f]) * istd[f]; }
}
extern "C" int oe_global_cmvn(const float* x, const float* mean, const float* istd, long n, int F, float* y, void* stream) {
    OE_REQUIRE(x && mean && istd && y && n > 0 && F > 0, "oe_global_cmvn: bad arguments");
    hipLaunchKernelGGL(cmvn_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, istd, n, F, y);
    OE_LAUNCH_CHECK("global_cmvn");
    return 0;
}

// out = dy * act'(pre)  (stand-alone Linear+activation backward)
__global__ void act_grad_kernel(const float* __restrict__ dy, const float* __restrict__ pre, long n, int act, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dy[i] * act_bwd(act, pre[i]);
}
extern "C" int oe_act_grad(const float* dy, const float* pre, long n, int act, float* out, void* stream) {
    OE_REQUIRE(dy && pre && out && n > 0, "oe_act_grad: bad arguments");
    hipLaunchKernelGGL(act_grad_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, pre, n, act, out);
    OE_LAUNCH_CHECK("act_grad");
    return 0;
}

// log_softmax over rows (ctc.py:56-64; asr_model.py:484-488), one wave per row
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, long rows, int V, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * V;
    float m = -INFINITY, s = 0.f;
    for (int i = lane; i < V; i += 64) {
        const float v = p[i];
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    if (m == -INFINITY) s = 0.f;
    wave_lse(m, s);
    const float lse = m + __logf(s);
    for (int i = lane; i < V; i += 64) out[row * V + i] = p[i] - lse;
}
extern "C" int oe_log_softmax(const float* x, long rows, int V, float* out, void* stream) {
    OE_REQUIRE(x && out && rows > 0 && V > 0, "oe_log_softmax: bad arguments");
    hipLaunchKernelGGL(log_softmax_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, V, out);
    OE_LAUNCH_CHECK("log_softmax");
    return 0;
}

// masked softmax over the last dim of a materialised score tensor (attention.py:83-90, the module-API method
// forward_attention; the hot path never builds this tensor - it runs the fused kernels of attention.hip).
// One wave per row (b, h, i).  y = softmax with mask==0 positions at -inf, then 0-filled (rows without a valid key: all 0);
// out = y * dropout mask (element index row * T2 + j).
__global__ __launch_bounds__(256) void masked_softmax_fwd_kernel(const float* __restrict__ s, const unsigned char* __restrict__ mask,
                                                                 long m_bs, long m_rs, int H, int T1, int T2, long rows, float p,
                                                                 unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                                 float* __restrict__ y, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % T1);
    const long b = row / ((long)T1 * H);
    const unsigned char* mrow = mask ? mask + b * m_bs + (long)i * m_rs : nullptr;
    const float* sp = s + row * T2;
    float m = -INFINITY, z = 0.f;
    for (int j = lane; j < T2; j += 64) {
        if (mrow && !mrow[j]) continue;
        const float v = sp[j];
        const float mn = fmaxf(m, v);
        z = z * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    if (m == -INFINITY) z = 0.f;
    wave_lse(m, z);
    const float inv = z > 0.f ? 1.f / z : 0.f;
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    const DropParams dp = drop_params(p);
    for (int j = lane; j < T2; j += 64) {
        const bool dead = (mrow && !mrow[j]) || m == -INFINITY;
        const float v = dead ? 0.f : __expf(sp[j] - m) * inv;
        y[row * T2 + j] = v;
        if (out != y) out[row * T2 + j] = p > 0.f ? v * drop_elem(seed, (unsigned long long)(row * T2 + j), dp) : v;
    }
}
// ds = y * (g - sum_j g_j y_j), g = dout * dropout mask
__global__ __launch_bounds__(256) void masked_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dout, int T2, long rows,
                                                                 float p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                                 float* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    const DropParams dp = drop_params(p);
    float acc = 0.f;
    for (int j = lane; j < T2; j += 64) {
        const long e = row * T2 + j;
        const float g = p > 0.f ? dout[e] * drop_elem(seed, (unsigned long long)e, dp) : dout[e];
        acc += g * y[e];
    }
    acc = wave_sum(acc);
    for (int j = lane; j < T2; j += 64) {
        const long e = row * T2 + j;
        const float g = p > 0.f ? dout[e] * drop_elem(seed, (unsigned long long)e, dp) : dout[e];
        ds[e] = y[e] * (g - acc);
    }
}
extern "C" int oe_masked_softmax_fwd(const float* scores, const unsigned char* mask, long mask_bstride, long mask_rstride, int B, int H,
                                     int T1, int T2, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                                     float* y, float* out, void* stream) {
    OE_REQUIRE(scores && y && out && B > 0 && H > 0 && T1 > 0 && T2 > 0 && drop_p >= 0.f && drop_p < 1.f, "oe_masked_softmax_fwd: bad arguments");
    OE_REQUIRE(drop_p == 0.f || out != y, "oe_masked_softmax_fwd: dropout needs a separate output buffer");
    const long rows = (long)B * H * T1;
    hipLaunchKernelGGL(masked_softmax_fwd_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, mask, mask_bstride,
                       mask_rstride, H, T1, T2, rows, drop_p, seed, seed_dev, y, out);
    OE_LAUNCH_CHECK("masked_softmax_fwd");
    return 0;
}
extern "C" int oe_masked_softmax_bwd(const float* y, const float* dout, long rows, int T2, float drop_p, unsigned long long seed,
                                     const unsigned long long* seed_dev, float* dscores, void* stream) {
    OE_REQUIRE(y && dout && dscores && rows > 0 && T2 > 0 && drop_p >= 0.f && drop_p < 1.f, "oe_masked_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dout, T2, rows, drop_p, seed,
                       seed_dev, dscores);
    OE_LAUNCH_CHECK("masked_softmax_bwd");
    return 0;
}

// log_softmax(x)[row, idx] without the (rows, V) log-probability tensor: what attention rescoring reads from the decoders'
// and the LM's outputs (asr_model.py:504-528: the hypothesis' token and <eos> at every position).  One wave per row, the
// same log-sum-exp arithmetic as log_softmax_kernel (bit-identical values); one read of the logits, 4-8 bytes written.
__global__ __launch_bounds__(256) void logprob_gather_kernel(const float* __restrict__ x, long rows, int V, const long long* __restrict__ idx_a,
                                                             int idx_b, float* __restrict__ out_a, float* __restrict__ out_b) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * V;
    float m = -INFINITY, s = 0.f;
    for (int i = lane; i < V; i += 64) {
        const float v = p[i];
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    if (m == -INFINITY) s = 0.f;
    wave_lse(m, s);
    const float lse = m + __logf(s);
    if (lane == 0) {
        const long long a = idx_a[row];
        out_a[row] = (a >= 0 && a < V) ? p[a] - lse : 0.f;
        if (out_b) out_b[row] = p[idx_b] - lse;
    }
}
extern "C" int oe_logprob_gather(const float* x, long rows, int V, const long long* idx_a, int idx_b, float* out_a, float* out_b, void* stream) {
    OE_REQUIRE(x && idx_a && out_a && rows > 0 && V > 0 && (!out_b || (idx_b >= 0 && idx_b < V)), "oe_logprob_gather: bad arguments");
    hipLaunchKernelGGL(logprob_gather_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, V, idx_a, idx_b, out_a, out_b);
    OE_LAUNCH_CHECK("logprob_gather");
    return 0;
}

// Per-row top-k, optionally of the row's log-softmax (asr_model.py:251, 358: `logp.topk(beam_size)` after log_softmax; :258
// `scores.topk`).  One wave per row: the row is staged in LDS while the online log-sum-exp runs (the same arithmetic as
// log_softmax_kernel; the values equal log_softmax -> topk to the last bit or two), every lane remembers the best of its own strided
// elements, and k rounds pick the wave-wide best (ties: lowest index), strike it out and rescan only the winning lane.
// Sorted descending like torch.topk.  HBM-bound: one read of the row, k values + k int64 indices written.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long rows, int V, int k, int log_softmax,
                                                        float* __restrict__ vals, long long* __restrict__ idx) {
    extern __shared__ float topk_rows_s[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows) return;                                     // wave-uniform; no block barrier below
    float* r = topk_rows_s + (size_t)wave * V;
    const float* p = x + row * V;
    float m = -INFINITY, s = 0.f, bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < V; i += 64) {
        const float v = p[i];
        r[i] = v;
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
        if (v == v && (v > bv || bi == 0x7fffffff)) { bv = v; bi = i; }
    }
    if (m == -INFINITY) s = 0.f;
    wave_lse(m, s);
    const float lse = log_softmax ? m + __logf(s) : 0.f;
    for (int j = 0; j < k; ++j) {
        const float mv = wave_max(bv);
        int cand = (bv == mv) ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) {
            vals[row * k + j] = mv - lse;
            idx[row * k + j] = cand == 0x7fffffff ? 0 : cand;
        }
        if (bi == cand && cand != 0x7fffffff) {                  // the winner strikes its element out and rescans its own stride
            r[bi] = __builtin_nanf("");                          // struck: never compares (so -inf entries stay selectable)
            bv = -INFINITY;
            bi = 0x7fffffff;
            for (int i = lane; i < V; i += 64) {
                const float v = r[i];
                if (v == v && (v > bv || bi == 0x7fffffff)) { bv = v; bi = i; }
            }
        }
    }
}
extern "C" int oe_topk_rows(const float* x, long rows, int V, int k, int log_softmax, float* vals, long long* idx, void* stream) {
    OE_REQUIRE(x && vals && idx && rows > 0 && V > 0 && k > 0 && k <= V, "oe_topk_rows: bad arguments (rows=%ld V=%d k=%d)", rows, V, k);
    OE_REQUIRE(V <= 40000, "oe_topk_rows: a row of %d floats does not fit in LDS", V);
    const int waves = V <= 10000 ? 4 : V <= 20000 ? 2 : 1;
    const size_t lds = (size_t)waves * V * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;                               // once per process: opt in to more than 64 KB of dynamic LDS
        if (!raised) {
            if (hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                oe_set_error("oe_topk_rows: cannot raise the dynamic LDS limit");
                return -1;
            }
            raised = true;
        }
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3(oe_cdiv(rows, waves)), dim3(64 * waves), lds, (hipStream_t)stream, x, rows, V, k, log_softmax,
                       vals, idx);
    OE_LAUNCH_CHECK("topk_rows");
    return 0;
}

// Decoder token bookkeeping of a training step in ONE launch (asr_model.py:162-176: add_sos_eos of the labels and of their
// reversal, reverse_pad_list, the target mask) - as index arithmetic in torch it was ~70 four-microsecond launches that the
// captured graph ran ahead of the encoder.  One block per utterance.  Semantics (common.py:61-132, mask.py:9-69), fixed
// width W = L + 1:
//   kept(y)  = the entries of a row that are != ignore_id, in order (n of them)
//   ys_in    = [sos, kept..., eos, eos, ...]            ys_out   = [kept..., eos, ignore, ...]
//   r        = the first min(len, L) entries of the row reversed, the rest ignore_id;  r_ys_in / r_ys_out = the same from r
//   mask[b, i, j] = (j < len + 1) && (j <= i)
__global__ __launch_bounds__(64) void att_inputs_kernel(const int* __restrict__ ys, const int* __restrict__ lens, int L, int sos, int eos,
                                                        int ignore_id, long long* __restrict__ ys_in, long long* __restrict__ ys_out,
                                                        long long* __restrict__ r_in, long long* __restrict__ r_out,
                                                        unsigned char* __restrict__ mask) {
    extern __shared__ int att_s[];                      // kept tokens of the row, then of its reversal
    int* fwd = att_s;
    int* rev = att_s + L;
    __shared__ int n_fwd, n_rev;
    const int b = blockIdx.x, W = L + 1;
    const int* row = ys + (long)b * L;
    const int len = min(max(lens[b], 0), L);
    if (threadIdx.x == 0) {
        int n = 0;
        for (int i = 0; i < L; ++i)
            if (row[i] != ignore_id) fwd[n++] = row[i];
        n_fwd = n;
        n = 0;
        for (int i = len - 1; i >= 0; --i)              // reverse_pad_list, then the same compaction
            if (row[i] != ignore_id) rev[n++] = row[i];
        n_rev = n;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < W; j += 64) {
        const long o = (long)b * W + j;
        ys_in[o] = j == 0 ? sos : (j - 1 < n_fwd ? fwd[j - 1] : eos);
        ys_out[o] = j < n_fwd ? fwd[j] : (j == n_fwd ? eos : ignore_id);
        if (r_in) {
            r_in[o] = j == 0 ? sos : (j - 1 < n_rev ? rev[j - 1] : eos);
            r_out[o] = j < n_rev ? rev[j] : (j == n_rev ? eos : ignore_id);
        }
    }
    const int valid = lens[b] + 1;
    for (int e = threadIdx.x; e < W * W; e += 64) {
        const int i = e / W, j = e - i * W;
        mask[(long)b * W * W + e] = (j < valid && j <= i) ? 1 : 0;
    }
}
extern "C" int oe_att_inputs(const int* ys_pad, const int* ys_lens, int B, int L, int sos, int eos, int ignore_id, long long* ys_in,
                             long long* ys_out, long long* r_ys_in, long long* r_ys_out, unsigned char* tgt_mask, void* stream) {
    OE_REQUIRE(ys_pad && ys_lens && ys_in && ys_out && tgt_mask && B > 0 && L >= 0, "oe_att_inputs: bad arguments");
    OE_REQUIRE((r_ys_in == nullptr) == (r_ys_out == nullptr), "oe_att_inputs: the reversed outputs come together");
    OE_REQUIRE(L <= 8000, "oe_att_inputs: label rows of %d entries do not fit in LDS", L);
    hipLaunchKernelGGL(att_inputs_kernel, dim3(B), dim3(64), (size_t)2 * max(L, 1) * sizeof(int), (hipStream_t)stream, ys_pad, ys_lens, L, sos,
                       eos, ignore_id, ys_in, ys_out, r_ys_in, r_ys_out, tgt_mask);
    OE_LAUNCH_CHECK("att_inputs");
    return 0;
}

// y = act(x) (stand-alone activation module, swish.py:15-17)
__global__ void act_fwd_kernel(const float* __restrict__ x, long n, int act, float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = act_fwd(act, x[i]);
}
extern "C" int oe_act_fwd(const float* x, long n, int act, float* y, void* stream) {
    OE_REQUIRE(x && y && n > 0, "oe_act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(oe_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, act, y);
    OE_LAUNCH_CHECK("act_fwd");
    return 0;
}
