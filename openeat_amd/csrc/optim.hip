// Optimiser step over the flat parameter arena: global gradient norm,
// clip_grad_norm_ and Adam fused (executor.py:58-63 + torch.optim.Adam
// defaults, train.py:195).  HBM-bound: 16 B read + 12 B written per parameter
// for Adam, 4 B read for the norm.
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define NORM_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    const long nv = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (long i = (nv << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
}

extern "C" size_t oe_grad_norm_workspace_floats(void) { return NORM_BLOCKS; }

extern "C" int oe_grad_norm(const float* g, long n, float* workspace, float* norm_out, void* stream) {
    OE_REQUIRE(g && workspace && norm_out && n > 0, "oe_grad_norm: bad arguments");
    OE_REQUIRE((((uintptr_t)g) & 15) == 0, "oe_grad_norm: gradient arena must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)min((long)NORM_BLOCKS, (long)oe_cdiv(n, 1024));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, workspace);
    OE_LAUNCH_CHECK("sumsq_partial");
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, workspace, nb, norm_out);
    OE_LAUNCH_CHECK("norm_final");
    return 0;
}

// state[0] = step count (float), incremented only when the update is applied.
__global__ void adam_tick_kernel(const float* __restrict__ total_norm, float* __restrict__ state) {
    if (isfinite(*total_norm)) state[0] += 1.f;
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ lr_dev, float lr_host,
                                                    float beta1, float beta2, float eps, float max_norm,
                                                    const float* __restrict__ total_norm, const float* __restrict__ state) {
    const float tn = total_norm ? *total_norm : 0.f;
    if (!isfinite(tn)) return;                                  // executor.py:59-60: skip the step
    float coef = 1.f;
    if (total_norm && max_norm > 0.f) coef = fminf(1.f, max_norm / (tn + 1e-6f));
    const float step = state[0];
    const float lr = lr_dev ? *lr_dev : lr_host;
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long nv = n >> 2;
    if (i < nv) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
#define ADAM1(F) { const float gg = gv.F * coef; mv.F = beta1 * mv.F + (1.f - beta1) * gg; vv.F = beta2 * vv.F + (1.f - beta2) * gg * gg; \
                   pv.F -= step_size * mv.F / (sqrtf(vv.F) * inv_sqrt_bc2 + eps); }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (i == 0) {
        for (long j = nv << 2; j < n; ++j) {
            const float gg = g[j] * coef;
            m[j] = beta1 * m[j] + (1.f - beta1) * gg;
            v[j] = beta2 * v[j] + (1.f - beta2) * gg * gg;
            p[j] -= step_size * m[j] / (sqrtf(v[j]) * inv_sqrt_bc2 + eps);
        }
    }
}

extern "C" int oe_adam_step(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float lr, float beta1,
                            float beta2, float eps, float max_norm, const float* total_norm, float* state, void* stream) {
    OE_REQUIRE(p && g && m && v && state && n > 0, "oe_adam_step: bad arguments");
    OE_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "oe_adam_step: arenas must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // state[1] is never written (stays 0, finite): used as the "norm" when no clipping is requested
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, total_norm ? total_norm : state + 1, state);
    OE_LAUNCH_CHECK("adam_tick");
    hipLaunchKernelGGL(adam_kernel, dim3(oe_cdiv((n + 3) / 4, 256)), dim3(256), 0, st, p, g, m, v, n, lr_dev, lr,
                       beta1, beta2, eps, max_norm, total_norm, state);
    OE_LAUNCH_CHECK("adam");
    return 0;
}
