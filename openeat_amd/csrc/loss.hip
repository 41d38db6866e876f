// Label-smoothed cross entropy head: KL(true_dist || softmax(x)) per token with
// the t*log t constant kept, its gradient, and token accuracy, fused so that
// neither true_dist nor log_softmax is ever materialised.
// Replaces /root/reference/openeat/modules/label_smoothing_loss.py:58-91 and
// /root/reference/openeat/utils/common.py:135-157 (th_accuracy).
// HBM-bound: logits read once (second touch hits L2), gradient written once.
#include "oe_common.h"
#include "../../include/openeat_hip.h"

// one wave per row
__global__ __launch_bounds__(256) void lsm_rows_kernel(float* __restrict__ x, long ldv, long rows, int V,
                                                        const long long* __restrict__ target, int ignore_id, float smoothing,
                                                        const float* __restrict__ count_valid, float denom_host, float gscale,
                                                        int write_grad, float* __restrict__ row_loss, int* __restrict__ row_hit) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + row * ldv;
    const long long tg = target[row];
    const bool vec = ((((uintptr_t)p) & 15) == 0);
    const int nv = vec ? (V >> 2) : 0;
    if (tg == ignore_id) {
        if (write_grad) {
            for (int i = lane; i < nv; i += 64) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = (nv << 2) + lane; i < V; i += 64) p[i] = 0.f;
        }
        if (lane == 0) { row_loss[row] = 0.f; row_hit[row] = -1; }
        return;
    }
    float m = -INFINITY, s = 0.f, sx = 0.f, bv = -INFINITY;
    int bi = 0x7fffffff;
    auto visit = [&](float v, int i) {
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
        sx += v;
        if (v > bv) { bv = v; bi = i; }
    };
    for (int i = lane; i < nv; i += 64) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        visit(v.x, 4 * i); visit(v.y, 4 * i + 1); visit(v.z, 4 * i + 2); visit(v.w, 4 * i + 3);
    }
    for (int i = (nv << 2) + lane; i < V; i += 64) visit(p[i], i);
    if (m == -INFINITY) s = 0.f;
    wave_lse(m, s);
    sx = wave_sum(sx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const float lse = m + __logf(s);
    const float conf = 1.f - smoothing;
    const float e = smoothing / (float)(V - 1);
    const float xy = p[tg];
    if (lane == 0) {
        float loss = 0.f;
        if (conf > 0.f) loss += conf * (__logf(conf) - (xy - lse));
        if (e > 0.f) loss += e * ((float)(V - 1) * __logf(e) - (sx - xy - (float)(V - 1) * lse));
        row_loss[row] = loss;
        row_hit[row] = (bi == (int)tg) ? 1 : 0;
    }
    if (write_grad) {
        const float denom = count_valid ? fmaxf(*count_valid, 1.f) : denom_host;
        const float g = gscale / denom;
        const int ti = (int)tg;
        for (int i = lane; i < nv; i += 64) {
            float4 v = reinterpret_cast<const float4*>(p)[i];
            const int c = 4 * i;
            v.x = (__expf(v.x - lse) - (c == ti ? conf : e)) * g;
            v.y = (__expf(v.y - lse) - (c + 1 == ti ? conf : e)) * g;
            v.z = (__expf(v.z - lse) - (c + 2 == ti ? conf : e)) * g;
            v.w = (__expf(v.w - lse) - (c + 3 == ti ? conf : e)) * g;
            reinterpret_cast<float4*>(p)[i] = v;
        }
        for (int i = (nv << 2) + lane; i < V; i += 64) p[i] = (__expf(p[i] - lse) - (i == ti ? conf : e)) * g;
    }
}

__global__ void lsm_count_kernel(const long long* __restrict__ target, long rows, int ignore_id, float* __restrict__ out) {
    __shared__ float sh[4];
    float c = 0.f;
    for (long i = threadIdx.x; i < rows; i += 256) c += (target[i] != ignore_id) ? 1.f : 0.f;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}

// out[0] = sum(row_loss)/denom, out[1] = #correct, out[2] = #valid   (fixed order: deterministic)
__global__ void lsm_reduce_kernel(const float* __restrict__ row_loss, const int* __restrict__ row_hit, long rows,
                                  const float* __restrict__ count_valid, float denom_host, float* __restrict__ out) {
    __shared__ float sh[3][4];
    float l = 0.f, c = 0.f, n = 0.f;
    for (long i = threadIdx.x; i < rows; i += 256) {
        l += row_loss[i];
        const int h = row_hit[i];
        if (h >= 0) { n += 1.f; c += (float)h; }
    }
    l = wave_sum(l); c = wave_sum(c); n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = l; sh[1][threadIdx.x >> 6] = c; sh[2][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float denom = count_valid ? fmaxf(*count_valid, 1.f) : denom_host;
        out[0] = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / denom;
        out[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        out[2] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    }
}

extern "C" size_t oe_lsm_workspace_bytes(long rows) { return (size_t)rows * 8 + 64; }

extern "C" int oe_lsm_loss_fused(float* logits, long ldv, long rows, int V, const long long* target, int ignore_id, float smoothing,
                                 int normalize_length, float batch_size, float grad_scale, int write_grad, float* out3,
                                 void* workspace, void* stream) {
    OE_REQUIRE(logits && target && out3 && workspace, "oe_lsm_loss_fused: null pointer");
    OE_REQUIRE(rows > 0 && V > 1 && ldv >= V && smoothing >= 0.f && smoothing < 1.f && batch_size > 0.f,
               "oe_lsm_loss_fused: bad arguments rows=%ld V=%d", rows, V);
    hipStream_t st = (hipStream_t)stream;
    float* count = reinterpret_cast<float*>(workspace);
    float* row_loss = count + 16;
    int* row_hit = reinterpret_cast<int*>(row_loss + rows);
    const float* cnt = nullptr;
    if (normalize_length) {
        hipLaunchKernelGGL(lsm_count_kernel, dim3(1), dim3(256), 0, st, target, rows, ignore_id, count);
        OE_LAUNCH_CHECK("lsm_count");
        cnt = count;
    }
    hipLaunchKernelGGL(lsm_rows_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, st, logits, ldv, rows, V, target, ignore_id,
                       smoothing, cnt, batch_size, grad_scale, write_grad, row_loss, row_hit);
    OE_LAUNCH_CHECK("lsm_rows");
    hipLaunchKernelGGL(lsm_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, row_hit, rows, cnt, batch_size, out3);
    OE_LAUNCH_CHECK("lsm_reduce");
    return 0;
}
