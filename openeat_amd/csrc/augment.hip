// Feature-level augmentation on the device: SpecAugment masks and spec-substitute of
// /root/reference/openeat/dataset/feature_processor.py:10-64, applied to the padded (B, Tmax, F) batch in place.
// The random draws stay on the host in the reference's own order (python `random`), so for a given seed the
// result is bit-identical to the reference; the kernels only move / zero data.  HBM-bound: one read-modify-write
// of the touched rows.
#include "oe_common.h"
#include "../../include/openeat_hip.h"

// t_masks (B, nt, 2) / f_masks (B, nf, 2): [start, end) per mask, end already clipped by the host.
__global__ __launch_bounds__(256) void spec_mask_kernel(float* __restrict__ x, const int* __restrict__ nframes, int Tmax, int F,
                                                         const int* __restrict__ t_masks, int nt, const int* __restrict__ f_masks, int nf,
                                                         float value) {
    const int b = blockIdx.y;
    const int Tb = nframes ? min(nframes[b], Tmax) : Tmax;
    const long n = (long)Tb * F;
    float* base = x + (long)b * Tmax * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i / F), f = (int)(i - (long)t * F);
        bool hit = false;
        for (int k = 0; k < nt; ++k) hit |= (t >= t_masks[(b * nt + k) * 2] && t < t_masks[(b * nt + k) * 2 + 1]);
        for (int k = 0; k < nf; ++k) hit |= (f >= f_masks[(b * nf + k) * 2] && f < f_masks[(b * nf + k) * 2 + 1]);
        if (hit) base[i] = value;
    }
}

// subs (B, ns, 3): (start, end, pos): rows [start, end) <- rows [start - pos, end - pos), one after the other
// (each substitution sees the previous ones, overlapping source/destination read before written: numpy semantics).
__global__ __launch_bounds__(256) void spec_substitute_kernel(float* __restrict__ x, int Tmax, int F, const int* __restrict__ subs, int ns,
                                                               int max_rows) {
    extern __shared__ float sh[];                     // max_rows * F
    const int b = blockIdx.x;
    float* base = x + (long)b * Tmax * F;
    for (int k = 0; k < ns; ++k) {
        const int start = subs[(b * ns + k) * 3], end = subs[(b * ns + k) * 3 + 1], pos = subs[(b * ns + k) * 3 + 2];
        const int n = max(0, min(end - start, max_rows)) * F;
        for (int i = threadIdx.x; i < n; i += 256) sh[i] = base[(long)(start - pos) * F + i];
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) base[(long)start * F + i] = sh[i];
        __syncthreads();
    }
}

// Feature dither (dataset.py:197-201): x += (u - 0.5) * a with u ~ U[0,1) per element, on each utterance's own frames.
// The reference draws u with numpy's global generator; here it is Philox (seed, element index): same distribution, not
// the same numbers.
__global__ __launch_bounds__(256) void feature_dither_kernel(float* __restrict__ x, const int* __restrict__ nframes, int Tmax, int F,
                                                              float a, unsigned long long seed) {
    const int b = blockIdx.y;
    const int Tb = nframes ? min(nframes[b], Tmax) : Tmax;
    const long n4 = ((long)Tb * F + 3) / 4;
    float* base = x + (long)b * Tmax * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint4 r = philox4(seed, (unsigned long long)b * Tmax * F / 4 + i);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long k = i * 4 + e;
            if (k < (long)Tb * F) base[k] += ((float)(w[e] >> 8) * (1.0f / 16777216.0f) - 0.5f) * a;
        }
    }
}

extern "C" int oe_feature_dither(float* x, const int* nframes, int B, int Tmax, int F, float a, unsigned long long seed, void* stream) {
    OE_REQUIRE(x && B > 0 && Tmax > 0 && F > 0, "oe_feature_dither: bad arguments");
    if (a == 0.f) return 0;
    const int nb = (int)min((long)oe_cdiv((long)Tmax * F / 4 + 1, 256), 64L);
    hipLaunchKernelGGL(feature_dither_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, nframes, Tmax, F, a, seed);
    OE_LAUNCH_CHECK("oe_feature_dither");
    return 0;
}

extern "C" int oe_spec_augment(float* x, const int* nframes, int B, int Tmax, int F, const int* t_masks, int nt, const int* f_masks,
                               int nf, float value, void* stream) {
    OE_REQUIRE(x && B > 0 && Tmax > 0 && F > 0 && nt >= 0 && nf >= 0 && (nt == 0 || t_masks) && (nf == 0 || f_masks),
               "oe_spec_augment: bad arguments");
    if (nt == 0 && nf == 0) return 0;
    const int nb = (int)min((long)oe_cdiv((long)Tmax * F, 256), 64L);
    hipLaunchKernelGGL(spec_mask_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, nframes, Tmax, F, t_masks, nt, f_masks, nf, value);
    OE_LAUNCH_CHECK("oe_spec_augment");
    return 0;
}

extern "C" int oe_spec_substitute(float* x, int B, int Tmax, int F, const int* subs, int ns, int max_rows, void* stream) {
    OE_REQUIRE(x && B > 0 && Tmax > 0 && F > 0 && ns >= 0 && (ns == 0 || subs) && max_rows > 0, "oe_spec_substitute: bad arguments");
    OE_REQUIRE((size_t)max_rows * F * sizeof(float) <= 64 * 1024, "oe_spec_substitute: max_rows * F = %d floats exceed the LDS staging buffer",
               max_rows * F);
    if (ns == 0) return 0;
    hipLaunchKernelGGL(spec_substitute_kernel, dim3(B), dim3(256), (size_t)max_rows * F * sizeof(float), (hipStream_t)stream, x, Tmax, F, subs,
                       ns, max_rows);
    OE_LAUNCH_CHECK("oe_spec_substitute");
    return 0;
}
