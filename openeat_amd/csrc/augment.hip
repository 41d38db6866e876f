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

// ------------------------------------------------------------- speed perturbation ----
// /root/reference/openeat/dataset/audio_processor.py:20-35: sox `speed s` + `rate sr` = the waveform read s times faster
// and resampled back to sr: out[i] = x(i * s), band-limited to min(1, 1/s) of the input Nyquist.  sox is not in the
// reference tree (no parity pin; SURVEY 8f rank 2 asks for distribution parity): this is a Hann-windowed sinc
// interpolator, cutoff 0.95 * min(1, 1/s), SP_ZEROS zero crossings each side, normalised to unit DC gain.
// One output sample per thread; utterance b reads its own speed, speed 1 copies.  HBM-bound (reads hit L1/L2).
#define SP_ZEROS 16
__global__ __launch_bounds__(256) void speed_perturb_kernel(const float* __restrict__ x, long ld_in, const int* __restrict__ n_in,
                                                             const float* __restrict__ speed, float* __restrict__ y, long ld_out,
                                                             const int* __restrict__ n_out, int Nmax_out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nmax_out) return;
    float* yo = y + (long)b * ld_out;
    const int no = n_out[b], ni = n_in[b];
    if (i >= no) { yo[i] = 0.f; return; }                       // padding of the batch
    const float* xi = x + (long)b * ld_in;
    const float s = speed[b];
    if (s == 1.0f) { yo[i] = i < ni ? xi[i] : 0.f; return; }
    const double p = (double)i * (double)s;                     // position in input samples
    const int c = (int)floor(p);
    const float frac = (float)(p - (double)c);
    const float fc = 0.95f * fminf(1.f, 1.f / s);               // cutoff relative to the input Nyquist
    const float half = (float)SP_ZEROS / fc;                    // window half width in input samples
    const int R = (int)ceilf(half);
    float acc = 0.f, wsum = 0.f;
    for (int k = -R + 1; k <= R; ++k) {
        const float d = (float)k - frac;                        // tap position relative to p
        if (fabsf(d) >= half) continue;
        const float a = 3.14159265358979f * fc * d;
        const float sinc = fabsf(a) < 1e-6f ? 1.f : sinf(a) / a;
        const float w = sinc * (0.5f + 0.5f * cosf(3.14159265358979f * d / half));
        wsum += w;
        const int j = c + k;
        if (j >= 0 && j < ni) acc += w * xi[j];
    }
    yo[i] = acc / wsum;
}

extern "C" int oe_speed_perturb(const float* wav, long ld_in, const int* n_in, const float* speed, int B, int Nmax_out,
                                float* out, long ld_out, const int* n_out, void* stream) {
    OE_REQUIRE(wav && n_in && speed && out && n_out, "oe_speed_perturb: null pointer");
    OE_REQUIRE(B > 0 && Nmax_out > 0 && ld_out >= Nmax_out && ld_in > 0, "oe_speed_perturb: bad shape B=%d Nmax_out=%d ld_out=%ld", B,
               Nmax_out, ld_out);
    hipLaunchKernelGGL(speed_perturb_kernel, dim3(oe_cdiv(Nmax_out, 256), B), dim3(256), 0, (hipStream_t)stream, wav, ld_in, n_in, speed,
                       out, ld_out, n_out, Nmax_out);
    OE_LAUNCH_CHECK("speed_perturb");
    return 0;
}
