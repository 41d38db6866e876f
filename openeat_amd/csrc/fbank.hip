// Kaldi-compatible log-mel filterbank on device, one wavefront per frame:
// framing (snip_edges) -> *scale -> DC removal -> pre-emphasis -> povey window ->
// 512-point FFT in LDS (Stockham radix-2) -> power spectrum -> triangular mel
// filters -> log(max(., eps)) [-> global CMVN].  Plus per-utterance mean/std
// normalisation.  Replaces the torchaudio.compliance.kaldi.fbank call of
// /root/reference/openeat/dataset/dataset.py:93-100 and
// /root/reference/openeat/dataset/feature_processor.py:5-8.
// HBM-bound: reads B*N*4 bytes of audio (each sample is touched by 2.5 frames,
// served by L2), writes B*T*n_mel*4 bytes.
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define FB_NFFT 512
#define FB_WAVES 4

__global__ __launch_bounds__(64 * FB_WAVES) void fbank_kernel(const float* __restrict__ wav, const int* __restrict__ nsamples,
                                                              int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
                                                              float scale, float preemph, const float* __restrict__ window,
                                                              const float* __restrict__ twiddle, const int* __restrict__ mel_start,
                                                              const int* __restrict__ mel_off, const float* __restrict__ mel_w,
                                                              float floor_eps, const float* __restrict__ cmvn_mean,
                                                              const float* __restrict__ cmvn_istd, float* __restrict__ out) {
    __shared__ float2 bufA[FB_WAVES][FB_NFFT];
    __shared__ float2 bufB[FB_WAVES][FB_NFFT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long frame = (long)blockIdx.x * FB_WAVES + wave;
    const bool live_frame = frame < (long)B * Tmax;
    const int b = live_frame ? (int)(frame / Tmax) : 0;
    const int t = live_frame ? (int)(frame % Tmax) : 0;
    const int ns = nsamples ? nsamples[b] : (int)wav_stride;
    const int Tb = ns < win ? 0 : 1 + (ns - win) / hop;
    const bool live = live_frame && t < Tb;
    float2* A = bufA[wave];
    float2* Bf = bufB[wave];
    // ---- load + scale, frame mean
    float part = 0.f;
    const float* src = wav + (long)b * wav_stride + (long)t * hop;
    for (int i = lane; i < FB_NFFT; i += 64) {
        float v = 0.f;
        if (live && i < win) v = src[i] * scale;
        A[i].x = v;
        part += v;
    }
    const float mean = wave_sum(part) / (float)win;
    __syncthreads();
    // ---- DC removal, pre-emphasis (first sample uses itself), window, zero pad
    for (int i = lane; i < FB_NFFT; i += 64) {
        float v = 0.f;
        if (i < win) {
            const float cur = A[i].x - mean;
            const float prev = A[i > 0 ? i - 1 : 0].x - mean;
            v = (cur - preemph * prev) * window[i];
        }
        Bf[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    // ---- 512-point FFT, Stockham autosort radix-2: Bf -> A -> Bf ...
    float2* s0 = Bf;
    float2* s1 = A;
#pragma unroll 1
    for (int Ns = 1; Ns < FB_NFFT; Ns <<= 1) {
        for (int j = lane; j < FB_NFFT / 2; j += 64) {
            const int k = j & (Ns - 1);
            const float2 w = reinterpret_cast<const float2*>(twiddle)[k * (FB_NFFT / 2 / Ns)];
            const float2 u = s0[j];
            const float2 x1 = s0[j + FB_NFFT / 2];
            const float2 v = make_float2(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
            const int base = ((j - k) << 1) + k;          // (j / Ns) * 2Ns + k
            s1[base] = make_float2(u.x + v.x, u.y + v.y);
            s1[base + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2* tmp = s0; s0 = s1; s1 = tmp;
    }
    // result in s0; power spectrum of bins 0..256 into s1[].x
    for (int i = lane; i <= FB_NFFT / 2; i += 64) {
        const float2 c = s0[i];
        s1[i].x = c.x * c.x + c.y * c.y;
    }
    __syncthreads();
    if (!live_frame) return;
    float* dst = out + ((long)b * Tmax + t) * n_mel;
    for (int m = lane; m < n_mel; m += 64) {
        float r = 0.f;
        if (live) {
            const int st = mel_start[m], o0 = mel_off[m], o1 = mel_off[m + 1];
            float acc = 0.f;
            for (int q = o0; q < o1; ++q) acc += mel_w[q] * s1[st + (q - o0)].x;
            r = __logf(fmaxf(acc, floor_eps));
            if (cmvn_mean) r = (r - cmvn_mean[m]) * cmvn_istd[m];
        }
        dst[m] = r;      // frames past the utterance end are zero, like pad_sequence (dataset.py:217)
    }
}

extern "C" int oe_fbank(const float* wav, const int* nsamples, int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
                        float scale, float preemph, const float* window, const float* twiddle, const int* mel_start,
                        const int* mel_off, const float* mel_w, float floor_eps, const float* cmvn_mean,
                        const float* cmvn_istd, float* out, void* stream) {
    OE_REQUIRE(wav && window && twiddle && mel_start && mel_off && mel_w && out, "oe_fbank: null pointer");
    OE_REQUIRE(B > 0 && Tmax > 0 && win > 0 && win <= FB_NFFT && hop > 0 && n_mel > 0, "oe_fbank: bad shape (window must be <= %d samples)", FB_NFFT);
    OE_REQUIRE(nsamples || wav_stride >= (long)(Tmax - 1) * hop + win, "oe_fbank: Tmax frames do not fit in wav_stride samples");
    OE_REQUIRE((cmvn_mean == nullptr) == (cmvn_istd == nullptr), "oe_fbank: cmvn mean/istd must come together");
    const long frames = (long)B * Tmax;
    hipLaunchKernelGGL(fbank_kernel, dim3(oe_cdiv(frames, FB_WAVES)), dim3(64 * FB_WAVES), 0, (hipStream_t)stream, wav, nsamples,
                       B, wav_stride, Tmax, win, hop, n_mel, scale, preemph, window, twiddle, mel_start, mel_off, mel_w, floor_eps,
                       cmvn_mean, cmvn_istd, out);
    OE_LAUNCH_CHECK("fbank");
    return 0;
}

// ---- per-utterance normalisation: (x - mean_t) / std_t over the utterance's own frames (ddof = 0, no epsilon)
__global__ __launch_bounds__(256) void utt_norm_kernel(float* __restrict__ x, const int* __restrict__ nframes, int Tmax, int F) {
    __shared__ float sh[2][4][64];
    const int b = blockIdx.x;
    const int f = blockIdx.y * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int Tb = nframes ? min(nframes[b], Tmax) : Tmax;
    float* base = x + (long)b * Tmax * F;
    float s = 0.f;
    if (f < F) for (int t = ry; t < Tb; t += 4) s += base[(long)t * F + f];
    sh[0][ry][threadIdx.x & 63] = s;
    __syncthreads();
    const float mean = (sh[0][0][threadIdx.x & 63] + sh[0][1][threadIdx.x & 63] + sh[0][2][threadIdx.x & 63] + sh[0][3][threadIdx.x & 63]) / (float)max(Tb, 1);
    float q = 0.f;
    if (f < F) for (int t = ry; t < Tb; t += 4) { const float d = base[(long)t * F + f] - mean; q += d * d; }
    sh[1][ry][threadIdx.x & 63] = q;
    __syncthreads();
    const float var = (sh[1][0][threadIdx.x & 63] + sh[1][1][threadIdx.x & 63] + sh[1][2][threadIdx.x & 63] + sh[1][3][threadIdx.x & 63]) / (float)max(Tb, 1);
    const float inv = 1.f / sqrtf(var);
    if (f < F) for (int t = ry; t < Tb; t += 4) base[(long)t * F + f] = (base[(long)t * F + f] - mean) * inv;
}

extern "C" int oe_utt_normalize(float* x, const int* nframes, int B, int Tmax, int F, void* stream) {
    OE_REQUIRE(x && B > 0 && Tmax > 0 && F > 0, "oe_utt_normalize: bad arguments");
    hipLaunchKernelGGL(utt_norm_kernel, dim3(B, oe_cdiv(F, 64)), dim3(256), 0, (hipStream_t)stream, x, nframes, Tmax, F);
    OE_LAUNCH_CHECK("utt_normalize");
    return 0;
}
