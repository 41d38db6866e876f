// Kaldi-compatible log-mel filterbank on device, one wavefront per frame:
// framing (snip_edges) -> *scale -> DC removal -> pre-emphasis -> povey window ->
// power-of-two FFT in LDS (Stockham radix-2; 512 points for 25 ms @ 16 kHz) -> power spectrum -> triangular mel
// filters -> log(max(., eps)) [-> global CMVN].  Plus per-utterance mean/std
// normalisation.  Replaces the torchaudio.compliance.kaldi.fbank call of
// /root/reference/openeat/dataset/dataset.py:93-100 and
// /root/reference/openeat/dataset/feature_processor.py:5-8.
// HBM-bound: reads B*N*4 bytes of audio (each sample is touched by 2.5 frames,
// served by L2), writes B*T*n_mel*4 bytes.
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define FB_WAVES 4
#define FB_MAX_NFFT 1024      // window <= 1024 samples (dataset.py:93-100 passes the corpus's sample_frequency: 8 kHz -> 256 points)

// Each wave works on its own pair of LDS buffers, so the stages are ordered by a wave-level fence (the wave's LDS
// operations complete in order; s_waitcnt makes its writes visible to its own later reads) - a block barrier would make
// the four frames of a block wait for each other at every one of the nine FFT stages.
#define FB_WAVE_SYNC()                                            \
    do {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
        __builtin_amdgcn_wave_barrier();                          \
    } while (0)
template <int FB_NFFT>
__global__ __launch_bounds__(64 * FB_WAVES) void fbank_kernel(const float* __restrict__ wav, const int* __restrict__ nsamples,
                                                              int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
                                                              float scale, float preemph, const float* __restrict__ window,
                                                              const float* __restrict__ twiddle, const int* __restrict__ mel_start,
                                                              const int* __restrict__ mel_off, const float* __restrict__ mel_w,
                                                              float floor_eps, const float* __restrict__ cmvn_mean,
                                                              const float* __restrict__ cmvn_istd, float* __restrict__ out,
                                                              float dither, unsigned long long dither_seed) {
    __shared__ float2 bufA[FB_WAVES][FB_NFFT];
    __shared__ float2 bufB[FB_WAVES][FB_NFFT];
    // the FFT's twiddle factors, once per block: read from global memory inside the butterfly loop they were 36 dependent
    // round trips per frame (four per stage) - most of the kernel's 159 us at 32 x 998 frames
    __shared__ float2 tw[FB_NFFT / 2];
    for (int i = threadIdx.x; i < FB_NFFT / 2; i += 64 * FB_WAVES) tw[i] = reinterpret_cast<const float2*>(twiddle)[i];
    __syncthreads();
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
        if (live && i < win) {
            v = src[i] * scale;
            if (dither != 0.f) {
                // kaldi's waveform dither (dataset.py:98: dither=wav_dither): independent N(0, dither^2) noise on every sample
                // of every frame's window, after scaling.  Box-Muller on two hashed uniforms of (frame, sample).
                // (Philox, not the dropout hash: on consecutive counters that one leaves a lag-1 correlation of -0.6 % between
                // the Gaussians, a measurable spectral tilt)
                const uint4 h = philox4(dither_seed, (unsigned long long)frame * FB_NFFT + i);
                const float u1 = ((float)(h.x >> 8) + 0.5f) * (1.f / 16777216.f), u2 = (float)(h.y >> 8) * (1.f / 16777216.f);
                v += dither * sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
            }
        }
        A[i].x = v;
        part += v;
    }
    const float mean = wave_sum(part) / (float)win;
    FB_WAVE_SYNC();
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
    FB_WAVE_SYNC();
    // ---- FB_NFFT-point FFT, Stockham autosort radix-2: Bf -> A -> Bf ...
    float2* s0 = Bf;
    float2* s1 = A;
#pragma unroll 1
    for (int Ns = 1; Ns < FB_NFFT; Ns <<= 1) {
        for (int j = lane; j < FB_NFFT / 2; j += 64) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * (FB_NFFT / 2 / Ns)];
            const float2 u = s0[j];
            const float2 x1 = s0[j + FB_NFFT / 2];
            const float2 v = make_float2(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
            const int base = ((j - k) << 1) + k;          // (j / Ns) * 2Ns + k
            s1[base] = make_float2(u.x + v.x, u.y + v.y);
            s1[base + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        FB_WAVE_SYNC();
        float2* tmp = s0; s0 = s1; s1 = tmp;
    }
    // result in s0; power spectrum of bins 0..FB_NFFT/2 into s1[].x
    for (int i = lane; i <= FB_NFFT / 2; i += 64) {
        const float2 c = s0[i];
        s1[i].x = c.x * c.x + c.y * c.y;
    }
    FB_WAVE_SYNC();
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
    return oe_fbank_dither(wav, nsamples, B, wav_stride, Tmax, win, hop, n_mel, scale, preemph, window, twiddle, mel_start, mel_off,
                           mel_w, floor_eps, cmvn_mean, cmvn_istd, 0.f, 0ull, out, stream);
}

extern "C" int oe_fbank_dither(const float* wav, const int* nsamples, int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
                               float scale, float preemph, const float* window, const float* twiddle, const int* mel_start,
                               const int* mel_off, const float* mel_w, float floor_eps, const float* cmvn_mean,
                               const float* cmvn_istd, float dither, unsigned long long seed, float* out, void* stream) {
    OE_REQUIRE(wav && window && twiddle && mel_start && mel_off && mel_w && out, "oe_fbank: null pointer");
    OE_REQUIRE(B > 0 && Tmax > 0 && win > 0 && win <= FB_MAX_NFFT && hop > 0 && n_mel > 0, "oe_fbank: bad shape (window must be <= %d samples)", FB_MAX_NFFT);
    OE_REQUIRE(nsamples || wav_stride >= (long)(Tmax - 1) * hop + win, "oe_fbank: Tmax frames do not fit in wav_stride samples");
    OE_REQUIRE((cmvn_mean == nullptr) == (cmvn_istd == nullptr), "oe_fbank: cmvn mean/istd must come together");
    const long frames = (long)B * Tmax;
    // FFT length = the window rounded up to a power of two (kaldi's round_to_power_of_two); `twiddle` holds its nfft / 2 factors
#define FB_LAUNCH(NF) hipLaunchKernelGGL(fbank_kernel<NF>, dim3(oe_cdiv(frames, FB_WAVES)), dim3(64 * FB_WAVES), 0, (hipStream_t)stream, wav, nsamples, \
                       B, wav_stride, Tmax, win, hop, n_mel, scale, preemph, window, twiddle, mel_start, mel_off, mel_w, floor_eps,                  \
                       cmvn_mean, cmvn_istd, out, dither, seed)
    if (win <= 128) FB_LAUNCH(128); else if (win <= 256) FB_LAUNCH(256); else if (win <= 512) FB_LAUNCH(512); else FB_LAUNCH(1024);
#undef FB_LAUNCH
    OE_LAUNCH_CHECK("fbank");
    return 0;
}

// ---- per-utterance normalisation: (x - mean_t) / std_t over the utterance's own frames (ddof = 0, no epsilon)
// 1024 threads = 16 row groups x 64 mel bins; every pass keeps four loads per thread in flight.  (With 256 threads and one
// load at a time the three passes were 250 dependent round trips each: 131 us for 10 MB.)
#define UN_RG 16
__global__ __launch_bounds__(1024) void utt_norm_kernel(float* __restrict__ x, const int* __restrict__ nframes, int Tmax, int F) {
    __shared__ float sh[2][UN_RG][64];
    const int b = blockIdx.x;
    const int cx = threadIdx.x & 63;
    const int f = blockIdx.y * 64 + cx;
    const int ry = threadIdx.x >> 6;
    const int Tb = nframes ? min(nframes[b], Tmax) : Tmax;
    float* base = x + (long)b * Tmax * F + (f < F ? f : 0);
    const bool on = f < F;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = ry;
    if (on) {
        for (; t + 3 * UN_RG < Tb; t += 4 * UN_RG) {
            s0 += base[(long)t * F]; s1 += base[(long)(t + UN_RG) * F]; s2 += base[(long)(t + 2 * UN_RG) * F]; s3 += base[(long)(t + 3 * UN_RG) * F];
        }
        for (; t < Tb; t += UN_RG) s0 += base[(long)t * F];
    }
    sh[0][ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < UN_RG; ++r) m += sh[0][r][cx];
    const float mean = m / (float)max(Tb, 1);
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (on) {
        for (t = ry; t + 3 * UN_RG < Tb; t += 4 * UN_RG) {
            const float d0 = base[(long)t * F] - mean, d1 = base[(long)(t + UN_RG) * F] - mean;
            const float d2 = base[(long)(t + 2 * UN_RG) * F] - mean, d3 = base[(long)(t + 3 * UN_RG) * F] - mean;
            q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
        }
        for (; t < Tb; t += UN_RG) { const float d = base[(long)t * F] - mean; q0 += d * d; }
    }
    sh[1][ry][cx] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < UN_RG; ++r) v += sh[1][r][cx];
    const float inv = 1.f / sqrtf(v / (float)max(Tb, 1));
    if (on) {
        for (t = ry; t + 3 * UN_RG < Tb; t += 4 * UN_RG) {
            const float a0 = base[(long)t * F], a1 = base[(long)(t + UN_RG) * F], a2 = base[(long)(t + 2 * UN_RG) * F], a3 = base[(long)(t + 3 * UN_RG) * F];
            base[(long)t * F] = (a0 - mean) * inv; base[(long)(t + UN_RG) * F] = (a1 - mean) * inv;
            base[(long)(t + 2 * UN_RG) * F] = (a2 - mean) * inv; base[(long)(t + 3 * UN_RG) * F] = (a3 - mean) * inv;
        }
        for (; t < Tb; t += UN_RG) base[(long)t * F] = (base[(long)t * F] - mean) * inv;
    }
}

extern "C" int oe_utt_normalize(float* x, const int* nframes, int B, int Tmax, int F, void* stream) {
    OE_REQUIRE(x && B > 0 && Tmax > 0 && F > 0, "oe_utt_normalize: bad arguments");
    hipLaunchKernelGGL(utt_norm_kernel, dim3(B, oe_cdiv(F, 64)), dim3(64 * UN_RG), 0, (hipStream_t)stream, x, nframes, Tmax, F);
    OE_LAUNCH_CHECK("utt_normalize");
    return 0;
}
