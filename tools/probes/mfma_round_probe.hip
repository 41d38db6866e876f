// Probe: how does v_mfma_f32_32x32x16_bf16 round when small products meet a large accumulator?
// D = C + sum_{k<16} a*b with every A element = a, every B element = b, every C element = c.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* in, float* out, int n) {
    for (int t = 0; t < n; ++t) {
        const float a = in[3 * t], b = in[3 * t + 1], c = in[3 * t + 2];
        bf16x8 fa, fb;
        for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)a; fb[e] = (__bf16)b; }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = acc[0];
    }
}
// mixed: half the k-slots carry +a*b, one slot carries a big value
__global__ void probe_mixed(const float* in, float* out, int n) {
    for (int t = 0; t < n; ++t) {
        const float small = in[3 * t], big = in[3 * t + 1], c = in[3 * t + 2];
        bf16x8 fa, fb;
        // lane half 0 holds k = 0..7, half 1 holds k = 8..15: slot 0 of half 0 = big, others small
        for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)small; fb[e] = (__bf16)1.0f; }
        if ((threadIdx.x >> 5) == 0) fa[0] = (__bf16)big;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = acc[0];
    }
}
int main() {
    const int N = 14;
    float h[3 * N] = {
        0x1p-10f, 0x1p-10f, 1.f,          // products 2^-20 x16 = 2^-16
        0x1p-12f, 0x1p-12f, 1.f,          // 2^-24 x16 = 2^-20
        0x1p-13f, 0x1p-13f, 1.f,          // 2^-26 x16 = 2^-22
        0x1p-14f, 0x1p-13f, 1.f,          // 2^-27 x16 = 2^-23 (1 ulp)
        0x1p-14f, 0x1p-14f, 1.f,          // 2^-28 x16 = 2^-24 (half ulp: tie)
        0x1.8p-14f, 0x1p-14f, 1.f,        // 1.5*2^-28 x16 = 1.5*2^-24 (0.75 ulp -> RNE 1 ulp, trunc 0)
        0x1.8p-13f, 0x1p-14f, 1.f,        // 1.5*2^-27 x16 = 1.5 ulp -> RNE 2 ulp, trunc 1
        -0x1.8p-13f, 0x1p-14f, 1.f,       // -1.5 ulp(of values below 1: ulp 2^-24 => -3 small ulps, exact)
        -0x1.8p-14f, 0x1p-14f, 1.f,       // -0.75 * 2^-23 = -1.5 * 2^-24: exact below 1
        0x1.8p-13f, 0x1p-14f, 32.f,       // sum 1.5*2^-23 vs c = 32 (ulp 2^-18): far below
        0x1p-9f, 0x1p-9f, 32.f,           // 2^-18 x16 = 2^-14 = 16 ulp
        0x1.8p-11f, 0x1p-11f, 32.f,       // 1.5*2^-22 x 16 = 1.5*2^-18 = 1.5 ulp(32) -> RNE 2, trunc 1
        0x1.cp-11f, 0x1p-11f, 32.f,       // 1.75*2^-22 x 16 = 1.75 ulp -> RNE 2, trunc 1
        0x1.8p-12f, 0x1p-11f, 32.f,       // 0.75 ulp -> RNE 1, trunc 0
    };
    float *din, *dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, N * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, dout, N);
    float o[N];
    hipMemcpy(o, dout, N * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < N; ++t) {
        const double ideal = (double)h[3 * t + 2] + 16.0 * (double)h[3 * t] * (double)h[3 * t + 1];
        const float c = h[3 * t + 2];
        const double ulp = ldexp(1.0, ilogbf(c) - 23);
        printf("case %2d: c=%g  sum=%.4f ulp   got - c = %.4f ulp   (RNE of ideal: %.4f ulp)\n", t, c, (ideal - c) / ulp, ((double)o[t] - c) / ulp,
               ((double)(float)ideal - c) / ulp);
    }
    // mixed magnitudes inside one MFMA: one big product + 15 small ones, c = 0
    const int M2 = 6;
    float g[3 * M2] = {
        0x1p-20f, 1.f, 0.f,      // big 1, 15 x 2^-20: sum = 1 + 15*2^-20 exact representable
        0x1p-24f, 1.f, 0.f,      // 15 * 2^-24 = 0.9375 * 2^-20: representable (1 + 15*2^-24 needs 24 bits: 2^-24 is half ulp at 1 -> 15*2^-24 = 7.5 ulp)
        0x1p-26f, 1.f, 0.f,      // 15 * 2^-26 = 1.875 ulp
        0x1p-27f, 1.f, 0.f,      // 15 * 2^-27 = 0.9375 ulp
        0x1p-28f, 1.f, 0.f,      // 0.47 ulp
        0x1p-30f, 1.f, 0.f,
    };
    hipMemcpy(din, g, sizeof(g), hipMemcpyHostToDevice);
    probe_mixed<<<1, 64>>>(din, dout, M2);
    hipMemcpy(o, dout, M2 * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < M2; ++t) {
        const double ideal = (double)g[3 * t + 1] + 15.0 * (double)g[3 * t];
        printf("mixed %d: small=%g  ideal-1 = %.4f ulp   got-1 = %.4f ulp\n", t, g[3 * t], (ideal - 1.0) / ldexp(1.0, -23), ((double)o[t] - 1.0) / ldexp(1.0, -23));
    }
    return 0;
}
