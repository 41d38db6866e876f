// Position-wise feed forward as ONE kernel:  y = residual + beta * drop_out( W2 drop_in(act(W1 x + b1)) + b2 )
// (/root/reference/openeat/modules/positionwise_feed_forward.py:36-43 inside encoder_layer.py:81-83,104-106 and
// decoder_layer.py:104-106), bf16 matrix cores, precision 1 (bf16 products) or 3 (hi*hi + hi*lo + lo*hi, fp32-grade).
//
// Why: as two GEMM launches the (rows x ff) intermediate travels to HBM and back (config 2: 7936 x 1024 fp32 = 32.5 MB
// written twice - pre-activation and activation - and read once), each launch pays its own prologue / epilogue, and the
// fp32 operands are split to bf16 on every fragment use.  Measured in the step: 37 + 39 us per feed-forward.
//
// How (everything is computed TRANSPOSED, so that the intermediate never leaves registers):
//   * a block owns 32 rows m of x; its 4 waves (one per SIMD, the whole 512-register file each) split the ff axis: wave w
//     takes the 32-wide ff tiles ft = w, w + 4, ... and accumulates a PARTIAL y^T over them (summed through LDS at the end);
//   * per ff tile:  h^T[fc, m] = W1[fc, :] . x^T[:, m]   - A = rows of W1, B = the block's x rows, held as fragments in
//     registers for the whole kernel (row m on the lane, 8 consecutive features per k-slot group);
//     the accumulator has fc on its rows and m on the lane: +b1, activation, dropout happen in place, and its registers
//     8s..8s+7 ARE the B fragment of the next product (k-slots in accumulator-row order, as the attention kernels do
//     with P):   y^T[c, m] += W2[c, fc] . a^T[fc, m]   - A = rows of W2 with its k-slots permuted to that order;
//   * the weights are pre-split once per optimizer step into bf16 hi / lo planes IN FRAGMENT ORDER (oe_ffn_pack_weights:
//     one 1 KiB piece = the 64 lanes' 8 elements of one MFMA operand): a wave's weight stream is then a linear run of
//     pieces that travels L2 -> LDS by LDS-DMA into a wave-private 4-slot ring (no block barrier in the main loop,
//     counted vmcnt, three 8 KiB stages in flight) and is read back with one conflict-free ds_read_b128 per fragment -
//     no conversion instruction anywhere in the K-loops;
//   * the pre-activation and the dropped activation, which the (still unfused) backward reads, leave through a
//     wave-private LDS patch as whole 128-byte row segments.
// Cost model at config 2 (d = 256, ff = 1024, 7936 rows = 248 blocks): 768 MFMAs per wave = 24.6 k cycles (11.7 us)
// if the matrix pipe never waited; 2 MiB of weight pieces per block from L2 (every block streams all of W1 and W2).
//
// Supported: d in {128, 256}, ff a multiple of 128, activation relu / swish; anything else stays on the two-GEMM path.
#include <stdlib.h>
#include "gemm_common.h"          // f32x16, static_for
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FFN_THREADS 256
#define FFN_ROWS 32

__device__ __forceinline__ int ffn_acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

// ---- weight packing ------------------------------------------------------------------------------------------------------
// W1 (ff, d) -> [ft][ks][plane][lane][8]:  W1[32 ft + (lane & 31)][16 ks + 8 (lane >> 5) + e]
// W2 (d, ff) -> [ft][dt][s][plane][lane][8]:  W2[32 dt + (lane & 31)][32 ft + 16 s + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)]
// plane 0 = bf16(x), plane 1 = bf16(x - plane 0) (only with precision 3).
// TRANSPOSED = false: w1 = W1 (ff, d), w2 = W2 (d, ff) - the forward's operands.
// TRANSPOSED = true (the backward's operands, oe_ffn_bwd): the first packed stream is W2^T in W1's role (dH = dY W2) and
// the second W1^T in W2's role (dX = dH W1): the same fragment layouts read through swapped strides - a lane's eight
// elements are then strided in memory, but consecutive lanes (rows of the fragment) stay on consecutive addresses.
template <bool TRANSPOSED>
__device__ __forceinline__ void ffn_pack_piece(const float* __restrict__ w1, const float* __restrict__ w2, int d, int ff, int planes,
                                               __bf16* __restrict__ w1p, __bf16* __restrict__ w2p, long piece, int lane) {
    const int KS = d / 16, DT = d / 32, FT = ff / 32;
    const long n1 = (long)FT * KS, n2 = (long)FT * DT * 2;
    float x[8];
    __bf16* dst;
    if (piece < n1) {
        const int ft = (int)(piece / KS), ks = (int)(piece % KS);
        const int row = 32 * ft + (lane & 31), col = 16 * ks + 8 * (lane >> 5);          // element (row, col + e) of the (ff, d) operand
        if (!TRANSPOSED) {
            const float* src = w1 + (long)row * d + col;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = w2[(long)(col + e) * ff + row];               // W2^T[row][col + e]
        }
        dst = w1p + (piece * planes) * 512 + lane * 8;
    } else if (piece < n1 + n2) {
        const long q = piece - n1;
        const int s = (int)(q & 1), dt = (int)((q >> 1) % DT), ft = (int)((q >> 1) / DT);
        const int row = 32 * dt + (lane & 31), col = 32 * ft + 16 * s + 4 * (lane >> 5);   // elements (row, col + {0..3, 8..11}) of the (d, ff) operand
        if (!TRANSPOSED) {
            const float* src = w2 + (long)row * ff + col;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 8);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = w1[(long)(col + (e & 3) + 8 * (e >> 2)) * d + row];     // W1^T[row][col + ..]
        }
        dst = w2p + (q * planes) * 512 + lane * 8;
    } else {
        return;
    }
    // planes 1 / 2: bf16(x), bf16(x - p0) (precision 1 / 3); planes 3: the three exact pieces of precision 6 (oe_common.h)
    oe_bf16x8 pl[3];
    oe_split8<3>(x, pl);
    *reinterpret_cast<oe_bf16x8*>(dst) = pl[0];
    if (planes >= 2) *reinterpret_cast<oe_bf16x8*>(dst + 512) = pl[1];
    if (planes >= 3) *reinterpret_cast<oe_bf16x8*>(dst + 1024) = pl[2];
}
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int d, int ff,
                                                       int planes, __bf16* __restrict__ w1p, __bf16* __restrict__ w2p) {
    ffn_pack_piece<TRANSPOSED>(w1, w2, d, ff, planes, w1p, w2p, (long)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);   // one wave per fragment
}
// Many feed-forwards' weights in ONE launch (oe_ffn_pack_weights_table): entry e = nine 64-bit words { W1, W2, packed W1, packed W2,
// packed W2^T (backward stream 1), packed W1^T (backward stream 2), d, ff, planes }; blockIdx.y = entry, blockIdx.z = orientation
// (0: forward pair, 1: backward pair; a null destination skips it, an all-zero entry is skipped whole).  The plane count is the
// ENTRY's (its buffers were sized for it: a table may hold feed-forwards registered under different precisions).  The table lives
// in device memory: a captured graph holds the launch.
__global__ __launch_bounds__(256) void ffn_pack_table_kernel(const long long* __restrict__ table) {
    const long long* e = table + (long)blockIdx.y * 9;
    const int planes = (int)e[8];
    if (!e[0] || !e[1] || planes < 1) return;
    const float* w1 = reinterpret_cast<const float*>(e[0]);
    const float* w2 = reinterpret_cast<const float*>(e[1]);
    const int d = (int)e[6], ff = (int)e[7];
    const long piece = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.z == 0) {
        if (e[2] && e[3]) ffn_pack_piece<false>(w1, w2, d, ff, planes, reinterpret_cast<__bf16*>(e[2]), reinterpret_cast<__bf16*>(e[3]), piece, threadIdx.x & 63);
    } else {
        if (e[4] && e[5]) ffn_pack_piece<true>(w1, w2, d, ff, planes, reinterpret_cast<__bf16*>(e[4]), reinterpret_cast<__bf16*>(e[5]), piece, threadIdx.x & 63);
    }
}

// ---- the fused kernel ------------------------------------------------------------------------------------------------------
struct FfnParams {
    const float* x; long ldx;
    const __bf16* w1p; const float* b1;
    const __bf16* w2p; const float* b2;
    float* pre; float* act_out;                  // (rows, ff) outputs for backward, either may be null
    const float* residual; long ldr; float beta;
    float* y; long ldy;
    int rows, ff, act;
    float p_in; unsigned long long seed_in;
    float p_out; unsigned long long seed_out;
    const unsigned long long* seed_dev;
};

template <int TERMS> struct WFrag { bf16x8 hi, lo; };
template <int TERMS>
__device__ __forceinline__ f32x16 ffn_mma(const WFrag<TERMS>& a, const WFrag<TERMS>& b, f32x16 c) {
    if (TERMS == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}
template <int TERMS>
__device__ __forceinline__ void ffn_split(const float (&x)[8], WFrag<TERMS>& f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f.hi[e] = (__bf16)x[e];
        if (TERMS == 3) f.lo[e] = (__bf16)(x[e] - (float)f.hi[e]);
    }
}

// D = model width (128 / 256); NOUT = number of (rows, ff) outputs written per ff tile (0, 1 or 2).
//
// Weight fragments go L2 -> REGISTERS directly: the packed layout makes every fragment one fully coalesced 1 KiB
// wave-load (global_load_dwordx4), four register sets rotate (three stages = 24 KiB per wave in flight ahead of the MFMAs)
// and hipcc counts the waits itself.  The first version streamed the pieces through a wave-private LDS-DMA ring: alone the
// stream ran at ~57 GB/s per CU (35 us for the 2 MiB), the MFMAs alone take 12 us, together 67 us - DMA issue, LDS-DMA
// writes, fragment reads and MFMAs of one wave serialise on a SIMD that holds nothing else (tools/ffn_bench.py).
// BWD = true is the feed-forward's input gradient on the same skeleton (oe_ffn_bwd): x = dY (after the output dropout /
// scale), stream 1 = W2^T, epilogue 1 = dH = (dY W2) * dropout mask * act'(pre) with the forward's pre-activation read back
// through the patch and dH written out (the weight gradient of W1 needs it), stream 2 = W1^T, epilogue 2 = plain store of dX.
template <int D, int TERMS, int NOUT, bool BWD = false>
__global__ __launch_bounds__(FFN_THREADS) void ffn_fwd_kernel(FfnParams p) {
    constexpr int PL = TERMS == 3 ? 2 : 1;
    constexpr int KS = D / 16, DT = D / 32;
    constexpr int FR = 4;                                            // fragments per stage (GEMM1: k-steps, GEMM2: (dt, s) pairs)
    constexpr int NS1 = KS / FR, NS2 = (DT * 2) / FR;
    constexpr int NSTG = NS1 + NS2;                                  // stages per ff tile
    constexpr int STAGE_BYTES = FR * PL * 1024;
    constexpr int NSET = 4;                                          // register sets; NSET - 1 stages in flight
    static_assert(NS1 >= 1 && NS2 >= 1 && KS % FR == 0 && (DT * 2) % FR == 0 && NSTG % NSET == 0, "stage split");
    __shared__ __attribute__((aligned(16))) float part_s[4][FFN_ROWS * D];     // partial y of each wave, [m][(c + 4 m) mod D]
    __shared__ __attribute__((aligned(16))) float patch_s[4][32 * 36];
    extern __shared__ __attribute__((aligned(16))) float b1_s[];     // ff floats

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, lk = lane >> 5;
    const long m0 = (long)blockIdx.x * FFN_ROWS;
    const long mrow = min(m0 + lq, (long)p.rows - 1);               // rows past the end re-read the last one; never stored
    const int FT = p.ff / 32;
    const int my_tiles = (FT - wave + 3) / 4;                        // ff tiles wave, wave + 4, ...
    const int total = my_tiles * NSTG;                               // stages of this wave's stream
    // every block streams the SAME weights: block b starts its walk b tiles further on, so that the blocks of an XCD do not
    // all ask the same L2 channels for the same lines at the same time
    const int rot = blockIdx.x % my_tiles;

    for (int i = threadIdx.x; i < p.ff; i += FFN_THREADS) b1_s[i] = p.b1 ? p.b1[i] : 0.f;

    // ---- x^T fragments: row mrow, features 16 ks + 8 lk .. + 7
    WFrag<TERMS> xf[KS];
    {
        const float* xr = p.x + mrow * p.ldx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + 16 * ks + 8 * lk), b = *reinterpret_cast<const float4*>(xr + 16 * ks + 8 * lk + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            ffn_split<TERMS>(v, xf[ks]);
        }
    }
    __syncthreads();                                                 // b1_s

    const char* w1b = reinterpret_cast<const char*>(p.w1p) + lane * 16;
    const char* w2b = reinterpret_cast<const char*>(p.w2p) + lane * 16;
    const long w1_tile_bytes = (long)KS * PL * 1024, w2_tile_bytes = (long)DT * 2 * PL * 1024;
    // stage m of the wave's stream: tile (m / NSTG), GEMM1 stages first
    auto load_stage = [&](int m, WFrag<TERMS> (&f)[FR]) {
        const int ti = m / NSTG, within = m % NSTG;
        const int ft = wave + 4 * ((ti + rot) % my_tiles);
        const char* src = within < NS1 ? w1b + ft * w1_tile_bytes + (long)within * STAGE_BYTES
                                       : w2b + ft * w2_tile_bytes + (long)(within - NS1) * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            f[j].hi = *reinterpret_cast<const bf16x8*>(src + (j * PL) * 1024);
            if (TERMS == 3) f[j].lo = *reinterpret_cast<const bf16x8*>(src + (j * PL + 1) * 1024);
        }
    };
    WFrag<TERMS> fr[NSET][FR];
    static_for<0, NSET - 1>([&](auto k_c) {
        constexpr int k = decltype(k_c)::value;
        if (k < total) load_stage(k, fr[k]);
    });

    f32x16 yacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[t][r] = 0.f;
    const unsigned long long sd = p.seed_dev ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull;
    const unsigned long long seed_in = p.seed_in + sd, seed_out = p.seed_out + sd;
    const DropParams dp_in = drop_params(p.p_in), dp_out = drop_params(p.p_out);
    float* patch = patch_s[wave];

    int n = 0;                                                       // stage whose MFMAs come next; n % NSET == its set (NSTG % NSET == 0)
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int ft = wave + 4 * ((ti + rot) % my_tiles);
        // ---- h^T tile = W1[ft] . x^T
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        float4 prev[4];                                              // BWD: this tile's pre-activation rows, issued ahead of the product
        if (BWD) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const long gr = min(m0 + ps * 8 + (lane >> 3), (long)p.rows - 1);
                prev[ps] = *reinterpret_cast<const float4*>(p.pre + gr * p.ff + ft * 32 + (lane & 7) * 4);
            }
        }
        static_for<0, NS1>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int cu = st % NSET, pf = (st + NSET - 1) % NSET;
            if (n + NSET - 1 < total) load_stage(n + NSET - 1, fr[pf]);     // into the set of stage n - 1
#pragma unroll
            for (int j = 0; j < FR; ++j) hacc = ffn_mma<TERMS>(fr[cu][j], xf[st * FR + j], hacc);
            ++n;
        });
        // ---- epilogue 1 (in place, fc on the accumulator rows): + b1, [pre ->], activation, dropout, [act ->], split
        float hv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 b4 = *reinterpret_cast<const float4*>(&b1_s[ft * 32 + 8 * g4 + 4 * lk]);
            hv[4 * g4] = hacc[4 * g4] + b4.x; hv[4 * g4 + 1] = hacc[4 * g4 + 1] + b4.y;
            hv[4 * g4 + 2] = hacc[4 * g4 + 2] + b4.z; hv[4 * g4 + 3] = hacc[4 * g4 + 3] + b4.w;
        }
        auto store_tile = [&](float* out) {                          // out (rows, ff): the tile's 32 x 32 block as 128-byte row segments
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[lq * 36 + ffn_acc_row(r, lk)] = hv[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                if (m0 + row < p.rows)
                    *reinterpret_cast<float4*>(out + (m0 + row) * p.ff + ft * 32 + c4) = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
            }
        };
        if (BWD) {
            // the pre-activation tile, coalesced into the patch and read back transposed (fc on the registers, m on the lane)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<float4*>(&patch[(ps * 8 + (lane >> 3)) * 36 + (lane & 7) * 4]) = prev[ps];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] *= act_bwd(p.act, patch[lq * 36 + ffn_acc_row(r, lk)]);
        }
        if (!BWD && NOUT >= 1 && p.pre) store_tile(p.pre);
        if (BWD) {
        } else if (p.act == OE_ACT_SWISH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] *= sigmoidf_(hv[r]);
        } else if (p.act == OE_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = fmaxf(hv[r], 0.f);
        }
        if (p.p_in > 0.f) {
            const unsigned long long e0 = (unsigned long long)(m0 + lq) * p.ff + ft * 32 + 4 * lk;     // element (m, fc) of the (rows, ff) tensor
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 h = drop_hash4(seed_in, (e0 + 8 * g4) >> 2);
                hv[4 * g4] *= drop_field(h.x, 0, dp_in); hv[4 * g4 + 1] *= drop_field(h.x, 1, dp_in);
                hv[4 * g4 + 2] *= drop_field(h.y, 0, dp_in); hv[4 * g4 + 3] *= drop_field(h.y, 1, dp_in);
            }
        }
        if ((BWD || NOUT == 2) && p.act_out) store_tile(p.act_out);
        WFrag<TERMS> af[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = hv[8 * s + e];
            ffn_split<TERMS>(v, af[s]);
        }
        // ---- y^T += W2[:, ft] . a^T
        static_for<0, NS2>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int cu = (NS1 + st) % NSET, pf = (NS1 + st + NSET - 1) % NSET;
            if (n + NSET - 1 < total) load_stage(n + NSET - 1, fr[pf]);
#pragma unroll
            for (int j = 0; j < FR; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int frx = st * FR + j;                          // (dt, s) = (frx >> 1, frx & 1)
                yacc[frx >> 1] = ffn_mma<TERMS>(fr[cu][j], af[frx & 1], yacc[frx >> 1]);
            }
            ++n;
        });
    }
    // ---- sum the four partial y^T through LDS, epilogue 2, store rows
    constexpr int PART_PITCH = D;
    float* part = &part_s[0][0];
    {
        float* mine = part_s[wave];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[lq * PART_PITCH + ((t * 32 + ffn_acc_row(r, lk) + 4 * lq) & (D - 1))] = yacc[t][r];
    }
    __syncthreads();
    constexpr int C4 = D / 4;                                        // float4 columns per row
    for (int i = threadIdx.x; i < FFN_ROWS * C4; i += FFN_THREADS) {
        const int row = i / C4, c4 = (i % C4) * 4;
        const long gr = m0 + row;
        if (gr >= p.rows) continue;
        const int pc = (c4 + 4 * row) & (D - 1);
        float4 v = *reinterpret_cast<const float4*>(&part[row * PART_PITCH + pc]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 u = *reinterpret_cast<const float4*>(&part[w * (FFN_ROWS * PART_PITCH) + row * PART_PITCH + pc]);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (p.b2) { v.x += p.b2[c4]; v.y += p.b2[c4 + 1]; v.z += p.b2[c4 + 2]; v.w += p.b2[c4 + 3]; }
        if (p.p_out > 0.f) {
            const uint2 h = drop_hash4(seed_out, ((unsigned long long)gr * D + c4) >> 2);
            v.x *= drop_field(h.x, 0, dp_out); v.y *= drop_field(h.x, 1, dp_out);
            v.z *= drop_field(h.y, 0, dp_out); v.w *= drop_field(h.y, 1, dp_out);
        }
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual) res = *reinterpret_cast<const float4*>(p.residual + gr * p.ldr + c4);
        v = make_float4(res.x + p.beta * v.x, res.y + p.beta * v.y, res.z + p.beta * v.z, res.w + p.beta * v.w);
        *reinterpret_cast<float4*>(p.y + gr * p.ldy + c4) = v;
    }
}

// precision 6 (three planes, six terms): csrc/ffn6.hip
int oe_ffn6_supported(int d, int ff);
int oe_ffn6_run(const oe_ffn_args* a, bool bwd, void* stream);
static int ffn_planes(int precision) { return precision == 6 ? 3 : precision == 3 ? 2 : 1; }

extern "C" size_t oe_ffn_packed_bytes(int d, int ff, int precision) {
    return (size_t)d * ff * 2 * ffn_planes(precision);
}

extern "C" int oe_ffn_supported(int d, int ff, int precision, int act) {
    if (!(act == OE_ACT_NONE || act == OE_ACT_RELU || act == OE_ACT_SWISH)) return 0;
    if (precision == 6) return oe_ffn6_supported(d, ff);
    return (d == 128 || d == 256) && ff > 0 && ff % 128 == 0 && ff <= 8192 && (precision == 1 || precision == 3);
}

static int ffn_pack(const float* w1, const float* w2, int d, int ff, int precision, void* w1p, void* w2p, bool transposed, void* stream) {
    OE_REQUIRE(w1 && w2 && w1p && w2p, "oe_ffn_pack_weights: null pointer");
    OE_REQUIRE(oe_ffn_supported(d, ff, precision, 0), "oe_ffn_pack_weights: unsupported shape d=%d ff=%d precision=%d", d, ff, precision);
    OE_REQUIRE(((((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)w1p) | ((uintptr_t)w2p)) & 15) == 0, "oe_ffn_pack_weights: 16-byte alignment required");
    const long pieces = (long)(ff / 32) * (d / 16) + (long)(ff / 32) * (d / 32) * 2;
    if (transposed)
        hipLaunchKernelGGL(ffn_pack_kernel<true>, dim3(oe_cdiv(pieces, 4)), dim3(256), 0, (hipStream_t)stream, w1, w2, d, ff,
                           ffn_planes(precision), (__bf16*)w1p, (__bf16*)w2p);
    else
        hipLaunchKernelGGL(ffn_pack_kernel<false>, dim3(oe_cdiv(pieces, 4)), dim3(256), 0, (hipStream_t)stream, w1, w2, d, ff,
                           ffn_planes(precision), (__bf16*)w1p, (__bf16*)w2p);
    OE_LAUNCH_CHECK("oe_ffn_pack_weights");
    return 0;
}
extern "C" int oe_ffn_pack_weights_table(const void* table, int n, int max_d, int max_ff, int precision, void* stream) {
    OE_REQUIRE(table && n > 0, "oe_ffn_pack_weights_table: empty table");
    OE_REQUIRE(max_d >= 32 && max_d % 32 == 0 && max_ff >= 32 && max_ff % 32 == 0 && (precision == 1 || precision == 3 || precision == 6),
               "oe_ffn_pack_weights_table: bad launch geometry d=%d ff=%d precision=%d", max_d, max_ff, precision);
    const long pieces = (long)(max_ff / 32) * (max_d / 16) + (long)(max_ff / 32) * (max_d / 32) * 2;       // the largest entry's count
    hipLaunchKernelGGL(ffn_pack_table_kernel, dim3(oe_cdiv(pieces, 4), n, 2), dim3(256), 0, (hipStream_t)stream, (const long long*)table);
    OE_LAUNCH_CHECK("oe_ffn_pack_weights_table");
    return 0;
}
extern "C" int oe_ffn_pack_weights(const float* w1, const float* w2, int d, int ff, int precision, void* w1p, void* w2p, void* stream) {
    return ffn_pack(w1, w2, d, ff, precision, w1p, w2p, false, stream);
}
extern "C" int oe_ffn_pack_weights_bwd(const float* w1, const float* w2, int d, int ff, int precision, void* w2tp, void* w1tp, void* stream) {
    return ffn_pack(w1, w2, d, ff, precision, w2tp, w1tp, true, stream);
}

template <int D, int TERMS>
static int ffn_launch(const FfnParams& p, int nout, hipStream_t st) {
    const dim3 grid(oe_cdiv(p.rows, FFN_ROWS)), block(FFN_THREADS);
    const size_t dyn = (size_t)p.ff * sizeof(float);
    if (nout == 2) hipLaunchKernelGGL((ffn_fwd_kernel<D, TERMS, 2>), grid, block, dyn, st, p);
    else if (nout == 1) hipLaunchKernelGGL((ffn_fwd_kernel<D, TERMS, 1>), grid, block, dyn, st, p);
    else hipLaunchKernelGGL((ffn_fwd_kernel<D, TERMS, 0>), grid, block, dyn, st, p);
    OE_LAUNCH_CHECK("oe_ffn_fwd");
    return 0;
}

extern "C" int oe_ffn_fwd(const oe_ffn_args* a, void* stream) {
    OE_REQUIRE(a && a->x && a->w1p && a->w2p && a->y, "oe_ffn_fwd: null pointer");
    OE_REQUIRE(oe_ffn_supported(a->d, a->ff, a->precision, a->act), "oe_ffn_fwd: unsupported d=%d ff=%d precision=%d act=%d", a->d, a->ff,
               a->precision, a->act);
    OE_REQUIRE(a->rows > 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && (!a->residual || a->ldr % 4 == 0), "oe_ffn_fwd: bad rows / strides");
    OE_REQUIRE(((((uintptr_t)a->x) | ((uintptr_t)a->y) | ((uintptr_t)a->residual) | ((uintptr_t)a->pre_out) | ((uintptr_t)a->act_out) |
                 ((uintptr_t)a->w1p) | ((uintptr_t)a->w2p)) & 15) == 0, "oe_ffn_fwd: 16-byte alignment required");
    OE_REQUIRE(a->drop_in >= 0.f && a->drop_in < 1.f && a->drop_out >= 0.f && a->drop_out < 1.f, "oe_ffn_fwd: dropout rate out of range");
    OE_REQUIRE(!(a->act_out && !a->pre_out), "oe_ffn_fwd: act_out needs pre_out (the store count is a compile-time constant)");
    if (a->precision == 6) return oe_ffn6_run(a, false, stream);
    FfnParams p{};
    p.x = a->x; p.ldx = a->ldx; p.w1p = (const __bf16*)a->w1p; p.b1 = a->b1; p.w2p = (const __bf16*)a->w2p; p.b2 = a->b2;
    p.pre = a->pre_out; p.act_out = a->act_out; p.residual = a->residual; p.ldr = a->ldr; p.beta = a->beta; p.y = a->y; p.ldy = a->ldy;
    p.rows = a->rows; p.ff = a->ff; p.act = a->act; p.p_in = a->drop_in; p.seed_in = a->seed_in; p.p_out = a->drop_out; p.seed_out = a->seed_out;
    p.seed_dev = a->seed_dev;
    const int nout = a->act_out ? 2 : a->pre_out ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (a->d == 256) return a->precision == 3 ? ffn_launch<256, 3>(p, nout, st) : ffn_launch<256, 1>(p, nout, st);
    return a->precision == 3 ? ffn_launch<128, 3>(p, nout, st) : ffn_launch<128, 1>(p, nout, st);
}

// The feed-forward's input gradient in one launch (autograd of positionwise_feed_forward.py:43):
//   dH = (dY W2) * dropout mask(drop_in, seed_in) * act'(pre),   dX = dH W1
// oe_ffn_args fields as used here: x = dY (rows, d), w1p / w2p = the two streams of oe_ffn_pack_weights_bwd, pre_out = the
// forward's pre-activation (rows, ff) - an INPUT -, act_out = dH (rows, ff) output, y = dX (rows, d); b1, b2, residual,
// drop_out must be unset.
extern "C" int oe_ffn_bwd(const oe_ffn_args* a, void* stream) {
    OE_REQUIRE(a && (a->x || a->ln.dy) && a->w1p && a->w2p && a->y && a->pre_out && a->act_out, "oe_ffn_bwd: null pointer");
    OE_REQUIRE(!a->ln.dy || a->precision == 6, "oe_ffn_bwd: the LayerNorm-backward prologue exists in precision 6 only");
    OE_REQUIRE(oe_ffn_supported(a->d, a->ff, a->precision, a->act), "oe_ffn_bwd: unsupported d=%d ff=%d precision=%d act=%d", a->d, a->ff,
               a->precision, a->act);
    OE_REQUIRE(!a->b1 && !a->b2 && !a->residual && a->drop_out == 0.f && a->beta == 1.f, "oe_ffn_bwd: bias / residual / output dropout do not apply");
    OE_REQUIRE(a->rows > 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0, "oe_ffn_bwd: bad rows / strides");
    OE_REQUIRE(((((uintptr_t)a->x) | ((uintptr_t)a->y) | ((uintptr_t)a->pre_out) | ((uintptr_t)a->act_out) | ((uintptr_t)a->w1p) |
                 ((uintptr_t)a->w2p)) & 15) == 0, "oe_ffn_bwd: 16-byte alignment required");
    OE_REQUIRE(a->drop_in >= 0.f && a->drop_in < 1.f, "oe_ffn_bwd: dropout rate out of range");
    if (a->precision == 6) return oe_ffn6_run(a, true, stream);
    FfnParams p{};
    p.x = a->x; p.ldx = a->ldx; p.w1p = (const __bf16*)a->w1p; p.w2p = (const __bf16*)a->w2p;
    p.pre = a->pre_out; p.act_out = a->act_out; p.beta = 1.f; p.y = a->y; p.ldy = a->ldy;
    p.rows = a->rows; p.ff = a->ff; p.act = a->act; p.p_in = a->drop_in; p.seed_in = a->seed_in; p.seed_dev = a->seed_dev;
    const dim3 grid(oe_cdiv(p.rows, FFN_ROWS)), block(FFN_THREADS);
    const size_t dyn = (size_t)p.ff * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (a->d == 256) {
        if (a->precision == 3) hipLaunchKernelGGL((ffn_fwd_kernel<256, 3, 1, true>), grid, block, dyn, st, p);
        else hipLaunchKernelGGL((ffn_fwd_kernel<256, 1, 1, true>), grid, block, dyn, st, p);
    } else {
        if (a->precision == 3) hipLaunchKernelGGL((ffn_fwd_kernel<128, 3, 1, true>), grid, block, dyn, st, p);
        else hipLaunchKernelGGL((ffn_fwd_kernel<128, 1, 1, true>), grid, block, dyn, st, p);
    }
    OE_LAUNCH_CHECK("oe_ffn_bwd");
    return 0;
}
