// Position-wise feed forward as ONE kernel in the REFERENCE-PRECISION arithmetic (precision 6: three exact bf16 pieces per
// operand, six MFMA terms per product - oe_common.h):   y = residual + beta * drop_out( W2 drop_in(act(W1 x + b1)) + b2 )
// (/root/reference/openeat/modules/positionwise_feed_forward.py:36-43 inside encoder_layer.py:81-83,104-106) and, on the same
// skeleton, its input gradient  dH = (dY W2) * mask * act'(pre),  dX = dH W1  (autograd of the same lines).
//
// Why a kernel of its own (ffn.hip is precision 1 / 3, d <= 256, everything in registers): with three planes the block's x
// fragments alone would be 192 registers per lane and the partial y^T of a wave 128-256 more.  And why fuse at all: as two
// GEMM launches the (rows, ff) pre-activation AND activation go to HBM and the activation comes back (config 2: 64 MB written,
// 32 MB read per feed-forward, 24 of them per step), each launch pays its own ramp for 4 GFLOP, and in the step the pair takes
// 100 us (47.5 + 52.6, profiles/r03_experiments.md) against 72 us alone.
//
// Design (everything transposed, as ffn.hip: weights are the MFMA A operand, activations the B operand with the row on the lane):
//   * a block owns BM = 32 RT rows (RT = 2 at d <= 256: every weight fragment that leaves L2 feeds 12 MFMAs instead of 6 - the
//     kernel is bound by that stream: 3 MB of weight planes per block at d = 256, ff = 1024);
//   * the block's x rows are split ONCE into three bf16 planes in LDS ([plane][row][d + 8]: the 16-byte pad puts the 16 rows of
//     a ds_read_b128 lane group on 16 different bank quads);
//   * the ff axis is walked in chunks of 128 (four 32-wide tiles).  Per chunk, wave w (one per SIMD, the whole register file)
//       GEMM 1:  h^T tile w = W1[tile] . x^T  for both row tiles  (A from the packed weight stream, B = x fragments from LDS)
//       epilogue 1 on the accumulators: + b1, pre-activation out (through a wave-private LDS patch, 128-byte row segments),
//                activation, dropout, split into three planes -> LDS, in the B-fragment order of GEMM 2 (lane-linear 16-byte
//                pieces: the accumulator's registers 8s..8s+7 ARE k-slots 8 lk..8 lk + 7 of step s, ffn.hip's k-slot permutation)
//       GEMM 2:  y^T[d tiles of wave w] += W2[those d tiles, chunk] . h^T   - the waves split the OUTPUT columns here, so every
//                wave reads all four h tiles of the chunk from LDS and nobody holds a partial sum: no cross-wave reduction,
//                64 accumulator registers instead of 256;
//     two raw s_barriers per chunk (h written -> read; read -> rewritten), 384 MFMAs per wave between them;
//   * the weight fragments come pre-split IN FRAGMENT ORDER (oe_ffn_pack_weights, 3 planes: one 3 KiB piece = the 64 lanes' 8
//     elements of one MFMA operand, three planes) straight L2 -> registers, four rotating register sets of two fragments
//     (18 KiB per wave in flight across both GEMMs, the barriers and the chunk boundaries; hipcc counts the vmcnt);
//   * epilogue 2: each wave's own 64 output columns leave through its LDS patch as 128-byte row segments: + b2, dropout, scaled
//     residual.
// Supported: d in {128, 256} (64-row blocks) and 512 (32-row blocks: 96 KiB of x planes), ff a multiple of 128.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

// Shape of the weight ring: NSET register sets of FR pieces (3 KiB each), NSET - 1 stages in flight per wave.  Measured on MI355X
// (tools/probes/ring_depth.py + the step, same box, profiles/r04_experiments.md): (FR, NSET) = (2, 4) - 144 KiB per block in flight, the
// round's first choice - 15.94-15.96 ms/step, (2, 2) 15.85-15.89, (4, 2) 15.97, (1, 8) 15.86-15.88, (1, 2) slower feed-forward,
// (2, 8) spills; **(1, 4)** - 72 KiB in flight, 48 ring registers instead of 96 - 15.70-15.77, and faster at every shape tried (d = 512:
// 332 -> 310 us backward; 25 472 rows: 214 -> 203 us).  The kernels' intake is a rate, not a latency: more bytes in flight buy nothing.
#ifndef OE_F6_NSET
#define OE_F6_NSET 4
#endif
#ifndef OE_F6_FR
#define OE_F6_FR 1
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// LayerNorm-backward prologue (oe_ln_prologue of the header): see ln_bwd_rows_to_planes
struct LnPro {
    const float* dy; const float* x; const float* stats; const float* gamma; const float* add;
    float* dx; float* g; float* ws;
    float g_alpha, g_p; unsigned long long g_seed; const unsigned char* g_rowmask; const unsigned char* ln_rowmask;
    const float* beta; const float* gamma2; const float* stats2; float* ws2;     // pair (gamma2 != nullptr): see ln_bwd_rows_to_planes
};
static LnPro ln_pro_of(const oe_ln_prologue& a) {
    LnPro q;
    q.dy = a.dy; q.x = a.x; q.stats = a.stats; q.gamma = a.gamma; q.add = a.add; q.dx = a.dx; q.g = a.g; q.ws = a.ws;
    q.g_alpha = a.g_alpha; q.g_p = a.g_p; q.g_seed = a.g_seed; q.g_rowmask = a.g_rowmask; q.ln_rowmask = a.ln_rowmask;
    q.beta = a.beta; q.gamma2 = a.gamma2; q.stats2 = a.stats2; q.ws2 = a.ws2;
    return q;
}
static const char* ln_pro_check(const oe_ln_prologue& a) {
    if (!(a.x && a.stats && a.gamma && a.dx && a.g && a.ws)) return "null pointer";
    if (((((uintptr_t)a.dy) | ((uintptr_t)a.x) | ((uintptr_t)a.gamma) | ((uintptr_t)a.add) | ((uintptr_t)a.dx) | ((uintptr_t)a.g)) & 15) != 0) return "16-byte alignment required";
    if (!(a.g_p >= 0.f && a.g_p < 1.f)) return "dropout rate out of range";
    if (a.gamma2 && !(a.beta && a.stats2 && a.ws2 && !a.ln_rowmask)) return "pair: beta / stats2 / ws2 missing (and no row mask)";
    if (((((uintptr_t)a.beta) | ((uintptr_t)a.gamma2)) & 15) != 0) return "16-byte alignment required";
    return nullptr;
}

// LayerNorm-FORWARD prologue (oe_lnf_prologue of the header): see ln_fwd_rows_to_planes
struct LnfPro {
    const float* x; const float* gamma; const float* beta; float eps; float* y; float* stats; const unsigned char* rowmask;
    const float* gamma2; const float* beta2; float eps2; float* u; float* stats2;      // pair (gamma2 != nullptr): y = LN2(u), u = LN1(x)
};
static LnfPro lnf_pro_of(const oe_lnf_prologue& a) {
    LnfPro q;
    q.x = a.x; q.gamma = a.gamma; q.beta = a.beta; q.eps = a.eps; q.y = a.y; q.stats = a.stats; q.rowmask = a.rowmask;
    q.gamma2 = a.gamma2; q.beta2 = a.beta2; q.eps2 = a.eps2; q.u = a.u; q.stats2 = a.stats2;
    return q;
}
static const char* lnf_pro_check(const oe_lnf_prologue& a) {
    if (!(a.gamma && a.beta && a.y && a.stats)) return "null pointer";
    if (((((uintptr_t)a.x) | ((uintptr_t)a.gamma) | ((uintptr_t)a.beta) | ((uintptr_t)a.y)) & 15) != 0) return "16-byte alignment required";
    if (a.gamma2 && !(a.beta2 && a.stats2 && !a.rowmask)) return "pair: beta2 / stats2 missing (and no row mask)";
    if (((((uintptr_t)a.gamma2) | ((uintptr_t)a.beta2) | ((uintptr_t)a.u)) & 15) != 0) return "16-byte alignment required";
    return nullptr;
}

struct Ffn6Params {
    const float* x; long ldx;
    const unsigned char* w1p; const float* b1;
    const unsigned char* w2p; const float* b2;
    float* pre; float* act_out;                  // (rows, ff): fwd outputs (either may be null); BWD: pre is an INPUT, act_out = dH
    const float* residual; long ldr; float beta;
    float* y; long ldy;
    int rows, ff, act;
    float p_in; unsigned long long seed_in;
    float p_out; unsigned long long seed_out;
    const unsigned long long* seed_dev;
    LnPro ln;                                    // BWD with LNP = 1: the rows of x (= dY) are made by a LayerNorm backward
    LnfPro lnf;                                  // forward with LNP = 2: the rows of x are made by a LayerNorm forward
};

// derivative of the three activations the kernel admits (oe_ffn_supported): none, relu, swish - act_bwd's full table (tanh, erf, ...)
// is too much code for the unrolled epilogue: hipcc then keeps a loop and the tile's registers become a private array in LDS
__device__ __forceinline__ float f6_act_bwd(int act, float x) {
    const float s = sigmoidf_(x);
    const float sw = s * (1.f + x * (1.f - s));
    return act == OE_ACT_SWISH ? sw : (act == OE_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : 1.f);
}
struct F6 { bf16x8 p[3]; };

// Diagnostic build only (-DOE_GEMM_STAMPS, tools/ffn6_stamps.py, tools/rowgemm6_stamps.py): s_memtime stamps of the first and the last
// block's waves at the phase boundaries, written to a (2 x 8 x 128)-word buffer nothing else reads.  No stamp exists in the shipped library.
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* f6_stamp_buf = nullptr;
extern "C" int oe_ffn6_set_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(f6_stamp_buf), &buf, sizeof(buf));
}
#define F6_STAMP(slot)                                                                                                   \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        unsigned long long t_;                                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (f6_stamp_buf && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (threadIdx.x & 63) == 0 && (slot) < 128)     \
            f6_stamp_buf[(blockIdx.x == 0 ? 0 : 1024) + (threadIdx.x >> 6) * 128 + (slot)] = t_;                        \
    } while (0)
#else
#define F6_STAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ int f6_acc_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
__device__ __forceinline__ void f6_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void f6_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Epilogue 1 of one 32 x 32 h^T tile (fc on the accumulator rows, row m on the lane) - a function with BY-VALUE arguments, not a
// lambda: captured by reference, the prefetched pre-activation rows were kept in memory (a private array promoted to LDS, each load
// waited for with vmcnt(0) and parked there).  q0..q3: forward = the bias of the tile's columns 8 g + 4 lk .. + 3 (g = 0..3);
// backward = rows 8 ps + (lane >> 3), columns 4 (lane & 7) .. + 3 of the forward's pre-activation tile (ps = 0..3).
// hv: the finished tile (activation x mask / dH) in accumulator order, for f6_write_planes.
template <bool BWD, int NOUT>
__device__ __forceinline__ void f6_epilogue1(const Ffn6Params& p, const f32x16& hacc, float4 q0, float4 q1, float4 q2, float4 q3, long mrow, int ft,
                                             int lane, float* patch, unsigned long long seed_in, const DropParams& dp_in, float (&hv)[16]) {
    const int lq = lane & 31, lk = lane >> 5;
    auto store_tile = [&](float* out) {                              // the tile's 32 x 32 block of a (rows, ff) tensor as 128-byte row segments
        f6_wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[lq * 36 + f6_acc_row(r, lk)] = hv[r];
        f6_wave_sync();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            if (mrow + row < p.rows)
                *reinterpret_cast<float4*>(out + (mrow + row) * p.ff + ft * 32 + c4) = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
        }
    };
    if (!BWD) {
        hv[0] = hacc[0] + q0.x; hv[1] = hacc[1] + q0.y; hv[2] = hacc[2] + q0.z; hv[3] = hacc[3] + q0.w;
        hv[4] = hacc[4] + q1.x; hv[5] = hacc[5] + q1.y; hv[6] = hacc[6] + q1.z; hv[7] = hacc[7] + q1.w;
        hv[8] = hacc[8] + q2.x; hv[9] = hacc[9] + q2.y; hv[10] = hacc[10] + q2.z; hv[11] = hacc[11] + q2.w;
        hv[12] = hacc[12] + q3.x; hv[13] = hacc[13] + q3.y; hv[14] = hacc[14] + q3.z; hv[15] = hacc[15] + q3.w;
        if (NOUT >= 1 && p.pre) store_tile(p.pre);
        if (p.act == OE_ACT_SWISH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] *= sigmoidf_(hv[r]);
        } else if (p.act == OE_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = fmaxf(hv[r], 0.f);
        }
    } else {
        // the pre-activation tile, coalesced into the patch and read back transposed
        f6_wave_sync();
        float* pr = &patch[(lane >> 3) * 36 + (lane & 7) * 4];
        *reinterpret_cast<float4*>(pr) = q0;
        *reinterpret_cast<float4*>(pr + 8 * 36) = q1;
        *reinterpret_cast<float4*>(pr + 16 * 36) = q2;
        *reinterpret_cast<float4*>(pr + 24 * 36) = q3;
        f6_wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = hacc[r] * f6_act_bwd(p.act, patch[lq * 36 + f6_acc_row(r, lk)]);
    }
    if (p.p_in > 0.f) {
        const unsigned long long e0 = (unsigned long long)(mrow + lq) * p.ff + ft * 32 + 4 * lk;     // element (m, fc) of the (rows, ff) tensor
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uint2 h = drop_hash4(seed_in, (e0 + 8 * g4) >> 2);
            hv[4 * g4] *= drop_field(h.x, 0, dp_in); hv[4 * g4 + 1] *= drop_field(h.x, 1, dp_in);
            hv[4 * g4 + 2] *= drop_field(h.y, 0, dp_in); hv[4 * g4 + 3] *= drop_field(h.y, 1, dp_in);
        }
    }
    if ((BWD || NOUT == 2) && p.act_out) store_tile(p.act_out);
}
// three planes of a finished tile, as the two B fragments (k-steps s = 0, 1) of GEMM 2: lane-linear 16-byte pieces.  Only after
// EVERY row tile's patch traffic is over: the patch lives in the same LDS region as these planes.
__device__ __forceinline__ void f6_write_planes(const float (&hv)[16], unsigned char* hdst, int s_stride, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = hv[8 * s + e];
        oe_bf16x8 pl[3];
        oe_split8<3>(v, pl);
#pragma unroll
        for (int pn = 0; pn < 3; ++pn) *reinterpret_cast<oe_bf16x8*>(hdst + s * s_stride + pn * 1024 + lane * 16) = pl[pn];
    }
}

// D = model width, RT = 32-row tiles per block, BWD = the input gradient, NOUT = (rows, ff) tensors written per tile (fwd: 0 / 1 / 2),
// NG = wave groups.  NG = 2 (RT = 1): eight waves, two per SIMD; group g walks the chunks g, g + 2, ... with an h buffer of its own
// and runs ONE BARRIER BEHIND the other group, so that in every interval between two barriers one group is in its matrix phase
// (GEMM 2 of a chunk + GEMM 1 of its next) while the other is in its vector phase (epilogue 1): with one wave per SIMD the
// epilogues (a quarter of the block's cycles, tools/ffn6_stamps.py) and every wait on the weight stream stand in front of the
// matrix pipe; with two co-resident waves of opposite phase the hardware interleaves them.  The groups' partial y^T meet in LDS.
// The LayerNorm backward in front of a row-block kernel as the way its 32 rows reach LDS (eight waves, D = 256):
//     dx = add + LN'(dy; x, stats, gamma),   g = g_alpha * dropmask(g_seed) * g_rowmask * dx
// are oe_layernorm_bwd_dx_drop's two outputs, element for element (same per-lane float4 order, same wave sums, same mask words); a
// wave owns rows wv, wv + 8, wv + 16, wv + 24 of the block exactly as a wave of layernorm_bwd_kernel owns whole rows.  dx and g
// are written (the residual stream and the weight gradient read them), g is split into the block's three planes at xs, and the
// per-block partial sums of the LayerNorm's parameter gradients go to q.ws in layernorm_bwd_kernel's layout (one slot per 16
// rows) for the same reduction launch.  `red`: 32 KiB of LDS nothing else uses yet ([8 waves][2 slots][2][D] floats).  Contains
// one __syncthreads(); the caller's own barrier behind the planes must follow.
// The LayerNorm FORWARD in front of a row-block kernel (the pre-norm of a residual block, encoder_layer.py:79-80, 86-87, 92-93, 103-104)
// as the way its 32 rows reach LDS: y = (x - mean) rstd gamma + beta per row, a wave per row exactly as layernorm_fwd_kernel (same
// two-pass statistics, same per-lane order); y and the (mean, rstd) pairs are written (the weight gradient and the LayerNorm's backward
// read them), y is split into the block's planes.  rowmask: rows with 0 give y = 0 (convolution.py:88-89).  No barrier of its own.
template <int D, bool PAIR>
__device__ __forceinline__ void ln_fwd_rows_to_planes(const LnfPro& q, int rows, long m0, int wv, int lane, unsigned char* xs) {
    static_assert(D == 256, "one float4 per lane and row");
    constexpr int BM = 32, XP = D + 8;
    const float4 g = reinterpret_cast<const float4*>(q.gamma)[lane], bb = reinterpret_cast<const float4*>(q.beta)[lane];
    float4 g2 = g, bb2 = bb;
    if (PAIR) { g2 = reinterpret_cast<const float4*>(q.gamma2)[lane]; bb2 = reinterpret_cast<const float4*>(q.beta2)[lane]; }
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(q.x + min(m0 + wv + 8 * k, (long)rows - 1) * D)[lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = wv + 8 * k;
        const long grow = m0 + row;
        const bool valid = grow < rows;
        const float mean = wave_sum(v[k].x + v[k].y + v[k].z + v[k].w) / D;
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, e = v[k].w - mean;
        const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + e * e) / D + q.eps);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(valid && q.rowmask && !q.rowmask[grow]))
            o = make_float4((v[k].x - mean) * rstd * g.x + bb.x, (v[k].y - mean) * rstd * g.y + bb.y,
                            (v[k].z - mean) * rstd * g.z + bb.z, (v[k].w - mean) * rstd * g.w + bb.w);
        if (valid && lane == 0) { q.stats[grow * 2] = mean; q.stats[grow * 2 + 1] = rstd; }
        if (PAIR) {
            // the second norm on the first one's output (layernorm_fwd_kernel's PAIR): o = LN2(u); u itself leaves too when wanted
            if (valid && q.u) reinterpret_cast<float4*>(q.u + grow * D)[lane] = o;
            const float mean2 = wave_sum(o.x + o.y + o.z + o.w) / D;
            const float a2 = o.x - mean2, b2 = o.y - mean2, c2 = o.z - mean2, e2 = o.w - mean2;
            const float rstd2 = rsqrtf(wave_sum(a2 * a2 + b2 * b2 + c2 * c2 + e2 * e2) / D + q.eps2);
            if (valid && lane == 0) { q.stats2[grow * 2] = mean2; q.stats2[grow * 2 + 1] = rstd2; }
            o = make_float4((o.x - mean2) * rstd2 * g2.x + bb2.x, (o.y - mean2) * rstd2 * g2.y + bb2.y,
                            (o.z - mean2) * rstd2 * g2.z + bb2.z, (o.w - mean2) * rstd2 * g2.w + bb2.w);
        }
        if (valid) reinterpret_cast<float4*>(q.y + grow * D)[lane] = o;
        const float xq[4] = {o.x, o.y, o.z, o.w};
        oe_bf16x4v pl[3];
        oe_split4<3>(xq, pl);
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4v*>(xs + ((size_t)(n * BM + row) * XP + 4 * lane) * 2) = pl[n];
    }
}

// PAIR (layernorm_bwd_kernel's PAIR): two norms back to back, y2 = LN2(u), u = LN1(x) - dy is the gradient of y2, `add` the gradient
// that reaches u on its other path, u is recomputed from x and LN1's statistics (q.stats, q.gamma, q.beta); the row first goes
// through LN2's backward (q.gamma2, q.stats2; its parameter partials to q.ws2), the result through LN1's.
template <int D, bool PAIR>
__device__ __forceinline__ void ln_bwd_rows_to_planes(const LnPro& q, const unsigned long long* seed_dev, int rows, long m0, int wv, int lane,
                                                      unsigned char* xs, float* red) {
    static_assert(D == 256, "one float4 per lane and row");
    constexpr int BM = 32, XP = D + 8;
    constexpr int NV = D / 256;
    const float4* g4 = reinterpret_cast<const float4*>(q.gamma);
    float4 gam[NV], dg[2][NV], db[2][NV];
    float4 bet[PAIR ? NV : 1], gam2[PAIR ? NV : 1], dg2[2][PAIR ? NV : 1], db2[2][PAIR ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gam[j] = g4[lane + 64 * j];
        dg[0][j] = dg[1][j] = db[0][j] = db[1][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PAIR) {
            bet[j] = reinterpret_cast<const float4*>(q.beta)[lane + 64 * j];
            gam2[j] = reinterpret_cast<const float4*>(q.gamma2)[lane + 64 * j];
            dg2[0][j] = dg2[1][j] = db2[0][j] = db2[1][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const DropParams g_dpar = drop_params(q.g_p);
    const unsigned long long g_seed_eff = q.g_seed + (seed_dev ? *seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    // this wave's four rows: wv, wv + 8 (slot 0 of the partial sums), wv + 16, wv + 24 (slot 1); all their loads first
    float4 dyv[4][NV], xv[4][NV], av[4][NV];
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long rc = min(m0 + wv + 8 * k, (long)rows - 1);
        mean[k] = q.stats[rc * 2];
        rstd[k] = q.stats[rc * 2 + 1];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            dyv[k][j] = reinterpret_cast<const float4*>(q.dy + rc * D)[i];
            xv[k][j] = reinterpret_cast<const float4*>(q.x + rc * D)[i];
            av[k][j] = q.add ? reinterpret_cast<const float4*>(q.add + rc * D)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = wv + 8 * k;
        const long grow = m0 + row;
        const bool valid = grow < rows;
        const bool live = valid && !(q.ln_rowmask && !q.ln_rowmask[grow]);      // a masked row: dx = add, nothing for gamma / beta
        if (PAIR) {
            // LN2's backward on this row first: u = LN1(x) recomputed, dyv <- add + LN2'(dy); `add` is used up
            const long rc = min(grow, (long)rows - 1);
            const float mean2 = q.stats2[rc * 2], rstd2 = q.stats2[rc * 2 + 1];
            float4 g2[NV], xh2[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float4 v = xv[k][j];
                const float4 u = make_float4((v.x - mean[k]) * rstd[k] * gam[j].x + bet[j].x, (v.y - mean[k]) * rstd[k] * gam[j].y + bet[j].y,
                                             (v.z - mean[k]) * rstd[k] * gam[j].z + bet[j].z, (v.w - mean[k]) * rstd[k] * gam[j].w + bet[j].w);
                xh2[j] = make_float4((u.x - mean2) * rstd2, (u.y - mean2) * rstd2, (u.z - mean2) * rstd2, (u.w - mean2) * rstd2);
                float4 t = dyv[k][j];
                if (!live) { t = make_float4(0.f, 0.f, 0.f, 0.f); xh2[j] = t; }
                g2[j] = make_float4(t.x * gam2[j].x, t.y * gam2[j].y, t.z * gam2[j].z, t.w * gam2[j].w);
                s1 += g2[j].x + g2[j].y + g2[j].z + g2[j].w;
                s2 += g2[j].x * xh2[j].x + g2[j].y * xh2[j].y + g2[j].z * xh2[j].z + g2[j].w * xh2[j].w;
                float4& dgs = dg2[k >> 1][j];
                float4& dbs = db2[k >> 1][j];
                dgs.x += t.x * xh2[j].x; dgs.y += t.y * xh2[j].y; dgs.z += t.z * xh2[j].z; dgs.w += t.w * xh2[j].w;
                dbs.x += t.x; dbs.y += t.y; dbs.z += t.z; dbs.w += t.w;
            }
            const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
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
        float4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 t = dyv[k][j];
            const float4 v = xv[k][j];
            xh[j] = make_float4((v.x - mean[k]) * rstd[k], (v.y - mean[k]) * rstd[k], (v.z - mean[k]) * rstd[k], (v.w - mean[k]) * rstd[k]);
            if (!live) { t = make_float4(0.f, 0.f, 0.f, 0.f); xh[j] = t; }
            g[j] = make_float4(t.x * gam[j].x, t.y * gam[j].y, t.z * gam[j].z, t.w * gam[j].w);
            s1 += g[j].x + g[j].y + g[j].z + g[j].w;
            s2 += g[j].x * xh[j].x + g[j].y * xh[j].y + g[j].z * xh[j].z + g[j].w * xh[j].w;
            float4& dgs = dg[k >> 1][j];
            float4& dbs = db[k >> 1][j];
            dgs.x += t.x * xh[j].x; dgs.y += t.y * xh[j].y; dgs.z += t.z * xh[j].z; dgs.w += t.w * xh[j].w;
            dbs.x += t.x; dbs.y += t.y; dbs.z += t.z; dbs.w += t.w;
        }
        const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            float4 o = av[k][j];
            if (live) {
                o.x += rstd[k] * (g[j].x - c1 - xh[j].x * c2);
                o.y += rstd[k] * (g[j].y - c1 - xh[j].y * c2);
                o.z += rstd[k] * (g[j].z - c1 - xh[j].z * c2);
                o.w += rstd[k] * (g[j].w - c1 - xh[j].w * c2);
            }
            float4 gq = make_float4(o.x * q.g_alpha, o.y * q.g_alpha, o.z * q.g_alpha, o.w * q.g_alpha);
            const unsigned long long e0 = (unsigned long long)grow * D + 4 * i;
            if (q.g_p > 0.f) {
                const uint4 r = drop_words8(g_seed_eff, e0 >> 3);
                const bool hi = (e0 >> 2) & 1;                 // second half of the call's eight fields
                const unsigned wa = hi ? r.z : r.x, wb = hi ? r.w : r.y;
                gq.x *= drop_field(wa, 0, g_dpar); gq.y *= drop_field(wa, 1, g_dpar);
                gq.z *= drop_field(wb, 0, g_dpar); gq.w *= drop_field(wb, 1, g_dpar);
            }
            if (valid && q.g_rowmask && !q.g_rowmask[grow]) gq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                reinterpret_cast<float4*>(q.dx + grow * D)[i] = o;
                reinterpret_cast<float4*>(q.g + grow * D)[i] = gq;
            }
            const float xq[4] = {gq.x, gq.y, gq.z, gq.w};
            oe_bf16x4v pl[3];
            oe_split4<3>(xq, pl);
#pragma unroll
            for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4v*>(xs + ((size_t)(n * BM + row) * XP + 4 * i) * 2) = pl[n];
        }
    }
    // the parameter-gradient partials of the block's two 16-row slots: the waves' sums meet in the (still unused) patches
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            reinterpret_cast<float4*>(red + ((wv * 2 + sl) * 2 + 0) * D)[lane + 64 * j] = dg[sl][j];
            reinterpret_cast<float4*>(red + ((wv * 2 + sl) * 2 + 1) * D)[lane + 64 * j] = db[sl][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * D; c += 512) {                           // c = (slot * 2 + which) * D + column
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * 4 * D + c];
        const long slot = 2 * (long)blockIdx.x + c / (2 * D);
        if (slot * 16 < rows) q.ws[slot * 2 * D + (c % (2 * D))] = sum;
    }
    if (PAIR) {
        __syncthreads();                                                        // everyone has read the first norm's sums
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                reinterpret_cast<float4*>(red + ((wv * 2 + sl) * 2 + 0) * D)[lane + 64 * j] = dg2[sl][j];
                reinterpret_cast<float4*>(red + ((wv * 2 + sl) * 2 + 1) * D)[lane + 64 * j] = db2[sl][j];
            }
        __syncthreads();
        for (int c = threadIdx.x; c < 4 * D; c += 512) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[w * 4 * D + c];
            const long slot = 2 * (long)blockIdx.x + c / (2 * D);
            if (slot * 16 < rows) q.ws2[slot * 2 * D + (c % (2 * D))] = sum;
        }
    }
}

template <int D, int RT, bool BWD, int NOUT, int NG, int LNP = 0>
__global__ __launch_bounds__(256 * NG, (NG == 2 || (RT == 1 && D <= 256)) ? 2 : 1) void ffn6_kernel(Ffn6Params p) {
    static_assert(NG == 1 || (NG == 2 && RT == 1), "two wave groups: 32-row blocks");
    static_assert(LNP == 0 || (D == 256 && RT == 1 && NG == 2 && BWD == (LNP == 1)), "LayerNorm prologues: the eight-wave 32-row kernels at d = 256");
    constexpr int BM = 32 * RT;
    constexpr int KS = D / 16, DT = D / 32, DPW = DT / 4;            // k-steps of GEMM 1, output tiles, output tiles per wave
    constexpr int FR = OE_F6_FR, NSET = OE_F6_NSET;
    constexpr int NS1 = KS / FR, NS2 = DPW * 8 / FR, NSTG = NS1 + NS2;
    static_assert(DT % 4 == 0 && KS % FR == 0 && NSTG % NSET == 0, "stage split");
    constexpr int XP = D + 8;                                        // bf16 elements per x row in LDS
    constexpr int X_BYTES = 3 * BM * XP * 2;
    constexpr int H_WAVE = 2 * RT * 3 * 1024;                        // one wave's h tile: [s][rt][plane][lane][8] bf16
    constexpr int H_BYTES = 4 * H_WAVE;
    static_assert(H_WAVE >= 32 * 36 * 4, "the wave's h region doubles as its 32 x 36 fp32 patch");
    constexpr int PIECE = 3 * 1024;                                  // one packed fragment: three planes
    constexpr int NT = RT * DPW;                                     // output tiles (32 x 32) per wave
    static_assert(NG == 1 || (NT % 2 == 0 && 8 * (NT / 2) * 4096 <= X_BYTES), "the groups' exchange buffer reuses the x planes");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[X_BYTES + NG * H_BYTES];
    unsigned char* xs = lds;

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = NG == 2 ? (wv >> 2) : 0;                         // wave group; `wave` = this wave's role inside it
    const int wave = wv & 3;
    unsigned char* hs = lds + X_BYTES + grp * H_BYTES;
    const int lq = lane & 31, lk = lane >> 5;
    const long m0 = (long)blockIdx.x * BM;
    const int nchunks = p.ff / 128;
    const int rot = blockIdx.x % nchunks;                            // blocks start their walk over the weights at different chunks

    // ---- the block's rows -> three bf16 planes in LDS (rows past the end re-read the last one; never stored)
    if constexpr (LNP == 2) {
        // ... made by the pre-norm in front of this feed-forward (or by norm_final + that pre-norm: the pair form writes u, the
        // residual this kernel's epilogue 2 reads back from global memory - same block, behind the barriers below)
        if (p.lnf.gamma2) ln_fwd_rows_to_planes<D, true>(p.lnf, p.rows, m0, wv, lane, xs);
        else ln_fwd_rows_to_planes<D, false>(p.lnf, p.rows, m0, wv, lane, xs);
    } else if constexpr (LNP == 1) {
        // ... made by the LayerNorm backward that precedes this feed-forward's backward (ln_bwd_rows_to_planes); the partial sums
        // meet in the h regions, which nothing uses before the first chunk's epilogue
        static_assert(LNP != 1 || 8 * 2 * 2 * D * 4 <= NG * H_BYTES, "the prologue's partial sums reuse the h regions");
        if (p.ln.gamma2) ln_bwd_rows_to_planes<D, true>(p.ln, p.seed_dev, p.rows, m0, wv, lane, xs, reinterpret_cast<float*>(lds + X_BYTES));
        else ln_bwd_rows_to_planes<D, false>(p.ln, p.seed_dev, p.rows, m0, wv, lane, xs, reinterpret_cast<float*>(lds + X_BYTES));
    } else {
        constexpr int C4 = D / 4;
        for (int i = threadIdx.x; i < BM * C4; i += 256 * NG) {
            const int row = i / C4, c4 = (i - row * C4) * 4;
            const long gr = min(m0 + row, (long)p.rows - 1);
            const float4 v = *reinterpret_cast<const float4*>(p.x + gr * p.ldx + c4);
            const float xv[4] = {v.x, v.y, v.z, v.w};
            oe_bf16x4v pl[3];
            oe_split4<3>(xv, pl);
#pragma unroll
            for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4v*>(xs + ((size_t)(n * BM + row) * XP + c4) * 2) = pl[n];
        }
    }

    // ---- the weight stream.  Stage w (0 .. NSTG - 1) of a chunk: FR packed fragments - GEMM 1's first (k-steps FR w + j of this
    // wave's ff tile), then GEMM 2's ([dtl][ftl][s] order).  Which fragments a stage holds is a COMPILE-TIME property of w, the
    // chunk only moves two wave-uniform base pointers: no branch, no division in the loop (a runtime stage decode cost 50 branches
    // per chunk and cut the loop into scheduling regions of a dozen MFMAs).
    const unsigned char* w1l = p.w1p + lane * 16 + (long)wave * KS * PIECE;            // + chunk * 4 KS PIECE
    const unsigned char* w2l = p.w2p + lane * 16 + (long)wave * DPW * 2 * PIECE;      // + chunk * 4 DT 2 PIECE
    constexpr long W1_CHUNK = 4L * KS * PIECE, W2_CHUNK = 4L * DT * 2 * PIECE;
    auto load_stage = [&](auto w_c, int c, F6 (&f)[FR]) {
        constexpr int w = decltype(w_c)::value;
        const unsigned char* b1p = w1l + c * W1_CHUNK;
        const unsigned char* b2p = w2l + c * W2_CHUNK;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const unsigned char* src;
            if constexpr (w < NS1) {
                src = b1p + (long)(w * FR + j) * PIECE;
            } else {
                const int fidx = (w - NS1) * FR + j;                                   // [dtl][ftl][s]
                const int dtl = fidx >> 3, ftl = (fidx >> 1) & 3, s = fidx & 1;
                src = b2p + (long)(((ftl * DT + dtl) * 2) + s) * PIECE;
            }
            f[j].p[0] = *reinterpret_cast<const bf16x8*>(src); f[j].p[1] = *reinterpret_cast<const bf16x8*>(src + 1024);
            f[j].p[2] = *reinterpret_cast<const bf16x8*>(src + 2048);
        }
    };
    auto chunk_at = [&](int ci) {                                    // the ci-th chunk of this block's walk (past the end: the last one again)
        int c = min(ci, nchunks - 1) + rot;
        return c >= nchunks ? c - nchunks : c;
    };
    F6 fr[NSET][FR];
    {
        const int c0 = chunk_at(grp);
        static_for<0, NSET - 1>([&](auto k_c) { load_stage(k_c, c0, fr[decltype(k_c)::value]); });
    }

    f32x16 yacc[RT][DPW];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int t = 0; t < DPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[a][t][r] = 0.f;
    const unsigned long long sd = p.seed_dev ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull;
    const unsigned long long seed_in = p.seed_in + sd, seed_out = p.seed_out + sd;
    const DropParams dp_in = drop_params(p.p_in), dp_out = drop_params(p.p_out);
    unsigned char* hmine = hs + wave * H_WAVE;
    float* patch = reinterpret_cast<float*>(hmine);

    // x fragment (B operand of GEMM 1): row 32 rt + lq, features 16 ks + 8 lk .. + 7
    auto x_frag = [&](int rt, int ks, F6& f) {
        const unsigned char* a = xs + ((size_t)(rt * 32 + lq) * XP + 16 * ks + 8 * lk) * 2;
#pragma unroll
        for (int n = 0; n < 3; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(a + (size_t)n * BM * XP * 2);
    };
    F6_STAMP(0);
    __syncthreads();                                                 // x planes (nothing else is in flight in LDS yet)
    F6_STAMP(1);
    if (NG == 2 && grp == 1) f6_lds_barrier();                       // the stagger: group 1 runs one barrier behind group 0

    for (int ci = grp; ci < nchunks; ci += NG) {
        const int c = chunk_at(ci), c_next = chunk_at(ci + NG);
        const int ft = 4 * c + wave;
        // ---- bias / pre-activation of this tile, issued ahead of the product (named registers: see f6_epilogue1)
        float4 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
        qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!BWD) {
            if (p.b1) {
                const float* bp = p.b1 + ft * 32 + 4 * lk;
                qa0 = *reinterpret_cast<const float4*>(bp); qa1 = *reinterpret_cast<const float4*>(bp + 8); qa2 = *reinterpret_cast<const float4*>(bp + 16); qa3 = *reinterpret_cast<const float4*>(bp + 24);
            }
        } else {
            const long col = ft * 32 + (lane & 7) * 4;
            const long last = (long)p.rows - 1;
            const long r0 = m0 + (lane >> 3);
            qa0 = *reinterpret_cast<const float4*>(p.pre + min(r0, last) * p.ff + col);
            qa1 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 8, last) * p.ff + col);
            qa2 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 16, last) * p.ff + col);
            qa3 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 24, last) * p.ff + col);
            if constexpr (RT > 1) {
                qb0 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 32, last) * p.ff + col);
                qb1 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 40, last) * p.ff + col);
                qb2 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 48, last) * p.ff + col);
                qb3 = *reinterpret_cast<const float4*>(p.pre + min(r0 + 56, last) * p.ff + col);
            }
        }
        // ---- GEMM 1: h^T tile = W1[ft] . x^T, both row tiles
        f32x16 hacc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[rt][r] = 0.f;
        F6 xf[2][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) x_frag(rt, 0, xf[0][rt]);
        static_for<0, NS1>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int cu = st % NSET, pf = (st + NSET - 1) % NSET;
            constexpr int ahead = st + NSET - 1;                     // stage prefetched now: unconditional (a branch here makes hipcc wait vmcnt(0))
            load_stage(std::integral_constant<int, ahead % NSTG>{}, ahead < NSTG ? c : c_next, fr[pf]);
            __builtin_amdgcn_sched_barrier(0);                      // nothing crosses: left alone, hipcc sinks each prefetch load to its first use
            //                                                         (shorter live range) and waits vmcnt(0) there - a load-use chain, not a ring
            static_for<0, FR>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                constexpr int ks = st * FR + j;
                if constexpr (ks + 1 < KS) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) x_frag(rt, ks + 1, xf[(ks + 1) & 1][rt]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) hacc[rt] = oe_mma_terms<6>(fr[cu][j], xf[ks & 1][rt], hacc[rt]);
            });
        });
        F6_STAMP(2 + 5 * (ci / NG));                                 // GEMM 1 done
        // every wave of the group has finished reading the previous chunk's h tiles (its GEMM 2) before anyone overwrites them
        // (unconditional: with two groups every barrier is one group's "read" and the other's "written" barrier)
        f6_lds_barrier();
        F6_STAMP(3 + 5 * (ci / NG));
        // ---- epilogue 1, per row tile: first everything that goes through the patch, then the planes (same LDS region)
        float hv0[16], hv1[16];
        f6_epilogue1<BWD, NOUT>(p, hacc[0], qa0, qa1, qa2, qa3, m0, ft, lane, patch, seed_in, dp_in, hv0);
        if constexpr (RT > 1) {
            if (BWD) f6_epilogue1<BWD, NOUT>(p, hacc[1], qb0, qb1, qb2, qb3, m0 + 32, ft, lane, patch, seed_in, dp_in, hv1);
            else f6_epilogue1<BWD, NOUT>(p, hacc[1], qa0, qa1, qa2, qa3, m0 + 32, ft, lane, patch, seed_in, dp_in, hv1);
        }
        f6_wave_sync();
        f6_write_planes(hv0, hmine, RT * 3 * 1024, lane);
        if constexpr (RT > 1) f6_write_planes(hv1, hmine + 3 * 1024, RT * 3 * 1024, lane);
        F6_STAMP(4 + 5 * (ci / NG));                                 // epilogue 1 done
        f6_lds_barrier();                                            // the chunk's four h tiles are in LDS
        F6_STAMP(5 + 5 * (ci / NG));
        // ---- GEMM 2: y^T[this wave's d tiles] += W2[d tiles, chunk] . h^T
        auto h_frag = [&](int ftl, int s, int rt, F6& f) {
            const unsigned char* a = hs + ftl * H_WAVE + ((s * RT + rt) * 3) * 1024 + lane * 16;
#pragma unroll
            for (int pn = 0; pn < 3; ++pn) f.p[pn] = *reinterpret_cast<const bf16x8*>(a + pn * 1024);
        };
        F6 hf[2][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) h_frag(0, 0, rt, hf[0][rt]);
        static_for<0, NS2>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int cu = (NS1 + st) % NSET, pf = (NS1 + st + NSET - 1) % NSET;
            constexpr int ahead = NS1 + st + NSET - 1;
            load_stage(std::integral_constant<int, ahead % NSTG>{}, ahead < NSTG ? c : c_next, fr[pf]);
            __builtin_amdgcn_sched_barrier(0);                      // nothing crosses: left alone, hipcc sinks each prefetch load to its first use
            //                                                         (shorter live range) and waits vmcnt(0) there - a load-use chain, not a ring
            static_for<0, FR>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                constexpr int fidx = st * FR + j;
                constexpr int dtl = fidx >> 3;
                if constexpr (fidx + 1 < DPW * 8) {
                    constexpr int nx = fidx + 1;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) h_frag((nx >> 1) & 3, nx & 1, rt, hf[nx & 1][rt]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) yacc[rt][dtl] = oe_mma_terms<6>(fr[cu][j], hf[fidx & 1][rt], yacc[rt][dtl]);
            });
        });
        F6_STAMP(6 + 5 * (ci / NG));                                 // GEMM 2 done
    }
    F6_STAMP(2 + 5 * (nchunks / NG));                                // last GEMM 2 done
    if (NG == 2 && grp == 0) f6_lds_barrier();                       // (group 1 made this one at its start)
    f6_lds_barrier();                                                // the last chunk's h tiles have been read: the regions are patches now

    // ---- two groups: each wave hands the tiles its partner (same role, other group) finishes to it through LDS (the x planes are
    // dead by now) and adds the partner's share of its own: tile t belongs to group t % 2
    if constexpr (NG == 2) {
        float* ex = reinterpret_cast<float*>(xs);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if ((t & 1) != grp) {
                float* dst = ex + (size_t)((grp * 4 + wave) * (NT / 2) + (t >> 1)) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r * 64] = yacc[t / DPW][t % DPW][r];
            }
        f6_lds_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if ((t & 1) == grp) {
                const float* src = ex + (size_t)(((1 - grp) * 4 + wave) * (NT / 2) + (t >> 1)) * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) yacc[t / DPW][t % DPW][r] += src[r * 64];
            }
    }

    // ---- epilogue 2: this wave's 32 DPW output columns of every row tile (two groups: the tiles of its own parity)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int dtl = 0; dtl < DPW; ++dtl) {
            if (NG == 2 && ((rt * DPW + dtl) & 1) != grp) continue;
            const int dt = wave * DPW + dtl;
            f6_wave_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[lq * 36 + f6_acc_row(r, lk)] = yacc[rt][dtl][r];
            f6_wave_sync();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                const long gr = m0 + rt * 32 + row;
                if (gr >= p.rows) continue;
                const int col = dt * 32 + c4;
                float4 v = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
                if (p.b2) { v.x += p.b2[col]; v.y += p.b2[col + 1]; v.z += p.b2[col + 2]; v.w += p.b2[col + 3]; }
                if (p.p_out > 0.f) {
                    const uint2 h = drop_hash4(seed_out, ((unsigned long long)gr * D + col) >> 2);
                    v.x *= drop_field(h.x, 0, dp_out); v.y *= drop_field(h.x, 1, dp_out);
                    v.z *= drop_field(h.y, 0, dp_out); v.w *= drop_field(h.y, 1, dp_out);
                }
                float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.residual) res = *reinterpret_cast<const float4*>(p.residual + gr * p.ldr + col);
                v = make_float4(res.x + p.beta * v.x, res.y + p.beta * v.y, res.z + p.beta * v.z, res.w + p.beta * v.w);
                *reinterpret_cast<float4*>(p.y + gr * p.ldy + col) = v;
            }
        }
    F6_STAMP(3 + 5 * (nchunks / NG));
}

// ---- one Linear as a row-block GEMM (oe_rowgemm6): the first half of the kernel above on its own ------------------------------
//   out[rows, NO] = residual + beta * rowmask * dropout( in[rows, D] @ Wg[NO, D]^T + bias )
// for the K = d (256 / 512) Linears of an encoder layer (attention.py:56-58,97 linear_q/k/v/out, convolution.py:79-111 pointwise
// convs) and their input gradients.  As tiled GEMM launches these 1-3 GFLOP problems are all ramp: a 128 x 128 block walks eight
// K-tiles between a cold start and an epilogue (23-36 us in the step for 7936 rows, 45-88 TFLOP/s).  Here a block owns 32 rows for
// the WHOLE reduction: the rows' three bf16 planes sit in LDS (split once), each wave computes 32 x 32 output tiles transposed
// (Wg's packed fragments are the A operand, straight L2 -> registers through the four-set ring; the rows are the B operand), the
// epilogue leaves through a wave-private patch as 128-byte row segments.  Eight waves in two groups take alternate 128-column
// chunks: no barrier after the first, the waves drift apart and one's epilogue runs under another's MFMAs.
struct Row6Params {
    const float* x; long ldx;
    const unsigned char* wp; const float* bias;
    const float* residual; long ldr; float beta;
    const unsigned char* rowmask;
    float* y; long ldy;
    int rows, no;
    float p_out; unsigned long long seed_out;
    const unsigned long long* seed_dev;
    int k, act; float* preact_out; const float* actgrad_in; long ld_aux;     // (tile form only)
    LnPro ln;                                    // LayerNorm-backward prologue (row-block form, LNP = 1): the rows of x are MADE
    LnfPro lnf;                                  // LayerNorm-forward prologue (LNP = 2)
    // LayerNorm-backward EPILOGUE (LNE): the product's rows go through the backward of a LayerNorm (+ activation) whose output the
    // forward GEMM consumed; le_dx receives the result instead of y
    const float* le_x; const float* le_stats; const float* le_gamma; const float* le_beta; int le_act; float* le_dx; float* le_ws;
};

// LNP (LayerNorm-backward prologue): the GEMM's input rows are the gradient a pre-norm residual block's backward STARTS from,
//     g = g_alpha * dropmask(g_seed) * dx,   dx = ln_add + LN'(ln_dy; ln_x, ln_stats, ln_gamma)
// (oe_layernorm_bwd_dx_drop's two outputs, element for element: same per-lane float4 order, same wave sums, same mask words).  A
// block owns 32 whole rows and a wave owns four of them, exactly as in layernorm_bwd_kernel, so the LayerNorm backward that used
// to be a launch of its own in front of this GEMM (10.8 us + a dependent-launch gap, 24 times per step at config 2) becomes the way
// the rows reach LDS: dx and g are written (the residual stream and the weight gradient read them), g is split into the planes, and
// the per-block partial sums of the LayerNorm's parameter gradients go to ln_ws in layernorm_bwd_kernel's layout (one slot per 16
// rows: rows 0-15 and 16-31 of the block) for the same reduction launch.
// LNE (LayerNorm-backward epilogue, n = 256): the product is the gradient of z = act(LN(yc)) - the conv module's norm + activation
// between the depthwise convolution and pointwise_conv2 (convolution.py:107-111), whose input gradient this launch computes - and
// what the next kernel wants is the gradient of yc: the block owns its 32 rows whole (eight waves x 32 columns), so the LayerNorm
// backward runs on the accumulators' way out: t = dz act'(xh gamma + beta), g = t gamma, row sums of g and g xh across the eight
// waves through LDS (one barrier), dx = rstd (g - c1 - xh c2); the parameter-gradient partials go to le_ws in layernorm_bwd_kernel's
// layout.  dz itself is never written.
template <int D, int LNP = 0, bool LNE = false>
__global__ __launch_bounds__(512, 2) void rowgemm6_kernel(Row6Params p) {
    constexpr int BM = 32, NG = 2;
    constexpr int KS = D / 16;
    constexpr int FR = OE_F6_FR, NSET = OE_F6_NSET;
    constexpr int NSTG = KS / FR;
    static_assert(KS % FR == 0 && NSTG % NSET == 0, "stage split");
    constexpr int XP = D + 8;
    constexpr int X_BYTES = 3 * BM * XP * 2;
    constexpr int PATCH = 32 * 36 * 4;
    constexpr int PIECE = 3 * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[X_BYTES + 8 * PATCH + (LNE ? 8 * 32 * 2 * 4 : 0)];
    unsigned char* xs = lds;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wv >> 2, wave = wv & 3;
    const int lq = lane & 31, lk = lane >> 5;
    const long m0 = (long)blockIdx.x * BM;
    const int nchunks = p.no / 128;
    const int rot = blockIdx.x % nchunks;
    float* patch = reinterpret_cast<float*>(lds + X_BYTES + wv * PATCH);
    const unsigned char* wl = p.wp + lane * 16 + (long)wave * KS * PIECE;
    constexpr long W_CHUNK = 4L * KS * PIECE;
    auto load_stage = [&](auto w_c, int c, F6 (&f)[FR]) {
        constexpr int w = decltype(w_c)::value;
        const unsigned char* b1p = wl + c * W_CHUNK;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const unsigned char* src = b1p + (long)(w * FR + j) * PIECE;
            f[j].p[0] = *reinterpret_cast<const bf16x8*>(src); f[j].p[1] = *reinterpret_cast<const bf16x8*>(src + 1024);
            f[j].p[2] = *reinterpret_cast<const bf16x8*>(src + 2048);
        }
    };
    auto chunk_at = [&](int ci) {
        int c = min(ci, nchunks - 1) + rot;
        return c >= nchunks ? c - nchunks : c;
    };
    F6 fr[NSET][FR];
    F6_STAMP(0);
    {
        const int c0 = chunk_at(grp);
        static_for<0, NSET - 1>([&](auto k_c) { load_stage(k_c, c0, fr[decltype(k_c)::value]); });
    }
    // (the rows after the ring's first stages are on their way: both round trips overlap)
    if constexpr (LNP == 2) {
        ln_fwd_rows_to_planes<D, false>(p.lnf, p.rows, m0, wv, lane, xs);
    } else if constexpr (LNP == 1) {
        static_assert(8 * 2 * 2 * D * 4 <= 8 * PATCH, "the prologue's partial sums reuse the (still unused) patches");
        if (p.ln.gamma2) ln_bwd_rows_to_planes<D, true>(p.ln, p.seed_dev, p.rows, m0, wv, lane, xs, reinterpret_cast<float*>(lds + X_BYTES));
        else ln_bwd_rows_to_planes<D, false>(p.ln, p.seed_dev, p.rows, m0, wv, lane, xs, reinterpret_cast<float*>(lds + X_BYTES));
    } else {
        constexpr int C4 = D / 4;
        for (int i = threadIdx.x; i < BM * C4; i += 512) {
            const int row = i / C4, c4 = (i - row * C4) * 4;
            const long gr = min(m0 + row, (long)p.rows - 1);
            const float4 v = *reinterpret_cast<const float4*>(p.x + gr * p.ldx + c4);
            const float xv[4] = {v.x, v.y, v.z, v.w};
            oe_bf16x4v pl[3];
            oe_split4<3>(xv, pl);
#pragma unroll
            for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4v*>(xs + ((size_t)(n * BM + row) * XP + c4) * 2) = pl[n];
        }
    }
    const unsigned long long seed_out = p.seed_out + (p.seed_dev ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const DropParams dp_out = drop_params(p.p_out);
    auto x_frag = [&](int ks, F6& f) {
        const unsigned char* a = xs + ((size_t)lq * XP + 16 * ks + 8 * lk) * 2;
#pragma unroll
        for (int n = 0; n < 3; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(a + (size_t)n * BM * XP * 2);
    };
    F6_STAMP(1);
    __syncthreads();                                                 // the rows' planes: the only block-wide barrier
    F6_STAMP(2);

    for (int ci = grp; ci < nchunks; ci += NG) {
        const int c = chunk_at(ci), c_next = chunk_at(ci + NG);
        const int ft = 4 * c + wave;                                 // this wave's 32 output columns: 32 ft ..
        float4 q0, q1, q2, q3;
        q0 = q1 = q2 = q3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            const float* bp = p.bias + ft * 32 + 4 * lk;
            q0 = *reinterpret_cast<const float4*>(bp); q1 = *reinterpret_cast<const float4*>(bp + 8);
            q2 = *reinterpret_cast<const float4*>(bp + 16); q3 = *reinterpret_cast<const float4*>(bp + 24);
        }
        // the residual's row segments of this tile, ahead of the product (their round trip would stand in front of the stores)
        float4 r0, r1, r2, r3;
        r0 = r1 = r2 = r3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual) {
            const long last = (long)p.rows - 1;
            const float* rp = p.residual + ft * 32 + (lane & 7) * 4;
            const long rr = m0 + (lane >> 3);
            r0 = *reinterpret_cast<const float4*>(rp + min(rr, last) * p.ldr); r1 = *reinterpret_cast<const float4*>(rp + min(rr + 8, last) * p.ldr);
            r2 = *reinterpret_cast<const float4*>(rp + min(rr + 16, last) * p.ldr); r3 = *reinterpret_cast<const float4*>(rp + min(rr + 24, last) * p.ldr);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        F6 xf[2];
        x_frag(0, xf[0]);
        static_for<0, NSTG>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int cu = st % NSET, pf = (st + NSET - 1) % NSET;
            constexpr int ahead = st + NSET - 1;
            load_stage(std::integral_constant<int, ahead % NSTG>{}, ahead < NSTG ? c : c_next, fr[pf]);
            __builtin_amdgcn_sched_barrier(0);                      // (the prefetch stays here: ffn6_kernel)
            static_for<0, FR>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                constexpr int ks = st * FR + j;
                if constexpr (ks + 1 < KS) x_frag(ks + 1, xf[(ks + 1) & 1]);
                acc = oe_mma_terms<6>(fr[cu][j], xf[ks & 1], acc);
            });
        });
        F6_STAMP(3 + 3 * (ci / NG));
        // ---- epilogue: + bias on the accumulators (output column on the rows), then row segments through the patch
        float hv[16];
        hv[0] = acc[0] + q0.x; hv[1] = acc[1] + q0.y; hv[2] = acc[2] + q0.z; hv[3] = acc[3] + q0.w;
        hv[4] = acc[4] + q1.x; hv[5] = acc[5] + q1.y; hv[6] = acc[6] + q1.z; hv[7] = acc[7] + q1.w;
        hv[8] = acc[8] + q2.x; hv[9] = acc[9] + q2.y; hv[10] = acc[10] + q2.z; hv[11] = acc[11] + q2.w;
        hv[12] = acc[12] + q3.x; hv[13] = acc[13] + q3.y; hv[14] = acc[14] + q3.z; hv[15] = acc[15] + q3.w;
        f6_wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[lq * 36 + f6_acc_row(r, lk)] = hv[r];
        f6_wave_sync();
        F6_STAMP(4 + 3 * (ci / NG));
        if constexpr (LNE) {
            static_assert(D == 256, "row sums over eight waves x 32 columns");
            // Every wave of the block is past its MFMA loop before any of them starts this epilogue: a second line of defence against the
            // packed-fp32 hazard described at `xh` below (the barrier alone also hides it - that is how it was first worked around - and
            // costs nothing measurable; which other v_pk_* pairs could be hit next to another wave's MFMAs is not characterised).
#ifndef OE_LNE_REPRO
            __syncthreads();
#endif
            float (*rowsum)[32][2] = reinterpret_cast<float (*)[32][2]>(lds + X_BYTES + 8 * PATCH);     // [8 waves][32 rows][2]
            const int c4 = (lane & 7) * 4, col = ft * 32 + c4;
            const float4 gm = *reinterpret_cast<const float4*>(p.le_gamma + col), bt = *reinterpret_cast<const float4*>(p.le_beta + col);
            float4 tt[4], xh[4], gg[4];
            float rs[4];
            float4 dgs[2], dbs[2];
            dgs[0] = dgs[1] = dbs[0] = dbs[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3);
                const long gr = m0 + row;
                const bool valid = gr < p.rows;
                const long rc = min(gr, (long)p.rows - 1);
                const float4 v = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
                const float4 xv = *reinterpret_cast<const float4*>(p.le_x + rc * D + col);
                const float mean = p.le_stats[rc * 2], rstd = p.le_stats[rc * 2 + 1];
                rs[ps] = rstd;
#ifndef OE_LNE_REPRO
                // (x - mean) rstd on ONE-FLOAT instructions.  What hipcc makes of the plain expression - v_pk_add_f32 (x.xy - mean, op_sel_hi /
                // neg) and, two instructions later, v_pk_mul_f32 (.. x rstd, op_sel:[0,1]) on the register pair the statistics were loaded
                // into - returned +-0 in the LOW half (.x, .z) for lanes 48-63 of the first pass whenever the other wave of the SIMD was
                // still issuing MFMAs: rows 6 and 7 of a block wrong in ~10 % of the blocks, different ones every launch; inputs verified
                // right in the same run, the high halves always right (tools/probes/dbg_lne.py with -DOE_LNE_REPRO, profiles/r04_experiments.md).
                {
                    float e[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm("v_sub_f32 %0, %1, %2" : "=v"(e[q]) : "v"(e[q]), "v"(mean));
                        asm("v_mul_f32 %0, %1, %2" : "=v"(e[q]) : "v"(e[q]), "v"(rstd));
                    }
                    xh[ps] = make_float4(e[0], e[1], e[2], e[3]);
                }
#else
                xh[ps] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
#endif
                float4 t = v;
                if (p.le_act) {
                    t.x *= f6_act_bwd(p.le_act, xh[ps].x * gm.x + bt.x); t.y *= f6_act_bwd(p.le_act, xh[ps].y * gm.y + bt.y);
                    t.z *= f6_act_bwd(p.le_act, xh[ps].z * gm.z + bt.z); t.w *= f6_act_bwd(p.le_act, xh[ps].w * gm.w + bt.w);
                }
                if (!valid) { t = make_float4(0.f, 0.f, 0.f, 0.f); xh[ps] = t; }
                tt[ps] = t;
                gg[ps] = make_float4(t.x * gm.x, t.y * gm.y, t.z * gm.z, t.w * gm.w);
                float s1 = gg[ps].x + gg[ps].y + gg[ps].z + gg[ps].w;
                float s2 = gg[ps].x * xh[ps].x + gg[ps].y * xh[ps].y + gg[ps].z * xh[ps].z + gg[ps].w * xh[ps].w;
                s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
                s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
                s1 += __shfl_xor(s1, 4, 64); s2 += __shfl_xor(s2, 4, 64);
                if ((lane & 7) == 0) { rowsum[wv][row][0] = s1; rowsum[wv][row][1] = s2; }
                float4& dgq = dgs[ps >> 1];
                float4& dbq = dbs[ps >> 1];
                dgq.x += t.x * xh[ps].x; dgq.y += t.y * xh[ps].y; dgq.z += t.z * xh[ps].z; dgq.w += t.w * xh[ps].w;
                dbq.x += t.x; dbq.y += t.y; dbq.z += t.z; dbq.w += t.w;
            }
            __syncthreads();                                         // (n = 256: every wave runs this body exactly once)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3);
                const long gr = m0 + row;
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { c1 += rowsum[w][row][0]; c2 += rowsum[w][row][1]; }
                c1 /= D; c2 /= D;
                const float4 o = make_float4(rs[ps] * (gg[ps].x - c1 - xh[ps].x * c2), rs[ps] * (gg[ps].y - c1 - xh[ps].y * c2),
                                             rs[ps] * (gg[ps].z - c1 - xh[ps].z * c2), rs[ps] * (gg[ps].w - c1 - xh[ps].w * c2));
                if (gr < p.rows) *reinterpret_cast<float4*>(p.le_dx + gr * D + col) = o;
            }
            // parameter-gradient partials: this wave's 32 columns over the block's two 16-row slots
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                float4 a = dgs[sl], b = dbs[sl];
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
                    b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
                }
                const long slot = 2 * (long)blockIdx.x + sl;
                if (lane < 8 && slot * 16 < p.rows) {
                    *reinterpret_cast<float4*>(p.le_ws + slot * 2 * D + col) = a;
                    *reinterpret_cast<float4*>(p.le_ws + slot * 2 * D + D + col) = b;
                }
            }
            F6_STAMP(5 + 3 * (ci / NG));
            continue;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            const long gr = m0 + row;
            if (gr >= p.rows) continue;
            const int col = ft * 32 + c4;
            float4 v = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
            if (p.p_out > 0.f) {
                const uint2 h = drop_hash4(seed_out, ((unsigned long long)gr * p.no + col) >> 2);
                v.x *= drop_field(h.x, 0, dp_out); v.y *= drop_field(h.x, 1, dp_out);
                v.z *= drop_field(h.y, 0, dp_out); v.w *= drop_field(h.y, 1, dp_out);
            }
            if (p.rowmask && !p.rowmask[gr]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.residual || p.beta != 1.f) {
                const float4 res = ps == 0 ? r0 : ps == 1 ? r1 : ps == 2 ? r2 : r3;
                v = make_float4(res.x + p.beta * v.x, res.y + p.beta * v.y, res.z + p.beta * v.z, res.w + p.beta * v.w);
            }
            *reinterpret_cast<float4*>(p.y + gr * p.ldy + col) = v;
        }
        F6_STAMP(5 + 3 * (ci / NG));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K-PHASED form of the row-block GEMM: k = 256 nph (nph = 3, 4), n <= 256 - the input gradient through the fused q / k / v
// projection (7936 x 256 <- 768: 36.6 us as 248 tiles of 128 x 64 walking 24 K-tiles, 85 TFLOP/s).  The rows of one phase (256
// columns of x) sit in LDS as planes; the next phase's rows are already in registers while this phase's MFMAs run and replace
// them behind a barrier pair; every wave keeps ONE accumulator tile through all phases (hence n <= 256: two groups x four waves);
// the packed weight stream of a column tile is contiguous over the whole reduction, so the register ring runs straight through
// the phase boundaries.
__global__ __launch_bounds__(512, 2) void rowgemm6p_kernel(Row6Params p) {
    constexpr int D = 256, BM = 32;
    constexpr int KS = D / 16;
    constexpr int FR = OE_F6_FR, NSET = OE_F6_NSET;
    constexpr int NSTG = KS / FR;
    constexpr int XP = D + 8;
    constexpr int X_BYTES = 3 * BM * XP * 2;
    constexpr int PATCH = 32 * 36 * 4;
    constexpr int PIECE = 3 * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[X_BYTES + 8 * PATCH];
    unsigned char* xs = lds;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wv >> 2, wave = wv & 3;
    const int lq = lane & 31, lk = lane >> 5;
    const long m0 = (long)blockIdx.x * BM;
    const long last = (long)p.rows - 1;
    const int nchunks = p.no / 128;                                  // 1 or 2 (host check)
    const int nph = p.k / D;
    const int kst = p.k / 16;                                        // pieces per column tile
    const bool active = grp < nchunks;                               // n = 128: the second group only helps with the rows
    const int ft = 4 * (active ? grp : 0) + wave;
    float* patch = reinterpret_cast<float*>(lds + X_BYTES + wv * PATCH);
    const unsigned char* wl = p.wp + lane * 16 + (long)ft * kst * PIECE;
    auto load_stage = [&](auto w_c, int ph, F6 (&f)[FR]) {
        constexpr int w = decltype(w_c)::value;
        const unsigned char* b1p = wl + (long)ph * KS * PIECE;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const unsigned char* src = b1p + (long)(w * FR + j) * PIECE;
            f[j].p[0] = *reinterpret_cast<const bf16x8*>(src); f[j].p[1] = *reinterpret_cast<const bf16x8*>(src + 1024);
            f[j].p[2] = *reinterpret_cast<const bf16x8*>(src + 2048);
        }
    };
    F6 fr[NSET][FR];
    static_for<0, NSET - 1>([&](auto k_c) { load_stage(k_c, 0, fr[decltype(k_c)::value]); });
    // this thread's four float4 of a phase's rows: element i = threadIdx.x + 512 q -> row i / 64, columns 4 (i % 64) ..
    float4 xr[4];
    auto load_x = [&](int ph) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + 512 * q;
            xr[q] = *reinterpret_cast<const float4*>(p.x + min(m0 + (i >> 6), last) * p.ldx + ph * D + (i & 63) * 4);
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + 512 * q;
            const int row = i >> 6, c4 = (i & 63) * 4;
            const float xv[4] = {xr[q].x, xr[q].y, xr[q].z, xr[q].w};
            oe_bf16x4v pl[3];
            oe_split4<3>(xv, pl);
#pragma unroll
            for (int n = 0; n < 3; ++n) *reinterpret_cast<oe_bf16x4v*>(xs + ((size_t)(n * BM + row) * XP + c4) * 2) = pl[n];
        }
    };
    load_x(0);
    const unsigned long long seed_out = p.seed_out + (p.seed_dev ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const DropParams dp_out = drop_params(p.p_out);
    float4 q0, q1, q2, q3;
    q0 = q1 = q2 = q3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
        const float* bp = p.bias + ft * 32 + 4 * lk;
        q0 = *reinterpret_cast<const float4*>(bp); q1 = *reinterpret_cast<const float4*>(bp + 8);
        q2 = *reinterpret_cast<const float4*>(bp + 16); q3 = *reinterpret_cast<const float4*>(bp + 24);
    }
    float4 r0, r1, r2, r3;
    r0 = r1 = r2 = r3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.residual) {
        const float* rp = p.residual + ft * 32 + (lane & 7) * 4;
        const long rr = m0 + (lane >> 3);
        r0 = *reinterpret_cast<const float4*>(rp + min(rr, last) * p.ldr); r1 = *reinterpret_cast<const float4*>(rp + min(rr + 8, last) * p.ldr);
        r2 = *reinterpret_cast<const float4*>(rp + min(rr + 16, last) * p.ldr); r3 = *reinterpret_cast<const float4*>(rp + min(rr + 24, last) * p.ldr);
    }
    auto x_frag = [&](int ks, F6& f) {
        const unsigned char* a = xs + ((size_t)lq * XP + 16 * ks + 8 * lk) * 2;
#pragma unroll
        for (int n = 0; n < 3; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(a + (size_t)n * BM * XP * 2);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ph = 0; ph < nph; ++ph) {
        if (ph > 0) __syncthreads();                                 // every wave is done with the previous phase's planes
        store_x();
        if (ph + 1 < nph) load_x(ph + 1);                            // in flight under this phase's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (active) {
            const int ph_next = min(ph + 1, nph - 1);                // (past the end: the last phase's stages again, into sets nobody reads)
            F6 xf[2];
            x_frag(0, xf[0]);
            static_for<0, NSTG>([&](auto st_c) {
                constexpr int st = decltype(st_c)::value;
                constexpr int cu = st % NSET, pf = (st + NSET - 1) % NSET;
                constexpr int ahead = st + NSET - 1;
                load_stage(std::integral_constant<int, ahead % NSTG>{}, ahead < NSTG ? ph : ph_next, fr[pf]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, FR>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    constexpr int ks = st * FR + j;
                    if constexpr (ks + 1 < KS) x_frag(ks + 1, xf[(ks + 1) & 1]);
                    acc = oe_mma_terms<6>(fr[cu][j], xf[ks & 1], acc);
                });
            });
        }
    }
    if (!active) return;
    float hv[16];
    hv[0] = acc[0] + q0.x; hv[1] = acc[1] + q0.y; hv[2] = acc[2] + q0.z; hv[3] = acc[3] + q0.w;
    hv[4] = acc[4] + q1.x; hv[5] = acc[5] + q1.y; hv[6] = acc[6] + q1.z; hv[7] = acc[7] + q1.w;
    hv[8] = acc[8] + q2.x; hv[9] = acc[9] + q2.y; hv[10] = acc[10] + q2.z; hv[11] = acc[11] + q2.w;
    hv[12] = acc[12] + q3.x; hv[13] = acc[13] + q3.y; hv[14] = acc[14] + q3.z; hv[15] = acc[15] + q3.w;
    f6_wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[lq * 36 + f6_acc_row(r, lk)] = hv[r];
    f6_wave_sync();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        const long gr = m0 + row;
        if (gr >= p.rows) continue;
        const int col = ft * 32 + c4;
        float4 v = *reinterpret_cast<const float4*>(&patch[row * 36 + c4]);
        if (p.p_out > 0.f) {
            const uint2 h = drop_hash4(seed_out, ((unsigned long long)gr * p.no + col) >> 2);
            v.x *= drop_field(h.x, 0, dp_out); v.y *= drop_field(h.x, 1, dp_out);
            v.z *= drop_field(h.y, 0, dp_out); v.w *= drop_field(h.y, 1, dp_out);
        }
        if (p.rowmask && !p.rowmask[gr]) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual || p.beta != 1.f) {
            const float4 res = ps == 0 ? r0 : ps == 1 ? r1 : ps == 2 ? r2 : r3;
            v = make_float4(res.x + p.beta * v.x, res.y + p.beta * v.y, res.z + p.beta * v.z, res.w + p.beta * v.w);
        }
        *reinterpret_cast<float4*>(p.y + gr * p.ldy + col) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TILE FORM of the same product for FEW rows (the decoders' Linears: 992 rows at config 2; the relative-position projection: 248).
// There the row-block form is one latency chain per block - 31 blocks that each stream the whole packed matrix (393 KiB for
// 256 x 256: 13.6 us, as long as for 7936 rows) - and a tiled launch is 16 blocks walking eight K-tiles (14 us).  Here a block owns
// ONE 32 x 32 output tile and its eight waves split the REDUCTION: wave w takes k in [w k/8, (w+1) k/8) - its weight pieces
// (k/128 of them, contiguous in the packed stream) and its x fragments (fp32 straight from global, 32 bytes per lane and step,
// split in registers) are all requested at once, then k/128 six-term products, the eight partial tiles meet in LDS, every wave
// finishes four rows.  One round trip of loads, one barrier: (rows/32) (n/32) blocks of ~5 us.  The epilogue is oe_gemm_f32's
// element for element (bias -> pre-activation out -> activation, or times act'(aux) -> dropout -> dead rows -> residual + beta x),
// so the feed-forward's first GEMM and the input gradient through it qualify too.
template <int KSW>
__global__ __launch_bounds__(512, 2) void rowtile6_kernel(Row6Params p) {
    constexpr int PIECE = 3 * 1024;
    __shared__ __attribute__((aligned(16))) float red[8][32 * 36];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, lk = lane >> 5;
    const long m0 = (long)blockIdx.x * 32;
    const int ft = blockIdx.y;
    const long last = (long)p.rows - 1;
    // this wave's weight pieces and x fragments: everything in flight at once
    const unsigned char* wl = p.wp + ((long)ft * (8 * KSW) + wv * KSW) * PIECE + lane * 16;
    const float* xr = p.x + min(m0 + lq, last) * p.ldx + wv * (16 * KSW) + 8 * lk;
    F6 wf[KSW];
    float4 xa[KSW], xb[KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        xa[j] = *reinterpret_cast<const float4*>(xr + 16 * j);
        xb[j] = *reinterpret_cast<const float4*>(xr + 16 * j + 4);
    }
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
#pragma unroll
        for (int n = 0; n < 3; ++n) wf[j].p[n] = *reinterpret_cast<const bf16x8*>(wl + (long)j * PIECE + n * 1024);
    }
    // the two output elements this lane finishes: row 4 wv + (lane >> 4), columns 2 (lane & 15), + 1 of the tile
    const int er = 4 * wv + (lane >> 4), ec = 2 * (lane & 15);
    const long gr = m0 + er;
    const long grc = min(gr, last);
    const int col = ft * 32 + ec;
    float2 bias2 = make_float2(0.f, 0.f), res2 = make_float2(0.f, 0.f), aux2 = make_float2(0.f, 0.f);
    if (p.bias) bias2 = *reinterpret_cast<const float2*>(p.bias + col);
    if (p.residual) res2 = *reinterpret_cast<const float2*>(p.residual + grc * p.ldr + col);
    if (p.actgrad_in) aux2 = *reinterpret_cast<const float2*>(p.actgrad_in + grc * p.ld_aux + col);
    const bool dead = p.rowmask && !p.rowmask[grc];
    const unsigned long long seed_out = p.seed_out + (p.seed_dev ? *p.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const DropParams dp_out = drop_params(p.p_out);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const float xv[8] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w, xb[j].x, xb[j].y, xb[j].z, xb[j].w};
        oe_bf16x8 pl[3];
        oe_split8<3>(xv, pl);
        F6 xf;
#pragma unroll
        for (int n = 0; n < 3; ++n) xf.p[n] = pl[n];
        acc = oe_mma_terms<6>(wf[j], xf, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][lq * 36 + f6_acc_row(r, lk)] = acc[r];
    __syncthreads();
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float2 t = *reinterpret_cast<const float2*>(&red[w][er * 36 + ec]);
        v0 += t.x; v1 += t.y;
    }
    if (gr > last) return;
    float x0 = v0 + bias2.x, x1 = v1 + bias2.y;
    if (p.preact_out) *reinterpret_cast<float2*>(p.preact_out + gr * p.ld_aux + col) = make_float2(x0, x1);
    if (p.actgrad_in) {
        x0 *= f6_act_bwd(p.act, aux2.x); x1 *= f6_act_bwd(p.act, aux2.y);
    } else if (p.act == OE_ACT_RELU) {
        x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f);
    } else if (p.act == OE_ACT_SWISH) {
        x0 *= sigmoidf_(x0); x1 *= sigmoidf_(x1);
    }
    if (p.p_out > 0.f) {
        const uint2 h = drop_hash4(seed_out, ((unsigned long long)gr * p.no + col) >> 2);
        const unsigned hw = (col & 2) ? h.y : h.x;
        x0 *= drop_field(hw, 0, dp_out); x1 *= drop_field(hw, 1, dp_out);
    }
    if (dead) { x0 = 0.f; x1 = 0.f; }
    *reinterpret_cast<float2*>(p.y + gr * p.ldy + col) = make_float2(res2.x + p.beta * x0, res2.y + p.beta * x1);
}

// one matrix W (rows R, cols Cc, row-major) -> the A-operand fragments of Wg = W (transposed = 0: Wg is (R, Cc)) or of Wg = W^T
// (transposed = 1: Wg is (Cc, R)): [tile of 32 Wg rows][k-step of 16][plane][lane][8], as ffn_pack_kernel's first stream
__global__ __launch_bounds__(256) void row6_pack_table_kernel(const long long* __restrict__ table) {
    const long long* e = table + (long)blockIdx.y * 6;
    const float* w = reinterpret_cast<const float*>(e[0]);
    __bf16* dst0 = reinterpret_cast<__bf16*>(e[1]);
    const int R = (int)e[2], Cc = (int)e[3], ldw = (int)e[4], tr = (int)e[5];
    const int NOg = tr ? Cc : R, Kg = tr ? R : Cc;                  // Wg is (NOg, Kg)
    const int KSg = Kg / 16;
    const long piece = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (piece >= (long)(NOg / 32) * KSg) return;
    const int lane = threadIdx.x & 63;
    const int ft = (int)(piece / KSg), ks = (int)(piece % KSg);
    const int row = 32 * ft + (lane & 31), col = 16 * ks + 8 * (lane >> 5);     // Wg[row][col + e]
    float x[8];
    if (!tr) {
        const float* src = w + (long)row * ldw + col;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = w[(long)(col + q) * ldw + row];
    }
    oe_bf16x8 pl[3];
    oe_split8<3>(x, pl);
    __bf16* dst = dst0 + piece * 1536 + lane * 8;
    *reinterpret_cast<oe_bf16x8*>(dst) = pl[0];
    *reinterpret_cast<oe_bf16x8*>(dst + 512) = pl[1];
    *reinterpret_cast<oe_bf16x8*>(dst + 1024) = pl[2];
}

extern "C" int oe_rowgemm6_supported(int k, int n) {
    static const int kphase = getenv("OE_ROWGEMM_KPHASE") ? atoi(getenv("OE_ROWGEMM_KPHASE")) : 1;      // tuning: 0 = off
    if (kphase && (k == 768 || k == 1024) && (n == 128 || n == 256)) return 1;       // K-phased kernel: one accumulator tile per wave
    return (k == 256 || k == 512) && n >= 128 && n % 128 == 0 && n <= 8192;
}
// which kernel oe_rowgemm6 runs for a problem: 2 = tile form (few rows: k a multiple of 128 up to 1024, n a multiple of 32),
// 1 = row-block form, 0 = neither
static int row6_tile_max_rows = getenv("OE_ROWTILE_MAX_ROWS") ? atoi(getenv("OE_ROWTILE_MAX_ROWS")) : 2048;
extern "C" int oe_rowgemm6_form(int rows, int k, int n) {
    const int ksw = k / 128;
    if (rows > 0 && rows <= row6_tile_max_rows && k % 128 == 0 && (ksw == 2 || ksw == 4 || ksw == 6 || ksw == 8) && n >= 32 && n % 32 == 0 &&
        n <= 32 * 65535)
        return 2;
    return oe_rowgemm6_supported(k, n) ? 1 : 0;
}

// table: device array of n entries of six 64-bit words { W, packed, R, Cc, ld, transposed }; max_pieces: the largest entry's
// (Wg rows / 32) * (Wg cols / 16)
extern "C" int oe_rowgemm6_pack_table(const void* table, int n, long max_pieces, void* stream) {
    OE_REQUIRE(table && n > 0 && max_pieces > 0, "oe_rowgemm6_pack_table: empty table");
    hipLaunchKernelGGL(row6_pack_table_kernel, dim3(oe_cdiv(max_pieces, 4), n), dim3(256), 0, (hipStream_t)stream, (const long long*)table);
    OE_LAUNCH_CHECK("oe_rowgemm6_pack_table");
    return 0;
}

extern "C" int oe_rowgemm6(const oe_rowgemm_args* a, void* stream) {
    OE_REQUIRE(a && (a->x || a->ln.dy || a->lnf.x) && a->wp && (a->y || a->lne.dx), "oe_rowgemm6: null pointer");
    const int form = a->rows > 0 ? oe_rowgemm6_form(a->rows, a->k, a->n) : 0;
    OE_REQUIRE(form != 0, "oe_rowgemm6: unsupported rows=%d k=%d n=%d", a->rows, a->k, a->n);
    const bool has_act = a->act != OE_ACT_NONE || a->preact_out || a->actgrad_in;
    OE_REQUIRE(form == 2 || !has_act, "oe_rowgemm6: activation / pre-activation epilogue only in the tile form (rows <= %d)", row6_tile_max_rows);
    OE_REQUIRE(a->act == OE_ACT_NONE || a->act == OE_ACT_RELU || a->act == OE_ACT_SWISH, "oe_rowgemm6: activation %d is not fused", a->act);
    OE_REQUIRE(!has_act || !(a->preact_out || a->actgrad_in) || (a->ld_aux % 2 == 0 && ((((uintptr_t)a->preact_out) | ((uintptr_t)a->actgrad_in)) & 7) == 0),
               "oe_rowgemm6: pre-activation / act-grad source must be 8-byte aligned with an even row stride");
    OE_REQUIRE(a->rows > 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && (!a->residual || a->ldr % 4 == 0), "oe_rowgemm6: bad rows / strides");
    if (a->ln.dy) {
        OE_REQUIRE(form == 1 && a->k == 256, "oe_rowgemm6: the LayerNorm-backward prologue exists in the row-block form at k = 256 only");
        const char* why = ln_pro_check(a->ln);
        OE_REQUIRE(why == nullptr, "oe_rowgemm6: LayerNorm prologue: %s", why);
    }
    if (a->lnf.x) {
        OE_REQUIRE(form == 1 && a->k == 256 && !a->ln.dy && !a->lnf.gamma2, "oe_rowgemm6: the (single) LayerNorm-forward prologue exists in the row-block form at k = 256 only");
        const char* why = lnf_pro_check(a->lnf);
        OE_REQUIRE(why == nullptr, "oe_rowgemm6: LayerNorm-forward prologue: %s", why);
    }
    OE_REQUIRE(((((uintptr_t)a->x) | ((uintptr_t)a->y) | ((uintptr_t)a->residual) | ((uintptr_t)a->wp) | ((uintptr_t)a->bias)) & 15) == 0,
               "oe_rowgemm6: 16-byte alignment required");
    OE_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f, "oe_rowgemm6: dropout rate out of range");
    Row6Params p{};
    p.x = a->x; p.ldx = a->ldx; p.wp = (const unsigned char*)a->wp; p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.beta = a->beta;
    p.rowmask = a->rowmask; p.y = a->y; p.ldy = a->ldy; p.rows = a->rows; p.no = a->n; p.p_out = a->drop_p; p.seed_out = a->seed; p.seed_dev = a->seed_dev;
    p.k = a->k; p.act = a->act; p.preact_out = a->preact_out; p.actgrad_in = a->actgrad_in; p.ld_aux = a->ld_aux;
    p.ln = ln_pro_of(a->ln);
    p.lnf = lnf_pro_of(a->lnf);
    if (a->lne.dx) {
        OE_REQUIRE(form == 1 && a->k == 256 && a->n == 256 && !a->lnf.x, "oe_rowgemm6: the LayerNorm-backward epilogue exists for 256 <- 256 in the row-block form only");
        OE_REQUIRE(a->lne.x && a->lne.stats && a->lne.gamma && a->lne.beta && a->lne.ws, "oe_rowgemm6: LayerNorm epilogue: null pointer");
        OE_REQUIRE(!a->bias && a->drop_p == 0.f && !a->rowmask && !a->residual && a->beta == 1.f, "oe_rowgemm6: LayerNorm epilogue: no other epilogue feature applies");
        OE_REQUIRE(a->lne.act == OE_ACT_NONE || a->lne.act == OE_ACT_RELU || a->lne.act == OE_ACT_SWISH, "oe_rowgemm6: LayerNorm epilogue: activation %d", a->lne.act);
        OE_REQUIRE(((((uintptr_t)a->lne.x) | ((uintptr_t)a->lne.gamma) | ((uintptr_t)a->lne.beta) | ((uintptr_t)a->lne.dx) | ((uintptr_t)a->lne.ws)) & 15) == 0,
                   "oe_rowgemm6: LayerNorm epilogue: 16-byte alignment required");
        p.le_x = a->lne.x; p.le_stats = a->lne.stats; p.le_gamma = a->lne.gamma; p.le_beta = a->lne.beta; p.le_act = a->lne.act;
        p.le_dx = a->lne.dx; p.le_ws = a->lne.ws;
    }
    if (form == 2) {
        const dim3 tgrid(oe_cdiv(a->rows, 32), a->n / 32), tblock(512);
        hipStream_t st = (hipStream_t)stream;
        switch (a->k / 128) {
            case 2: hipLaunchKernelGGL((rowtile6_kernel<2>), tgrid, tblock, 0, st, p); break;
            case 4: hipLaunchKernelGGL((rowtile6_kernel<4>), tgrid, tblock, 0, st, p); break;
            case 6: hipLaunchKernelGGL((rowtile6_kernel<6>), tgrid, tblock, 0, st, p); break;
            default: hipLaunchKernelGGL((rowtile6_kernel<8>), tgrid, tblock, 0, st, p); break;
        }
        OE_LAUNCH_CHECK("oe_rowgemm6 (tile form)");
        return 0;
    }
    const dim3 grid(oe_cdiv(a->rows, 32)), block(512);
    if (a->k > 512) hipLaunchKernelGGL(rowgemm6p_kernel, grid, block, 0, (hipStream_t)stream, p);
    else if (a->lne.dx && a->ln.dy) hipLaunchKernelGGL((rowgemm6_kernel<256, 1, true>), grid, block, 0, (hipStream_t)stream, p);
    else if (a->lne.dx) hipLaunchKernelGGL((rowgemm6_kernel<256, 0, true>), grid, block, 0, (hipStream_t)stream, p);
    else if (a->ln.dy) hipLaunchKernelGGL((rowgemm6_kernel<256, 1>), grid, block, 0, (hipStream_t)stream, p);
    else if (a->lnf.x) hipLaunchKernelGGL((rowgemm6_kernel<256, 2>), grid, block, 0, (hipStream_t)stream, p);
    else if (a->k == 512) hipLaunchKernelGGL((rowgemm6_kernel<512>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((rowgemm6_kernel<256>), grid, block, 0, (hipStream_t)stream, p);
    OE_LAUNCH_CHECK("oe_rowgemm6");
    return 0;
}

template <int D, int RT, int NG>
static int ffn6_launch(const Ffn6Params& p, bool bwd, int nout, hipStream_t st) {
    const dim3 grid(oe_cdiv(p.rows, 32 * RT)), block(256 * NG);
    if constexpr (D == 256 && RT == 1 && NG == 2) {
        if (bwd && p.ln.dy) {
            hipLaunchKernelGGL((ffn6_kernel<D, RT, true, 1, NG, 1>), grid, block, 0, st, p);
            OE_LAUNCH_CHECK("oe_ffn_bwd (precision 6, LayerNorm-backward prologue)");
            return 0;
        }
        if (!bwd && p.lnf.x) {
            if (nout == 2) hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 2, NG, 2>), grid, block, 0, st, p);
            else if (nout == 1) hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 1, NG, 2>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 0, NG, 2>), grid, block, 0, st, p);
            OE_LAUNCH_CHECK("oe_ffn_fwd (precision 6, LayerNorm prologue)");
            return 0;
        }
    }
    OE_REQUIRE(!p.ln.dy && !p.lnf.x, "oe_ffn: the LayerNorm prologues exist at d = 256 in the two-group block shape only");
    if (bwd) hipLaunchKernelGGL((ffn6_kernel<D, RT, true, 1, NG>), grid, block, 0, st, p);
    else if (nout == 2) hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 2, NG>), grid, block, 0, st, p);
    else if (nout == 1) hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 1, NG>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((ffn6_kernel<D, RT, false, 0, NG>), grid, block, 0, st, p);
    OE_LAUNCH_CHECK(bwd ? "oe_ffn_bwd (precision 6)" : "oe_ffn_fwd (precision 6)");
    return 0;
}

int oe_ffn6_supported(int d, int ff) { return (d == 128 || d == 256 || d == 512) && ff > 0 && ff % 128 == 0 && ff <= 8192; }

// Block shape (OE_FFN6_MODE forces one: 1 = 64 rows / 4 waves, 2 = 32 rows / 4 waves, 3 = 32 rows / 8 waves in two groups).
// Measured on MI355X (tools/ffn6_bench.py, profiles/r04_ffn6.md): the kernel is bound by the weight stream (3 MB of planes per
// block at d = 256) and by what one wave per SIMD cannot overlap; two staggered groups are the default wherever they exist.
static int ffn6_mode_v = getenv("OE_FFN6_MODE") ? atoi(getenv("OE_FFN6_MODE")) : 0;
static int ffn6_mode() { return ffn6_mode_v; }
extern "C" int oe_ffn6_config(int mode) {
    if (mode >= 0) ffn6_mode_v = mode;
    return ffn6_mode_v;
}

// called by ffn.hip's oe_ffn_fwd / oe_ffn_bwd for precision 6 (arguments already validated there)
int oe_ffn6_run(const oe_ffn_args* a, bool bwd, void* stream) {
    Ffn6Params p{};
    p.x = a->x; p.ldx = a->ldx; p.w1p = (const unsigned char*)a->w1p; p.b1 = a->b1; p.w2p = (const unsigned char*)a->w2p; p.b2 = a->b2;
    p.pre = a->pre_out; p.act_out = a->act_out; p.residual = a->residual; p.ldr = a->ldr; p.beta = a->beta; p.y = a->y; p.ldy = a->ldy;
    p.rows = a->rows; p.ff = a->ff; p.act = a->act; p.p_in = a->drop_in; p.seed_in = a->seed_in; p.p_out = a->drop_out; p.seed_out = a->seed_out;
    p.seed_dev = a->seed_dev;
    if (a->ln.dy) {
        const char* why = ln_pro_check(a->ln);
        OE_REQUIRE(bwd && why == nullptr, "oe_ffn: LayerNorm prologue: %s", bwd ? why : "backward only");
        p.ln = ln_pro_of(a->ln);
    }
    if (a->lnf.x) {
        const char* why = lnf_pro_check(a->lnf);
        OE_REQUIRE(!bwd && why == nullptr, "oe_ffn: LayerNorm-forward prologue: %s", bwd ? "forward only" : why);
        p.lnf = lnf_pro_of(a->lnf);
    }
    const int nout = a->act_out ? 2 : a->pre_out ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int mode = ffn6_mode();
    const bool two_groups = (a->ff % 256 == 0) && mode != 1 && mode != 2;
    if (a->d == 512) return two_groups ? ffn6_launch<512, 1, 2>(p, bwd, nout, st) : ffn6_launch<512, 1, 1>(p, bwd, nout, st);
    if (a->d == 256) {
        if (two_groups) return ffn6_launch<256, 1, 2>(p, bwd, nout, st);
        return mode == 2 ? ffn6_launch<256, 1, 1>(p, bwd, nout, st) : ffn6_launch<256, 2, 1>(p, bwd, nout, st);
    }
    return ffn6_launch<128, 2, 1>(p, bwd, nout, st);
}
