// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one
// rounding per product) with fused epilogues.  One kernel serves
//   forward   y = x W^T            (A row-major [M][K], B row-major [N][K])
//   dgrad     dx = dy W            (A row-major,        B k-major   [K][N])
//   wgrad     dW = dy^T x          (A k-major [K][M],   B k-major   [K][N], split-K)
//   conv2     implicit GEMM of the 3x3/s2 subsampling conv over an NHWC input
//             (im2col addresses generated in the loader; forward and wgrad)
//
// Tiling: 256 threads = 4 waves (2x2); each wave owns TM x TN tiles of 32x32
// accumulators.  Both operands are staged k-major in LDS (As[BK][BM+4]) so a
// wave's MFMA operand read is 32 consecutive dwords per half-wave: conflict
// free.  K-tile 16, LDS double-buffered, next tile's global loads issued
// before the MFMA block of the current one (register staging).
//
// Roofline: MFMA-bound.  fp32-input MFMA peak on gfx950 = 157.3 TFLOP/s.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"


#define BK 16
#define LDS_PAD 4

template <int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, bool GATHER_A, bool GATHER_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(OperandDesc A, OperandDesc B, float* __restrict__ C, long ldc,
                                                        int M, int N, int K, int k_chunk, int gx, int gy, EpiParams ep) {
    // XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs (private L2s); give each
    // XCD one contiguous range of tiles (x fastest) so blocks sharing an operand strip hit the same L2.
    int tile_x, tile_y, tile_z;
    {
        const int nblk = gridDim.x, id = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, j = id >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_x = swz % gx;
        tile_y = (swz / gx) % gy;
        tile_z = swz / (gx * gy);
    }
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA = BM + LDS_PAD, LDB = BN + LDS_PAD;
    constexpr int LDS_FLOATS = (2 * BK * (LDA + LDB) > 4 * 32 * 36) ? 2 * BK * (LDA + LDB) : 4 * 32 * 36;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    auto As = [&](int buf) -> float* { return lds + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return lds + 2 * BK * LDA + buf * (BK * LDB); };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)tile_y * BM, n0 = (long)tile_x * BN;
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    // ---- per-thread load slots -------------------------------------------------
    // row-major operand tile (rows x 16): slot -> (row = s/4, kq = s%4), 4 k-values
    // k-major operand tile  (16 x rows): slot -> (k = s/(rows/4), rq = s%(rows/4)), 4 rows
    constexpr int A_SLOTS = BM * BK / 4 / 256, B_SLOTS = BN * BK / 4 / 256;
    long a_fix[A_SLOTS];  // the part of the address that does not change with the k-tile
    long b_fix[B_SLOTS];
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
        int slot = tid + s * 256;
        if (!A_KMAJOR) { long r = m0 + slot / 4; a_fix[s] = (r < M) ? addr_row<GATHER_A>(A, r) : -1; }
        else { long c = m0 + (slot % (BM / 4)) * 4; a_fix[s] = addr_col<false>(A, c); }
    }
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
        int slot = tid + s * 256;
        if (!B_KMAJOR) { long r = n0 + slot / 4; b_fix[s] = (r < N) ? addr_row<false>(B, r) : -1; }
        else { long c = n0 + (slot % (BN / 4)) * 4; b_fix[s] = addr_col<GATHER_B>(B, c); }
    }

    float4 a_reg[A_SLOTS], b_reg[B_SLOTS];

    auto load_tiles = [&](int kt) {
        const int k0 = k_begin + kt * BK;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            int slot = tid + s * 256;
            if (!A_KMAJOR) {
                int k = k0 + (slot & 3) * 4;
                int nv = (a_fix[s] < 0) ? 0 : max(0, min(4, k_end - k));
                a_reg[s] = nv ? load4(A.p + a_fix[s] + addr_col<GATHER_A>(A, k), nv, A.vec_ok) : make_float4(0, 0, 0, 0);
            } else {
                int k = k0 + slot / (BM / 4);
                long c = m0 + (slot % (BM / 4)) * 4;
                int nv = (k < k_end) ? (int)max(0L, min(4L, (long)M - c)) : 0;
                a_reg[s] = nv ? load4(A.p + addr_row<false>(A, k) + a_fix[s], nv, A.vec_ok) : make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            int slot = tid + s * 256;
            if (!B_KMAJOR) {
                int k = k0 + (slot & 3) * 4;
                int nv = (b_fix[s] < 0) ? 0 : max(0, min(4, k_end - k));
                b_reg[s] = nv ? load4(B.p + b_fix[s] + addr_col<false>(B, k), nv, B.vec_ok) : make_float4(0, 0, 0, 0);
            } else {
                int k = k0 + slot / (BN / 4);
                long c = n0 + (slot % (BN / 4)) * 4;
                int nv = (k < k_end) ? (int)max(0L, min(4L, (long)N - c)) : 0;
                b_reg[s] = nv ? load4(B.p + addr_row<GATHER_B>(B, k) + b_fix[s], nv, B.vec_ok) : make_float4(0, 0, 0, 0);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            int slot = tid + s * 256;
            if (!A_KMAJOR) {
                int row = slot / 4, kq = (slot & 3) * 4;
                float* d = As(buf) + kq * LDA + row;
                d[0] = a_reg[s].x; d[LDA] = a_reg[s].y; d[2 * LDA] = a_reg[s].z; d[3 * LDA] = a_reg[s].w;
            } else {
                int k = slot / (BM / 4), rq = (slot % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(As(buf) + k * LDA + rq) = a_reg[s];
            }
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            int slot = tid + s * 256;
            if (!B_KMAJOR) {
                int row = slot / 4, kq = (slot & 3) * 4;
                float* d = Bs(buf) + kq * LDB + row;
                d[0] = b_reg[s].x; d[LDB] = b_reg[s].y; d[2 * LDB] = b_reg[s].z; d[3 * LDB] = b_reg[s].w;
            } else {
                int k = slot / (BN / 4), rq = (slot % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(Bs(buf) + k * LDB + rq) = b_reg[s];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();

    const int lrow = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* as = As(buf) + wm * (32 * TM) + lrow;
        const float* bs = Bs(buf) + wn * (32 * TN) + lrow;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(kk * 2 + lk) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(kk * 2 + lk) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<TM, TN>(acc, lds, C, ldc, M, N, m0, n0, ep, tile_z);
}

template <int TM, int TN, bool AK, bool BKM, bool GA, bool GB>
static int launch(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int splitk,
                  const EpiParams& ep, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    int kc = oe_cdiv(oe_cdiv(K, splitk), BK) * BK;
    if (kc <= 0) kc = BK;
    int nz = oe_cdiv(K, kc);
    if (nz < 1) nz = 1;
    const int gx = oe_cdiv(N, BN), gy = oe_cdiv(M, BM);
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, AK, BKM, GA, GB>), dim3(gx * gy * nz), dim3(256), 0, st, A, B, C, ldc, M, N, K, kc, gx, gy, ep);
    OE_LAUNCH_CHECK("oe_gemm_f32");
    return 0;
}

static bool vec_ok(const void* p, long ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

extern "C" int oe_gemm_f32(const oe_gemm_args* g, void* stream) {
    // an operand may exist as pre-split planes ALONE (a / b NULL with a_planes / b_planes given, precision 6): the launch then fails
    // if the pre-split kernel does not take the problem - no other kernel could read it
    OE_REQUIRE(g && g->c && (g->a || (g->a_planes && g->b_planes && g->precision == 6)) && (g->b || (g->a_planes && g->b_planes && g->precision == 6)),
               "oe_gemm_f32: null operand");
    OE_REQUIRE(g->m > 0 && g->n > 0 && g->k >= 0, "oe_gemm_f32: bad shape m=%d n=%d k=%d", g->m, g->n, g->k);
    OE_REQUIRE(g->split_k >= 1, "oe_gemm_f32: split_k must be >= 1");
    OE_REQUIRE(!(g->split_k > 1 && !g->atomic_out), "oe_gemm_f32: split_k > 1 needs atomic_out");
    OE_REQUIRE(!(g->atomic_out && (g->act || g->residual || g->preact_out || g->actgrad_in || g->drop_p > 0.f || g->beta != 1.f)),
               "oe_gemm_f32: atomic_out supports only alpha/bias epilogues");
    OE_REQUIRE(g->drop_p >= 0.f && g->drop_p < 1.f, "oe_gemm_f32: drop_p out of range");
    OE_REQUIRE(g->act == OE_ACT_NONE || g->act == OE_ACT_RELU || g->act == OE_ACT_SWISH,
               "oe_gemm_f32: only relu (1) and swish (2) are fused into the epilogue; apply act %d with oe_act_fwd / oe_act_grad", g->act);
    hipStream_t st = (hipStream_t)stream;
    OperandDesc A{}, B{};
    A.p = g->a; A.ld = g->lda; A.vec_ok = vec_ok(g->a, g->lda);
    B.p = g->b; B.ld = g->ldb; B.vec_ok = vec_ok(g->b, g->ldb);
    const bool ga = g->conv_gather == OE_GATHER_A, gb = g->conv_gather == OE_GATHER_B;
    if (ga || gb) {
        OperandDesc& X = ga ? A : B;
        X.T1 = g->conv_t1; X.F1 = g->conv_f1; X.T2 = g->conv_t2; X.F2 = g->conv_f2; X.C = g->conv_c;
        X.KS = g->conv_k > 0 ? g->conv_k : 3;             // 0 = the 3x3 stride-2 conv of Conv2dSubsampling4/8
        X.S = g->conv_s > 0 ? g->conv_s : 2;
        OE_REQUIRE(X.C > 0 && X.C % 4 == 0, "oe_gemm_f32: conv gather needs C %% 4 == 0");
        const int KH = g->conv_kh > 0 ? g->conv_kh : X.KS;             // kernel height; X.KS is the width
        OE_REQUIRE(X.T1 >= KH && X.F1 >= X.KS && X.T2 >= 1 && X.F2 >= 1 && X.T2 <= (X.T1 - KH) / X.S + 1 && X.F2 <= (X.F1 - X.KS) / X.S + 1,
                   "oe_gemm_f32: conv gather dims inconsistent");
        OE_REQUIRE(g->conv_kh <= 0 || ga, "oe_gemm_f32: conv_kh is for a gather on A");
        OE_REQUIRE(gb ? (X.T2 == (X.T1 - KH) / X.S + 1 && X.F2 == (X.F1 - X.KS) / X.S + 1) : true, "oe_gemm_f32: conv gather dims inconsistent");
        X.vec_ok = ((uintptr_t)X.p % 16 == 0);
        OE_REQUIRE(ga ? (!g->a_kmajor && g->k == KH * X.KS * X.C) : (g->b_kmajor && g->n == X.KS * X.KS * X.C),
                   "oe_gemm_f32: conv gather layout mismatch");
    }
    EpiParams ep{};
    ep.alpha = g->alpha; ep.alpha_dev = g->alpha_dev; ep.bias = g->bias; ep.act = g->act;
    ep.preact_out = g->preact_out; ep.actgrad_in = g->actgrad_in; ep.ld_aux = g->ld_aux ? g->ld_aux : g->ldc;
    ep.actgrad_bf16 = g->actgrad_bf16;
    OE_REQUIRE(!g->actgrad_bf16 || (g->actgrad_in && g->act == OE_ACT_RELU && !g->preact_out),
               "oe_gemm_f32: actgrad_bf16 is for the ReLU mask (only the sign of the source is read), without preact_out");
    ep.drop_p = g->drop_p; ep.seed = g->seed; ep.seed_dev = g->seed_dev; ep.rowmask = g->rowmask;
    ep.residual = g->residual; ep.ldr = g->ldr ? g->ldr : g->ldc; ep.beta = g->beta; ep.res_row_mod = g->res_row_mod;
    ep.accumulate = g->accumulate; ep.atomic = g->atomic_out; ep.a_colsum = g->a_colsum;
    ep.scatter = g->out_scatter; ep.sc_t1 = g->sc_t1; ep.sc_f1 = g->sc_f1; ep.sc_t2 = g->sc_t2; ep.sc_f2 = g->sc_f2; ep.sc_s = g->sc_s;
    ep.c_planes = (__bf16*)g->c_planes; ep.c_pstride = g->c_plane_stride; ep.ld_cp = g->ldcp ? g->ldcp : g->ldc;
    OE_REQUIRE(!g->c_planes || (!g->atomic_out && !g->accumulate && !g->out_scatter && ep.ld_cp % 4 == 0 && g->c_plane_stride % 4 == 0 &&
                                (((uintptr_t)g->c_planes) & 7) == 0),
               "oe_gemm_f32: c_planes needs a plain (non-atomic, non-scattered) output, 8-byte aligned planes and strides of whole 4-element groups");
    if (g->out_scatter) {
        OE_REQUIRE(!g->atomic_out && !g->accumulate && !g->preact_out && !g->residual && !g->rowmask && g->drop_p <= 0.f,
                   "oe_gemm_f32: out_scatter supports alpha / bias / activation / act-grad epilogues only");
        OE_REQUIRE(g->sc_t2 >= 1 && g->sc_f2 >= 1 && g->sc_s >= 1 && (long)(g->sc_t2 - 1) * g->sc_s < g->sc_t1 && (long)(g->sc_f2 - 1) * g->sc_s < g->sc_f1 &&
                   g->m % ((long)g->sc_t2 * g->sc_f2) == 0, "oe_gemm_f32: out_scatter dims inconsistent");
    }
    OE_REQUIRE(!g->a_colsum || (g->precision != 0 && g->a_kmajor && g->conv_gather != OE_GATHER_A), "oe_gemm_f32: a_colsum needs the bf16 path and a k-major A");
    const int M = g->m, N = g->n, K = g->k, sk = g->split_k;
    OE_REQUIRE(g->precision == 0 || g->precision == 1 || g->precision == 3 || g->precision == 6,
               "oe_gemm_f32: precision must be 0 (fp32), 1 (bf16), 3 (bf16x3) or 6 (bf16x6)");
    OE_REQUIRE(g->conv_korder == 0 || (g->conv_korder == 1 && ga && g->precision == 6 && g->a_planes && g->b_planes && sk <= 1 && A.C % 32 == 0),
               "oe_gemm_f32: conv_korder 1 needs a gather on A, precision 6, pre-split operands, no split of the reduction and C %% 32 == 0");
    if (g->precision == 6 && g->a_planes && g->b_planes) {       // pre-split operands: tiles by LDS-DMA, no conversion in the loop
        const int r = oe_gemm_pl_try(A, B, g->a_planes, g->a_plane_stride, g->b_planes, g->b_plane_stride, g->c, g->ldc, M, N, K, sk, ep,
                                     g->a_kmajor, g->b_kmajor, ga, gb, st, g->conv_korder, g->planes_k_padded != 0);
        if (r != 1) return r;
        OE_REQUIRE(g->conv_korder == 0, "oe_gemm_f32: conv_korder 1, but the pre-split kernel does not take this problem (M=%d N=%d K=%d) - "
                                        "B is laid out for it alone", M, N, K);
    }
    if (g->precision == 6 && g->b_planes && !g->a_planes && g->a && !ga && !gb && !g->a_kmajor) {     // the weight operand pre-split, the activation fp32
        const int r = oe_gemm_hyb_try(A, B, g->b_planes, g->b_plane_stride, g->c, g->ldc, M, N, K, sk, ep, g->b_kmajor, st);
        if (r != 1) return r;
    }
    OE_REQUIRE(g->a && g->b, "oe_gemm_f32: an operand exists as bf16 planes only, but the pre-split kernel does not take this problem "
                             "(M=%d N=%d K=%d, a_kmajor=%d b_kmajor=%d gather=%d)", M, N, K, g->a_kmajor, g->b_kmajor, g->conv_gather);
    if (g->precision) return oe_gemm_bf16_dispatch(A, B, g->c, g->ldc, M, N, K, sk, ep, g->a_kmajor, g->b_kmajor, ga, gb, g->precision, st);
    const long b22 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128) * sk, b12 = (long)oe_cdiv(M, 64) * oe_cdiv(N, 128) * sk;
    const int tile = (b22 >= 200 && M >= 128 && N >= 128) ? 22 : (b12 >= 160 && N >= 128) ? 12 : 11;
#define OE_DISPATCH(AK, BKM, GA, GB)                                                                       \
    do {                                                                                                   \
        if (tile == 22) return launch<2, 2, AK, BKM, GA, GB>(A, B, g->c, g->ldc, M, N, K, sk, ep, st);     \
        if (tile == 12) return launch<1, 2, AK, BKM, GA, GB>(A, B, g->c, g->ldc, M, N, K, sk, ep, st);     \
        return launch<1, 1, AK, BKM, GA, GB>(A, B, g->c, g->ldc, M, N, K, sk, ep, st);                     \
    } while (0)
    if (!g->a_kmajor && !g->b_kmajor) { if (ga) { OE_DISPATCH(false, false, true, false); } else { OE_DISPATCH(false, false, false, false); } }
    if (!g->a_kmajor && g->b_kmajor) { OE_DISPATCH(false, true, false, false); }
    if (g->a_kmajor && g->b_kmajor) { if (gb) { OE_DISPATCH(true, true, false, true); } else { OE_DISPATCH(true, true, false, false); } }
    OE_REQUIRE(false, "oe_gemm_f32: layout a_kmajor=1,b_kmajor=0 is not supported");
#undef OE_DISPATCH
}
