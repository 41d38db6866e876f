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
#include "oe_common.h"
#include "../../include/openeat_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define LDS_PAD 4

struct OperandDesc {
    const float* p;
    long ld;
    int vec_ok;     // 16-byte vector loads are legal (pointer/ld alignment)
    // conv2 im2col gather (NHWC input (B,T1,F1,C), 3x3 stride 2, output (B,T2,F2)):
    int T1, F1, T2, F2, C;
};

template <bool GATHER>
__device__ __forceinline__ long addr_row(const OperandDesc& d, long r) {
    if (!GATHER) return r * d.ld;
    int f = (int)(r % d.F2);
    long q = r / d.F2;
    int t = (int)(q % d.T2);
    long b = q / d.T2;
    return ((b * d.T1 + 2 * t) * (long)d.F1 + 2 * f) * d.C;
}
template <bool GATHER>
__device__ __forceinline__ long addr_col(const OperandDesc& d, long c) {
    if (!GATHER) return c;
    int seg = 3 * d.C;
    int kh = (int)(c / seg);
    return (long)kh * d.F1 * d.C + (c - (long)kh * seg);
}

// Load 4 consecutive logical elements (along the contiguous direction) with
// bounds: n_valid in [0,4] elements are in range.
__device__ __forceinline__ float4 load4(const float* p, int n_valid, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_valid >= 4 && vec_ok) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (n_valid > 0) v.x = p[0];
        if (n_valid > 1) v.y = p[1];
        if (n_valid > 2) v.z = p[2];
        if (n_valid > 3) v.w = p[3];
    }
    return v;
}

struct EpiParams {
    float alpha;
    const float* alpha_dev;
    const float* bias;
    int act;
    float* preact_out;
    const float* actgrad_in;
    long ld_aux;
    float drop_p;
    unsigned long long seed;
    const unsigned long long* seed_dev;
    const unsigned char* rowmask;
    const float* residual;
    long ldr;
    int res_row_mod;
    float beta;
    int accumulate;
    int atomic;
};

template <int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, bool GATHER_A, bool GATHER_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(OperandDesc A, OperandDesc B, float* __restrict__ C, long ldc,
                                                        int M, int N, int K, int k_chunk, EpiParams ep) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA = BM + LDS_PAD, LDB = BN + LDS_PAD;
    constexpr int LDS_FLOATS = (2 * BK * (LDA + LDB) > 4 * 32 * 36) ? 2 * BK * (LDA + LDB) : 4 * 32 * 36;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    auto As = [&](int buf) -> float* { return lds + buf * (BK * LDA); };
    auto Bs = [&](int buf) -> float* { return lds + 2 * BK * LDA + buf * (BK * LDB); };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)blockIdx.y * BM, n0 = (long)blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_chunk;
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

    // ---- epilogue --------------------------------------------------------------
    // Each wave parks one 32x32 accumulator tile at a time in its own LDS patch
    // and re-reads it row-major, so that every global access of the epilogue
    // (C, residual, pre-activation, act-grad input) is a coalesced float4 row.
    float alpha = ep.alpha;
    if (ep.alpha_dev) alpha *= *ep.alpha_dev;
    const float inv_keep = ep.drop_p > 0.f ? 1.f / (1.f - ep.drop_p) : 1.f;
    const unsigned long long seed = ep.seed + (ep.seed_dev ? *ep.seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
    const bool first_split = (blockIdx.z == 0);
    constexpr int EP_LD = 36;
    float* patch = lds + wave * (32 * EP_LD);
    const bool c_vec = (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0) && !ep.atomic;
    const bool aux_vec = (ep.ld_aux % 4 == 0) && (((uintptr_t)ep.preact_out & 15) == 0) && (((uintptr_t)ep.actgrad_in & 15) == 0);
    const bool res_vec = (ep.ldr % 4 == 0) && (((uintptr_t)ep.residual & 15) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * EP_LD + lrow] = acc[i][j][r];
            __syncthreads();
            const long row_base = m0 + wm * (32 * TM) + i * 32;
            const long col = n0 + wn * (32 * TN) + j * 32 + (lane & 7) * 4;
            const int ncol = (int)max(0L, min(4L, (long)N - col));
            float bias4[4] = {0.f, 0.f, 0.f, 0.f};
            if (ep.bias && first_split) for (int e = 0; e < ncol; ++e) bias4[e] = ep.bias[col + e];
            for (int pass = 0; pass < 4; ++pass) {
                const int lr = pass * 8 + (lane >> 3);
                const long row = row_base + lr;
                if (row >= M || ncol == 0) continue;
                const float4 t4 = *reinterpret_cast<const float4*>(patch + lr * EP_LD + (lane & 7) * 4);
                float v[4] = {t4.x, t4.y, t4.z, t4.w};
                const bool full = (ncol == 4);
                float aux[4] = {0.f, 0.f, 0.f, 0.f}, res[4] = {0.f, 0.f, 0.f, 0.f};
                if (ep.actgrad_in) {
                    const float* ap = ep.actgrad_in + row * ep.ld_aux + col;
                    if (full && aux_vec) { float4 a4 = *reinterpret_cast<const float4*>(ap); aux[0] = a4.x; aux[1] = a4.y; aux[2] = a4.z; aux[3] = a4.w; }
                    else for (int e = 0; e < ncol; ++e) aux[e] = ap[e];
                }
                if (ep.residual) {
                    const float* rp = ep.residual + (ep.res_row_mod > 0 ? row % ep.res_row_mod : row) * ep.ldr + col;
                    if (full && res_vec) { float4 r4 = *reinterpret_cast<const float4*>(rp); res[0] = r4.x; res[1] = r4.y; res[2] = r4.z; res[3] = r4.w; }
                    else for (int e = 0; e < ncol; ++e) res[e] = rp[e];
                }
                const bool row_dead = ep.rowmask && !ep.rowmask[row];
                float pre[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e] * alpha + bias4[e];
                    pre[e] = x;
                    if (ep.actgrad_in) x *= act_bwd(ep.act, aux[e]);
                    else x = act_fwd(ep.act, x);
                    if (ep.drop_p > 0.f) x *= dropout_scale(seed, (unsigned long long)(row * N + col + e), ep.drop_p, inv_keep);
                    if (row_dead) x = 0.f;
                    x = res[e] + ep.beta * x;
                    v[e] = x;
                }
                if (ep.preact_out) {
                    float* pp = ep.preact_out + row * ep.ld_aux + col;
                    if (full && aux_vec) *reinterpret_cast<float4*>(pp) = make_float4(pre[0], pre[1], pre[2], pre[3]);
                    else for (int e = 0; e < ncol; ++e) pp[e] = pre[e];
                }
                float* dst = C + row * ldc + col;
                if (ep.atomic) {
                    for (int e = 0; e < ncol; ++e) atomicAdd(dst + e, v[e]);
                } else if (full && c_vec) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (ep.accumulate) { const float4 c4 = *reinterpret_cast<const float4*>(dst); o.x += c4.x; o.y += c4.y; o.z += c4.z; o.w += c4.w; }
                    *reinterpret_cast<float4*>(dst) = o;
                } else {
                    for (int e = 0; e < ncol; ++e) dst[e] = ep.accumulate ? dst[e] + v[e] : v[e];
                }
            }
        }
    }
}

template <int TM, int TN, bool AK, bool BKM, bool GA, bool GB>
static int launch(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int splitk,
                  const EpiParams& ep, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    int kc = oe_cdiv(oe_cdiv(K, splitk), BK) * BK;
    if (kc <= 0) kc = BK;
    int nz = oe_cdiv(K, kc);
    if (nz < 1) nz = 1;
    dim3 grid(oe_cdiv(N, BN), oe_cdiv(M, BM), nz);
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, AK, BKM, GA, GB>), grid, dim3(256), 0, st, A, B, C, ldc, M, N, K, kc, ep);
    OE_LAUNCH_CHECK("oe_gemm_f32");
    return 0;
}

static bool vec_ok(const void* p, long ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

extern "C" int oe_gemm_f32(const oe_gemm_args* g, void* stream) {
    OE_REQUIRE(g && g->a && g->b && g->c, "oe_gemm_f32: null operand");
    OE_REQUIRE(g->m > 0 && g->n > 0 && g->k >= 0, "oe_gemm_f32: bad shape m=%d n=%d k=%d", g->m, g->n, g->k);
    OE_REQUIRE(g->split_k >= 1, "oe_gemm_f32: split_k must be >= 1");
    OE_REQUIRE(!(g->split_k > 1 && !g->atomic_out), "oe_gemm_f32: split_k > 1 needs atomic_out");
    OE_REQUIRE(!(g->atomic_out && (g->act || g->residual || g->preact_out || g->actgrad_in || g->drop_p > 0.f || g->beta != 1.f)),
               "oe_gemm_f32: atomic_out supports only alpha/bias epilogues");
    OE_REQUIRE(g->drop_p >= 0.f && g->drop_p < 1.f, "oe_gemm_f32: drop_p out of range");
    hipStream_t st = (hipStream_t)stream;
    OperandDesc A{}, B{};
    A.p = g->a; A.ld = g->lda; A.vec_ok = vec_ok(g->a, g->lda);
    B.p = g->b; B.ld = g->ldb; B.vec_ok = vec_ok(g->b, g->ldb);
    const bool ga = g->conv_gather == OE_GATHER_A, gb = g->conv_gather == OE_GATHER_B;
    if (ga || gb) {
        OperandDesc& X = ga ? A : B;
        X.T1 = g->conv_t1; X.F1 = g->conv_f1; X.T2 = g->conv_t2; X.F2 = g->conv_f2; X.C = g->conv_c;
        OE_REQUIRE(X.C > 0 && X.C % 4 == 0, "oe_gemm_f32: conv gather needs C %% 4 == 0");
        OE_REQUIRE(X.T2 == (X.T1 - 3) / 2 + 1 && X.F2 == (X.F1 - 3) / 2 + 1, "oe_gemm_f32: conv gather dims inconsistent");
        X.vec_ok = ((uintptr_t)X.p % 16 == 0);
        OE_REQUIRE(ga ? (!g->a_kmajor && g->k == 9 * X.C) : (g->b_kmajor && g->n == 9 * X.C),
                   "oe_gemm_f32: conv gather layout mismatch");
    }
    EpiParams ep{};
    ep.alpha = g->alpha; ep.alpha_dev = g->alpha_dev; ep.bias = g->bias; ep.act = g->act;
    ep.preact_out = g->preact_out; ep.actgrad_in = g->actgrad_in; ep.ld_aux = g->ld_aux ? g->ld_aux : g->ldc;
    ep.drop_p = g->drop_p; ep.seed = g->seed; ep.seed_dev = g->seed_dev; ep.rowmask = g->rowmask;
    ep.residual = g->residual; ep.ldr = g->ldr ? g->ldr : g->ldc; ep.beta = g->beta; ep.res_row_mod = g->res_row_mod;
    ep.accumulate = g->accumulate; ep.atomic = g->atomic_out;
    const int M = g->m, N = g->n, K = g->k, sk = g->split_k;
    const long blocks128 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128) * sk;
    const bool big = blocks128 >= 320 && M >= 128 && N >= 128;
#define OE_DISPATCH(AK, BKM, GA, GB)                                                                   \
    return big ? launch<2, 2, AK, BKM, GA, GB>(A, B, g->c, g->ldc, M, N, K, sk, ep, st)                \
               : launch<1, 1, AK, BKM, GA, GB>(A, B, g->c, g->ldc, M, N, K, sk, ep, st)
    if (!g->a_kmajor && !g->b_kmajor) { if (ga) { OE_DISPATCH(false, false, true, false); } else { OE_DISPATCH(false, false, false, false); } }
    if (!g->a_kmajor && g->b_kmajor) { OE_DISPATCH(false, true, false, false); }
    if (g->a_kmajor && g->b_kmajor) { if (gb) { OE_DISPATCH(true, true, false, true); } else { OE_DISPATCH(true, true, false, false); } }
    OE_REQUIRE(false, "oe_gemm_f32: layout a_kmajor=1,b_kmajor=0 is not supported");
#undef OE_DISPATCH
}
