// GEMM on operands that are already split into bf16 planes (hi, and lo = bf16(x - hi) for the 3-term
// fp32-grade product):  C[m][n] = sum_k A[m][k] * B[n][k], both operands row-major with k contiguous.
//
// The fp32-operand kernels (gemm_bf16.hip, gemm_dma.hip) spend most of a K-tile converting fragments - every
// element of A is split once per column block and wave that touches it.  Here the split is done once per tensor
// (oe_split_bf16, or by the kernel that produces the tensor), the planes travel global -> LDS by LDS-DMA exactly
// as they sit in memory, and a fragment is one ds_read_b128 per plane: the K-loop is MFMA + LDS traffic only.
// Weight gradients and x @ W take the same kernel on transposed planes (oe_split_bf16 transpose = 1), so one
// operand form covers every GEMM of the path.
//
// LDS image per stage and plane: [rows][32] bf16 (64 B rows); one DMA piece = 16 rows.  The 16-byte chunk c of row r
// is stored at slot c ^ ((r >> 2) & 3) - applied to the per-lane source address (the DMA destination is lane-linear)
// and again on the read, which makes the ds_read_b128 of 16 consecutive rows conflict free.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define PBK 32

__device__ __forceinline__ void dma16p(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_dma_and_barrier_p() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N_OUTSTANDING) : "memory");
}

template <int TM, int TN, int TERMS, int NST>
__global__ __launch_bounds__(256) void gemm_planes_kernel(const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al, long lda,
                                                           const __bf16* __restrict__ Bh, const __bf16* __restrict__ Bl, long ldb,
                                                           float* __restrict__ C, long ldc, int M, int N, int K, int k_chunk,
                                                           int gx, int gy, EpiParams ep) {
    int tile_x, tile_y, tile_z;
    {   // XCD-aware tile order (see gemm_bf16.hip)
        const int nblk = gridDim.x, id = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, j = id >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_x = swz % gx;
        tile_y = (swz / gx) % gy;
        tile_z = swz / (gx * gy);
    }
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int NP = (TERMS == 3) ? 2 : 1;                          // planes per operand
    constexpr int A_EL = BM * PBK, B_EL = BN * PBK;                    // bf16 elements per plane tile
    constexpr int STAGE_EL = NP * (A_EL + B_EL);
    constexpr int PA = BM / 64, PB = BN / 64;                          // DMA pieces (16 rows) per wave, plane and tile
    constexpr int LPT = NP * (PA + PB);
    constexpr int LDS_BYTES = (NST * STAGE_EL * 2 > 4 * 32 * 36 * 4) ? NST * STAGE_EL * 2 : 4 * 32 * 36 * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[LDS_BYTES];
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)tile_y * BM, n0 = (long)tile_x * BN;
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin) / PBK;

    // element offsets (shared by the hi and lo plane) of this lane's pieces; rows past the edge re-read the last row
    long offA[PA], offB[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = 16 * (wave * PA + j) + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        offA[j] = min(m0 + row, (long)M - 1) * lda + k_begin + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = 16 * (wave * PB + j) + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        offB[j] = min(n0 + row, (long)N - 1) * ldb + k_begin + chunk * 8;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    auto issue = [&](int stage) {
        const unsigned st = lds_base + (unsigned)(stage * STAGE_EL) * 2u;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const unsigned d = st + (unsigned)((wave * PA + j) * 512) * 2u;
            dma16p(Ah + offA[j], d);
            if (TERMS == 3) dma16p(Al + offA[j], d + A_EL * 2u);
            offA[j] += PBK;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const unsigned d = st + (unsigned)(NP * A_EL + (wave * PB + j) * 512) * 2u;
            dma16p(Bh + offB[j], d);
            if (TERMS == 3) dma16p(Bl + offB[j], d + B_EL * 2u);
            offB[j] += PBK;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t);

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 2) & 3;          // rows of a fragment are tile rows 32*i + frow: the key only needs frow
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = min(nk, kt + NST - 1) - (kt + 1);
        if (NST >= 4 && younger >= 2) wait_dma_and_barrier_p<2 * LPT>();
        else if (NST >= 3 && younger >= 1) wait_dma_and_barrier_p<LPT>();
        else wait_dma_and_barrier_p<0>();
        if (kt + NST - 1 < nk) issue((kt + NST - 1) % NST);
        const __bf16* ah = lds + (kt % NST) * STAGE_EL + (wm * 32 * TM + frow) * PBK;
        const __bf16* bh = lds + (kt % NST) * STAGE_EL + NP * A_EL + (wn * 32 * TN + frow) * PBK;
#pragma unroll
        for (int ks = 0; ks < PBK / 16; ++ks) {
            const int slot = ((2 * ks + fhalf) ^ fkey) * 8;
            bf16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * PBK + slot);
                if (TERMS == 3) fal[i] = *reinterpret_cast<const bf16x8*>(ah + A_EL + i * 32 * PBK + slot);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fbh[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * PBK + slot);
                if (TERMS == 3) fbl[j] = *reinterpret_cast<const bf16x8*>(bh + B_EL + j * 32 * PBK + slot);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (TERMS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    gemm_epilogue<TM, TN>(acc, reinterpret_cast<float*>(lds_raw), C, ldc, M, N, m0, n0, ep, tile_z);
}

template <int TM, int TN, int TERMS, int NST>
static int launch_planes(const oe_gemm_args* g, const EpiParams& ep, const void* ah, const void* al, const void* bh, const void* bl,
                         int kc, int nz, hipStream_t st) {
    const int gx = oe_cdiv(g->n, 64 * TN), gy = oe_cdiv(g->m, 64 * TM);
    hipLaunchKernelGGL((gemm_planes_kernel<TM, TN, TERMS, NST>), dim3(gx * gy * nz), dim3(256), 0, st, (const __bf16*)ah, (const __bf16*)al,
                       g->lda, (const __bf16*)bh, (const __bf16*)bl, g->ldb, g->c, g->ldc, g->m, g->n, g->k, kc, gx, gy, ep);
    OE_LAUNCH_CHECK("oe_gemm_planes");
    return 0;
}

extern "C" int oe_gemm_planes(const oe_gemm_args* g, const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo,
                              void* stream) {
    OE_REQUIRE(g && a_hi && b_hi && g->c, "oe_gemm_planes: null operand");
    OE_REQUIRE(g->precision == 1 || g->precision == 3, "oe_gemm_planes: precision must be 1 or 3");
    OE_REQUIRE(g->precision == 1 || (a_lo && b_lo), "oe_gemm_planes: precision 3 needs the lo planes");
    OE_REQUIRE(g->m > 0 && g->n > 0 && g->k > 0 && g->k % PBK == 0, "oe_gemm_planes: k=%d must be a positive multiple of %d", g->k, PBK);
    OE_REQUIRE(g->lda % 8 == 0 && g->ldb % 8 == 0 && ((uintptr_t)a_hi % 16 == 0) && ((uintptr_t)b_hi % 16 == 0) &&
               ((uintptr_t)a_lo % 16 == 0) && ((uintptr_t)b_lo % 16 == 0), "oe_gemm_planes: planes must be 16-byte aligned with ld %% 8 == 0");
    OE_REQUIRE(!g->a_kmajor && !g->b_kmajor && g->conv_gather == OE_GATHER_NONE && !g->a_colsum,
               "oe_gemm_planes: operands are row-major planes (transpose when splitting)");
    OE_REQUIRE(g->split_k >= 1 && !(g->split_k > 1 && !g->atomic_out), "oe_gemm_planes: split_k > 1 needs atomic_out");
    EpiParams ep{};
    ep.alpha = g->alpha; ep.alpha_dev = g->alpha_dev; ep.bias = g->bias; ep.act = g->act;
    ep.preact_out = g->preact_out; ep.actgrad_in = g->actgrad_in; ep.ld_aux = g->ld_aux ? g->ld_aux : g->ldc;
    ep.drop_p = g->drop_p; ep.seed = g->seed; ep.seed_dev = g->seed_dev; ep.rowmask = g->rowmask;
    ep.residual = g->residual; ep.ldr = g->ldr ? g->ldr : g->ldc; ep.beta = g->beta; ep.res_row_mod = g->res_row_mod;
    ep.accumulate = g->accumulate; ep.atomic = g->atomic_out; ep.a_colsum = nullptr;
    int kc = oe_cdiv(oe_cdiv(g->k, g->split_k), PBK) * PBK;
    const int nz = oe_cdiv(g->k, kc);
    const long b22 = (long)oe_cdiv(g->m, 128) * oe_cdiv(g->n, 128) * nz;
    int tile = (b22 >= 200 && g->m >= 128 && g->n >= 128) ? 22 : 11;
    static const int forced_tile = getenv("OE_GEMM_TILE") ? atoi(getenv("OE_GEMM_TILE")) : 0;
    static const int nst = getenv("OE_PLANES_NST") ? atoi(getenv("OE_PLANES_NST")) : 4;
    if (forced_tile == 22 || forced_tile == 11) tile = forced_tile;
    hipStream_t st = (hipStream_t)stream;
#define OE_PL(TT, NS)                                                                                           \
    do {                                                                                                        \
        if (tile == 22) return launch_planes<2, 2, TT, NS>(g, ep, a_hi, a_lo, b_hi, b_lo, kc, nz, st);          \
        return launch_planes<1, 1, TT, NS>(g, ep, a_hi, a_lo, b_hi, b_lo, kc, nz, st);                          \
    } while (0)
    if (g->precision == 3) { if (nst == 2) OE_PL(3, 2); else if (nst == 3) OE_PL(3, 3); else OE_PL(3, 4); }
    if (nst == 2) OE_PL(1, 2); else if (nst == 3) OE_PL(1, 3); else OE_PL(1, 4);
#undef OE_PL
}

// ---- fp32 -> bf16 planes ------------------------------------------------------
// hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  transpose = 0: out[r][c]; 1: out[c][r] (32x32 tiles via LDS).
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, long ldx, long rows, int cols,
                                                          __bf16* __restrict__ hi, __bf16* __restrict__ lo, long ldo) {
    const int c4 = cols >> 2;
    const long total = rows * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        bf16x4 h, l;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(hi + r * ldo + c) = h;
        if (lo) {
            l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
            l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
            *reinterpret_cast<bf16x4*>(lo + r * ldo + c) = l;
        }
    }
}

// 64 (rows) x 64 (cols) tile per block: coalesced float4 reads along cols, 8-byte writes along rows
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ x, long ldx, long rows, int cols,
                                                               __bf16* __restrict__ hi, __bf16* __restrict__ lo, long ldo) {
    __shared__ float t[64][65];
    const long r0 = (long)blockIdx.y * 64;
    const int c0 = blockIdx.x * 64;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int rr = p * 16 + (threadIdx.x >> 4), cc = (threadIdx.x & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + rr < rows && c0 + cc < cols) v = *reinterpret_cast<const float4*>(x + (r0 + rr) * ldx + c0 + cc);
        t[rr][cc] = v.x; t[rr][cc + 1] = v.y; t[rr][cc + 2] = v.z; t[rr][cc + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int cc = p * 16 + (threadIdx.x >> 4), rr = (threadIdx.x & 15) * 4;      // output row = input column
        if (c0 + cc < cols && r0 + rr < rows) {
            const float a = t[rr][cc], b = t[rr + 1][cc], c = t[rr + 2][cc], d = t[rr + 3][cc];
            bf16x4 h, l;
            h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
            *reinterpret_cast<bf16x4*>(hi + (long)(c0 + cc) * ldo + r0 + rr) = h;
            if (lo) {
                l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
                l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
                *reinterpret_cast<bf16x4*>(lo + (long)(c0 + cc) * ldo + r0 + rr) = l;
            }
        }
    }
}

extern "C" int oe_split_bf16(const float* x, long ldx, long rows, int cols, int transpose, void* hi, void* lo, long ld_out,
                             void* stream) {
    OE_REQUIRE(x && hi && rows > 0 && cols > 0, "oe_split_bf16: bad arguments");
    OE_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16 == 0), "oe_split_bf16: x must be 16-byte aligned with cols, ldx %% 4 == 0");
    OE_REQUIRE(ld_out % 4 == 0 && ((uintptr_t)hi % 8 == 0) && ((uintptr_t)lo % 8 == 0), "oe_split_bf16: planes must be 8-byte aligned with ld %% 4 == 0");
    OE_REQUIRE(!transpose || rows % 4 == 0, "oe_split_bf16: transposed planes need rows %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    if (!transpose) {
        const long total = rows * (cols / 4);
        const int nb = (int)min((long)oe_cdiv(total, 256), 256L * 16);
        hipLaunchKernelGGL(split_rows_kernel, dim3(nb), dim3(256), 0, st, x, ldx, rows, cols, (__bf16*)hi, (__bf16*)lo, ld_out);
    } else {
        hipLaunchKernelGGL(split_transpose_kernel, dim3(oe_cdiv(cols, 64), oe_cdiv(rows, 64)), dim3(256), 0, st, x, ldx, rows, cols,
                           (__bf16*)hi, (__bf16*)lo, ld_out);
    }
    OE_LAUNCH_CHECK("oe_split_bf16");
    return 0;
}
