// bf16-MFMA GEMM whose operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4):
// no staging registers, no ds_write pass, and a 4-deep tile ring kept in flight across the
// K-loop barriers with counted vmcnt waits.  The GEMMs of this path are short serial chains
// (K = 256 is eight K-tiles): the register-staged kernel in gemm_bf16.hip exposes one memory
// round trip per K-tile, here the ring hides all but the first.
//
// Operands stay fp32 in HBM *and* in LDS; the bf16 split (hi, or hi + lo for the 3-term
// fp32-grade product) happens on the fragment in registers, right before the MFMA.
//
// LDS image per stage (fp32):
//   row-major operand (k contiguous):  [rows][32]  one DMA piece = 8 rows x 128 B.  The 16-byte
//       chunk c of row r is stored at slot c ^ ((r >> 1) & 7): the permutation is applied to the
//       per-lane SOURCE address (the DMA destination is lane-linear) and again on the read, so the
//       two ds_read_b128 of a fragment are bank-conflict free.
//   k-major operand (rows contiguous): [32][rows]  linear; a fragment is eight ds_read_b32 with
//       consecutive lanes on consecutive rows (conflict free).
// Only interior, 16-byte aligned problems come here (host check in oe_gemm_dma_try); everything
// else - ragged edges, the conv2 im2col gather - stays on gemm_bf16.hip.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define DBK 32      // K-tile

#ifdef OE_GEMM_STAMPS
extern "C" int oe_debug_set_stamp_buffer(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(oe_stamp_buf), &p, sizeof(p));
}
#endif

// one LDS-DMA piece: 64 lanes x 16 B from per-lane `src` to the wave-uniform LDS byte address `dst`
// (hardware adds lane*16).  M0 carries the destination and is compiler-reserved: save / restore it
// inside the statement.  hipcc does not count this load; completion is our counted vmcnt below.
__device__ __forceinline__ void dma16(const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N_OUTSTANDING) : "memory");
}

template <int NPL> struct DFrag { bf16x8 p[NPL]; };
template <int NPL>
__device__ __forceinline__ void split8(const float (&x)[8], DFrag<NPL>& f) { oe_split8<NPL>(x, f.p); }

// fragment (32 rows x 16 k) of one operand tile: lane -> row (lane & 31), k half (lane >> 5)
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void read_frag(const float* tile, int row, int half, int ks, float (&x)[8]) {
    if (!KMAJOR) {
        const int slot = (4 * ks + 2 * half) ^ ((row >> 1) & 7);
        const float4 v0 = *reinterpret_cast<const float4*>(tile + row * DBK + slot * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(tile + row * DBK + (slot ^ 1) * 4);
        x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w;
        x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    } else {
        const float* p = tile + (ks * 16 + half * 8) * ROWS + row;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e * ROWS];
    }
}

// NST = ring depth (tiles resident in LDS)
// floor(k / d) for 0 <= k < 2^24 through the float reciprocal (one correction step); rd = 1.0f / d
__device__ __forceinline__ int div_small(int k, int d, float rd, int& rem) {
    int q = (int)((float)k * rd);
    rem = k - q * d;
    if (rem >= d) { ++q; rem -= d; }
    if (rem < 0) { --q; rem += d; }
    return q;
}

// GATHER_B (weight gradient of the stride-2 3x3 conv2, both operands k-major): row k of B is the im2col row of
// output position k = (b, t2, f2) of an NHWC activation; only its start address differs from a plain row, and a
// DMA piece takes a per-lane source address anyway, so the gather costs two small divisions per piece and tile.
template <int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int TERMS, int NST, bool GATHER_B = false>
__global__ __launch_bounds__(256, (NST == 2 && TM * TN <= 4) ? 2 : 1) void gemm_dma_kernel(const float* __restrict__ Ap, long lda, const float* __restrict__ Bp, long ldb,
                                                        float* __restrict__ C, long ldc, int M, int N, int K, int k_chunk,
                                                        int gx, int gy, EpiParams ep, OperandDesc Bd) {
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
    constexpr int A_FLOATS = BM * DBK, B_FLOATS = BN * DBK, STAGE_FLOATS = A_FLOATS + B_FLOATS;
    constexpr int PA = BM / 32, PB = BN / 32;          // DMA pieces (1 KiB) per wave per tile
    constexpr int LPT = PA + PB;                        // vmcnt units per tile
    constexpr int LDS_FLOATS = (NST * STAGE_FLOATS > 4 * 32 * 36) ? NST * STAGE_FLOATS : 4 * 32 * 36;
    __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

    OE_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)tile_y * BM, n0 = (long)tile_x * BN;
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin) / DBK;

    // per-lane source pointers of this wave's pieces for the first tile
    const float* srcA[PA];
    const float* srcB[PB];
    int kposB[PB];                      // GATHER_B: output position (= k index) of this lane's row in the next tile
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int g = wave * PA + j;
        // rows / columns past the matrix edge re-read the last valid ones: their products land in accumulator
        // rows the (bounds-checked) epilogue of an edge block never stores
        if (!A_KMAJOR) {
            const int row = 8 * g + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            srcA[j] = Ap + min(m0 + row, (long)M - 1) * lda + k_begin + chunk * 4;
        } else {
            const int off = g * 256 + lane * 4;
            srcA[j] = Ap + (long)(k_begin + off / BM) * lda + min(m0 + (off % BM), (long)M - 4);
        }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int g = wave * PB + j;
        if (!B_KMAJOR) {
            const int row = 8 * g + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            srcB[j] = Bp + min(n0 + row, (long)N - 1) * ldb + k_begin + chunk * 4;
        } else if (!GATHER_B) {
            const int off = g * 256 + lane * 4;
            srcB[j] = Bp + (long)(k_begin + off / BN) * ldb + min(n0 + (off % BN), (long)N - 4);
        } else {
            const int off = g * 256 + lane * 4;
            kposB[j] = k_begin + off / BN;
            srcB[j] = Bp + addr_col<true>(Bd, n0 + (off % BN));        // column part: fixed for the whole block
        }
    }
    const float rF2 = GATHER_B ? 1.0f / (float)Bd.F2 : 0.f, rT2 = GATHER_B ? 1.0f / (float)Bd.T2 : 0.f;
    const long stepA = A_KMAJOR ? (long)DBK * lda : DBK, stepB = B_KMAJOR ? (long)DBK * ldb : DBK;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;        // LDS byte address of the ring
    auto issue = [&](int stage) {
        const unsigned sa = lds_base + (unsigned)(stage * STAGE_FLOATS + wave * PA * 256) * 4u;
        const unsigned sb = lds_base + (unsigned)(stage * STAGE_FLOATS + A_FLOATS + wave * PB * 256) * 4u;
#pragma unroll
        for (int j = 0; j < PA; ++j) { dma16(srcA[j], sa + j * 1024u); srcA[j] += stepA; }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (!GATHER_B) { dma16(srcB[j], sb + j * 1024u); srcB[j] += stepB; }
            else {
                int f, t;
                const int q = div_small(kposB[j], Bd.F2, rF2, f);
                const int b = div_small(q, Bd.T2, rT2, t);
                dma16(srcB[j] + (((long)b * Bd.T1 + Bd.S * t) * Bd.F1 + Bd.S * f) * Bd.C, sb + j * 1024u);
                kposB[j] += DBK;
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

    const bool do_csum = A_KMAJOR && ep.a_colsum != nullptr && tile_x == 0 && wn == 0;   // wave-uniform
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t);
    OE_STAMP(1);

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the younger tiles' pieces are outstanding; the barrier then
        // (a) publishes every wave's pieces of tile kt and (b) retires all reads of tile kt-1's stage
        const int younger = min(nk, kt + NST - 1) - (kt + 1);
        if (NST >= 4 && younger >= 2) wait_dma_and_barrier<2 * LPT>();
        else if (younger >= 1) wait_dma_and_barrier<LPT>();
        else wait_dma_and_barrier<0>();
        if (kt == 0) OE_STAMP(2);
        if (kt + NST - 1 < nk) issue((kt + NST - 1) % NST);      // into the stage tile kt-1 just left
        const float* at = lds + (kt % NST) * STAGE_FLOATS + (A_KMAJOR ? wm * 32 * TM : wm * 32 * TM * DBK);
        const float* bt = lds + (kt % NST) * STAGE_FLOATS + A_FLOATS + (B_KMAJOR ? wn * 32 * TN : wn * 32 * TN * DBK);
#pragma unroll
        for (int ks = 0; ks < DBK / 16; ++ks) {
            constexpr int NPL = oe_npl<TERMS>::N;
            DFrag<NPL> fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float x[8];
                read_frag<A_KMAJOR, BM>(at + (A_KMAJOR ? i * 32 : i * 32 * DBK), frow, fhalf, ks, x);
                if (A_KMAJOR && do_csum) csum[i] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                split8<NPL>(x, fa[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float x[8];
                read_frag<B_KMAJOR, BN>(bt + (B_KMAJOR ? j * 32 : j * 32 * DBK), frow, fhalf, ks, x);
                split8<NPL>(x, fb[j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = oe_mma_terms<TERMS>(fa[i], fb[j], acc[i][j]);
        }
    }
    OE_STAMP(3);
    if (A_KMAJOR && do_csum) {
        float al = ep.alpha;
        if (ep.alpha_dev) al *= *ep.alpha_dev;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
            const long row = m0 + wm * 32 * TM + i * 32 + frow;
            if (fhalf == 0 && row < M) atomicAdd(ep.a_colsum + row, v * al);
        }
    }
    gemm_epilogue<TM, TN>(acc, lds, C, ldc, M, N, m0, n0, ep, tile_z);
    OE_STAMP(4);
}

template <int TM, int TN, bool AK, bool BKM, int TERMS, int NST, bool GB = false>
static int launch_dma(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int kc, int nz,
                      const EpiParams& ep, hipStream_t st) {
    const int gx = oe_cdiv(N, 64 * TN), gy = oe_cdiv(M, 64 * TM);
    hipLaunchKernelGGL((gemm_dma_kernel<TM, TN, AK, BKM, TERMS, NST, GB>), dim3(gx * gy * nz), dim3(256), 0, st, A.p, A.ld, B.p, B.ld, C, ldc,
                       M, N, K, kc, gx, gy, ep, B);
    OE_LAUNCH_CHECK("oe_gemm (bf16 mfma, lds-dma)");
    return 0;
}

// Returns 1 when the problem does not qualify (caller falls through to the register-staged kernel),
// 0 on a successful launch, <0 / hip error otherwise.  `tile` = 10*TM + TN of the block (2x2 waves of
// TM x TN 32x32 tiles): 42 = 256x128, 22 = 128x128, 21 = 128x64, 11 = 64x64.
int oe_gemm_dma_try(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk, const EpiParams& ep,
                    bool a_kmajor, bool b_kmajor, bool gather_b, int terms, int tile, hipStream_t st) {
    // OE_GEMM_DMA: 0 = never, 1 = where it measured faster (default), 2 = wherever the problem qualifies (tuning)
    static const int mode = getenv("OE_GEMM_DMA") ? atoi(getenv("OE_GEMM_DMA")) : 1;
    if (!mode) return 1;
    if (a_kmajor && !b_kmajor) return 1;
    // Measured on MI355X (tools/gemm_bench.py, M = 7936): the ring wins where the register-staged kernel pays a
    // transpose per K-tile and the K-loop is long - weight gradients (both operands k-major) and x @ W with
    // K >= 512; short-K x @ W^T problems are epilogue/launch bound and run faster at 2-4 co-resident blocks per CU.
    // ... and problems too small to put more than one 64x64 block on a CU: nothing overlaps a block's own memory
    // latency there except the ring (decoder-side GEMMs, M = B*(L+1) rows; linear_pos, M = T rows).
    // ... and very wide outputs (the vocabulary projections): many column blocks re-read the same rows of A.
    const bool small = (long)oe_cdiv(M, 64) * oe_cdiv(N, 64) * sk <= 256;
    const bool wide = N >= 2048 && !a_kmajor;
    // ... and, with six MFMA terms per product (precision 6), every output at least 512 wide: the register-staged kernel's
    // exposed K-tile round trip costs more once the matrix work per tile doubles (7936 x 1024 x 256: 32.4 against 43.8 us,
    // x 768: 29.7 / 41.9, x 512: 21.0 / 27.8, dY W 7936 x 1024 x 256: 36.2 / 52.1; N = 256 outputs measured equal or slower)
    // ... and narrower outputs once 128 x 128 tiles still make more than a round of the chip (the 64 x 16 s batch, 25472 rows:
    // x 256 x 1024 135 against 200 us, x 256 x 256 30 / 63, dY W 25472 x 256 x 1024 121 / 141 with 64 x 64 tiles)
    const long b22 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128) * sk;
    const bool six = terms == 6 && !a_kmajor && M >= 128 && N >= 128 && (N >= 512 || b22 >= 300);
    // ... and 128 x 64 tiles for the long reductions into narrow outputs that stay below that (7936 x 256 x 1024: 41.3 against
    // 45.8 us x W^T, 45.2 / 49.1 dY W; x 768: 33.3 / 36.0)
    const bool six_tall = terms == 6 && !a_kmajor && !six && M >= 128 && N >= 64 && K >= 512 && (long)oe_cdiv(M, 128) * oe_cdiv(N, 64) * sk >= 200;
    if (mode == 1 && !(a_kmajor && b_kmajor) && !(b_kmajor && K >= 512) && !small && !wide && !six && !six_tall) return 1;
    static const bool tile_forced = getenv("OE_GEMM_TILE") && atoi(getenv("OE_GEMM_TILE")) != 0;     // tuning (tools/gemm_bench.py)
    if (!tile_forced && six && b22 >= (N >= 512 ? 200 : 300)) tile = 22;
    if (!tile_forced && mode == 1 && six_tall) tile = 21;
    if (tile == 12) tile = 11;
    const int bm = 64 * (tile / 10), bn = 64 * (tile % 10);
    // pieces are 16 bytes: K a multiple of the K-tile; a k-major operand's row length (M resp. N) a multiple of 4.
    // Ragged M / N edges are clamped in the kernel; the gather path keeps whole tiles.
    if (!A.vec_ok || !B.vec_ok || K % DBK || M < 4 || N < 4) return 1;
    if ((a_kmajor && M % 4) || (b_kmajor && N % 4)) return 1;
    if (gather_b && (M % bm || N % bn)) return 1;
    if (gather_b) {
        // conv weight gradient: a column tile must stay inside one kernel row (KS*C contiguous floats),
        // positions must be exact in float (div_small), C a multiple of 4 for the 16-byte pieces
        if (!(a_kmajor && b_kmajor) || (B.KS * B.C) % bn || B.C % 4 || K >= (1 << 24) || (tile != 22 && tile != 11)) return 1;
        int kc = oe_cdiv(oe_cdiv(K, sk), DBK) * DBK;
        if (kc <= 0) kc = DBK;
        const int nz = oe_cdiv(K, kc);
        if (terms == 6) {
            if (tile == 22) return launch_dma<2, 2, true, true, 6, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
            return launch_dma<1, 1, true, true, 6, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
        }
        if (terms == 3) {
            if (tile == 22) return launch_dma<2, 2, true, true, 3, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
            return launch_dma<1, 1, true, true, 3, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
        }
        if (tile == 22) return launch_dma<2, 2, true, true, 1, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
        return launch_dma<1, 1, true, true, 1, 4, true>(A, B, C, ldc, M, N, K, kc, nz, ep, st);
    }
    int kc = oe_cdiv(oe_cdiv(K, sk), DBK) * DBK;
    if (kc <= 0) kc = DBK;
    const int nz = oe_cdiv(K, kc);
    // ring depth: long reductions keep three tiles in flight at one 128x128 block per CU (96 KiB of LDS); short ones
    // (K of a few tiles) do better with two stages = two blocks per CU, whose prologues/epilogues overlap
    static const int forced_nst = getenv("OE_DMA_NST") ? atoi(getenv("OE_DMA_NST")) : 0;       // tuning
    const int nst = forced_nst ? forced_nst : (tile == 11 ? 4 : (kc >= 512 ? 3 : 2));   // 64x64 tiles: 4 stages are still 64 KiB
#define OE_DMA_T(AK, BKM, T)                                                                                         \
    do {                                                                                                             \
        if (tile == 42) return launch_dma<4, 2, AK, BKM, T, 3>(A, B, C, ldc, M, N, K, kc, nz, ep, st);               \
        if (tile == 24) return launch_dma<2, 4, AK, BKM, T, 3>(A, B, C, ldc, M, N, K, kc, nz, ep, st);               \
        if (tile == 22 && nst == 2) return launch_dma<2, 2, AK, BKM, T, 2>(A, B, C, ldc, M, N, K, kc, nz, ep, st);   \
        if (tile == 22 && nst == 3) return launch_dma<2, 2, AK, BKM, T, 3>(A, B, C, ldc, M, N, K, kc, nz, ep, st);   \
        if (tile == 22) return launch_dma<2, 2, AK, BKM, T, 4>(A, B, C, ldc, M, N, K, kc, nz, ep, st);               \
        if (tile == 21) return launch_dma<2, 1, AK, BKM, T, 4>(A, B, C, ldc, M, N, K, kc, nz, ep, st);               \
        if (tile == 11 && nst == 2) return launch_dma<1, 1, AK, BKM, T, 2>(A, B, C, ldc, M, N, K, kc, nz, ep, st);   \
        if (tile == 11 && nst == 3) return launch_dma<1, 1, AK, BKM, T, 3>(A, B, C, ldc, M, N, K, kc, nz, ep, st);   \
        if (tile == 11) return launch_dma<1, 1, AK, BKM, T, 4>(A, B, C, ldc, M, N, K, kc, nz, ep, st);               \
        return 1;                                                                                                    \
    } while (0)
#define OE_DMA(AK, BKM) do { if (terms == 6) OE_DMA_T(AK, BKM, 6); else if (terms == 3) OE_DMA_T(AK, BKM, 3); else OE_DMA_T(AK, BKM, 1); } while (0)
    if (!a_kmajor && !b_kmajor) OE_DMA(false, false);
    if (!a_kmajor && b_kmajor) OE_DMA(false, true);
    OE_DMA(true, true);
#undef OE_DMA
#undef OE_DMA_T
}
