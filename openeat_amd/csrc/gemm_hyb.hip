// Precision-6 GEMM with ONE pre-split operand:  C = A B^T (B row-major) or C = A B (B k-major), A = an fp32 activation,
// B = a WEIGHT that already exists as three bf16 planes (the parameter arena is split once per optimizer step:
// ParamArena.refresh_planes) - every torch.nn.Linear / 1x1 conv forward and input gradient of the path
// (/root/reference/openeat/modules/positionwise_feed_forward.py:43, attention.py:56-58,97, convolution.py:103,113 and their
// autograd).
//
// Why a third kernel beside gemm_dma.hip (both operands fp32, split on every fragment use) and gemm_pl.hip (both pre-split).
// PMC of the step (profiles/r03_gemm_pmc.md): the ring kernel issues 12.5-15 vector instructions per MFMA - the split of BOTH
// fragments, redone by every wave that uses them - and sits at 21 % MFMA busy with 40 % of its wave cycles in issue and 36 % in
// issue stalls: it is bound by the vector pipe.  The all-planes kernel has 3.3 vector instructions per MFMA but waits 43 % of its
// cycles for LDS-DMA fills (6 bytes per operand element) - and its activation planes cost their producers more than the GEMMs
// gain (profiles/r03_experiments.md).  The weights' planes cost one pass over the arena per step.  So: the activation tile travels
// as fp32 (4 bytes per element, split on the fragment as in the ring kernel), the weight tile as planes (no vector work): half
// the ring kernel's split instructions for 25 % more fill bytes.
//
// LDS image per stage: [A fp32: BM rows x 32 k, the ring kernel's XOR-swizzled row-major tile][B plane 0..2: gemm_pl.hip's
// row-major ([BN][32] bf16, 64-byte rows, chunk c of row r at c ^ ((r >> 2) & 3)) or k-major ([32][128] bf16 sub-tiles, chunk c of
// k-row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3))) tile].  Pieces are 1 KiB LDS-DMA transfers with counted vmcnt, NST stages.
// Interior, aligned problems only (host check); everything else stays on the other kernels.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define HBK 32      // K-tile

__device__ __forceinline__ void hy_dma16(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N_OUTSTANDING>
__device__ __forceinline__ void hy_wait_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N_OUTSTANDING) : "memory");
}

struct HFrag { bf16x8 p[3]; };

__device__ __forceinline__ void hy_split8(const float (&x)[8], HFrag& f) { oe_split8<3>(x, f.p); }
// A fragment (32 rows x 16 k) of the fp32 tile [rows][32] with 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (gemm_dma.hip)
__device__ __forceinline__ void hy_a_frag(const float* tile, int row, int half, int ks, HFrag& f) {
    const int slot = (4 * ks + 2 * half) ^ ((row >> 1) & 7);
    const float4 v0 = *reinterpret_cast<const float4*>(tile + row * HBK + slot * 4);
    const float4 v1 = *reinterpret_cast<const float4*>(tile + row * HBK + (slot ^ 1) * 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    hy_split8(x, f);
}
__device__ __forceinline__ int hy_fsw(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ int hy_gsw(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }
// B fragment of 32 rows (= output columns) x 16 k of a row-major plane tile [rows][32] bf16 (gemm_pl.hip::pl_row_frag)
__device__ __forceinline__ void hy_b_row_frag(const unsigned char* tile, int plane_bytes, int row, int half, int ks, HFrag& f) {
    const int pos = (2 * ks + half) ^ hy_fsw(row);
    const unsigned char* p = tile + row * (HBK * 2) + pos * 16;
#pragma unroll
    for (int n = 0; n < 3; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(p + n * plane_bytes);
}
// B fragment of 32 columns x 16 k-rows of a k-major plane tile (sub-tiles [32][128] bf16; gemm_pl.hip::pl_col_frag)
__device__ __forceinline__ void hy_b_col_frag(const unsigned char* tile, int plane_bytes, int col32, int ks, int lane, HFrag& f) {
    const int i = lane & 15, grp = lane >> 4;
    const int kr0 = 16 * ks + 8 * (grp >> 1) + (i >> 2);
    tile += (col32 >> 7) * (HBK * 256);
    col32 &= 127;
    const int ch = (col32 >> 3) + 2 * (grp & 1) + ((i & 3) >> 1);
    const unsigned char* p0 = tile + kr0 * 256 + 16 * (ch ^ hy_gsw(kr0)) + 8 * (i & 1);
    const unsigned char* p1 = tile + (kr0 + 4) * 256 + 16 * (ch ^ hy_gsw(kr0 + 4)) + 8 * (i & 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        union { s16x4 h[2]; bf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + n * plane_bytes));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1 + n * plane_bytes));
        f.p[n] = u.v;
    }
}

// Block = WM x 2 waves, each TM x TN accumulator tiles of 32 x 32: block tile (32 TM WM) x (64 TN).  BKM: B is k-major (64 TN must
// then be a multiple of 128).  WM = 4, TM = 1, TN = 4 (128 x 256, eight waves of 32 x 128): an activation element is split by two
// waves for 2 x 4 x 6 MFMAs each - a third of the 2 x 2 arrangement's split work per MFMA.
template <int TM, int TN, bool BKM, int NST, int WM = 2>
__global__ __launch_bounds__(WM * 128, (NST * (32 * TM * WM * HBK * 4 + 3 * 64 * TN * HBK * 2) <= 80 * 1024) ? 2 : 1)
void gemm_hyb_kernel(const float* __restrict__ Ap, long lda, const __bf16* __restrict__ Bp, long ldb, long b_pstride, float* __restrict__ C, long ldc,
                     int M, int N, int K, int gx, int gy, EpiParams ep) {
    int tile_x, tile_y;
    {   // XCD-aware tile order (gemm_bf16.hip)
        const int nblk = gridDim.x, id = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, j = id >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_x = swz % gx;
        tile_y = swz / gx;
    }
    constexpr int NW = WM * 2;
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    static_assert(!BKM || BN % 128 == 0, "k-major tiles are made of 128-column sub-tiles");
    constexpr int A_BYTES = BM * HBK * 4, B_T = BN * HBK * 2;         // fp32 A tile; one B plane tile
    constexpr int STAGE = A_BYTES + 3 * B_T;
    constexpr int PA = A_BYTES / 1024, PB = 3 * B_T / 1024;           // 1 KiB pieces per stage
    constexpr int PPW = (PA + PB) / NW;                               // per wave
    static_assert((PA + PB) % NW == 0, "piece count must split over the waves");
    constexpr int EPI_BYTES = NW * 32 * 36 * 4;
    constexpr int LDS_BYTES = (NST * STAGE > EPI_BYTES) ? NST * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)tile_y * BM, n0 = (long)tile_x * BN;
    const int nk = K / HBK;

    // this wave's pieces: g = wave * PPW + j; g < PA: A piece g (8 rows x 128 B), else B piece g - PA (plane, sub-piece)
    const unsigned char* src[PPW];
    unsigned dst[PPW];
    long step[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int g = wave * PPW + j;
        if (g < PA) {
            const int row = 8 * g + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            src[j] = reinterpret_cast<const unsigned char*>(Ap + min(m0 + row, (long)M - 1) * lda + chunk * 4);
            dst[j] = (unsigned)(g * 1024);
            step[j] = HBK * 4;
        } else {
            constexpr int PPL = B_T / 1024;                           // pieces per plane
            const int gg = g - PA, plane = gg / PPL, sub = gg - plane * PPL;
            const __bf16* base = Bp + plane * b_pstride;
            dst[j] = (unsigned)(A_BYTES + plane * B_T + sub * 1024);
            if (!BKM) {
                const int row = sub * 16 + (lane >> 2), cpos = lane & 3;
                const int csrc = cpos ^ hy_fsw(row);
                src[j] = reinterpret_cast<const unsigned char*>(base + min(n0 + row, (long)N - 1) * ldb + csrc * 8);
                step[j] = HBK * 2;
            } else {
                const int st = sub / (HBK / 4);
                const int krow = (sub - st * (HBK / 4)) * 4 + (lane >> 4), cpos = lane & 15;
                const int csrc = cpos ^ hy_gsw(krow);
                const long col = min(n0 + st * 128 + csrc * 8, (long)N - 8);
                src[j] = reinterpret_cast<const unsigned char*>(base + (long)krow * ldb + col);
                step[j] = (long)HBK * ldb * 2;
            }
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    int tiles_issued = 0;
    unsigned stage_off = 0;
    auto issue_piece = [&](int j) {
        const bool advance = tiles_issued + 1 < nk;                   // past the end the last tile is re-read into a stage nobody reads
        hy_dma16(src[j], lds_base + stage_off + dst[j]);
        src[j] += advance ? step[j] : 0;
    };
    auto end_issue = [&]() {
        ++tiles_issued;
        stage_off += STAGE;
        if (stage_off == NST * STAGE) stage_off = 0;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(j);
        end_issue();
    }

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the NST - 2 younger tiles' pieces are outstanding; the barrier publishes every wave's
        // pieces of tile kt and retires all reads of tile kt - 1's stage, which this iteration's issue overwrites
        hy_wait_and_barrier<(NST - 2) * PPW>();
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(j);
        end_issue();
        const unsigned char* st = lds + (kt % NST) * STAGE;
        const float* at = reinterpret_cast<const float*>(st) + wm * 32 * TM * HBK;
        const unsigned char* bt = st + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            HFrag fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!BKM) hy_b_row_frag(bt, B_T, wn * 32 * TN + j * 32 + frow, fhalf, ks, fb[j]);
                else hy_b_col_frag(bt, B_T, wn * 32 * TN + j * 32, ks, lane, fb[j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) hy_a_frag(at + i * 32 * HBK, frow, fhalf, ks, fa[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = oe_mma_terms<6>(fa[i], fb[j], acc[i][j]);
        }
    }
    // the surplus pieces (issued past the end of the range) must have landed before the epilogue reuses the LDS
    hy_wait_and_barrier<0>();
    gemm_epilogue<TM, TN, WM>(acc, reinterpret_cast<float*>(lds), C, ldc, M, N, m0, n0, ep, 0);
}

static long hyb_launches = 0;
extern "C" long oe_gemm_hyb_launches(void) { return hyb_launches; }

template <int TM, int TN, bool BKM, int NST, int WM = 2>
static int launch_hyb(const OperandDesc& A, const void* Bp, long ldb, long b_pstride, float* C, long ldc, int M, int N, int K, const EpiParams& ep,
                      hipStream_t st) {
    const int gx = oe_cdiv(N, 64 * TN), gy = oe_cdiv(M, 32 * TM * WM);
    hipLaunchKernelGGL((gemm_hyb_kernel<TM, TN, BKM, NST, WM>), dim3(gx * gy), dim3(WM * 128), 0, st, A.p, A.ld, (const __bf16*)Bp, ldb, b_pstride, C, ldc,
                       M, N, K, gx, gy, ep);
    OE_LAUNCH_CHECK("oe_gemm (bf16x6, weight planes)");
    ++hyb_launches;
    return 0;
}

// Returns 1 when the problem does not qualify (the caller goes on to the kernels that split both operands), 0 on a launch.
// B / Bp: the weight operand's fp32 descriptor (leading dimension) and plane 0 of its pre-split copy.
int oe_gemm_hyb_try(const OperandDesc& A, const OperandDesc& B, const void* Bp, long b_pstride, float* C, long ldc, int M, int N, int K, int sk,
                    const EpiParams& ep, bool b_kmajor, hipStream_t st) {
    static const int mode = getenv("OE_GEMM_HYB") ? atoi(getenv("OE_GEMM_HYB")) : 1;             // 0 = never (A/B comparisons)
    if (!mode || !Bp || sk > 1 || ep.atomic || ep.a_colsum) return 1;
    if (!A.vec_ok || K % HBK || K < 2 * HBK || M < 128 || N < 64) return 1;
    if ((((uintptr_t)Bp) & 15) || B.ld % 8 || b_pstride % 8) return 1;
    if (b_kmajor && (N % 8 || N < 128)) return 1;
    // the shapes the ring kernel takes at six terms (gemm_dma.hip::oe_gemm_dma_try): 128 x 128 tiles for outputs at least 512 wide
    // or more than a round of such tiles; 128 x 64 for long reductions into narrower outputs that still cover the chip
    const long b22 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128);
    const long b21 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 64);
    static const int forced = getenv("OE_HYB_TILE") ? atoi(getenv("OE_HYB_TILE")) : 0;           // tuning: 22 / 21
    int tile = 0;
    if (N >= 128 && (N >= 512 ? b22 >= 200 : b22 >= 300)) tile = 22;
    else if (!b_kmajor && K >= 512 && b21 >= 200) tile = 21;
    // (k-major B tiles are 128 columns wide: the long reductions into 256-wide outputs - 124 such tiles at config 2 - stay on the
    // ring kernel's 128 x 64 tiles)
    // 128 x 256 (eight waves) wherever the grid of such tiles is about one round of the chip or more - whole 256-wide rows at the
    // 16 s batches too (25472 x 256 x 1024: 83.5 us against 94.7 on 128 x 128 tiles and 111 on the ring; x W of the same: 79 / 91 / 114)
    const long b24 = (N % 256 == 0) ? (long)oe_cdiv(M, 128) * (N / 256) : 0;
    if (b24 >= 160 && (tile == 22 || tile == 21 || (b_kmajor && K >= 512))) tile = 24;
    if (forced) tile = forced;
    if (tile == 24 && N >= 256) {
        if (b_kmajor) return launch_hyb<1, 4, true, 2, 4>(A, Bp, B.ld, b_pstride, C, ldc, M, N, K, ep, st);
        return launch_hyb<1, 4, false, 2, 4>(A, Bp, B.ld, b_pstride, C, ldc, M, N, K, ep, st);
    }
    if (tile == 22) {
        if (b_kmajor) return launch_hyb<2, 2, true, 2>(A, Bp, B.ld, b_pstride, C, ldc, M, N, K, ep, st);
        return launch_hyb<2, 2, false, 2>(A, Bp, B.ld, b_pstride, C, ldc, M, N, K, ep, st);
    }
    if (tile == 21 && !b_kmajor) return launch_hyb<2, 1, false, 4>(A, Bp, B.ld, b_pstride, C, ldc, M, N, K, ep, st);
    return 1;
}
