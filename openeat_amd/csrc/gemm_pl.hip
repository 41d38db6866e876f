// GEMM on PRE-SPLIT operands: both matrices arrive as three bf16 planes in HBM (x = p0 + p1 + p2 exactly, oe_common.h),
// written ONCE by whoever produced the tensor (LayerNorm / GEMM epilogues / oe_split_planes; weights: once per optimizer
// step), and the product is the six-term sum of oe_mma_terms<6> - the fp32 product to one rounding (precision 6).
//
// Why a kernel of its own.  The other bf16 kernels take fp32 operands and split them on the way: gemm_bf16.hip once per
// staged element (global -> registers -> VALU split -> ds_write, one tile of prefetch: the memory round trip of every
// K-tile is exposed), gemm_dma.hip on every fragment use (LDS-DMA ring hides the memory, but the split is redone by each
// consuming wave: ~44 vector instructions per fragment at three pieces).  With six MFMAs per fragment pair the matrix
// pipe is where the time should go, so everything else leaves the loop: tiles travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write, no conversion), NST stages in flight across raw
// s_barriers with counted vmcnt, fragments are plain ds_read_b128 (row-major operands) or ds_read_b64_tr_b16 (k-major
// operands: gfx950's transposing LDS read) - the loop is DMA issue, LDS reads and MFMAs only.
//
// LDS image of one stage: [A plane 0..2][B plane 0..2], each plane tile
//   row-major operand (k contiguous): [rows][BK] bf16, BK*2-byte rows; the 16-byte chunk c of row r sits at position
//       c ^ f(r), f = (r >> 2) & 3 for 64-byte rows (BK 32), (r >> 3) & 1 for 32-byte rows (BK 16): conflict-free b128
//       fragment reads.  The permutation is applied to the per-lane SOURCE address (the DMA destination is lane-linear);
//   k-major operand (rows contiguous): [BK][128] bf16, 256-byte rows, chunk c of k-row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)):
//       conflict-free transposing reads of 4 k-rows x 16 columns per 16-lane group.  The address pattern hands lane half h
//       the k-slots 8h..8h+7 of a 16-deep step - the order the b128 read of a row-major operand has - so the two kinds mix.
// Ragged M / N edges re-read the last valid row / the last whole chunk (the bounds-checked epilogue never stores what that
// yields); K must be a multiple of BK (host check).
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct PlOperand {
    const __bf16* p;       // plane 0; plane n at p + n * plane_stride
    long ld;               // elements between consecutive rows of the stored matrix
    long plane_stride;
    int rows_total;        // row-major: rows of the logical operand (M or N); k-major: columns (M or N)
    // im2col gather (row-major A of a conv forward / parity-class input gradient, k-major B of a conv weight gradient)
    int T1, F1, T2, F2, C, KS, S;
    int korder, KH;        // gathered A: order of the reduction index (oe_gemm_args.conv_korder), window height
};

__device__ __forceinline__ void pl_dma16(const __bf16* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N_OUTSTANDING>
__device__ __forceinline__ void pl_wait_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N_OUTSTANDING) : "memory");
}

template <int BK> __device__ __forceinline__ int pl_fsw(int row) { return BK == 32 ? ((row >> 2) & 3) : ((row >> 3) & 1); }
__device__ __forceinline__ int pl_gsw(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }

struct PlFrag { bf16x8 p[3]; };

// fragment of 32 rows x 16 k of a row-major operand tile ([rows][BK] bf16 per plane, planes PLANE_BYTES apart)
template <int BK>
__device__ __forceinline__ void pl_row_frag(const unsigned char* tile, int plane_bytes, int row, int half, int ks, PlFrag& f) {
    const int pos = (2 * ks + half) ^ pl_fsw<BK>(row);
    const unsigned char* p = tile + row * (BK * 2) + pos * 16;
#pragma unroll
    for (int n = 0; n < 3; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(p + n * plane_bytes);
}
// fragment of 32 columns (col32 ..) x 16 k-rows (16 ks ..) of a k-major operand tile (per plane: sub-tiles of [BK][128] bf16,
// one per 128 columns, `sub_bytes` = BK * 256 apart)
__device__ __forceinline__ void pl_col_frag(const unsigned char* tile, int plane_bytes, int sub_bytes, int col32, int ks, int lane, PlFrag& f) {
    const int i = lane & 15, grp = lane >> 4;
    const int kr0 = 16 * ks + 8 * (grp >> 1) + (i >> 2);
    tile += (col32 >> 7) * sub_bytes;
    col32 &= 127;
    const int ch = (col32 >> 3) + 2 * (grp & 1) + ((i & 3) >> 1);
    const unsigned char* p0 = tile + kr0 * 256 + 16 * (ch ^ pl_gsw(kr0)) + 8 * (i & 1);
    const unsigned char* p1 = tile + (kr0 + 4) * 256 + 16 * (ch ^ pl_gsw(kr0 + 4)) + 8 * (i & 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        union { s16x4 h[2]; bf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + n * plane_bytes));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1 + n * plane_bytes));
        f.p[n] = u.v;
    }
}

// sum of the 24 bf16 values of a fragment (three planes x eight k-slots) added to c: v_dot2c_f32_bf16 against (1, 1)
__device__ __forceinline__ float pl_frag_sum(const PlFrag& f, float c) {
    bf16x2 ones;
    ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        union { bf16x8 v; bf16x2 q[4]; } u;
        u.v = f.p[n];
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_fdot2_f32_bf16(u.q[e], ones, c, false);
    }
    return c;
}

__device__ __forceinline__ int pl_div_small(int k, int d, float rd, int& rem) {     // floor(k / d), 0 <= k < 2^24 (gemm_dma.hip)
    int q = (int)((float)k * rd);
    rem = k - q * d;
    if (rem >= d) { ++q; rem -= d; }
    if (rem < 0) { --q; rem += d; }
    return q;
}

// GA: A is the im2col gather of a conv forward (row-major: rows = output positions, k = (kh, kw, ci)).
// GB: B is the im2col gather of a conv weight gradient (k-major: k = output position, columns = (kh, kw, ci)).
// Block = WM x 2 waves, each TM x TN accumulator tiles of 32 x 32: block tile (32 TM WM) x (64 TN).  WM = 4 puts two waves on
// every SIMD, so one wave's DMA issue and fragment reads run under its partner's MFMAs.
//
// The K-loop is ONE basic block: every iteration issues the DMA pieces of tile kt + NST - 1 (past the end of the range the
// source pointers stop advancing: the last tile is read again into a stage nobody reads - no branch, a constant vmcnt),
// and the pieces are placed BETWEEN the MFMA groups (a piece's issue cost hides in the matrix pipe's shadow instead of
// standing in front of the fragment reads).
template <int TM, int TN, int WM, bool AK, bool BKM, int BK, int NST, bool GA, bool GB>
__global__ __launch_bounds__(WM * 128, (WM == 2 && NST * 3 * (32 * TM * WM + 64 * TN) * BK * 2 <= 80 * 1024) ? 2 : (WM == 4 ? 2 : 1))
void gemm_pl_kernel(PlOperand A, PlOperand B, float* __restrict__ C, long ldc, int M, int N, int K, int k_chunk, int gx, int gy, EpiParams ep,
                    int k_valid, int m_base) {
    int tile_x, tile_y, tile_z;
    {   // XCD-aware tile order (gemm_bf16.hip)
        const int nblk = gridDim.x, id = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, j = id >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_x = swz % gx;
        tile_y = (swz / gx) % gy;
        tile_z = swz / (gx * gy);
    }
    constexpr int NW = WM * 2;                                       // waves per block
    constexpr int BM = 32 * TM * WM, BN = 64 * TN;
    static_assert(!AK || BM % 128 == 0, "k-major tiles are made of 128-column sub-tiles");
    static_assert(!BKM || BN % 128 == 0, "k-major tiles are made of 128-column sub-tiles");
    constexpr int A_T = BM * BK * 2, B_T = BN * BK * 2;              // bytes of one plane tile
    constexpr int STAGE = 3 * (A_T + B_T);
    constexpr int PA = A_T / 1024, PB = B_T / 1024;                  // DMA pieces (1 KiB) per plane tile
    constexpr int PTOT = 3 * (PA + PB), PPW = PTOT / NW;             // pieces per stage / per wave
    static_assert(PTOT % NW == 0 && A_T % 1024 == 0 && B_T % 1024 == 0, "piece count must split over the waves");
    constexpr int EPI_BYTES = NW * 32 * 36 * 4;                      // gemm_epilogue: one 32 x 36 fp32 patch per wave
    constexpr int LDS_BYTES = (NST * STAGE > EPI_BYTES) ? NST * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)m_base + (long)tile_y * BM, n0 = (long)tile_x * BN;      // m_base: first row of this launch (a row range of the problem)
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin) / BK;

    // ---- this wave's pieces: piece g = wave * PPW + j -> (operand, plane, sub-piece)
    const __bf16* src[PPW];
    unsigned dst[PPW];             // byte offset inside a stage (wave-uniform)
    long step[PPW];                // elements the source advances per K-tile (wave-uniform); gathers: see issue_piece()
    int kpos[PPW];                 // GB pieces: output position (= k index) of this lane's row in the next tile
    bool is_ga[PPW], is_gb[PPW];
    const float rF2 = GB ? 1.0f / (float)B.F2 : 0.f, rT2 = GB ? 1.0f / (float)B.T2 : 0.f;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int g = wave * PPW + j;
        const bool isA = g < 3 * PA;
        const int gg = isA ? g : g - 3 * PA;
        const int P = isA ? PA : PB;
        const int plane = gg / P, sub = gg - plane * P;
        const PlOperand& X = isA ? A : B;
        const bool km = isA ? AK : BKM;
        const long x0 = isA ? m0 : n0;
        dst[j] = (unsigned)((isA ? 0 : 3 * A_T) + plane * (isA ? A_T : B_T) + sub * 1024);
        is_ga[j] = GA && isA;
        is_gb[j] = GB && !isA;
        kpos[j] = 0;
        const __bf16* base = X.p + plane * X.plane_stride;
        if (!km) {
            constexpr int CPR = BK * 2 / 16, RPP = 1024 / (BK * 2);
            const int row = sub * RPP + lane / CPR, cpos = lane % CPR;
            const int csrc = cpos ^ pl_fsw<BK>(row);
            const long grow = min(x0 + row, (long)X.rows_total - 1);
            if (GA && isA) {
                const int f = (int)(grow % X.F2);
                const long q = grow / X.F2;
                const int t = (int)(q % X.T2);
                const long b = q / X.T2;
                src[j] = base + ((b * X.T1 + X.S * t) * (long)X.F1 + X.S * f) * X.C + csrc * 8;     // k offsets are added per tile
            } else {
                src[j] = base + grow * X.ld + k_begin + csrc * 8;
            }
            step[j] = BK;
        } else {
            // sub-tile (128 columns) s = sub / (BK / 4), k-rows 4 (sub % (BK / 4)) .. + 3 of it
            const int st = sub / (BK / 4);
            const int krow = (sub - st * (BK / 4)) * 4 + (lane >> 4), cpos = lane & 15;
            const int csrc = cpos ^ pl_gsw(krow);
            const long col = min(x0 + st * 128 + csrc * 8, (long)X.rows_total - 8);
            if (GB && !isA) {
                // columns = (kh, kw, ci) of the im2col row: linear inside a kernel row, + kh * F1 * C across rows
                const int seg = X.KS * X.C;
                const int kh = (int)(col / seg);
                src[j] = base + (long)kh * X.F1 * X.C + (col - (long)kh * seg);
                kpos[j] = k_begin + krow;
            } else {
                src[j] = base + (long)(k_begin + krow) * X.ld + col;
            }
            step[j] = (long)BK * X.ld;
        }
    }
    // running K position of a gathered A: kernel row kh and offset inside it (k = kh * KS * C + rem), wave-uniform.
    // Channel-chunk-major order (A.korder, k_begin = 0): 32 channels at a time, all KH x KS taps of them back to back -
    // ga_rem is then the column kw, ga_c the channel offset (chunk * 32 + the K-tile's half of the chunk when BK is 16)
    int ga_kh = 0, ga_rem = 0, ga_c = 0;
    if (GA && !A.korder) { const int seg = A.KS * A.C; ga_kh = k_begin / seg; ga_rem = k_begin - ga_kh * seg; }
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    int tiles_issued = 0;           // tiles whose pieces have been issued so far (wave-uniform); the last one is re-issued past the end
    unsigned stage_off = 0;         // stage the next issue writes
    long ga_off = 0;
    auto begin_issue = [&]() {
        if (!GA) ga_off = 0;
        else if (!A.korder) ga_off = (long)ga_kh * A.F1 * A.C + ga_rem;
        else ga_off = ((long)ga_kh * A.F1 + ga_rem) * A.C + ga_c;
    };
    auto issue_piece = [&](int j) {
        const unsigned sb = lds_base + stage_off;
        const bool advance = tiles_issued + 1 < nk;                  // wave-uniform: a select, not a branch
        if (is_ga[j]) {
            pl_dma16(src[j] + ga_off, sb + dst[j]);
        } else if (is_gb[j]) {
            int f, t;
            // (reduction padded to the K-tile, oe_gemm_args.planes_k_padded: positions past the last one read the last one - A is zero there)
            const int q = pl_div_small(min(kpos[j], k_valid - 1), B.F2, rF2, f);
            const int b = pl_div_small(q, B.T2, rT2, t);
            pl_dma16(src[j] + (((long)b * B.T1 + B.S * t) * B.F1 + B.S * f) * B.C, sb + dst[j]);
            kpos[j] += advance ? BK : 0;
        } else {
            pl_dma16(src[j], sb + dst[j]);
            src[j] += advance ? step[j] : 0;
        }
    };
    auto end_issue = [&]() {
        const bool advance = tiles_issued + 1 < nk;
        if (GA && advance) {
            if (!A.korder) { ga_rem += BK; if (ga_rem >= A.KS * A.C) { ga_rem -= A.KS * A.C; ++ga_kh; } }
            else if (BK < 32 && (ga_c & 31) + BK < 32) ga_c += BK;                    // the next part of the same chunk and tap
            else {
                ga_c &= ~31;                                                          // next tap of the chunk, or the next chunk
                if (++ga_rem == A.KS) { ga_rem = 0; if (++ga_kh == A.KH) { ga_kh = 0; ga_c += 32; } }
            }
        }
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

    // fused bias gradient of a weight gradient (k-major A = dY): column sums of A, taken from the fragments the MFMAs read
    const bool do_csum = AK && ep.a_colsum != nullptr && tile_x == 0 && wn == 0;        // wave-uniform
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {
        begin_issue();
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(j);
        end_issue();
    }

    constexpr int KS = BK / 16, NG = KS * TM * TN;                   // MFMA groups (six MFMAs each) per K-tile
    constexpr int PPG = (PPW + NG - 1) / NG;                         // pieces issued behind each of the first groups
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the NST - 2 younger tiles' pieces are outstanding; the barrier then (a) publishes
        // every wave's pieces of tile kt and (b) retires all reads of tile kt-1's stage, which this iteration's issue overwrites
        pl_wait_and_barrier<(NST - 2) * PPW>();
        const unsigned char* at = lds + (kt % NST) * STAGE;
        const unsigned char* bt = at + 3 * A_T;
        PlFrag fa[KS][TM], fb[KS][TN];
        // fragments are read one MFMA group ahead of their use: the first group's reads stand alone, every later group's
        // reads fly under the previous group's six MFMAs (all 18 reads up front would put their latency and the LDS's
        // 256 B / clock in front of the first MFMA of every wave of the CU at once)
        auto read_group = [&](auto gi) {
            constexpr int g = decltype(gi)::value;
            constexpr int ks = g / (TM * TN), i = (g / TN) % TM, j = g % TN;
            if constexpr (j == 0) {
                if (!AK) pl_row_frag<BK>(at, A_T, wm * 32 * TM + i * 32 + frow, fhalf, ks, fa[ks][i]);
                else pl_col_frag(at, A_T, BK * 256, wm * 32 * TM + i * 32, ks, lane, fa[ks][i]);
            }
            if constexpr (i == 0) {
                if (!BKM) pl_row_frag<BK>(bt, B_T, wn * 32 * TN + j * 32 + frow, fhalf, ks, fb[ks][j]);
                else pl_col_frag(bt, B_T, BK * 256, wn * 32 * TN + j * 32, ks, lane, fb[ks][j]);
            }
        };
        read_group(std::integral_constant<int, 0>{});
        begin_issue();
        static_for<0, NG>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            constexpr int ks = g / (TM * TN), i = (g / TN) % TM, j = g % TN;
            if constexpr (g + 1 < NG) read_group(std::integral_constant<int, g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[i][j] = oe_mma_terms<6>(fa[ks][i], fb[ks][j], acc[i][j]);
            if (AK && j == 0 && do_csum) csum[i] = pl_frag_sum(fa[ks][i], csum[i]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, PPG>([&](auto qi) {
                constexpr int pj = g * PPG + decltype(qi)::value;
                if constexpr (pj < PPW) issue_piece(pj);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        end_issue();
    }
    // the surplus pieces (issued past the end of the range) must have landed before the epilogue reuses the LDS
    pl_wait_and_barrier<0>();
    if (AK && do_csum) {
        float al = ep.alpha;
        if (ep.alpha_dev) al *= *ep.alpha_dev;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
            const long row = m0 + wm * 32 * TM + i * 32 + frow;
            if (fhalf == 0 && row < M) atomicAdd(ep.a_colsum + row, v * al);
        }
    }
    gemm_epilogue<TM, TN, WM>(acc, reinterpret_cast<float*>(lds), C, ldc, M, N, m0, n0, ep, tile_z);
}

static long pl_launches = 0;
extern "C" long oe_gemm_pl_launches(void) { return pl_launches; }

// dispatch knobs (tests and tools/pl_bench.py; -1 keeps a value): min_blocks = smallest grid the kernel accepts, tile =
// forced tile (22 / 11, 0 = automatic), bk = forced K-tile of the 128 x 128 tiles (16 / 32, 0 = automatic), waves = 8 or 4
static int pl_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static int pl_min_blocks = pl_env("OE_PL_MIN_BLOCKS", 96), pl_forced_tile = pl_env("OE_PL_TILE", 0), pl_forced_bk = pl_env("OE_PL_BK", 0),
           pl_waves = pl_env("OE_PL_WAVES", 8), pl_t44_min = pl_env("OE_PL_T44_MIN", 1024), pl_hybrid = pl_env("OE_PL_HYBRID", 1);
extern "C" int oe_gemm_pl_config(int min_blocks, int tile, int bk, int waves) {
    if (min_blocks >= 0) pl_min_blocks = min_blocks;
    if (tile >= 0) pl_forced_tile = tile;
    if (bk >= 0) pl_forced_bk = bk;
    if (waves >= 0) pl_waves = waves;
    return 0;
}

// whole rounds of 256 x 256 tiles in front of the 128 x 256 ones (tests / A-B runs): returns the previous setting
extern "C" int oe_gemm_pl_hybrid(int on) {
    const int was = pl_hybrid;
    if (on >= 0) pl_hybrid = on;
    return was;
}

static int pl_k_valid = 0;          // set by oe_gemm_pl_try around a launch whose K was padded (host-side, same thread as the launch)
template <int TM, int TN, int WM, bool AK, bool BKM, int BK, int NST, bool GA, bool GB>
static int launch_pl(const PlOperand& A_, const PlOperand& B, float* C, long ldc, int M, int N, int K, int sk, const EpiParams& ep, hipStream_t st,
                     int m_base = 0) {
    // rows m_base .. M - 1 of the problem (m_base a multiple of the tile height; A's rows stop at M for this launch)
    PlOperand A = A_;
    if (!AK) A.rows_total = M;
    int kc = oe_cdiv(oe_cdiv(K, sk), BK) * BK;
    if (kc <= 0) kc = BK;
    const int nz = oe_cdiv(K, kc);
    const int gx = oe_cdiv(N, 64 * TN), gy = oe_cdiv(M - m_base, 32 * TM * WM);
    hipLaunchKernelGGL((gemm_pl_kernel<TM, TN, WM, AK, BKM, BK, NST, GA, GB>), dim3(gx * gy * nz), dim3(WM * 128), 0, st, A, B, C, ldc, M, N, K, kc, gx, gy, ep,
                       pl_k_valid > 0 ? pl_k_valid : K, m_base);
    OE_LAUNCH_CHECK("oe_gemm (bf16x6 planes)");
    ++pl_launches;
    return 0;
}

// Returns 1 when the problem does not qualify (the caller goes on to the kernels that split fp32 operands themselves),
// 0 on a launch.  Ap / Bp: plane 0 of the operands' pre-split copies (plane strides in elements), same logical layout and
// leading dimensions as the fp32 operands they mirror.
static int pl_try_impl(const OperandDesc& A, const OperandDesc& B, const void* Ap, long a_pstride, const void* Bp, long b_pstride, float* C, long ldc,
                       int M, int N, int K, int sk, const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, hipStream_t st, int korder);
int oe_gemm_pl_try(const OperandDesc& A, const OperandDesc& B, const void* Ap, long a_pstride, const void* Bp, long b_pstride, float* C, long ldc,
                   int M, int N, int K, int sk, const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, hipStream_t st, int korder,
                   bool k_padded) {
    // a weight gradient over a reduction that is not a whole number of K-tiles: the caller vouches for zero rows of A (and readable
    // rows of B) up to the next multiple of 16 - run the padded length, gathers clamp to the last valid position
    if (k_padded && a_kmajor && b_kmajor && ep.atomic && K % 16 != 0 && K > 16) {
        pl_k_valid = K;
        const int r = pl_try_impl(A, B, Ap, a_pstride, Bp, b_pstride, C, ldc, M, N, (K + 15) / 16 * 16, sk, ep, a_kmajor, b_kmajor, ga, gb, st, korder);
        pl_k_valid = 0;
        return r;
    }
    return pl_try_impl(A, B, Ap, a_pstride, Bp, b_pstride, C, ldc, M, N, K, sk, ep, a_kmajor, b_kmajor, ga, gb, st, korder);
}
static int pl_try_impl(const OperandDesc& A, const OperandDesc& B, const void* Ap, long a_pstride, const void* Bp, long b_pstride, float* C, long ldc,
                       int M, int N, int K, int sk, const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, hipStream_t st, int korder) {
    static const int mode = getenv("OE_GEMM_PL") ? atoi(getenv("OE_GEMM_PL")) : 1;            // 0 = never (A/B comparisons)
    if (!mode || !Ap || !Bp) return 1;
    if (a_kmajor && !b_kmajor) return 1;
    if (ep.a_colsum && !a_kmajor) return 1;
    PlOperand a{}, b{};
    a.p = (const __bf16*)Ap; a.ld = A.ld; a.plane_stride = a_pstride; a.rows_total = M;
    b.p = (const __bf16*)Bp; b.ld = B.ld; b.plane_stride = b_pstride; b.rows_total = N;
    auto aligned = [](const void* p, long ld, long ps) { return (((uintptr_t)p) & 15) == 0 && ld % 8 == 0 && ps % 8 == 0; };
    if (ga) { a.T1 = A.T1; a.F1 = A.F1; a.T2 = A.T2; a.F2 = A.F2; a.C = A.C; a.KS = A.KS; a.S = A.S; a.ld = 0; a.korder = korder; a.KH = K / (A.KS * A.C); }
    if (korder && (!ga || sk > 1 || A.C % 32 || K != a.KH * A.KS * A.C)) return 1;
    if (gb) { b.T1 = B.T1; b.F1 = B.F1; b.T2 = B.T2; b.F2 = B.F2; b.C = B.C; b.KS = B.KS; b.S = B.S; b.ld = 0; }
    if (!aligned(Ap, ga ? 8 : A.ld, a_pstride) || !aligned(Bp, gb ? 8 : B.ld, b_pstride)) return 1;
    if ((a_kmajor && (M % 8 || M < 8)) || (b_kmajor && (N % 8 || N < 8))) return 1;
    if (ga && (a_kmajor || b_kmajor || A.C % 32 || (A.KS * A.C) % 32)) return 1;
    if (gb && (!(a_kmajor && b_kmajor) || B.C % 8 || (B.KS * B.C) % 128 || N % 128 || K >= (1 << 24))) return 1;
    // Tile and K-tile.  128 x 128 x 16 with three stages (72 KiB: two blocks per CU overlap each other's prologue and epilogue
    // on the short-K problems) while the grid still covers the chip; 128 x 128 x 32 (144 KiB, one block per CU) for long
    // reductions; 64 x 64 x 32 (72 KiB) where 128 x 128 tiles would leave CUs idle (N = 256 outputs).  k-major tiles are 128
    // columns wide, so a k-major operand pins its side of the tile.
    const int forced_bk = pl_forced_bk, forced_tile = pl_forced_tile;
    const long b22 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128) * sk;
    int tile = 22;
    if (!a_kmajor && !b_kmajor && !ga && b22 < 320) tile = 11;
    // 128 x 256: the conv2 GEMMs (~1200 blocks) and the wide-output Linears whose 128 x 256 grid is about one round of the chip
    // (7936 x 1024 x 256: 248 blocks, 30.6 us against 34.9 with 496 tiles of 128 x 128; 7936 x 768: 25.7 against 29.2)
    if (!a_kmajor && !b_kmajor && N % 256 == 0 && K % 32 == 0) {
        const long b24 = (long)oe_cdiv(M, 128) * (N / 256) * sk;
        if (b24 >= 512 || (N >= 768 && b24 >= 160)) tile = 24;
    }
    // 256 x 256 tiles (8 waves of 64 x 128, K-tile 16, three stages): a K-tile moves 48 KiB for 96 MFMAs per wave - half the
    // LDS-DMA bytes per MFMA of the 128 x 128 tile, which is what bounds that one (L2 -> LDS fill rate).  For the long,
    // wide problems whose grid still fills the chip: conv2 (forward, input and weight gradients), the 16 s batches.
    const long b44 = (long)oe_cdiv(M, 256) * oe_cdiv(N, 256);
    if (M >= 256 && N >= 256 && N % 256 == 0 && (!a_kmajor || M % 256 == 0) && K % 16 == 0 &&
        (ep.atomic ? b44 * (K / 2048) >= 200 : b44 >= pl_t44_min)) tile = 44;     // (conv2 forward, 589 such tiles = 2.3 rounds: 128 x 256 is faster)
    if (forced_tile) tile = forced_tile;
    if ((a_kmajor || b_kmajor) && tile != 22 && tile != 44) return 1;
    if (ga && tile != 22 && tile != 24 && tile != 44) return 1;
    if (tile == 44 && ep.atomic && !forced_tile) {
        // own split of the reduction: about one block per CU, K-ranges of at least 2048 (the output is accumulated with
        // atomics into a zeroed buffer whatever the caller's split was)
        sk = (int)max(1L, min((long)(K / 2048), 256 / b44));
    }
    // K-tile 32 wherever K allows (measured faster than 16 on every Linear shape, K = 256 included: half the barriers)
    int bk = (tile == 44) ? 16 : 32;                                               // tiles 11 and 24 exist with K-tiles of 32 only
    if (forced_bk && tile == 22) bk = forced_bk;
    if (K % bk) { if (tile == 22 && bk == 32 && K % 16 == 0) bk = 16; else return 1; }
    if (sk > 1 && (long)oe_cdiv(oe_cdiv(K, sk), bk) * bk * (sk - 1) >= K) return 1;          // a split would be left empty
    // too few blocks to occupy the chip: the splitting kernels have smaller tiles and split the reduction
    const int min_blocks = pl_min_blocks;
    if ((long)oe_cdiv(M, tile == 11 ? 64 : tile == 44 ? 256 : 128) * oe_cdiv(N, tile == 11 ? 64 : (tile == 24 || tile == 44) ? 256 : 128) * sk < min_blocks) return 1;
    if (tile == 44) {
#define OE_PL44(AK, BKM, GA, GB) return launch_pl<2, 4, 4, AK, BKM, 16, 3, GA, GB>(a, b, C, ldc, M, N, K, sk, ep, st)
        if (!a_kmajor && !b_kmajor) { if (ga) OE_PL44(false, false, true, false); else OE_PL44(false, false, false, false); }
        if (!a_kmajor && b_kmajor) OE_PL44(false, true, false, false);
        if (gb) OE_PL44(true, true, false, true);
        OE_PL44(true, true, false, false);
#undef OE_PL44
    }
    // 128 x 128 tiles: 8 waves (4 x 2 of 32 x 64 each, two per SIMD) by default; OE_PL_WAVES=4 takes the 2 x 2 arrangement
    const int waves = pl_waves;
#define OE_PL(AK, BKM, GA, GB)                                                                                        \
    do {                                                                                                              \
        if (waves == 8) {                                                                                             \
            if (bk == 32) return launch_pl<1, 2, 4, AK, BKM, 32, 3, GA, GB>(a, b, C, ldc, M, N, K, sk, ep, st);       \
            return launch_pl<1, 2, 4, AK, BKM, 16, 4, GA, GB>(a, b, C, ldc, M, N, K, sk, ep, st);                     \
        }                                                                                                             \
        if (bk == 32) return launch_pl<2, 2, 2, AK, BKM, 32, 3, GA, GB>(a, b, C, ldc, M, N, K, sk, ep, st);           \
        return launch_pl<2, 2, 2, AK, BKM, 16, 3, GA, GB>(a, b, C, ldc, M, N, K, sk, ep, st);                         \
    } while (0)
    if (!a_kmajor && !b_kmajor) {
        if (tile == 11) return launch_pl<1, 1, 2, false, false, 32, 3, false, false>(a, b, C, ldc, M, N, K, sk, ep, st);
        // 128 x 256 tiles (8 waves of 32 x 128, K-tile 32, two stages): a whole 256-wide output row per block - A is read
        // once instead of once per column tile and a K-tile moves 72 KiB for 96 MFMAs per wave instead of 48 KiB for 48.
        // Worth it where the grid still fills the chip (the conv2 forward / input-gradient GEMMs: ~1200 blocks)
        if (tile == 24 && K % 32 == 0) {
            // Whole rounds of 256 x 256 tiles first, the rest on 128 x 256: the big tile moves half the LDS-DMA bytes per MFMA (52-59 % matrix-
            // pipe busy against 45-50) but its grid is coarse - conv2 forward is 589 of them = 2.3 rounds of the chip, which as THREE
            // rounds lost to the five rounds of 128 x 256 tiles.  Two full rounds of the big tile + one round of the small one take
            // 2 x 1.84 + 1 = 4.7 small-tile rounds (measured: conv2 forward 1068 -> 1001 us, the input gradient's K = 1024 / 512 classes
            // 547 -> 507, 342 -> 327, 333 -> 324; its K = 256 class is 6 us SLOWER - sixteen K-tiles of 16 do not amortise the big tile's
            // epilogue - hence K >= 512).  (Rows [0, m1) and [m1, M) of the same problem: two launches, same operands.)
            int m1 = 0;
            if (pl_hybrid && sk == 1 && !forced_tile && K % 16 == 0 && K >= 512 && M >= 256) {
                const int gxn = N / 256;
                const long t44 = (long)(M / 256) * gxn;                       // whole big tiles
                const long full = t44 / 256;                                   // whole rounds of them
                if (full >= 1 && (full * 256) % gxn == 0) {
                    const long rows1 = full * 256 / gxn * 256;
                    const long rest24 = (long)oe_cdiv(M - rows1, 128) * gxn;
                    const double cost_a = (double)oe_cdiv((long)oe_cdiv(M, 128) * gxn, 256L);
                    const double cost_b = 1.85 * full + (double)oe_cdiv(rest24, 256L);
                    if (rows1 < M && cost_b < cost_a) m1 = (int)rows1;
                }
            }
            if (m1 > 0) {
                if (ga) launch_pl<2, 4, 4, false, false, 16, 3, true, false>(a, b, C, ldc, m1, N, K, 1, ep, st);
                else launch_pl<2, 4, 4, false, false, 16, 3, false, false>(a, b, C, ldc, m1, N, K, 1, ep, st);
            }
            if (ga) return launch_pl<1, 4, 4, false, false, 32, 2, true, false>(a, b, C, ldc, M, N, K, sk, ep, st, m1);
            return launch_pl<1, 4, 4, false, false, 32, 2, false, false>(a, b, C, ldc, M, N, K, sk, ep, st, m1);
        }
        if (ga) OE_PL(false, false, true, false); else OE_PL(false, false, false, false);
    }
    if (!a_kmajor && b_kmajor) OE_PL(false, true, false, false);
    if (gb) OE_PL(true, true, false, true);
    OE_PL(true, true, false, false);
#undef OE_PL
}
