// bf16-input / fp32-accumulate GEMM on the CDNA4 matrix cores
// (v_mfma_f32_32x32x16_bf16, dense peak ~2.5 PFLOP/s = 16x the fp32-input rate).
// Operands stay fp32 in HBM and are converted while they are staged into LDS, so
// every other kernel of the path is unchanged.  Two precisions:
//   terms = 1 : a*b ~ hi(a)*hi(b)                       (bf16 products, ~3e-3 relative)
//   terms = 3 : a*b ~ hi*hi + hi*lo + lo*hi, lo = bf16(x - hi(x))
//               (error ~2^-17 per product at 3/16 of the fp32-MFMA cost)
//   terms = 6 : three exact pieces per operand, six products (oe_common.h): the fp32 product to within one fp32
//               rounding at 6/16 of the fp32-MFMA cost - the arithmetic that stands in for the reference's fp32 GEMMs
// Same operand addressing (plain / conv2 im2col gather) and epilogue as gemm.hip.
//
// Tiling: 256 threads = 2x2 waves, each wave TM x TN tiles of 32x32; K-tile 32.
// LDS tiles are [row][k] bf16 with an 80-byte row pitch: the MFMA fragment read is one
// ds_read_b128 per lane and is bank-conflict free (rows r..r+15 start on 16 distinct
// 4-bank slots: 20*r mod 64).  k-major operands are transposed in registers on the way in
// (4x4 micro-blocks, lane -> (k-block, row-block) = (l&7, l>>3) so that both the 128-byte
// global segments and the ds_write_b64 pattern are conflict free).
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BK2 32
#define PITCH 40   // bf16 elements per LDS row (32 data + 8 pad) = 80 bytes

template <int NPL> struct Frag { bf16x8 p[NPL]; };

template <int NPL>
__device__ __forceinline__ void split4(const float4& v, bf16x4 (&pl)[NPL]) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 q[NPL];
        oe_split_bf16<NPL>(x[e], q);
#pragma unroll
        for (int n = 0; n < NPL; ++n) pl[n][e] = q[n];
    }
}
// the NPL pieces of four values to plane n at `dst + n * plane_stride`
template <int NPL>
__device__ __forceinline__ void store4(const float4& v, __bf16* dst, int plane_stride) {
    bf16x4 pl[NPL];
    split4<NPL>(v, pl);
#pragma unroll
    for (int n = 0; n < NPL; ++n) *reinterpret_cast<bf16x4*>(dst + n * plane_stride) = pl[n];
}

// One operand's staging state: ROWS x 32 tile.
//   row-major (k contiguous): slot -> (row = idx/8, kq = idx%8), one float4 along k
//   k-major (rows contiguous): micro-block idx -> kb = idx%8, mb = (idx/8)%8 + 8*(idx/64); four float4 along rows
template <int ROWS, bool KMAJOR, bool GATHER>
struct Stage {
    static constexpr int NIDX = KMAJOR ? ROWS * 2 : ROWS * 8;           // work items per tile
    static constexpr int NS = (NIDX + 255) / 256;                      // per thread
    static constexpr bool PARTIAL = (NIDX % 256) != 0;               // last slot only partly populated
    float4 reg[NS][KMAJOR ? 4 : 1];
    long fix[NS];
    const float* ptr[NS];     // fast path: address of this slot's data for the current K-tile
    float4 csum[NS];          // k-major only: running sum over k of this slot's 4 rows (fused bias gradient)
    int gf[NS], gt[NS]; long gb[NS];   // k-major im2col gather: (f, t, b) of this slot's next position (no divisions in the loop)

    __device__ __forceinline__ void init(const OperandDesc& d, long i0, long limit) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!KMAJOR) {
                const long r = i0 + idx / 8;
                fix[s] = (idx < NIDX && r < limit) ? addr_row<GATHER>(d, r) : -1;
            } else {
                const int mb = ((idx >> 3) & 7) + 8 * (idx >> 6);
                fix[s] = (idx < NIDX) ? addr_col<GATHER>(d, i0 + 4 * mb) : -1;
            }
        }
    }
    // position counters of the first row this slot will read (k-major gather only)
    __device__ __forceinline__ void init_gather_pos(const OperandDesc& d, int k0) {
        if (KMAJOR && GATHER) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const long r = k0 + 4 * ((threadIdx.x + s * 256) & 7);
                gf[s] = (int)(r % d.F2);
                const long q = r / d.F2;
                gt[s] = (int)(q % d.T2);
                gb[s] = q / d.T2;
            }
        }
    }
    // Fast path (tile fully inside the matrix, vector loads legal, no gather): plain pointers that
    // advance by a wave-uniform stride per K-tile - no bounds checks, no 64-bit index math in the loop.
    __device__ __forceinline__ void zero_csum() {
#pragma unroll
        for (int s = 0; s < NS; ++s) csum[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void add_csum() {       // call once per loaded K-tile (k-major operands)
        if (KMAJOR) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    csum[s].x += reg[s][i].x; csum[s].y += reg[s][i].y; csum[s].z += reg[s][i].z; csum[s].w += reg[s][i].w;
                }
        }
    }
    // out[i0 + 4*mb + j] += alpha * csum: lanes that differ only in kb (lane bits 0-2) hold partial sums
    __device__ __forceinline__ void flush_csum(float* out, long i0, long limit, float alpha) {
        if (KMAJOR) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float4 v = csum[s];
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
                    v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
                }
                const int idx = threadIdx.x + s * 256;
                if ((idx & 7) == 0 && idx < NIDX) {
                    const long c = i0 + 4 * (((idx >> 3) & 7) + 8 * (idx >> 6));
                    if (c < limit) atomicAdd(out + c, v.x * alpha);
                    if (c + 1 < limit) atomicAdd(out + c + 1, v.y * alpha);
                    if (c + 2 < limit) atomicAdd(out + c + 2, v.z * alpha);
                    if (c + 3 < limit) atomicAdd(out + c + 3, v.w * alpha);
                }
            }
        }
    }
    __device__ __forceinline__ void init_fast(const OperandDesc& d, long i0, int k0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!KMAJOR) ptr[s] = d.p + (i0 + (idx >> 3)) * d.ld + k0 + (idx & 7) * 4;
            else ptr[s] = d.p + (long)(k0 + 4 * (idx & 7)) * d.ld + i0 + 4 * (((idx >> 3) & 7) + 8 * (idx >> 6));
        }
    }
    // Fast path of a row-major im2col gather (conv forward, parity-class dgrad): one pointer per slot at its row's first
    // input position, plus a wave-uniform offset per K-tile.  C is a multiple of the K-tile, so a tile never straddles a
    // kernel row (addr_col is linear inside one); rows past `limit` (ragged last M-tile) re-read the last valid row - the
    // epilogue discards them.
    __device__ __forceinline__ void init_fast_gather(const OperandDesc& d, long i0, long limit) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            long r = i0 + (idx >> 3);
            if (r >= limit) r = limit - 1;
            ptr[s] = d.p + addr_row<true>(d, r) + (idx & 7) * 4;
        }
    }
    __device__ __forceinline__ void load_fast_gather(const OperandDesc& d, int k0) {
        const long co = addr_col<true>(d, k0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!PARTIAL || idx < NIDX) reg[s][0] = *reinterpret_cast<const float4*>(ptr[s] + co);
        }
    }
    __device__ __forceinline__ void load_fast(long ld) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!PARTIAL || idx < NIDX) {
                if (!KMAJOR) {
                    reg[s][0] = *reinterpret_cast<const float4*>(ptr[s]);
                    ptr[s] += BK2;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) reg[s][i] = *reinterpret_cast<const float4*>(ptr[s] + i * ld);
                    ptr[s] += BK2 * ld;
                }
            }
        }
    }
    __device__ __forceinline__ void load(const OperandDesc& d, long i0, long limit, int k0, int k_end) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!KMAJOR) {
                const int k = k0 + (idx & 7) * 4;
                const int nv = (fix[s] < 0) ? 0 : max(0, min(4, k_end - k));
                reg[s][0] = nv ? load4(d.p + fix[s] + addr_col<GATHER>(d, k), nv, d.vec_ok) : make_float4(0, 0, 0, 0);
            } else {
                const int kb = idx & 7;
                const int mb = ((idx >> 3) & 7) + 8 * (idx >> 6);
                const long c = i0 + 4 * mb;
                const int nvr = (fix[s] < 0) ? 0 : (int)max(0L, min(4L, limit - c));
                if (GATHER) {
                    // im2col rows: walk (f, t, b) forward one output position at a time, carry instead of divide
                    int f = gf[s], t = gt[s];
                    long bb = gb[s];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k0 + 4 * kb + i;
                        const long base = ((bb * d.T1 + d.S * t) * (long)d.F1 + d.S * f) * d.C;
                        reg[s][i] = (nvr && k < k_end) ? load4(d.p + base + fix[s], nvr, d.vec_ok) : make_float4(0, 0, 0, 0);
                        if (++f == d.F2) { f = 0; if (++t == d.T2) { t = 0; ++bb; } }
                    }
                    f += BK2 - 4;                                  // this slot's rows of the next K-tile start 32 further
                    while (f >= d.F2) { f -= d.F2; if (++t == d.T2) { t = 0; ++bb; } }
                    gf[s] = f; gt[s] = t; gb[s] = bb;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k0 + 4 * kb + i;
                        reg[s][i] = (nvr && k < k_end) ? load4(d.p + addr_row<GATHER>(d, k) + fix[s], nvr, d.vec_ok)
                                                       : make_float4(0, 0, 0, 0);
                    }
                }
            }
        }
    }
    // tile = plane 0 of this operand's LDS image; plane n follows at n * plane_stride elements
    template <int TERMS>
    __device__ __forceinline__ void store(__bf16* tile, int plane_stride) {
        constexpr int NPL = oe_npl<TERMS>::N;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = threadIdx.x + s * 256;
            if (!PARTIAL || idx < NIDX) {
            if (!KMAJOR) {
                const int row = idx >> 3, kq = (idx & 7) * 4;
                store4<NPL>(reg[s][0], tile + row * PITCH + kq, plane_stride);
            } else {
                const int kb = idx & 7;
                const int mb = ((idx >> 3) & 7) + 8 * (idx >> 6);
                const float4 r0 = reg[s][0], r1 = reg[s][1], r2 = reg[s][2], r3 = reg[s][3];
                const float4 t0 = make_float4(r0.x, r1.x, r2.x, r3.x), t1 = make_float4(r0.y, r1.y, r2.y, r3.y);
                const float4 t2 = make_float4(r0.z, r1.z, r2.z, r3.z), t3 = make_float4(r0.w, r1.w, r2.w, r3.w);
                __bf16* hp = tile + (4 * mb) * PITCH + 4 * kb;
                store4<NPL>(t0, hp, plane_stride);
                store4<NPL>(t1, hp + PITCH, plane_stride);
                store4<NPL>(t2, hp + 2 * PITCH, plane_stride);
                store4<NPL>(t3, hp + 3 * PITCH, plane_stride);
            }
            }
        }
    }
};

template <int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, bool GATHER_A, bool GATHER_B, int TERMS, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(OperandDesc A, OperandDesc B, float* __restrict__ C, long ldc,
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
    constexpr int A_TILE = BM * PITCH, B_TILE = BN * PITCH;            // bf16 elements
    constexpr int NT = oe_npl<TERMS>::N;                                // planes per operand: hi (+ lo | + mid, lo)
    constexpr int STAGE_ELEMS = NT * (A_TILE + B_TILE);
    constexpr int LDS_BYTES = (2 * STAGE_ELEMS * 2 > 4 * 32 * 36 * 4) ? 2 * STAGE_ELEMS * 2 : 4 * 32 * 36 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    __bf16* lds16 = reinterpret_cast<__bf16*>(lds_raw);
    auto a_hi = [&](int buf) { return lds16 + buf * STAGE_ELEMS; };                   // planes of A: stride A_TILE
    auto b_hi = [&](int buf) { return lds16 + buf * STAGE_ELEMS + NT * A_TILE; };     // planes of B: stride B_TILE

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)tile_y * BM, n0 = (long)tile_x * BN;
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin + BK2 - 1) / BK2;

    Stage<BM, A_KMAJOR, GATHER_A> sa;
    Stage<BN, B_KMAJOR, GATHER_B> sb;
    // FAST (chosen on the host): every tile interior, K-ranges multiples of the K-tile, vector loads legal
    constexpr bool fast = FAST;
    constexpr bool fast_ga = FAST && GATHER_A && !A_KMAJOR;          // A = im2col gather through plain pointers
    if (fast) {
        if (fast_ga) sa.init_fast_gather(A, m0, M); else sa.init_fast(A, m0, k_begin);
        sb.init_fast(B, n0, k_begin);
    } else {
        sa.init(A, m0, M);
        sb.init(B, n0, N);
        sa.init_gather_pos(A, k_begin);
        sb.init_gather_pos(B, k_begin);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_csum = A_KMAJOR && ep.a_colsum != nullptr && tile_x == 0;      // block-uniform
    if (A_KMAJOR) sa.zero_csum();
    if (nk > 0) {
        if (fast) { if (fast_ga) sa.load_fast_gather(A, k_begin); else sa.load_fast(A.ld); sb.load_fast(B.ld); }
        else { sa.load(A, m0, M, k_begin, k_end); sb.load(B, n0, N, k_begin, k_end); }
        if (do_csum) sa.add_csum();
        sa.template store<TERMS>(a_hi(0), A_TILE);
        sb.template store<TERMS>(b_hi(0), B_TILE);
    }
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            if (fast) { if (fast_ga) sa.load_fast_gather(A, k_begin + (kt + 1) * BK2); else sa.load_fast(A.ld); sb.load_fast(B.ld); }
            else { sa.load(A, m0, M, k_begin + (kt + 1) * BK2, k_end); sb.load(B, n0, N, k_begin + (kt + 1) * BK2, k_end); }
        }
        const __bf16* ah = a_hi(buf) + (wm * 32 * TM + frow) * PITCH + fk;
        const __bf16* bh = b_hi(buf) + (wn * 32 * TN + frow) * PITCH + fk;
#pragma unroll
        for (int ks = 0; ks < BK2 / 16; ++ks) {
            Frag<NT> fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < NT; ++n) fa[i].p[n] = *reinterpret_cast<const bf16x8*>(ah + n * A_TILE + i * 32 * PITCH + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) fb[j].p[n] = *reinterpret_cast<const bf16x8*>(bh + n * B_TILE + j * 32 * PITCH + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = oe_mma_terms<TERMS>(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
            if (do_csum) sa.add_csum();
            sa.template store<TERMS>(a_hi(buf ^ 1), A_TILE);
            sb.template store<TERMS>(b_hi(buf ^ 1), B_TILE);
        }
        __syncthreads();
    }
    if (do_csum) {
        float al = ep.alpha;
        if (ep.alpha_dev) al *= *ep.alpha_dev;
        sa.flush_csum(ep.a_colsum, m0, M, al);
    }
    gemm_epilogue<TM, TN>(acc, reinterpret_cast<float*>(lds_raw), C, ldc, M, N, m0, n0, ep, tile_z);
}

template <int TM, int TN, bool AK, bool BKM, bool GA, bool GB, int TERMS>
static int launch_bf16_(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int splitk,
                       const EpiParams& ep, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    int kc = oe_cdiv(oe_cdiv(K, splitk), BK2) * BK2;
    if (kc <= 0) kc = BK2;
    int nz = oe_cdiv(K, kc);
    if (nz < 1) nz = 1;
    const int gx = oe_cdiv(N, BN), gy = oe_cdiv(M, BM);
    const bool fast = !GA && !GB && A.vec_ok && B.vec_ok && (M % BM == 0) && (N % BN == 0) && (K % BK2 == 0) && (kc % BK2 == 0);
    // the im2col gather on a row-major A has a fast form too: interior N and K as above, any M (rows are clamped)
    const bool fast_ga = GA && !GB && !AK && A.vec_ok && B.vec_ok && A.C % BK2 == 0 && (N % BN == 0) && (K % BK2 == 0) && (kc % BK2 == 0);
    if (!GA && !GB && fast)
        hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, AK, BKM, false, false, TERMS, true>), dim3(gx * gy * nz), dim3(256), 0, st, A, B, C, ldc, M, N, K, kc, gx, gy, ep);
    else if (GA && !GB && !AK && fast_ga)
        hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, false, BKM, GA && !AK, false, TERMS, true>), dim3(gx * gy * nz), dim3(256), 0, st, A, B, C, ldc, M, N, K, kc, gx, gy, ep);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN, AK, BKM, GA, GB, TERMS, false>), dim3(gx * gy * nz), dim3(256), 0, st, A, B, C, ldc, M, N, K, kc, gx, gy, ep);
    OE_LAUNCH_CHECK("oe_gemm (bf16 mfma)");
    return 0;
}
#define launch_bf16 launch_bf16_

// called from oe_gemm_f32 (gemm.hip) when args->precision is 1 or 3
int oe_gemm_bf16_dispatch(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk,
                          const EpiParams& ep, bool a_kmajor, bool b_kmajor, bool ga, bool gb, int terms, hipStream_t st) {
    // largest tile that still gives the 256 CUs ~one block each
    const long b22 = (long)oe_cdiv(M, 128) * oe_cdiv(N, 128) * sk, b12 = (long)oe_cdiv(M, 64) * oe_cdiv(N, 128) * sk;
    // weight gradients (both operands k-major, long reduction): operand re-reads dominate, so take the
    // 128x128 tile as soon as the grid still covers the chip; elsewhere favour >= ~2 blocks per CU.
    const bool wg = a_kmajor && b_kmajor;
    int tile = (b22 >= (wg ? 200 : 400) && M >= 128 && N >= 128) ? 22 : (b12 >= 400 && N >= 128) ? 12 : 11;
    static const int forced_tile = getenv("OE_GEMM_TILE") ? atoi(getenv("OE_GEMM_TILE")) : 0;   // tuning aid (tools/gemm_bench.py)
    if (forced_tile) tile = forced_tile;
    if (wg && !ga && !gb) {   // weight gradients: bf16 planes in LDS, two K-groups of waves per block (gemm_tn.hip)
        const int r = oe_gemm_tn_planes_try(A, B, C, ldc, M, N, K, sk, ep, terms, st);
        if (r != 1) return r;
    }
    if (!ga) {   // interior, aligned problems: the LDS-DMA ring kernel
        const int r = oe_gemm_dma_try(A, B, C, ldc, M, N, K, sk, ep, a_kmajor, b_kmajor, gb, terms, tile, st);
        if (r != 1) return r;
    }
#define OE_DISP(AK, BKM, GA, GB)                                                                                 \
    do {                                                                                                         \
        if (terms == 6) {                                                                                        \
            if (tile == 22) return launch_bf16<2, 2, AK, BKM, GA, GB, 6>(A, B, C, ldc, M, N, K, sk, ep, st);     \
            if (tile == 12) return launch_bf16<1, 2, AK, BKM, GA, GB, 6>(A, B, C, ldc, M, N, K, sk, ep, st);     \
            return launch_bf16<1, 1, AK, BKM, GA, GB, 6>(A, B, C, ldc, M, N, K, sk, ep, st);                     \
        }                                                                                                        \
        if (terms == 3) {                                                                                        \
            if (tile == 22) return launch_bf16<2, 2, AK, BKM, GA, GB, 3>(A, B, C, ldc, M, N, K, sk, ep, st);     \
            if (tile == 12) return launch_bf16<1, 2, AK, BKM, GA, GB, 3>(A, B, C, ldc, M, N, K, sk, ep, st);     \
            return launch_bf16<1, 1, AK, BKM, GA, GB, 3>(A, B, C, ldc, M, N, K, sk, ep, st);                     \
        }                                                                                                        \
        if (tile == 22) return launch_bf16<2, 2, AK, BKM, GA, GB, 1>(A, B, C, ldc, M, N, K, sk, ep, st);         \
        if (tile == 12) return launch_bf16<1, 2, AK, BKM, GA, GB, 1>(A, B, C, ldc, M, N, K, sk, ep, st);         \
        return launch_bf16<1, 1, AK, BKM, GA, GB, 1>(A, B, C, ldc, M, N, K, sk, ep, st);                         \
    } while (0)
    if (!a_kmajor && !b_kmajor) { if (ga) OE_DISP(false, false, true, false); else OE_DISP(false, false, false, false); }
    if (!a_kmajor && b_kmajor) OE_DISP(false, true, false, false);
    if (a_kmajor && b_kmajor) { if (gb) OE_DISP(true, true, false, true); else OE_DISP(true, true, false, false); }
#undef OE_DISP
    oe_set_error("oe_gemm: layout a_kmajor=1,b_kmajor=0 is not supported");
    return -1;
}
