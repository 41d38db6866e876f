// Weight-gradient GEMM on bf16 planes:  C[m][n] (+)= alpha * sum_k A[k][m] * B[k][n]  with both operands k-major
// (the rows of dY and of the layer input run along the reduction: ops.gemm_tn, i.e. every `dW = dY^T x` of backward;
// the reference gets these from autograd's Linear backward, /root/reference/openeat/modules/*.py via torch.nn.Linear).
//
// Why a kernel of its own.  The LDS-DMA ring kernel (gemm_dma.hip) keeps fp32 tiles in LDS: a fragment of a k-major tile
// is eight ds_read_b32 plus a bf16 split (hi, lo: ~30 vector instructions) per fragment per wave, redone by every wave
// that uses the fragment, and a block is four waves = one per SIMD, so nothing overlaps a wave's LDS / split / MFMA
// phases.  The FFN weight gradients (1024 x 256 outputs over K = 7936) ran at 36 us = 112 TFLOP/s, 13 % of the bf16x3
// MFMA peak.  Here (the layout attention_bf16.hip measured at 2x on the same product shape):
//   * a block is 8 waves = two per SIMD on one 128 x 128 output tile: waves 0-3 and 4-7 are two K-GROUPS, each a 2 x 2
//     arrangement of 64 x 64 sub-tiles; group g takes k rows 16g .. 16g+15 of every 32-row chunk (an in-block split of
//     the reduction: twice the waves without twice the output traffic); the groups exchange half of their accumulators
//     through LDS at the end and each finishes half of the tiles;
//   * a chunk (32 k-rows x 128 columns of A and of B) goes global fp32 -> registers -> split ONCE -> bf16 hi / lo planes
//     in a double-buffered LDS image, one barrier per chunk, the next chunk's loads issued before this chunk's MFMAs;
//   * a fragment is two ds_read_b64_tr_b16 per plane (gfx950's transposing LDS read) - the k-slot order it produces is
//     the same for A and B, and a sum over k does not care about the order;
//   * the bias gradient (column sums of A = dY) rides on the staging registers.
// Image: [32 k-rows][128 + 16] bf16 per plane: the 288-byte pitch puts the four rows of a transposing read's 16-lane
// group on disjoint banks.  LDS: 2 buffers x 2 operands x (1 or 2) planes x 9216 B = 36 / 72 KiB.
// Split-K blocks accumulate with float atomics (the arena is zeroed at the start of the step), as before.
#include <stdlib.h>
#include "gemm_common.h"
#include "../../include/openeat_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define TN_KROWS 32
#define TN_COLS 128
#define TN_PITCH (TN_COLS + 16)
#define TN_PLANE (TN_KROWS * TN_PITCH)            // bf16 elements per plane
#define TN_THREADS 512

// Diagnostic build only (-DOE_GEMM_STAMPS, tools/tn_stamps.py): s_memtime sums per phase, wave 0 of every block.
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* tn_stamp_buf = nullptr;
extern "C" int oe_debug_set_tn_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tn_stamp_buf), &p, sizeof(p)); }
#define TN_NOW(var)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define TN_ACC(slot) do { unsigned long long t1_; TN_NOW(t1_); tn_acc[slot] += t1_ - tn_t; tn_t = t1_; } while (0)
#else
#define TN_NOW(var) do { } while (0)
#define TN_ACC(slot) do { } while (0)
#endif

template <int TERMS> struct TFrag { bf16x8 p[oe_npl<TERMS>::N]; };

// fragment of 32 columns (col32 ..) x 16 k-rows (16 s ..) of one operand image: lane l of a 16-lane group supplies the
// address of row (l >> 2), columns 4 (l & 3) .. of the group's 4-row x 16-column block and receives column (l & 15) of
// the four rows; groups 0/1 take k rows 0-3 (+8), groups 2/3 rows 4-7 (+8): lane -> (column lane & 31 of the tile ...
// the same map for both operands, which is all a reduction over k needs.
template <int TERMS>
__device__ __forceinline__ void tn_frag(const __bf16* img, int col32, int s, int lane, TFrag<TERMS>& f) {
    const int i = lane & 15, grp = lane >> 4;
    const __bf16* p = img + (16 * s + 4 * (grp >> 1) + (i >> 2)) * TN_PITCH + col32 + 16 * (grp & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int n = 0; n < oe_npl<TERMS>::N; ++n) {
        union { s16x4 h[2]; bf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + n * TN_PLANE));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + n * TN_PLANE + 8 * TN_PITCH));
        f.p[n] = u.v;
    }
}

template <int TERMS>
__device__ __forceinline__ f32x16 tn_mma(const TFrag<TERMS>& a, const TFrag<TERMS>& b, f32x16 c) {
    return oe_mma_terms<TERMS>(a, b, c);
}

// this thread's share of a chunk of one operand: rows (tid >> 5) and (tid >> 5) + 16, columns 4 (tid & 31) ..
struct TnRegs { float4 v[2]; };
__device__ __forceinline__ void tn_load(TnRegs& t, const float* base, long ld, int k0, int k_last) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = min(k0 + (int)(threadIdx.x >> 5) + 16 * i, k_last);          // rows past the range re-read the last one
        t.v[i] = *reinterpret_cast<const float4*>(base + (long)k * ld);
    }
}
template <int TERMS>
__device__ __forceinline__ void tn_store(const TnRegs& t, __bf16* img, int k0, int k_end) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (int)(threadIdx.x >> 5) + 16 * i;
        const bool live = k0 + row < k_end;                                          // ... and are zeroed here
        const float x[4] = {live ? t.v[i].x : 0.f, live ? t.v[i].y : 0.f, live ? t.v[i].z : 0.f, live ? t.v[i].w : 0.f};
        constexpr int NPL = oe_npl<TERMS>::N;
        bf16x4 pl[NPL];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 q[NPL];
            oe_split_bf16<NPL>(x[e], q);
#pragma unroll
            for (int n = 0; n < NPL; ++n) pl[n][e] = q[n];
        }
        __bf16* d = img + row * TN_PITCH + 4 * (threadIdx.x & 31);
#pragma unroll
        for (int n = 0; n < NPL; ++n) *reinterpret_cast<bf16x4*>(d + n * TN_PLANE) = pl[n];
    }
}

// One block's work: output tile (tile_x, tile_y) of one problem over the K range of split tile_z.
template <int TERMS>
__device__ __forceinline__ void tn_block(const float* __restrict__ Ap, long lda, const float* __restrict__ Bp, long ldb,
                                         float* __restrict__ C, long ldc, int M, int N, int Mr, int Nr, int K, int k_chunk,
                                         int tile_x, int tile_y, int tile_z, const EpiParams& ep) {
    constexpr int NPL = oe_npl<TERMS>::N;
    constexpr int OP_ELEMS = NPL * TN_PLANE;                // one operand's image
    constexpr int BUF_ELEMS = 2 * OP_ELEMS;                 // A then B
    constexpr int XCH_BYTES = 4 * 2 * 32 * 64 * 4;          // the accumulator exchange at the end overlays the images
    constexpr int IMG_ELEMS = (2 * BUF_ELEMS * 2 >= XCH_BYTES) ? 2 * BUF_ELEMS : XCH_BYTES / 2;
    __shared__ __attribute__((aligned(16))) __bf16 img[IMG_ELEMS];
    __shared__ float csum_s[TN_COLS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const long m0 = (long)tile_y * TN_COLS, n0 = (long)tile_x * TN_COLS;
    const int k_begin = tile_z * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nchunks = (k_end - k_begin + TN_KROWS - 1) / TN_KROWS;

    // this thread's column of the staging loads (clamped at the matrix edge: whole float4s;
    // what the clamp duplicates lands in accumulator columns the bounds-checked epilogue never stores)
    const int c4 = 4 * (threadIdx.x & 31);
    const float* a_src = Ap + min(m0 + c4, (long)Mr - 4);          // Mr, Nr: M, N rounded up to whole float4s (<= lda, ldb)
    const float* b_src = Bp + min(n0 + c4, (long)Nr - 4);
    const bool do_csum = ep.a_colsum != nullptr && tile_x == 0;          // block-uniform
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef OE_GEMM_STAMPS
    unsigned long long tn_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tn_t, tn_t0;
    TN_NOW(tn_t);
    tn_t0 = tn_t;
#endif
    TnRegs ra, rb;
    if (nchunks > 0) {
        tn_load(ra, a_src, lda, k_begin, k_end - 1);
        tn_load(rb, b_src, ldb, k_begin, k_end - 1);
        if (do_csum) {
            // rows past k_end are duplicates: count the live ones only
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (k_begin + (int)(threadIdx.x >> 5) + 16 * i < k_end) { cs[0] += ra.v[i].x; cs[1] += ra.v[i].y; cs[2] += ra.v[i].z; cs[3] += ra.v[i].w; }
        }
        tn_store<TERMS>(ra, img, k_begin, k_end);
        tn_store<TERMS>(rb, img + OP_ELEMS, k_begin, k_end);
    }
    __syncthreads();
    TN_ACC(0);                                   // first chunk: load + split + store + barrier
    for (int c = 0; c < nchunks; ++c) {
        const int kn = k_begin + (c + 1) * TN_KROWS;
        const bool more = c + 1 < nchunks;                                            // block-uniform
        if (more) {
            tn_load(ra, a_src, lda, kn, k_end - 1);
            tn_load(rb, b_src, ldb, kn, k_end - 1);
        }
        TN_ACC(1);                               // load issue
        const __bf16* ia = img + (c & 1) * BUF_ELEMS;
        const __bf16* ib = ia + OP_ELEMS;
        TFrag<TERMS> fa[2], fb[2];
        tn_frag<TERMS>(ia, wm * 64, grp, lane, fa[0]);
        tn_frag<TERMS>(ia, wm * 64 + 32, grp, lane, fa[1]);
        tn_frag<TERMS>(ib, wn * 64, grp, lane, fb[0]);
        tn_frag<TERMS>(ib, wn * 64 + 32, grp, lane, fb[1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = tn_mma<TERMS>(fa[i], fb[j], acc[i][j]);
        TN_ACC(2);                               // fragment reads + MFMA issue
#ifdef OE_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TN_ACC(3);                               // what is left of the loads' latency
#endif
        if (more) {
            if (do_csum) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (kn + (int)(threadIdx.x >> 5) + 16 * i < k_end) { cs[0] += ra.v[i].x; cs[1] += ra.v[i].y; cs[2] += ra.v[i].z; cs[3] += ra.v[i].w; }
            }
            __bf16* nx = img + ((c + 1) & 1) * BUF_ELEMS;
            tn_store<TERMS>(ra, nx, kn, k_end);
            tn_store<TERMS>(rb, nx + OP_ELEMS, kn, k_end);
        }
        TN_ACC(4);                               // split + LDS store
        __syncthreads();
        TN_ACC(5);                               // barrier (incl. this wave's MFMA drain)
    }

    float alpha = ep.alpha;
    if (ep.alpha_dev) alpha *= *ep.alpha_dev;

    // ---- bias gradient: the 16 threads holding a column (lanes l and l + 32 of every wave) -> LDS -> one atomic per column
    if (do_csum) {
        if (threadIdx.x < TN_COLS) csum_s[threadIdx.x] = 0.f;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = cs[e] + __shfl_xor(cs[e], 32, 64);
            if (lane < 32) atomicAdd(&csum_s[4 * lane + e], v);
        }
        __syncthreads();
        if (threadIdx.x < TN_COLS && m0 + threadIdx.x < M) atomicAdd(ep.a_colsum + m0 + threadIdx.x, csum_s[threadIdx.x] * alpha);
    }

    // ---- the two K-groups swap halves of their accumulators: group g keeps and finishes row tiles i = g of every wave's
    // 2 x 2 sub-tile.  Exchange area: [wave pair q][direction g][32 registers][64 lanes] floats over the images (the
    // loop's last barrier retired all image reads).
    float* xch = reinterpret_cast<float*>(img);
    const int q = wave & 3;
    {
        float* mine = xch + ((q * 2 + grp) * 32) * 64 + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = grp == 0 ? acc[1][j][r] : acc[0][j][r];
    }
    __syncthreads();
    f32x16 fin[2];
    {
        const float* theirs = xch + ((q * 2 + (1 - grp)) * 32) * 64 + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[j][r] = (grp == 0 ? acc[0][j][r] : acc[1][j][r]) + theirs[(j * 16 + r) * 64];
    }
    const int lrow = lane & 31, lk = lane >> 5;
    const bool interior = (m0 + TN_COLS <= M) && (n0 + TN_COLS <= N);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long col = n0 + wn * 64 + j * 32 + lrow;
        const long row0 = m0 + wm * 64 + grp * 32 + 4 * lk;
        float* base = C + row0 * ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (!interior && !(row0 + dr < M && col < N)) continue;
            const float v = fin[j][r] * alpha;
            if (ep.atomic) atomicAdd(base + dr * ldc, v);
            else if (ep.accumulate) base[dr * ldc] += v;
            else base[dr * ldc] = v;
        }
    }
#ifdef OE_GEMM_STAMPS
    TN_ACC(6);                                   // bias gradient + exchange + output
    if (tn_stamp_buf && threadIdx.x == 0 && blockIdx.x < 2048) {
        for (int i = 0; i < 7; ++i) tn_stamp_buf[blockIdx.x * 8 + i] = tn_acc[i];
        tn_stamp_buf[blockIdx.x * 8 + 7] = tn_t - tn_t0;
    }
#endif
}

// XCD-aware tile order (see gemm_bf16.hip): block id -> position in a grid of nblk blocks
__device__ __forceinline__ int tn_swizzle(int id, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, j = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int TERMS>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_planes_kernel(const float* __restrict__ Ap, long lda, const float* __restrict__ Bp, long ldb,
                                                                   float* __restrict__ C, long ldc, int M, int N, int Mr, int Nr, int K, int k_chunk,
                                                                   int gx, int gy, EpiParams ep) {
    const int swz = tn_swizzle(blockIdx.x, gridDim.x);
    tn_block<TERMS>(Ap, lda, Bp, ldb, C, ldc, M, N, Mr, Nr, K, k_chunk, swz % gx, (swz / gx) % gy, swz / (gx * gy), ep);
}

// Several weight gradients in ONE launch (oe_gemm_tn_grouped): the deferred weight gradients of a captured step are small
// (256..768 x 256 outputs over K = 7936: 2-6 tiles each), so alone each needs a 20-60-way split of the reduction to cover the
// chip and its atomic epilogue dominates; four of them together are one 28-tile problem at a 9-way split (41 us against 81).
// The table lives in device memory; a block finds its problem by its position in the running block count.
template <int TERMS>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_grouped_kernel(const oe_tn_problem* __restrict__ tab, int n_problems) {
    const int swz = tn_swizzle(blockIdx.x, gridDim.x);
    int p = 0;
    while (p + 1 < n_problems && tab[p + 1].block_start <= swz) ++p;         // block-uniform (scalar loads)
    const oe_tn_problem pr = tab[p];
    const int local = swz - pr.block_start;
    EpiParams ep{};
    ep.alpha = pr.alpha; ep.alpha_dev = pr.alpha_dev; ep.beta = 1.f; ep.atomic = 1; ep.a_colsum = pr.a_colsum;
    tn_block<TERMS>(pr.a, pr.lda, pr.b, pr.ldb, pr.c, pr.ldc, pr.m, pr.n, (pr.m + 3) / 4 * 4, (pr.n + 3) / 4 * 4, pr.k, pr.k_chunk,
                    local % pr.gx, (local / pr.gx) % pr.gy, local / (pr.gx * pr.gy), ep);
}

// Host side of the grouped launch: fill the launch geometry of every problem (tiles, split of the reduction, first block) for
// about `target_blocks` blocks in all.  Returns the total number of blocks, or -1 if a problem cannot take this kernel
// (operands must be 16-byte aligned with leading dimensions of whole float4s that cover the rounded-up row lengths).
extern "C" int oe_gemm_tn_grouped_plan(oe_tn_problem* problems, int n, int target_blocks) {
    OE_REQUIRE(problems && n > 0 && target_blocks > 0, "oe_gemm_tn_grouped_plan: bad arguments");
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        oe_tn_problem& p = problems[i];
        OE_REQUIRE(p.a && p.b && p.c && p.m >= 4 && p.n >= 4 && p.k >= 1, "oe_gemm_tn_grouped_plan: problem %d: bad shape / null pointer", i);
        OE_REQUIRE((((uintptr_t)p.a | (uintptr_t)p.b) & 15) == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 && p.lda >= (p.m + 3) / 4 * 4 &&
                       p.ldb >= (p.n + 3) / 4 * 4 && p.ldc >= p.n,
                   "oe_gemm_tn_grouped_plan: problem %d: operands not 16-byte aligned / leading dimensions too short", i);
        p.gx = oe_cdiv(p.n, TN_COLS);
        p.gy = oe_cdiv(p.m, TN_COLS);
        tiles += (long)p.gx * p.gy;
    }
    // never more than 16 ways: a 37-way split of a five-problem leftover group spent 264 us in its atomic epilogues
    const int split = (int)min(16L, max(1L, (target_blocks + tiles / 2) / tiles));
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        oe_tn_problem& p = problems[i];
        int nz = min(split, max(1, p.k / 128));
        p.k_chunk = oe_cdiv(oe_cdiv(p.k, nz), TN_KROWS) * TN_KROWS;
        p.nz = oe_cdiv(p.k, p.k_chunk);
        p.block_start = blocks;
        blocks += p.gx * p.gy * p.nz;
    }
    return blocks;
}

extern "C" int oe_gemm_tn_grouped(const oe_tn_problem* problems_dev, int n, int total_blocks, int precision, void* stream) {
    OE_REQUIRE(problems_dev && n > 0 && total_blocks > 0, "oe_gemm_tn_grouped: bad arguments");
    OE_REQUIRE(precision == 1 || precision == 3 || precision == 6, "oe_gemm_tn_grouped: precision must be 1 (bf16), 3 (bf16x3) or 6 (bf16x6)");
    if (precision == 6)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<6>), dim3(total_blocks), dim3(TN_THREADS), 0, (hipStream_t)stream, problems_dev, n);
    else if (precision == 3)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<3>), dim3(total_blocks), dim3(TN_THREADS), 0, (hipStream_t)stream, problems_dev, n);
    else
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<1>), dim3(total_blocks), dim3(TN_THREADS), 0, (hipStream_t)stream, problems_dev, n);
    OE_LAUNCH_CHECK("oe_gemm_tn_grouped");
    return 0;
}

// Returns 1 when the problem does not qualify (the caller goes on to the ring / register-staged kernels), 0 on a launch.
int oe_gemm_tn_planes_try(const OperandDesc& A, const OperandDesc& B, float* C, long ldc, int M, int N, int K, int sk, const EpiParams& ep,
                          int terms, hipStream_t st) {
    // OE_GEMM_TN_PLANES: 0 = never, 1 = where it measured faster (default), 2 = wherever the problem qualifies (tuning)
    static const int mode = getenv("OE_GEMM_TN_PLANES") ? atoi(getenv("OE_GEMM_TN_PLANES")) : 1;
    if (!mode) return 1;
    // staging loads are whole float4s: a row length that is not a multiple of 4 is fine when the leading dimension has
    // the room (the CTC head's vocabulary, 3246 columns in rows of 3248): the last float4 then reads the row's own padding,
    // and what it yields only reaches accumulator rows the bounds-checked output never stores
    const int Mr = (M + 3) / 4 * 4, Nr = (N + 3) / 4 * 4;
    if (!A.vec_ok || !B.vec_ok || A.ld < Mr || B.ld < Nr || M < 4 || N < 4 || K < 1) return 1;
    if (ep.bias || ep.act || ep.preact_out || ep.actgrad_in || ep.residual || ep.rowmask || ep.drop_p > 0.f || ep.beta != 1.f || ep.scatter) return 1;
    if (sk > 1 && !ep.atomic) return 1;
    const int gx = oe_cdiv(N, TN_COLS), gy = oe_cdiv(M, TN_COLS);
    // Measured on MI355X (tools/tn_bench.py, precision 3): 1024 x 256 over K = 7936 32.6 us against the ring kernel's 36.9,
    // 256 x 4864 102 against 113, K = 25472 65 against 75; outputs of fewer than ~12 tiles need so many splits to cover the
    // chip that the atomic epilogue (in-kernel stamps: 25 % of a 16-way split block, 66 % of a 64-way one) eats the gain.
    if (mode == 1 && (gx * gy < 12 || K < 2048)) return 1;
    // own split of the reduction: about one block per CU, chunks of at least 128 k-rows; never more splits than the
    // caller allowed when it asked for none (no atomics without permission)
    int nz = 1;
    static const int blocks_target = getenv("OE_TN_BLOCKS") ? atoi(getenv("OE_TN_BLOCKS")) : 256;      // tuning
    if (ep.atomic) {
        nz = max(1, blocks_target / (gx * gy));
        nz = min(nz, max(1, K / 128));
    }
    int kc = oe_cdiv(oe_cdiv(K, nz), TN_KROWS) * TN_KROWS;
    nz = oe_cdiv(K, kc);
    if (terms == 6)
        hipLaunchKernelGGL((gemm_tn_planes_kernel<6>), dim3(gx * gy * nz), dim3(TN_THREADS), 0, st, A.p, A.ld, B.p, B.ld, C, ldc, M, N, Mr, Nr, K, kc, gx, gy, ep);
    else if (terms == 3)
        hipLaunchKernelGGL((gemm_tn_planes_kernel<3>), dim3(gx * gy * nz), dim3(TN_THREADS), 0, st, A.p, A.ld, B.p, B.ld, C, ldc, M, N, Mr, Nr, K, kc, gx, gy, ep);
    else
        hipLaunchKernelGGL((gemm_tn_planes_kernel<1>), dim3(gx * gy * nz), dim3(TN_THREADS), 0, st, A.p, A.ld, B.p, B.ld, C, ldc, M, N, Mr, Nr, K, kc, gx, gy, ep);
    OE_LAUNCH_CHECK("oe_gemm (bf16 mfma, k-major planes)");
    return 0;
}
