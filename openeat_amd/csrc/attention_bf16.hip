// Fused multi-head attention on the bf16 matrix cores with the streamed operand kept in LDS as bf16 PLANES
// (attention.py:65-97,112-117,189-209 and their autograd; precision 1 = hi plane only, precision 3 = hi + lo planes and
// the three-term product hi*hi + hi*lo + lo*hi, precision 6 = three planes and the six-term product of oe_common.h for
// all three kernels - as oe_gemm_args.precision; the dK/dV kernel keeps four resident and four streamed images, which at
// three planes each is more than a CU's LDS: there precision 6 single-buffers the streamed pair and packs the resident
// images (ResImage) - 151 KiB; measured 77 against 121 us per layer's backward with the exact-fp32 dK/dV kernel at B = 32,
// T = 248, 406 / 614 us at B = 64, T = 398).
//
// What changed against the first generation of bf16 kernels (attention.hip, still used for small problems and for the
// exact-fp32 mode): those kept fp32 tiles in LDS and split every fragment to bf16 on EVERY use (each of the waves that
// consumed a tile redid the conversion), staged 32-key tiles behind two barriers each, and re-staged the whole key range
// once per 64 queries.  Here
//   * a block is 8 waves, two per SIMD: four 32-row tiles of the resident axis (queries for forward / dQ, keys for
//     dK/dV) x two halves of the streamed axis - wave w owns resident tile w & 3 and the streamed tiles of parity w >> 2;
//     the halves are merged through LDS at the end (online-softmax merge of (m, l, O) for the forward, plain sums for the
//     gradients).  One wave alone issues a vector instruction every 4 cycles and nothing overlaps its MFMA / softmax /
//     LDS phases; two waves per SIMD fill each other's gaps (in-kernel stamps: a 4-wave block spent its time in one
//     serial chain per tile).  The streamed tensors are staged once per 128 resident rows;
//   * the streamed axis moves in chunks of 64 rows (one tile per wave): global fp32 -> registers (issued before the
//     chunk's MFMAs, every load unconditional) -> bf16 hi / lo written ONCE into a double-buffered LDS image, one
//     barrier per chunk;
//   * row fragments (products over features) are one ds_read_b128 per plane; column fragments (products over the 32
//     keys / queries of a tile, whose k-slots follow the accumulator row map so that the score tile never leaves
//     registers) are two ds_read_b64_tr_b16 per plane on the SAME image - gfx950's transposing LDS read - instead of
//     eight scalar reads and a conversion;
//   * outputs leave through a wave-private LDS patch as whole rows (256 contiguous bytes per store instruction)
//     instead of one 4-byte store per lane and row.
// Image: [64 rows][DPAD + 8] bf16, i.e. a 144-byte pitch at DPAD = 64: conflict-free for the b128 row reads (16 rows
// land on 16 different 16-byte slots of the 256-byte bank row), 2-way on the transposed reads (they are a few per cent
// of the LDS traffic).  LDS per block at DPAD = 64, precision 3: 2 buffers x 2 tensors x 2 planes x 9216 B = 72 KiB.
//
// Index conventions, dropout mask definition, lse / delta exchange between the kernels: attn_common.h / attention.hip.
#include <stdlib.h>
#include "attn_common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

#define PL_ROWS 64                     // rows of the streamed axis per chunk = one 32-row tile per wave half
#define PL_THREADS 512                 // 8 waves: resident tile = wave & 3, streamed-tile parity = wave >> 2

// Diagnostic build only (-DOE_GEMM_STAMPS, tools/attn_stamps.py): s_memtime sums per phase of the forward kernel, written
// to a buffer nothing else reads.  No stamp exists in the shipped library.
#ifdef OE_GEMM_STAMPS
static __device__ unsigned long long* oe_planes_stamp_buf = nullptr;
extern "C" int oe_debug_set_attn_planes_stamp_buffer(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(oe_planes_stamp_buf), &p, sizeof(p));
}
#define PL_NOW(var)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define PL_ACC(slot) do { unsigned long long t1_; PL_NOW(t1_); pl_acc[slot] += t1_ - pl_t; pl_t = t1_; } while (0)
#else
#define PL_NOW(var) do { } while (0)
#define PL_ACC(slot) do { } while (0)
#endif

template <int TERMS> struct PFrag { bf16x8 p[oe_npl<TERMS>::N]; };

template <int TERMS>
__device__ __forceinline__ f32x16 pmma(const PFrag<TERMS>& a, const PFrag<TERMS>& b, f32x16 c) {
    return oe_mma_terms<TERMS>(a, b, c);
}
template <int TERMS>
__device__ __forceinline__ void psplit(const float (&x)[8], PFrag<TERMS>& f) {
    oe_split8<oe_npl<TERMS>::N>(x, f.p);
}

// One streamed tensor's image in LDS: `planes` consecutive [PL_ROWS][PITCH] bf16 arrays (hi, then lo).
template <int DPAD, int TERMS>
struct Plane {
    static constexpr int PITCH = DPAD + 8;
    static constexpr int PLANE_ELEMS = PL_ROWS * PITCH;
    static constexpr int NPL = oe_npl<TERMS>::N;
    static constexpr int ELEMS = PLANE_ELEMS * NPL;
    // row fragment: A[row][16 s + 8 g + e], e = 0..7
    static __device__ __forceinline__ void row_frag(const __bf16* img, int row, int s, int g, PFrag<TERMS>& f) {
        const __bf16* p = img + row * PITCH + 16 * s + 8 * g;
#pragma unroll
        for (int n = 0; n < NPL; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(p + n * PLANE_ELEMS);
    }
    // column fragment: A[row0 + acc_row(8 s + e, g)][col], e = 0..7 - two transposing reads of 4 rows x 16 columns per
    // 16-lane group: lane 4q + pp of a group supplies the address of row q, columns 4pp..4pp+3 of the group's block and
    // receives column (lane & 15) of the four rows.  Groups 0/1 (g = 0) and 2/3 (g = 1) take their own row blocks.
    static __device__ __forceinline__ void col_frag(const __bf16* img, int row0, int col32, int s, int lane, PFrag<TERMS>& f) {
        const int i = lane & 15, grp = lane >> 4;
        const int g = grp >> 1;
        const __bf16* p = img + (row0 + 16 * s + 4 * g + (i >> 2)) * PITCH + col32 + 16 * (grp & 1) + 4 * (i & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            union { s16x4 h[2]; bf16x8 v; } u;
            u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + n * PLANE_ELEMS));
            u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + n * PLANE_ELEMS + 8 * PITCH));
            f.p[n] = u.v;
        }
    }
};

// A chunk of PL_ROWS rows x DPAD features of an fp32 tensor on its way global -> registers -> LDS planes.
template <int DPAD>
struct ChunkRegs {
    static constexpr int N = PL_ROWS * (DPAD / 4) / PL_THREADS;       // float4 per thread: 2 (DPAD 64) or 1 (DPAD 32)
    float4 v[N];
};
// Fast path (vec: rows 16-byte aligned, D % 4 == 0): every load is UNCONDITIONAL - rows past the end re-read the last valid
// row, columns past D re-read the last valid float4, and chunk_store zeroes them by the same predicate.  (A load inside a
// divergent branch makes hipcc wait vmcnt(0) at the branch's join: eight loads then cost eight memory round trips.)
template <int DPAD>
__device__ __forceinline__ void chunk_load(ChunkRegs<DPAD>& t, const float* src, long rs, int r0, int nrows_total, int D, bool vec) {
    if (vec) {
#pragma unroll
        for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
            const int e = threadIdx.x + i * PL_THREADS;
            const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
            const int gr = min(r0 + row, nrows_total - 1), gc = min(c4, D - 4);
            t.v[i] = *reinterpret_cast<const float4*>(src + (long)gr * rs + gc);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
        const int e = threadIdx.x + i * PL_THREADS;
        const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gr = r0 + row;
        if (gr < nrows_total && c4 < D) {
            const float* p = src + (long)gr * rs + c4;
            val.x = p[0]; if (c4 + 1 < D) val.y = p[1]; if (c4 + 2 < D) val.z = p[2]; if (c4 + 3 < D) val.w = p[3];
        }
        t.v[i] = val;
    }
}
template <int DPAD, int TERMS>
__device__ __forceinline__ void chunk_store(const ChunkRegs<DPAD>& t, __bf16* img, int r0, int nrows_total, int D, float mul = 1.f) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    using P = Plane<DPAD, TERMS>;
#pragma unroll
    for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
        const int e = threadIdx.x + i * PL_THREADS;
        const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
        const bool live = (r0 + row < nrows_total) && (c4 < D);          // zero what chunk_load's fast path over-read
        const float x[4] = {live ? t.v[i].x * mul : 0.f, live ? t.v[i].y * mul : 0.f, live ? t.v[i].z * mul : 0.f, live ? t.v[i].w * mul : 0.f};
        bf16x4 pl[P::NPL];
        oe_split4<P::NPL>(x, pl);
        __bf16* d = img + row * P::PITCH + c4;
#pragma unroll
        for (int n = 0; n < P::NPL; ++n) *reinterpret_cast<bf16x4*>(d + n * P::PLANE_ELEMS) = pl[n];
    }
}

// The dK/dV kernel's RESIDENT images (the block's own 128 keys: K' * scale and V, read as row fragments only).  At three
// planes and DPAD = 64 the padded pitch does not fit beside the streamed images, so that case packs rows to 128 bytes and
// permutes the 16-byte chunks of a row instead: chunk c of row r sits at c ^ ((r >> 1) & 7) - the 16 lanes of a b128 read
// group (rows {0-3, 12-15, 20-27} + 32 k of one chunk index) then cover the 16 slots of the 256-byte bank row exactly once.
template <int DPAD, int TERMS>
struct ResImage {
    static constexpr bool SWZ = (TERMS == 6 && DPAD == 64);
    static constexpr int PITCH = SWZ ? DPAD : DPAD + 8;
    static constexpr int NPL = oe_npl<TERMS>::N;
    static constexpr int PLANE_ELEMS = PL_ROWS * PITCH;
    static constexpr int ELEMS = PLANE_ELEMS * NPL;
    static __device__ __forceinline__ int chunk_pos(int row, int c) { return SWZ ? (c ^ ((row >> 1) & 7)) : c; }
    static __device__ __forceinline__ void row_frag(const __bf16* img, int row, int s, int g, PFrag<TERMS>& f) {
        const __bf16* p = img + row * PITCH + 8 * chunk_pos(row, 2 * s + g);
#pragma unroll
        for (int n = 0; n < NPL; ++n) f.p[n] = *reinterpret_cast<const bf16x8*>(p + n * PLANE_ELEMS);
    }
    static __device__ __forceinline__ void store(const ChunkRegs<DPAD>& t, __bf16* img, int r0, int nrows_total, int D, float mul) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < ChunkRegs<DPAD>::N; ++i) {
            const int e = threadIdx.x + i * PL_THREADS;
            const int row = e / (DPAD / 4), c4 = (e % (DPAD / 4)) * 4;
            const bool live = (r0 + row < nrows_total) && (c4 < D);
            const float x[4] = {live ? t.v[i].x * mul : 0.f, live ? t.v[i].y * mul : 0.f, live ? t.v[i].z * mul : 0.f, live ? t.v[i].w * mul : 0.f};
            bf16x4 pl[NPL];
            oe_split4<NPL>(x, pl);
            __bf16* d = img + row * PITCH + 8 * chunk_pos(row, c4 >> 3) + (c4 & 4);
#pragma unroll
            for (int n = 0; n < NPL; ++n) *reinterpret_cast<bf16x4*>(d + n * PLANE_ELEMS) = pl[n];
        }
    }
};

// eight consecutive features of one row -> registers (two float4 when aligned); zeros past D or when !ok
__device__ __forceinline__ void load8f(const float* row, int d0, int D, bool vec, bool ok, float (&out)[8]) {
    if (vec) {          // block-uniform; `row` is a valid row also when !ok: the loads are unconditional, the zeroing is a select
        const int dc = min(d0, D - 8);
        const float4 a = *reinterpret_cast<const float4*>(row + dc), c = *reinterpret_cast<const float4*>(row + dc + 4);
        const bool live = ok && d0 + 8 <= D;
        out[0] = live ? a.x : 0.f; out[1] = live ? a.y : 0.f; out[2] = live ? a.z : 0.f; out[3] = live ? a.w : 0.f;
        out[4] = live ? c.x : 0.f; out[5] = live ? c.y : 0.f; out[6] = live ? c.z : 0.f; out[7] = live ? c.w : 0.f;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = (ok && d0 + e < D) ? row[d0 + e] : 0.f;
    }
}

// An accumulator set acc[DT] holding X^T (rows = features t*32 + acc_row(r, lk), column = this lane's row lq of the
// resident axis) leaves as rows of X: through a wave-private [32][DPAD + 1] fp32 patch, then one 4-byte store per lane
// with the 64 lanes on consecutive features of one row (256 contiguous bytes per instruction at DPAD 64).
template <int DPAD>
__device__ __forceinline__ void store_rows(float* patch, const f32x16 (&acc)[DPAD / 32], float mul, float* dst, long rs, int row0,
                                           int nrows_total, int D, int lane) {
    constexpr int PP = DPAD + 1;
    const int lq = lane & 31, lk = lane >> 5;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < DPAD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[lq * PP + t * 32 + acc_row(r, lk)] = acc[t][r] * mul;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (DPAD == 64) {
        for (int rr = 0; rr < 32; ++rr) {
            const int gr = row0 + rr;
            if (gr < nrows_total && lane < D) dst[(long)gr * rs + lane] = patch[rr * PP + lane];
        }
    } else {
        for (int rr = 0; rr < 32; rr += 2) {
            const int gr = row0 + rr + lk;
            if (gr < nrows_total && lq < D) dst[(long)gr * rs + lq] = patch[(rr + lk) * PP + lq];
        }
    }
}

// ------------------------------------------------------------------ forward (MODE 0) and dQ (MODE 1) --
// wave w of block x owns queries (4x + (w & 3)) * 32 .. + 31 of (b, h) and, of every 64-key chunk, key tile (w >> 2).
template <int DPAD, int TERMS, int MODE>
__global__ __launch_bounds__(PL_THREADS, 2) void attn_planes_q_kernel(AttnParams p) {
    using P = Plane<DPAD, TERMS>;
    constexpr int DT = DPAD / 32, KS = DPAD / 16;
    constexpr int IMG = P::ELEMS;                                   // bf16 elements of one tensor's image
    constexpr int MERGE_FLOATS = DPAD * 32 + 64;                    // per resident tile: O^T / dQ^T [DPAD][32] + m[32] + l[32]
    constexpr int PATCH_FLOATS = 32 * (DPAD + 1);
    constexpr int TAIL_BYTES = 4 * (MERGE_FLOATS + PATCH_FLOATS) * 4;
    constexpr int IMG_BYTES = 2 * 2 * IMG * 2;                      // [buffer][K | V]
    __shared__ __attribute__((aligned(16))) char lds_raw[(IMG_BYTES > TAIL_BYTES ? IMG_BYTES : TAIL_BYTES)];
    __shared__ __attribute__((aligned(16))) float kb_s[2][PL_ROWS]; // per-key bias of the chunk; -inf = key masked for every query
    __bf16* imgs = reinterpret_cast<__bf16*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave & 3, half = wave >> 2;
    const int lq = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + tile) * 32;
    const bool wave_live = q0 < p.T1;                               // wave-uniform
    const int qi = q0 + lq;
    const bool q_ok = qi < p.T1;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* kbp = p.k + (long)b * p.k_bs + h * p.D;
    const float* vbp = p.v + (long)b * p.v_bs + h * p.D;
    const long bh = (long)b * p.H + h;
    const bool vk = (p.D % 4 == 0) && (p.k_rs % 4 == 0) && ((((uintptr_t)kbp) & 15) == 0);
    const bool vv = (p.D % 4 == 0) && (p.v_rs % 4 == 0) && ((((uintptr_t)vbp) & 15) == 0);

    // Q^T fragments (B operand of S^T = K Q^T), pre-scaled; dQ also dO^T fragments, delta and lse of this lane's query
    PFrag<TERMS> qfr[KS], dofr[MODE == 1 ? KS : 1];
    float dpart = 0.f;
    const long orow = (long)b * p.o_bs + (long)(q_ok ? qi : 0) * p.o_rs + h * p.D;
    {
        const bool vq = (p.D % 8 == 0) && (p.q_rs % 4 == 0) && ((((uintptr_t)qb) & 15) == 0);
        const bool vo = (p.D % 8 == 0) && (p.o_rs % 4 == 0) && (p.o_bs % 4 == 0) &&
                        (MODE != 1 || (((((uintptr_t)p.d_o) | ((uintptr_t)p.o_in)) & 15) == 0));
        const float* qrow = qb + (long)(q_ok ? qi : 0) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d0 = 16 * s + 8 * lk;
            float x[8];
            load8f(qrow, d0, p.D, vq, q_ok, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= p.scale;
            psplit<TERMS>(x, qfr[s]);
            if (MODE == 1) {
                float y[8], ov[8];
                load8f(p.d_o + orow, d0, p.D, vo, q_ok, y);
                load8f(p.o_in + orow, d0, p.D, vo, q_ok, ov);
#pragma unroll
                for (int e = 0; e < 8; ++e) dpart += y[e] * ov[e];
                psplit<TERMS>(y, dofr[s]);
            }
        }
    }
    float m_run = NEG_INF, l_run = 0.f, lse_i = 0.f, delta_i = 0.f;
    if (MODE == 1) {
        delta_i = xhalf_sum(dpart);
        if (q_ok) {
            lse_i = p.lse[bh * p.T1 + qi];
            if (lk == 0 && half == 0) p.delta[bh * p.T1 + qi] = delta_i;
        }
    }
    f32x16 oacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    const DropParams dpar = drop_params(p.drop_p);
    const unsigned long long drop_row = (unsigned long long)(bh * p.T1 + (q_ok ? qi : 0));
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);
    const bool key_mask = p.mask && p.m_rs == 0;
    const unsigned char* mrow = (p.mask && !key_mask) ? p.mask + (long)b * p.m_bs + (long)(q_ok ? qi : 0) * p.m_rs : nullptr;

#ifdef OE_GEMM_STAMPS
    unsigned long long pl_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pl_t = 0, pl_t0 = 0;
    PL_NOW(pl_t0);
    pl_t = pl_t0;
#endif
    ChunkRegs<DPAD> kreg, vreg;
    float kb_raw = 0.f;
    unsigned char km_raw = 1;
    // wave 0 also fetches the chunk's per-key bias and key-mask bytes: unconditional loads at clamped indices, combined in commit
    const bool kb_wave = wave == 0;                                 // threadIdx.x < PL_ROWS
    auto prefetch = [&](int j0) {
        chunk_load<DPAD>(kreg, kbp, p.k_rs, j0, p.T2, p.D, vk);
        chunk_load<DPAD>(vreg, vbp, p.v_rs, j0, p.T2, p.D, vv);
        if (kb_wave) {
            const int kj = min(j0 + lane, p.T2 - 1);
            if (p.keybias) kb_raw = p.keybias[bh * p.T2 + kj];
            if (key_mask) km_raw = p.mask[(long)b * p.m_bs + kj];
        }
    };
    auto commit = [&](int buf, int j0) {
        chunk_store<DPAD, TERMS>(kreg, imgs + (buf * 2 + 0) * IMG, j0, p.T2, p.D);
        chunk_store<DPAD, TERMS>(vreg, imgs + (buf * 2 + 1) * IMG, j0, p.T2, p.D);
        if (kb_wave) kb_s[buf][lane] = (j0 + lane < p.T2 && km_raw != 0) ? kb_raw : NEG_INF;
    };
    // causal hint (forward only): no query of this block looks past key (blockIdx.x + 1) * 128 - 1, the mask zeroes the rest
    // anyway - the chunks that hold nothing but such keys are not visited (identical sums: their terms are exp(-inf) = 0)
    const int T2e = (MODE == 0 && p.causal) ? min(p.T2, (int)(blockIdx.x + 1) * 128) : p.T2;
    const int nchunks = (T2e + PL_ROWS - 1) / PL_ROWS;
    prefetch(0);
    PL_ACC(0);                                          // Q fragments + first prefetch issue
    commit(0, 0);
    __syncthreads();
    PL_ACC(1);                                          // first chunk lands + written + barrier
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) prefetch((c + 1) * PL_ROWS);
        PL_ACC(2);                                      // prefetch issue
        const __bf16* Ki = imgs + (buf * 2 + 0) * IMG;
        const __bf16* Vi = imgs + (buf * 2 + 1) * IMG;
        const int j0 = c * PL_ROWS + half * 32;         // this wave's key tile of the chunk
        if (wave_live && j0 < p.T2) {
            const int jt = half;
            // per-key bias of this lane's 16 accumulator rows (keys jt*32 + 8g + 4lk + 0..3): four b128 reads, issued ahead
            // of the S MFMAs; the full (B,T1,T2) mask bytes likewise (unconditional loads, block-uniform branch)
            float kbv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v4 = *reinterpret_cast<const float4*>(&kb_s[buf][jt * 32 + 8 * g4 + 4 * lk]);
                kbv[4 * g4] = v4.x; kbv[4 * g4 + 1] = v4.y; kbv[4 * g4 + 2] = v4.z; kbv[4 * g4 + 3] = v4.w;
            }
            unsigned char mbyte[16];
            if (mrow) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mbyte[r] = mrow[min(j0 + acc_row(r, lk), p.T2 - 1)];
            }
            // S^T[key, query]
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                PFrag<TERMS> a;
                P::row_frag(Ki, jt * 32 + lq, s, lk, a);
                sacc = pmma<TERMS>(a, qfr[s], sacc);
            }
            PL_ACC(3);                                  // K row fragments + S MFMAs (issue)
            float pr[16];
            float tmax = NEG_INF;
            if (mrow) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = (mbyte[r] == 0) ? NEG_INF : sacc[r] + kbv[r];      // kbv: -inf for keys past T2
                    pr[r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = sacc[r] + kbv[r];                     // -inf for keys past T2 / masked keys
                    pr[r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            }
            if (MODE == 0) {
                tmax = xhalf_max(tmax);
                const float m_new = fmaxf(m_run, tmax);
                const float corr = (m_new == NEG_INF) ? 1.f : __expf(m_run - m_new);
                // exp(s - m) = exp2(s * log2e - m * log2e): one fma + v_exp per element; a masked score (-inf) gives 0 by
                // itself as long as the subtrahend is finite (all keys masked so far: m_new = -inf, clamped)
                const float mL = fmaxf(m_new, -1e30f) * 1.4426950408889634f;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(pr[r], 1.4426950408889634f, -mL));
                    psum += e;
                    pr[r] = e;
                }
                psum = xhalf_sum(psum);
                l_run = l_run * corr + psum;
                m_run = m_new;
                if (!__all(corr == 1.f)) {              // the running maximum moved for some query of the wave
#pragma unroll
                    for (int t = 0; t < DT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[t][r] *= corr;
                }
                PL_ACC(4);                              // softmax (waits for the S MFMAs) + O rescale
                if (p.drop_p > 0.f) attn_drop_qlane_keep(seed_eff, drop_row, j0, lk, dpar, pr);   // unscaled: see the final multiply
                PL_ACC(5);                              // dropout
                // O^T[dv, query] += V^T[dv, key] P^T[key, query]
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                    PFrag<TERMS> pf;
                    psplit<TERMS>(x, pf);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        PFrag<TERMS> a;
                        P::col_frag(Vi, jt * 32, t * 32, s, lane, a);
                        oacc[t] = pmma<TERMS>(a, pf, oacc[t]);
                    }
                }
                PL_ACC(6);                              // P split + V column fragments + PV MFMAs (issue)
            } else {
                // P^T = exp(S^T - lse); dP^T[key, query] = V[key,:] . dO[query,:]
                f32x16 dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dpacc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    PFrag<TERMS> a;
                    P::row_frag(Vi, jt * 32 + lq, s, lk, a);
                    dpacc = pmma<TERMS>(a, dofr[s], dpacc);
                }
                float dmask[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dmask[r] = 1.f;
                if (p.drop_p > 0.f) attn_drop_qlane(seed_eff, drop_row, j0, lk, dpar, dmask);
                const float lseL = fmaxf(lse_i, -1e30f) * 1.4426950408889634f;      // finite: a masked score (-inf) then gives 0 by itself
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(pr[r], 1.4426950408889634f, -lseL));
                    pr[r] = pv * (dpacc[r] * dmask[r] - delta_i);     // dS^T
                }
                // dQ^T[d, query] += K^T[d, key] dS^T[key, query]
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = pr[8 * s + e];
                    PFrag<TERMS> df;
                    psplit<TERMS>(x, df);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        PFrag<TERMS> a;
                        P::col_frag(Ki, jt * 32, t * 32, s, lane, a);
                        oacc[t] = pmma<TERMS>(a, df, oacc[t]);
                    }
                }
            }
        }
        PL_ACC(7);                                      // (MFMA drain before the commit's VALU)
        if (c + 1 < nchunks) commit(buf ^ 1, (c + 1) * PL_ROWS);
        PL_ACC(8);                                      // next chunk: wait for the loads, convert, write
        __syncthreads();
        PL_ACC(9);                                      // barrier
    }
#ifdef OE_GEMM_STAMPS
    if (oe_planes_stamp_buf && MODE == 0 && lane == 0 && wave == 0 && blockIdx.x == 0 && blockIdx.y < 4 && blockIdx.z < 32) {
        unsigned long long* o = oe_planes_stamp_buf + (blockIdx.z * 4 + blockIdx.y) * 12;
        for (int i = 0; i < 10; ++i) o[i] = pl_acc[i];
        o[10] = pl_t - pl_t0;
    }
#endif
    // ---- merge the key halves (the images are dead: every wave is past the last barrier): half 1 parks (m, l, O^T) / dQ^T,
    // half 0 combines, then stores through its patch
    float* tail = reinterpret_cast<float*>(lds_raw);
    float* mb = tail + tile * MERGE_FLOATS;
    float* patch = tail + 4 * MERGE_FLOATS + tile * PATCH_FLOATS;
    if (half == 1 && wave_live) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mb[(t * 32 + acc_row(r, lk)) * 32 + lq] = oacc[t][r];
        if (MODE == 0 && lk == 0) { mb[DPAD * 32 + lq] = m_run; mb[DPAD * 32 + 32 + lq] = l_run; }
    }
    __syncthreads();
    if (half == 1 || !wave_live) return;
    float a1 = 1.f, a2 = 1.f;
    if (MODE == 0) {
        const float m2 = mb[DPAD * 32 + lq], l2 = mb[DPAD * 32 + 32 + lq];
        const float m = fmaxf(m_run, m2);
        a1 = (m_run == NEG_INF) ? 0.f : __expf(m_run - m);
        a2 = (m2 == NEG_INF) ? 0.f : __expf(m2 - m);
        l_run = l_run * a1 + l2 * a2;
        m_run = m;
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = oacc[t][r] * a1 + mb[(t * 32 + acc_row(r, lk)) * 32 + lq] * a2;
    if (MODE == 0) {
        const float mul = (l_run > 0.f) ? (p.drop_p > 0.f ? dpar.inv_keep : 1.f) / l_run : 0.f;
        if (q_ok && lk == 0) p.lse[bh * p.T1 + qi] = (l_run > 0.f) ? m_run + __logf(l_run) : NEG_INF;
        store_rows<DPAD>(patch, oacc, mul, p.o + (long)b * p.o_bs + h * p.D, p.o_rs, q0, p.T1, p.D, lane);
    } else {
        store_rows<DPAD>(patch, oacc, p.scale, p.dq + (long)b * p.q_bs + h * p.D, p.q_rs, q0, p.T1, p.D, lane);
    }
}

// ------------------------------------------------------------- dK / dV -------
// wave w of block x owns keys (4x + (w & 3)) * 32 .. + 31 of (b, h) and, of every 64-query chunk (Q and dO rows, lse, delta
// stream through the planes), query tile (w >> 2).
template <int DPAD, int TERMS>
__global__ __launch_bounds__(PL_THREADS, 2) void attn_planes_k_kernel(AttnParams p) {
    using P = Plane<DPAD, TERMS>;
    using R = ResImage<DPAD, TERMS>;
    constexpr int DT = DPAD / 32, KS = DPAD / 16;
    constexpr int IMG = P::ELEMS, RIMG = R::ELEMS;
    constexpr int MERGE_FLOATS = 2 * DPAD * 32 + 32;                // per resident tile: dK^T, dV^T [DPAD][32] + bias gradient [32]
    constexpr int PATCH_FLOATS = 32 * (DPAD + 1);
    constexpr int TAIL_BYTES = 4 * (MERGE_FLOATS + PATCH_FLOATS) * 4;
    // streamed images [buffer][Q | dO], then the block's own 128 keys, resident: [K' * scale | V][64-row half] - their row
    // fragments are re-read per query tile (2 x KS b128 reads per plane) instead of living in 64 registers per lane, which
    // is what lets two waves share a SIMD without spilling.  Three planes at DPAD = 64 (precision 6): the streamed pair is
    // single-buffered (the next chunk waits in registers until every wave has left the current one: a second barrier per
    // chunk) and the resident images are packed (ResImage) - 2 x 27 + 4 x 24 KiB = 150 KiB.
    constexpr bool SB = (TERMS == 6 && DPAD == 64);
    constexpr int NBUF = SB ? 1 : 2;
    constexpr int IMG_BYTES = (NBUF * 2 * IMG + 2 * 2 * RIMG) * 2;
    static_assert(IMG_BYTES + 2 * 2 * PL_ROWS * 4 <= 160 * 1024, "attention dK/dV: LDS image does not fit");
    __shared__ __attribute__((aligned(16))) char lds_raw[(IMG_BYTES > TAIL_BYTES ? IMG_BYTES : TAIL_BYTES)];
    __shared__ __attribute__((aligned(16))) float ld_s[2][2][PL_ROWS];   // [buffer][lse | delta]
    __bf16* imgs = reinterpret_cast<__bf16*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave & 3, half = wave >> 2;
    const int lj = lane & 31, lk = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int k0 = (blockIdx.x * 4 + tile) * 32;
    const bool wave_live = k0 < p.T2;
    const int kj = k0 + lj;
    const bool k_ok = kj < p.T2;
    const float* qb = p.q + (long)b * p.q_bs + h * p.D;
    const float* dob = p.d_o + (long)b * p.o_bs + h * p.D;
    const long bh = (long)b * p.H + h;
    const bool vq = (p.D % 4 == 0) && (p.q_rs % 4 == 0) && ((((uintptr_t)qb) & 15) == 0);
    const bool vo = (p.D % 4 == 0) && (p.o_rs % 4 == 0) && ((((uintptr_t)dob) & 15) == 0);
    __bf16* res = imgs + NBUF * 2 * IMG;                            // [(K', V)][half] resident images
    {
        const float* kbp = p.k + (long)b * p.k_bs + h * p.D;
        const float* vbp = p.v + (long)b * p.v_bs + h * p.D;
        const bool vk = (p.D % 4 == 0) && (p.k_rs % 4 == 0) && ((((uintptr_t)kbp) & 15) == 0);
        const bool vv = (p.D % 4 == 0) && (p.v_rs % 4 == 0) && ((((uintptr_t)vbp) & 15) == 0);
        const int kb0 = blockIdx.x * 128;
        ChunkRegs<DPAD> ra, rb, rc, rd;
        chunk_load<DPAD>(ra, kbp, p.k_rs, kb0, p.T2, p.D, vk);
        chunk_load<DPAD>(rb, kbp, p.k_rs, kb0 + 64, p.T2, p.D, vk);
        chunk_load<DPAD>(rc, vbp, p.v_rs, kb0, p.T2, p.D, vv);
        chunk_load<DPAD>(rd, vbp, p.v_rs, kb0 + 64, p.T2, p.D, vv);
        R::store(ra, res + 0 * RIMG, kb0, p.T2, p.D, p.scale);
        R::store(rb, res + 1 * RIMG, kb0 + 64, p.T2, p.D, p.scale);
        R::store(rc, res + 2 * RIMG, kb0, p.T2, p.D, 1.f);
        R::store(rd, res + 3 * RIMG, kb0 + 64, p.T2, p.D, 1.f);
    }
    const __bf16* Kres = res + (tile >> 1) * RIMG;
    const __bf16* Vres = res + (2 + (tile >> 1)) * RIMG;
    const int rrow = (tile & 1) * 32 + lj;                          // this lane's key in its resident image
    const float kbiasL = ((p.keybias && k_ok) ? p.keybias[bh * p.T2 + kj] : 0.f) * 1.4426950408889634f;
    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[t][r] = 0.f; dvacc[t][r] = 0.f; }
    float dbias = 0.f;
    const DropParams dpar = drop_params(p.drop_p);
    const unsigned long long seed_eff = eff_seed(p.seed, p.seed_dev);

    ChunkRegs<DPAD> qreg, oreg;
    float lse_next = 0.f, delta_next = 0.f;
    const bool ld_wave = wave == 0;
    auto prefetch = [&](int i0) {
        chunk_load<DPAD>(qreg, qb, p.q_rs, i0, p.T1, p.D, vq);
        chunk_load<DPAD>(oreg, dob, p.o_rs, i0, p.T1, p.D, vo);
        if (ld_wave) {
            const int qi = min(i0 + lane, p.T1 - 1);
            lse_next = p.lse[bh * p.T1 + qi];
            delta_next = p.delta[bh * p.T1 + qi];
        }
    };
    auto commit = [&](int buf, int i0) {
        chunk_store<DPAD, TERMS>(qreg, imgs + (buf * 2 + 0) * IMG, i0, p.T1, p.D);
        chunk_store<DPAD, TERMS>(oreg, imgs + (buf * 2 + 1) * IMG, i0, p.T1, p.D);
        if (ld_wave) { ld_s[buf][0][lane] = lse_next; ld_s[buf][1][lane] = delta_next; }
    };
    const int nchunks = (p.T1 + PL_ROWS - 1) / PL_ROWS;
    prefetch(0);
    commit(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = SB ? 0 : (c & 1);
        if (c + 1 < nchunks) prefetch((c + 1) * PL_ROWS);
        const __bf16* Qi = imgs + (buf * 2 + 0) * IMG;
        const __bf16* Oi = imgs + (buf * 2 + 1) * IMG;
        const int i0 = c * PL_ROWS + half * 32;         // this wave's query tile of the chunk
        if (wave_live && i0 < p.T1) {
            const int it = half;
            // lse / delta of this lane's 16 accumulator rows (queries it*32 + 8g + 4lk + 0..3) and the mask bytes, ahead of the MFMAs
            float lsev[16], delv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 a4 = *reinterpret_cast<const float4*>(&ld_s[buf][0][it * 32 + 8 * g4 + 4 * lk]);
                const float4 b4 = *reinterpret_cast<const float4*>(&ld_s[buf][1][it * 32 + 8 * g4 + 4 * lk]);
                lsev[4 * g4] = a4.x; lsev[4 * g4 + 1] = a4.y; lsev[4 * g4 + 2] = a4.z; lsev[4 * g4 + 3] = a4.w;
                delv[4 * g4] = b4.x; delv[4 * g4 + 1] = b4.y; delv[4 * g4 + 2] = b4.z; delv[4 * g4 + 3] = b4.w;
            }
            unsigned char mbyte[16];
            if (p.mask) {
                const unsigned char* mcol = p.mask + (long)b * p.m_bs + min(kj, p.T2 - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) mbyte[r] = mcol[(long)min(i0 + acc_row(r, lk), p.T1 - 1) * p.m_rs];
            }
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                PFrag<TERMS> a, bk;
                P::row_frag(Qi, it * 32 + lj, s, lk, a);
                R::row_frag(Kres, rrow, s, lk, bk);
                sacc = pmma<TERMS>(a, bk, sacc);
                P::row_frag(Oi, it * 32 + lj, s, lk, a);
                R::row_frag(Vres, rrow, s, lk, bk);
                dpacc = pmma<TERMS>(a, bk, dpacc);
            }
            float pd[16], ds[16], dsc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dsc[r] = 1.f;
            if (p.drop_p > 0.f) attn_drop_klane(seed_eff, (unsigned long long)(bh * p.T1), i0, kj, lk, dpar, dsc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool ok = k_ok && i0 + acc_row(r, lk) < p.T1;
                if (p.mask) ok = ok && mbyte[r] != 0;
                const float pv = ok ? __builtin_amdgcn_exp2f(fmaf(sacc[r], 1.4426950408889634f, fmaf(lsev[r], -1.4426950408889634f, kbiasL))) : 0.f;
                pd[r] = pv * dsc[r];                                                     // dropped attention weights
                ds[r] = pv * (dpacc[r] * dsc[r] - delv[r]);                              // dS
                dbias += ds[r];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[8], y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] = pd[8 * s + e]; y[e] = ds[8 * s + e]; }
                PFrag<TERMS> pf, df;
                psplit<TERMS>(x, pf);
                psplit<TERMS>(y, df);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    PFrag<TERMS> a;
                    P::col_frag(Oi, it * 32, t * 32, s, lane, a);
                    dvacc[t] = pmma<TERMS>(a, pf, dvacc[t]);
                    P::col_frag(Qi, it * 32, t * 32, s, lane, a);
                    dkacc[t] = pmma<TERMS>(a, df, dkacc[t]);
                }
            }
        }
        if (SB) __syncthreads();                        // every wave has left the only streamed buffer
        if (c + 1 < nchunks) commit(SB ? 0 : (buf ^ 1), (c + 1) * PL_ROWS);
        __syncthreads();
    }
    dbias = xhalf_sum(dbias);
    // ---- merge the query halves: half 1 parks dK^T, dV^T and the bias gradient, half 0 adds and stores
    float* tail = reinterpret_cast<float*>(lds_raw);
    float* mb = tail + tile * MERGE_FLOATS;
    float* patch = tail + 4 * MERGE_FLOATS + tile * PATCH_FLOATS;
    if (half == 1 && wave_live) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mb[(t * 32 + acc_row(r, lk)) * 32 + lj] = dkacc[t][r];
                mb[DPAD * 32 + (t * 32 + acc_row(r, lk)) * 32 + lj] = dvacc[t][r];
            }
        if (lk == 0) mb[2 * DPAD * 32 + lj] = dbias;
    }
    __syncthreads();
    if (half == 1 || !wave_live) return;
    dbias += mb[2 * DPAD * 32 + lj];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dkacc[t][r] += mb[(t * 32 + acc_row(r, lk)) * 32 + lj];
            dvacc[t][r] += mb[DPAD * 32 + (t * 32 + acc_row(r, lk)) * 32 + lj];
        }
    if (k_ok && p.dkeybias && lk == 0) p.dkeybias[bh * p.T2 + kj] = dbias;
    store_rows<DPAD>(patch, dkacc, p.scale, p.dk + (long)b * p.k_bs + h * p.D, p.k_rs, k0, p.T2, p.D, lane);
    store_rows<DPAD>(patch, dvacc, 1.f, p.dv + (long)b * p.v_bs + h * p.D, p.v_rs, k0, p.T2, p.D, lane);
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
// OE_ATTN_PLANES: 0 = never, 1 = where the structure fits (default): at least three 32-row tiles on the resident axis
// (a block is four waves of one tile each; smaller problems - the decoders' 31-token queries - keep the split-key kernels
// of attention.hip) and D <= 64.
static int planes_mode() {
    static const int mode = getenv("OE_ATTN_PLANES") ? atoi(getenv("OE_ATTN_PLANES")) : 1;
    return mode;
}
static bool planes_fit(int resident_rows) { return planes_mode() != 0 && resident_rows > 64; }

int oe_attn_planes_fwd_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T1) || (terms != 1 && terms != 3 && terms != 6)) return 1;
    dim3 grid(oe_cdiv(p.T1, 128), p.H, p.B);
    if (terms == 6) {          // three planes per streamed tensor: 110 KiB of LDS at D = 64
        if (p.D <= 32) hipLaunchKernelGGL((attn_planes_q_kernel<32, 6, 0>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 6, 0>), grid, dim3(PL_THREADS), 0, st, p);
        return 0;
    }
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<32, 3, 0>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<32, 1, 0>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<64, 3, 0>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 1, 0>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
int oe_attn_planes_dq_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T1) || (terms != 1 && terms != 3 && terms != 6)) return 1;
    dim3 grid(oe_cdiv(p.T1, 128), p.H, p.B);
    if (terms == 6) {
        if (p.D <= 32) hipLaunchKernelGGL((attn_planes_q_kernel<32, 6, 1>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 6, 1>), grid, dim3(PL_THREADS), 0, st, p);
        return 0;
    }
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<32, 3, 1>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<32, 1, 1>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_q_kernel<64, 3, 1>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_q_kernel<64, 1, 1>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
int oe_attn_planes_dkdv_try(const AttnParams& p, int terms, hipStream_t st) {
    if (!planes_fit(p.T2) || (terms != 1 && terms != 3 && terms != 6)) return 1;
    dim3 grid(oe_cdiv(p.T2, 128), p.H, p.B);
    if (terms == 6) {
        if (p.D <= 32) hipLaunchKernelGGL((attn_planes_k_kernel<32, 6>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_k_kernel<64, 6>), grid, dim3(PL_THREADS), 0, st, p);
        return 0;
    }
    if (p.D <= 32) {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_k_kernel<32, 3>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_k_kernel<32, 1>), grid, dim3(PL_THREADS), 0, st, p);
    } else {
        if (terms == 3) hipLaunchKernelGGL((attn_planes_k_kernel<64, 3>), grid, dim3(PL_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_planes_k_kernel<64, 1>), grid, dim3(PL_THREADS), 0, st, p);
    }
    return 0;
}
