/* openeat_hip.h - C ABI of libopeneat_hip.so (gfx950 / MI355X).
 *
 * The reference (TongtongSong/OpenEAT) owns no native code: its hot path is
 * Python nn.Modules whose arithmetic is done by aten kernels.  This library
 * is what replaces those aten calls.  Every entry point below names the
 * reference call site (file:line under /root/reference) whose device work it
 * performs; the Python mirror of the reference's module API
 * (openeat_amd/{models,modules,utils}) binds them through ctypes - see
 * INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, scalars; no framework types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - tensors are dense row-major fp32 unless stated; lengths/labels int32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued asynchronously on it; no call synchronises or allocates.
 *   - return 0 on success, non-zero on error (-1: invalid argument; >0: a
 *     hipError_t); oe_last_error() returns a thread-local message.
 *   - activations: 0 none, 1 relu, 2 swish (x*sigmoid(x)).
 */
#ifndef OPENEAT_HIP_H
#define OPENEAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* oe_last_error(void);
int oe_abi_version(void);

/* Capture hygiene (host only): while `origin` is capturing, how many of the `n_sides` streams hold captured work of the same
 * capture that the origin's next launch would NOT be ordered after (forks that have not been led back).  Returns that
 * count (0 = every fork rejoined; *unjoined_index = first offender), or -1 on error.  No reference counterpart: the
 * reference's step is not captured. */
int oe_capture_unjoined_streams(void* origin, void* const* sides, int n_sides, int* unjoined_index);

/* Diagnostic: when `stream` reaches this point, write the device's constant-rate wall clock (100 MHz ticks) into
 * buf[slot] (device memory).  Captured into a graph it times the replay's phases without a profiler. */
int oe_stamp(long long* buf, int slot, void* stream);

/* ------------------------------------------------------------------------- *
 * GEMM with fused epilogue.  C[m,n] = epi( alpha * sum_k A(m,k) B(n,k) )
 * Replaces aten::addmm/mm/bmm behind every torch.nn.Linear and 1x1 Conv1d of
 * the path: positionwise_feed_forward.py:43, attention.py:56-58,97,185,
 * subsampling.py:113, convolution.py:103,113, ctc.py:38, decoder.py:192, and
 * their backward (executor.py:56).  With conv_gather it is the implicit GEMM
 * of Conv2d(d,d,3,2) (subsampling.py:79) over an NHWC activation.
 *   a_kmajor/b_kmajor: 0 = operand stored [rows][k] (k contiguous),
 *                      1 = operand stored [k][rows].
 *   epilogue order: v = alpha*acc (+bias[n]) ; preact_out <- v ;
 *     v = actgrad_in ? v*act'(actgrad_in[m,n]) : act(v) ; dropout(drop_p,seed,
 *     index m*n_cols+n) ; rowmask[m]==0 -> 0 ; v = beta*v (+ residual[m',n],
 *     m' = m % res_row_mod when res_row_mod > 0: a (T,d) table broadcast over
 *     the batch, i.e. x*sqrt(d)+pe of embedding.py:59) ;
 *     C = v | C += v (accumulate) | atomicAdd (atomic_out, required when
 *     split_k > 1; C must hold the running value, e.g. zeros).
 *   beta must be set (1.0 = plain).  Dropout masks are a pure function of
 *   (seed + *seed_dev * golden, element index): the same arguments regenerate
 *   the mask in backward; seed_dev (optional device counter) lets a captured
 *   HIP graph draw fresh masks on every replay.
 * ------------------------------------------------------------------------- */
enum { OE_GATHER_NONE = 0, OE_GATHER_A = 1, OE_GATHER_B = 2 };

typedef struct oe_gemm_args {
    const float* a; long lda; int a_kmajor;
    const float* b; long ldb; int b_kmajor;
    float* c; long ldc;
    int m, n, k;
    int split_k;
    float alpha; const float* alpha_dev;
    const float* bias;
    int act;
    float* preact_out; const float* actgrad_in; long ld_aux;
    float drop_p; unsigned long long seed; const unsigned long long* seed_dev;
    const unsigned char* rowmask;
    const float* residual; long ldr; int res_row_mod; float beta;
    int accumulate; int atomic_out;
    int conv_gather; int conv_t1, conv_f1, conv_t2, conv_f2, conv_c;   /* implicit conv: input (B,t1,f1,c) NHWC -> (B,t2,f2) positions */
    float* a_colsum; /* optional (precision != 0, k-major A): a_colsum[m] += alpha * sum_k A(m,k), i.e. the bias
                        gradient fused into the weight-gradient GEMM (adders per address = split_k) */
    int precision;   /* 0: fp32-input MFMA (exact fp32 products); 1: bf16 inputs, fp32 accumulate;
                        3: 3-term bf16 split hi*hi+hi*lo+lo*hi (~2^-17 relative error per product);
                        6: 6-term bf16 split - three exact bf16 pieces per operand, h+m+l = x, products
                           hh+hm+mh+mm+hl+lh: what is dropped is < 2^-24 |a||b|, one fp32 rounding - the mode
                           that stands in for the reference's fp32 Linear / Conv products */
    int conv_k, conv_s;   /* kernel size / stride of the gathered conv; 0 = 3 / 2 (Conv2dSubsampling4/8); 5 / 3 is
                             Conv2dSubsampling6's second conv (subsampling.py:136) */
    int conv_kh;          /* kernel HEIGHT when it differs from conv_k (= width); 0 = square.  The gathered window may be
                             smaller than the input allows (conv_t2 <= (conv_t1 - kh) / s + 1): a sub-grid of positions */
    /* Output row scatter (0 = off): logical row r = (b, t, f) over (sc_t2, sc_f2) is stored at row
     * (b*sc_t1 + sc_s*t)*sc_f1 + sc_s*f of C (row stride ldc) and reads its actgrad_in row there too.  With conv_gather
     * on a zero-padded dy this makes the transposed (stride-2) convolution an implicit GEMM per output-parity class that
     * writes the input gradient in place (no column buffer): the (parity) base offsets go into the c / actgrad_in
     * pointers.  Not with atomic_out / accumulate / preact_out / residual / rowmask / dropout. */
    int out_scatter; int sc_t1, sc_f1, sc_t2, sc_f2, sc_s;
    /* Pre-split operands (precision 6 only; all optional, null = absent).  a_planes / b_planes: plane 0 of a copy of A / B
     * as three bf16 planes p0 + p1 + p2 = x (oe_split_planes, or a producer's planes output), same logical layout and
     * leading dimensions as a / b, plane n at + n * plane_stride ELEMENTS; when both are given and the problem qualifies
     * (16-byte alignment, leading dimensions and K multiples of 8 / the K-tile) the product runs on gemm_pl.hip - tiles
     * by LDS-DMA, no conversion in the loop.  b_planes ALONE (a_planes NULL; a row-major A, no gather, no split of the reduction):
     * the weight operand of a Linear taken from its pre-split copy, the activation split on the fragment (gemm_hyb.hip) - half the
     * conversion work of the all-fp32 kernels for nothing but the weights' one split per optimizer step.  c_planes: ALSO write the
     * output as planes (row stride ldcp): the next GEMM's operand without a pass over it. */
    const void* a_planes; long a_plane_stride;
    const void* b_planes; long b_plane_stride;
    void* c_planes; long c_plane_stride; long ldcp;
    /* conv_gather == OE_GATHER_A on pre-split operands only: order of the reduction index.  0 = (kh, kw, ci) as everywhere
     * else; 1 = channel-chunk major, k = ((ci / 32) * KH * KW + kh * KW + kw) * 32 + ci % 32 (C % 32 == 0; the caller lays B's
     * columns out in the same order): the K-loop then visits the KH * KW taps of one 32-channel chunk back to back, so the
     * overlapping windows of neighbouring output positions are re-read from L2 instead of HBM (conv2 forward at config 2:
     * 2.2 GB -> 1.0 GB fetched).  The launch fails if the pre-split kernel does not take the problem. */
    int conv_korder;
    /* actgrad_in points at bf16 values (ld_aux counts elements) instead of fp32 - with act = relu only, where nothing but the
     * sign of the source is read: plane 0 of an activation that exists as bf16 planes alone (the conv1 output of the
     * subsampling front end, whose fp32 copy - 636 MB at config 2 - is then never written or read). */
    int actgrad_bf16;
    /* Weight gradients (a_kmajor && b_kmajor, atomic_out) on pre-split operands whose reduction length k is NOT a multiple of 16
     * (ragged batches: k = valid rows of the step): set when BOTH planes buffers hold at least ceil(k / 16) * 16 rows and A's rows
     * beyond k are ZERO (openeat_amd.planes.alloc pads every buffer that way); a gathered B is read at clamped positions there.
     * The pre-split kernel then runs the padded length; without the flag such a k leaves it to the kernels that split in the loop
     * (configs[4], 24L d = 512: the conv2 weight gradient over k = 222 547 positions took 57.9 ms there, a third of the step). */
    int planes_k_padded;
} oe_gemm_args;

int oe_gemm_f32(const oe_gemm_args* args, void* stream);

/* x (rows, cols) fp32, row stride ld -> three bf16 planes p0 + p1 + p2 = x EXACTLY (round-to-nearest pieces of 8 significant
 * bits each), plane n at planes + n * plane_stride elements, row stride ldp: the pre-split operand format of oe_gemm_f32
 * precision 6 (a_planes / b_planes).  cols % 8 == 0, 16-byte aligned pointers.  Replaces the in-kernel splitting of
 * every consumer by one pass (Linear operands: positionwise_feed_forward.py:36-43, attention.py:36-63). */
int oe_split_planes(const float* x, long ld, long rows, long cols, void* planes, long ldp, long plane_stride, void* stream);
/* how many oe_gemm_f32 calls of this process ran on the pre-split kernel (gemm_pl.hip) so far: tests and tools check that a
 * problem they meant for it did not silently take the splitting kernels */
long oe_gemm_pl_launches(void);
/* launches of the weight-operand-only kernel (gemm_hyb.hip: b_planes alone) so far - tests assert a problem meant for it took it */
long oe_gemm_hyb_launches(void);
/* dispatch knobs of the pre-split kernel, for tests and tuning tools (-1 keeps a value): the smallest grid it accepts
 * (default 96 blocks: below that the splitting kernels' smaller tiles / split-K fill the chip better), a forced tile (22 =
 * 128 x 128, 11 = 64 x 64, 0 = automatic), a forced K-tile (16 / 32, 0 = automatic), waves per 128 x 128 block (8 / 4) */
int oe_gemm_pl_config(int min_blocks, int tile, int bk, int waves);
/* 1 (default, OE_PL_HYBRID): a row-major x row-major problem that runs on 128 x 256 tiles takes whole rounds of the chip on 256 x 256
 * tiles first (rows [0, m1)) and the rest on 128 x 256 (two launches, same operands) where that is fewer tile-rounds: the conv2 forward
 * and input-gradient GEMMs of subsampling.py:88-93.  -1 only reads.  Returns the previous setting. */
int oe_gemm_pl_hybrid(int on);

/* ------------------------------------------------------------------------- *
 * Position-wise feed forward as one kernel (positionwise_feed_forward.py:36-43 with the caller's residual / dropout of
 * encoder_layer.py:81-83,104-106, decoder_layer.py:104-106):
 *     y = residual + beta * drop_out( W2 . drop_in( act( W1 x + b1 ) ) + b2 )
 * on the bf16 matrix cores (precision 1 or 3, as oe_gemm_args.precision); the (rows, ff) intermediate stays in
 * registers.  The weights are consumed pre-split and in MFMA-fragment order: oe_ffn_pack_weights writes W1 (ff, d) and
 * W2 (d, ff) into w1p / w2p (oe_ffn_packed_bytes each) and must be re-run whenever the weights change.
 * oe_ffn_supported: d in {128, 256}, ff a multiple of 128, act 0 / 1 / 2 (none, relu, swish), precision 1 / 3 - anything
 * else is the caller's two oe_gemm_f32 launches.  pre_out / act_out (optional, (rows, ff) dense): the pre-activation
 * W1 x + b1 and the dropped activation, for the backward GEMMs; act_out needs pre_out.  Dropout masks are those of
 * oe_gemm_f32's epilogue on the same tensors (element index row * ff + col with seed_in, row * d + col with seed_out,
 * seed_dev mixed in the same way), so oe_gemm_f32 / oe_dropout_scale regenerate them in backward.
 * All pointers 16-byte aligned, ldx / ldr / ldy multiples of 4.
 * ------------------------------------------------------------------------- */
/* LayerNorm-backward PROLOGUE of a row-block kernel (oe_rowgemm6 at k = 256, oe_ffn_bwd at d = 256, precision 6): when dy != NULL the
 * kernel's input rows are not read but MADE, as
 *     dx = add + LN'(dy; x, stats (mean, rstd per row), gamma),   g = g_alpha * dropmask(g_p, g_seed, seed_dev) * g_rowmask * dx
 * i.e. oe_layernorm_bwd_dx_drop's two outputs (written to dx / g, both (rows, 256) contiguous), g feeding the kernel's product; ws
 * receives the parameter-gradient partials in oe_layernorm_bwd_workspace_floats' layout for oe_layernorm_param_reduce(_table).
 * (encoder_layer.py:79-106: the backward of every `x = residual + dropout(f(norm(x)))` starts from this g, and the LayerNorm
 * backward that makes it used to be a launch of its own.)  ln_rowmask: the LayerNorm's own row mask (rows with 0: dx = add, no
 * parameter gradient) or NULL.  The seed_dev of the enclosing argument block is the one mixed into g_seed. */
typedef struct oe_ln_prologue {
    const float* dy; const float* x; const float* stats; const float* gamma; const float* add;
    float* dx; float* g; float* ws;
    float g_alpha, g_p; unsigned long long g_seed; const unsigned char* g_rowmask; const unsigned char* ln_rowmask;
    /* PAIR (gamma2 != NULL): two norms back to back, y2 = LN2(u), u = LN1(x) (oe_layernorm_pair_bwd_dx_drop: encoder_layer.py:109-110
     * followed by the next layer's :79-80) - dy is then the gradient of y2, add the gradient that reaches u on its other path, x /
     * stats / gamma / beta belong to LN1 (u is recomputed), gamma2 / stats2 to LN2, ws2 receives LN2's parameter partials; no row mask. */
    const float* beta; const float* gamma2; const float* stats2; float* ws2;
} oe_ln_prologue;

/* LayerNorm-FORWARD prologue of a row-block kernel (oe_rowgemm6 at k = 256, oe_ffn_fwd at d = 256, precision 6): when x != NULL the
 * kernel's input rows are MADE as y = LayerNorm(x; gamma, beta, eps) (rows with rowmask 0: y = 0), written to y with the (mean, rstd)
 * pairs to stats - oe_layernorm_fwd's outputs - and fed to the product: the pre-norm of a residual block (encoder_layer.py:79-80,
 * 86-87, 92-93, 103-104) without a launch of its own. */
typedef struct oe_lnf_prologue {
    const float* x; const float* gamma; const float* beta; float eps;
    float* y; float* stats; const unsigned char* rowmask;
    /* PAIR (oe_ffn_fwd only; gamma2 != NULL): y = LN2(u; gamma2, beta2, eps2) with u = LN1(x; gamma, beta, eps) - norm_final of an encoder
     * layer and the next layer's first pre-norm (oe_layernorm_pair_fwd); u (optional output, the feed-forward's residual) and stats2 are
     * written too; no row mask. */
    const float* gamma2; const float* beta2; float eps2; float* u; float* stats2;
} oe_lnf_prologue;

/* LayerNorm-backward EPILOGUE of oe_rowgemm6 (row-block form, 256 <- 256): dx != NULL sends the product's rows dz through the backward of
 * z = act(LayerNorm(x; gamma, beta)) (act in OE_ACT_NONE / RELU / SWISH) with the forward's statistics - the conv module's norm +
 * activation between the depthwise convolution and pointwise_conv2 (convolution.py:107-111), whose input gradient the launch computes:
 * dx (rows, 256) receives the gradient of x, ws the parameter-gradient partials (oe_layernorm_bwd_workspace_floats' layout); y is not
 * written and no other epilogue feature applies. */
typedef struct oe_ln_epilogue {
    const float* x; const float* stats; const float* gamma; const float* beta; int act;
    float* dx; float* ws;
} oe_ln_epilogue;

typedef struct oe_ffn_args {
    const float* x; long ldx;                  /* (rows, d) */
    const void* w1p; const float* b1;          /* packed W1, bias (ff) or NULL */
    const void* w2p; const float* b2;          /* packed W2, bias (d) or NULL */
    int rows, d, ff, act, precision;
    float drop_in; unsigned long long seed_in;
    float drop_out; unsigned long long seed_out;
    const unsigned long long* seed_dev;
    float* pre_out; float* act_out;
    const float* residual; long ldr; float beta;
    float* y; long ldy;
    oe_ln_prologue ln;                          /* oe_ffn_bwd only (precision 6, d = 256): ln.dy != NULL makes the rows of dY (x is then ignored) */
    oe_lnf_prologue lnf;                        /* oe_ffn_fwd only (precision 6, d = 256): lnf.x != NULL makes the rows of x (x is then ignored) */
} oe_ffn_args;
size_t oe_ffn_packed_bytes(int d, int ff, int precision);
int oe_ffn_supported(int d, int ff, int precision, int act);
int oe_ffn_pack_weights(const float* w1, const float* w2, int d, int ff, int precision, void* w1p, void* w2p, void* stream);
int oe_ffn_fwd(const oe_ffn_args* args, void* stream);
/* The feed-forward's input gradient on the same kernel skeleton (autograd of positionwise_feed_forward.py:43):
 *   dH = (dY W2) * dropout mask(drop_in, seed_in) * act'(pre),  dX = dH W1.
 * oe_ffn_pack_weights_bwd packs W2^T in W1's role and W1^T in W2's role (buffers of oe_ffn_packed_bytes each).
 * oe_ffn_bwd reads oe_ffn_args as: x = dY (rows, d) after the output dropout / scale, w1p / w2p = those two streams,
 * pre_out = the forward's pre-activation (rows, ff) - an INPUT here -, act_out = dH (rows, ff) OUTPUT (the weight
 * gradient of W1 consumes it), y = dX (rows, d); b1, b2, residual unset, drop_out 0, beta 1. */
int oe_ffn_pack_weights_bwd(const float* w1, const float* w2, int d, int ff, int precision, void* w2t_packed, void* w1t_packed, void* stream);
int oe_ffn_bwd(const oe_ffn_args* a, void* stream);
/* The weights of n feed-forwards packed in ONE launch, both orientations (the optimizer moves every weight every step: 48 pack
 * launches per step at config 2 otherwise).  table: device array of n entries of NINE 64-bit words { W1, W2, w1p, w2p, w2t_packed,
 * w1t_packed, d, ff, planes } (device pointers as integers; planes = 1 / 2 / 3 for the precision the entry's buffers were sized for,
 * oe_ffn_packed_bytes; a null destination pair skips that orientation, an all-zero entry is skipped); max_d / max_ff: the largest
 * entry's sizes (launch geometry); precision: validated against max_d / max_ff only.  The table lives in device memory, so a captured graph can hold the launch. */
int oe_ffn_pack_weights_table(const void* table, int n, int max_d, int max_ff, int precision, void* stream);
/* precision 6 (csrc/ffn6.hip; d in {128, 256, 512}, three planes, oe_ffn_packed_bytes = 6 bytes per weight): block shape of the
 * fused kernel - 0 = automatic (32-row blocks of eight waves in two staggered groups wherever ff is a multiple of 256),
 * 1 = 64 rows / four waves (d <= 256), 2 = 32 rows / four waves, 3 = two groups; -1 only reads.  Returns the mode in force.
 * Tuning / tests only: results are identical in every mode up to the order of the fp32 sums over the ff axis. */
int oe_ffn6_config(int mode);

/* One Linear with a short reduction as a ROW-BLOCK GEMM in precision 6 (csrc/ffn6.hip, the fused feed-forward's first half alone):
 *     y[rows, n] = residual + beta * rowmask * dropout( x[rows, k] @ Wg[n, k]^T + bias ),   k in {256, 512}, n a multiple of 128;
 *     also k in {768, 1024} with n in {128, 256} (K-phased kernel: the input gradient through the fused q / k / v projection)
 * (attention.py:56-58,97 linear_q / k / v / out, convolution.py:79-111 pointwise convs, and - with Wg = W^T - their input
 * gradients).  wp: Wg as packed A-operand fragments (oe_rowgemm6_pack_table: 6 bytes per weight, refreshed whenever the weights
 * change).  Dropout / rowmask / residual as oe_gemm_f32's epilogue (mask element index row * n + col, seed_dev mixed in the same way).
 * All pointers 16-byte aligned, ldx / ldr / ldy multiples of 4. */
typedef struct oe_rowgemm_args {
    const float* x; long ldx;
    const void* wp; const float* bias;          /* packed Wg; bias (n) or NULL */
    int rows, k, n;
    float drop_p; unsigned long long seed; const unsigned long long* seed_dev;
    const unsigned char* rowmask;               /* (rows) or NULL: rows with 0 are zeroed before the residual is added */
    const float* residual; long ldr; float beta;
    float* y; long ldy;
    /* tile form only (oe_rowgemm6_form = 2), oe_gemm_f32's activation epilogue: act = OE_ACT_NONE / RELU / SWISH applied after the bias
     * (preact_out, optional: the value before it, row stride ld_aux), or - actgrad_in set - the product times act'(actgrad_in[row, col]) */
    int act; float* preact_out; const float* actgrad_in; long ld_aux;
    oe_ln_prologue ln;                          /* row-block form at k = 256: ln.dy != NULL makes the input rows (x is then ignored) */
    oe_lnf_prologue lnf;                        /* ... or lnf.x != NULL: a LayerNorm forward makes them */
    oe_ln_epilogue lne;                         /* lne.dx != NULL: the product leaves through a LayerNorm backward */
} oe_rowgemm_args;
int oe_rowgemm6_supported(int k, int n);
/* Which kernel oe_rowgemm6 runs: 1 = the row-block form above (k in {256, 512}, n % 128 == 0: one 32-row block streams the whole packed
 * matrix - for thousands of rows), 2 = the TILE form for few rows (rows <= 2048, k in {256, 512, 768, 1024}, n % 32 == 0: one block per
 * 32 x 32 output tile, its eight waves split the reduction - the decoders' Linears, decoder_layer.py:82-106), 0 = neither. */
int oe_rowgemm6_form(int rows, int k, int n);
int oe_rowgemm6(const oe_rowgemm_args* args, void* stream);
/* table: device array of n entries of six 64-bit words { W (device pointer), packed destination, R, Cc, row stride of W,
 * transposed }: W (R, Cc) fp32 -> the fragments of Wg = W (transposed 0: an (R, Cc) operand, x W^T) or Wg = W^T (transposed 1: a
 * (Cc, R) operand, dy W); max_pieces = the largest entry's (Wg rows / 32) * (Wg cols / 16).  One launch for every Linear of the model. */
int oe_rowgemm6_pack_table(const void* table, int n, long max_pieces, void* stream);

/* Several weight gradients  C_i (+)= alpha_i * A_i^T B_i  (A_i (k_i, m_i), B_i (k_i, n_i), both k-major, fp32) in ONE launch
 * of the bf16-planes kernel, accumulated atomically into C_i (ops.flush_wgrads: the deferred weight gradients of a captured
 * step; autograd of torch.nn.Linear in the reference).  oe_gemm_tn_grouped_plan fills the launch geometry fields of a HOST
 * array and returns the total block count (-1: a problem does not qualify: 16-byte aligned operands, leading dimensions
 * that are whole float4s and cover the rounded-up row lengths); the caller copies the array to device memory and passes it
 * to oe_gemm_tn_grouped.  a_colsum (optional): alpha_i * column sums of A_i are added to it (fused bias gradient). */
typedef struct oe_tn_problem {
    const float* a; const float* b; float* c; float* a_colsum; const float* alpha_dev;
    long lda, ldb, ldc;
    int m, n, k;
    float alpha;
    int k_chunk, gx, gy, nz, block_start, reserved;      /* filled by oe_gemm_tn_grouped_plan */
} oe_tn_problem;
int oe_gemm_tn_grouped_plan(oe_tn_problem* problems_host, int n, int target_blocks);
int oe_gemm_tn_grouped(const oe_tn_problem* problems_dev, int n, int total_blocks, int precision, void* stream);

/* column sums: out[n] (+)= alpha * sum_m x[m,n]  - bias gradients of every
 * Linear (autograd of aten::addmm).  alpha_dev optional device scalar. */
int oe_colsum_f32(const float* x, long ldx, int m, int n, float alpha, const float* alpha_dev,
                  float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * LayerNorm over the last dim (torch.nn.LayerNorm in encoder_layer.py:54-62,
 * encoder.py:204, convolution.py:61, decoder_layer.py:43-45, decoder.py:163).
 * rowmask (optional, [rows] bytes): rows with 0 produce an all-zero output
 * row - the masked_fill_ of convolution.py:88-89 fused into norm_conv.
 * stats (optional out): [rows][2] = (mean, rstd) kept for backward.
 * act: activation applied to the normalised output (convolution.py:110:
 * activation(norm(x))); backward then needs beta to rebuild the pre-activation.
 * ------------------------------------------------------------------------- */
/* as oe_layernorm_fwd, plus y_planes (optional): y also as three bf16 planes (rows, d), plane_stride elements apart */
int oe_layernorm_fwd_pl(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                        const unsigned char* rowmask, int act, float* y, float* stats, void* y_planes, long plane_stride, void* stream);
int oe_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                     const unsigned char* rowmask, int act, float* y, float* stats, void* stream);
/* dx = LN'(dy) (+ add, optional: the residual branch's gradient of the
 * pre-norm blocks, may alias dx); dgamma/dbeta ACCUMULATED atomically (caller
 * zeroes them; block partials in `workspace` are summed in a fixed order:
 * deterministic).  rowmask as in forward (masked rows: LN'(dy) = 0, no
 * dgamma/dbeta contribution). */
size_t oe_layernorm_bwd_workspace_floats(int rows, int d);
int oe_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                     const float* stats, int rows, int d, const unsigned char* rowmask, const float* add,
                     float* dx, float* dgamma, float* dbeta, float* workspace, void* stream);
/* Two LayerNorms back to back in one pass, y2 = LN2(LN1(x)): the norms at an encoder layer boundary
 * (/root/reference/openeat/modules/encoder_layer.py:109-110 `x = norm_final(x)` followed by the next layer's :79-80
 * `x = norm_ff_macaron(x)`, or encoder.py's after_norm behind the last layer).  y1 (LN1's output, the next block's residual) may be
 * NULL when nothing else reads it; stats1 / stats2 = (mean, rstd) per row of each norm.  The backward takes dy2 (gradient of
 * y2) and `add` (the gradient reaching y1 on its other path, NULL: none), recomputes y1 from x, and writes dx (plus the optional
 * dropped copy gout of oe_layernorm_bwd_dx_drop) and the partial parameter-gradient rows of LN1 / LN2 into workspace1 / workspace2
 * (oe_layernorm_bwd_workspace_floats(rows, d) floats each; summed by oe_layernorm_param_reduce[_table]). */
int oe_layernorm_pair_fwd(const float* x, const float* gamma1, const float* beta1, float eps1, const float* gamma2, const float* beta2,
                          float eps2, int rows, int d, float* y1, float* stats1, float* y2, float* stats2, void* stream);
int oe_layernorm_pair_bwd_dx_drop(const float* dy2, const float* x, const float* gamma1, const float* beta1, const float* stats1,
                                  const float* gamma2, const float* stats2, int rows, int d, const float* add, float* dx, float* gout,
                                  float g_alpha, float g_p, unsigned long long g_seed, const unsigned long long* g_seed_dev,
                                  const unsigned char* g_rowmask, float* workspace1, float* workspace2, void* stream);
/* The same split in two: oe_layernorm_bwd_dx writes dx and the per-block partial sums of the parameter gradients into
 * `workspace`; oe_layernorm_param_reduce_table reduces the partials of n such calls with ONE launch.  table (device
 * memory) holds 5 int64 words per call: { workspace pointer, rows, d, dgamma pointer, dbeta pointer } (a null workspace
 * pointer skips the entry); max_rows / max_d bound the entries' rows / d.  dgamma / dbeta are accumulated atomically.
 * Every workspace must stay untouched until the reduction has run.  (A captured graph can hold the launch while the
 * host fills the table after the capture: 93 reductions of 4.6 us each become one launch at config 2.) */
int oe_layernorm_bwd_dx(const float* dy, const float* x, const float* gamma, const float* beta, int act, const float* stats,
                        int rows, int d, const unsigned char* rowmask, const float* add, float* dx, float* workspace,
                        void* stream);
int oe_layernorm_param_reduce(const float* workspace, int rows, int d, float* dgamma, float* dbeta, void* stream);   /* one call's partials */
int oe_layernorm_param_reduce_table(const long long* table, int n, int max_rows, int max_d, void* stream);
/* oe_layernorm_bwd_dx with a second output gout = g_alpha * keep(g_seed, element)/(1 - g_p) * dx, rows with
 * g_rowmask[row] == 0 zeroed: oe_dropout_scale applied to dx (same mask definition, d % 8 == 0).  dx of a pre-norm
 * block's LayerNorm is the gradient of the PREVIOUS block's output `residual + out_scale * dropout(f(.))`
 * (encoder_layer.py:83,89,95,106), whose backward starts by applying exactly that mask and scale. */
int oe_layernorm_bwd_dx_drop(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                             const float* stats, int rows, int d, const unsigned char* rowmask, const float* add, float* dx,
                             float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                             const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace,
                             void* stream);
/* ... plus out_planes (optional): three bf16 planes (rows, d) of gout when gout is given, else of dx - the tensor the
 * previous block's GEMMs read as an operand (oe_gemm_args.a_planes) */
int oe_layernorm_bwd_dx_drop_pl(const float* dy, const float* x, const float* gamma, const float* beta, int act,
                                const float* stats, int rows, int d, const unsigned char* rowmask, const float* add, float* dx,
                                float* gout, float g_alpha, float g_p, unsigned long long g_seed,
                                const unsigned long long* g_seed_dev, const unsigned char* g_rowmask, float* workspace,
                                void* out_planes, long plane_stride, void* stream);

/* ------------------------------------------------------------------------- *
 * CTC head: log_softmax + CTCLoss(reduction='sum', zero_infinity=True) and
 * its gradient w.r.t. the logits in one pass structure
 * (ctc.py:38-45: ctc_lo -> log_softmax(2) -> ctc_loss -> / B;  backward via
 * executor.py:56).  blank = 0.
 *   logits  (B, T, ldv>=V) batch-major, row stride ldv
 *   hlens   (B) valid frames, targets (B, Lmax) int32 (entries >= tlen ignored)
 *   utt_weight (B) or NULL: per-utterance weight w_b (NULL = 1).  CTC(length_normalized_loss=True)
 *        (ctc.py:24-25: CTCLoss(reduction='mean')) is w_b = 1 / max(tlen_b, 1) and a further / B on the host.
 *   out: nll (B) unweighted, infeasible -> 0;  loss_sum[0] = sum_b w_b * nll_b,
 *        dlogits (B, T, ldv) = grad_scale * d loss_sum / d logits (may alias
 *        logits; padded frames / infeasible utterances exactly 0), or NULL.
 *   workspace: float[ oe_ctc_workspace_floats(B,T,Lmax) ].
 * ------------------------------------------------------------------------- */
size_t oe_ctc_workspace_floats(int B, int T, int Lmax);
/* oe_ctc_loss_fused launch form, for tests and tuning tools (-1 / 0 keeps a value): pipe_mode 0 = rows, alpha/beta, labels one
 * after the other (default); 1 = for large batches the recursion runs in `chunks` time chunks on two internal streams under
 * the dense passes' traffic (ordinary stream events; capturable; measured SLOWER on MI355X: each cross-stream edge costs
 * more than the chain time it hides); 2 = pipelined whatever the size. */
int oe_ctc_config(int pipe_mode, int chunks);
/* pipe_mode 3 / 4 (round 3): the OVERLAPPED form for large batches / always - row statistics first (a read-only pass over the
 * logits, or oe_ctc_loss_fused_stats' row_stats), then two launches of mixed blocks: [one block per utterance runs the
 * recursion | dense gradient rows of the first half of the frames] and [label fix-up of all frames | dense rows of the second
 * half], then a small launch that zeroes infeasible utterances and sums the loss.  No block waits for another. */
/* As oe_ctc_loss_fused; row_stats (optional): for every row of the logits and every group of 32 columns the pair (max, sum exp(x -
 * max)) over the group's valid columns, [B*T][stats_groups = ceil(V/32)] float2 - what oe_gemm_f32 leaves in
 * oe_gemm_args.row_stats when it produced the logits: the overlapped form then needs no statistics pass over the logits. */
int oe_ctc_loss_fused_stats(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                            int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll, float* loss_sum,
                            float* dlogits, float* workspace, const float* row_stats, int stats_groups, void* stream);
int oe_ctc_loss_fused(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                      int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll, float* loss_sum,
                      float* dlogits, float* workspace, void* stream);

/* CTC greedy search on device (asr_model.py:318-325 + common.py:187-196):
 * argmax over V per frame (lowest index wins ties, = topk(1)), frames >= hlen
 * become `eos`, repeats merged, blanks dropped.  out_tokens (B, T) int32,
 * padded with -1; out_lens (B). */
int oe_ctc_greedy(const float* logits, long ldv, int B, int T, int V, const int* hlens, int eos,
                  int* frame_best, int* out_tokens, int* out_lens, void* stream);

/* ------------------------------------------------------------------------- *
 * Fused multi-head attention (scores -> mask -> softmax -> 0-fill -> dropout
 * -> .V), forward and backward, for attention.py:65-97,112-117,189-209.
 *   q (B,T1,H,D), k/v (B,T2,H,D), out (B,T1,H,D): element (b,t,h,d) at
 *   base + b*bstride + t*rstride + h*D + d (so q/k/v may be slices of one
 *   fused projection buffer).  D <= 64.
 *   mask: bytes, (b,i,j) at mask + b*mask_bstride + i*mask_rstride + j;
 *         mask_rstride = 0 for a (B,1,T2) key mask.  0 = masked out.  Rows
 *         with no valid key produce zeros (the reference's 0-fill).
 *   keybias (B,H,T2): added to the scaled score (relative-position term,
 *         see oe_relpos_prepare); NULL for plain attention.
 *   scores = scale * q.k + keybias;  lse (B,H,T1) natural-log normaliser.
 *   dropout acts on the normalised weights, index ((b*H+h)*T1+i)*T2+j.
 * backward: needs out, lse, d_out; writes dq, dk, dv (same strides as q,k,v),
 *   dkeybias (B,H,T2) if keybias was given; delta (B,H,T1) is scratch.
 * ------------------------------------------------------------------------- */
typedef struct oe_attn_args {
    const float* q; long q_bstride, q_rstride;
    const float* k; long k_bstride, k_rstride;
    const float* v; long v_bstride, v_rstride;
    float* out; long o_bstride, o_rstride;
    float* lse;
    const unsigned char* mask; long mask_bstride, mask_rstride;
    const float* keybias;
    int B, H, T1, T2, D;
    float scale;
    float drop_p; unsigned long long seed; const unsigned long long* seed_dev;
    /* backward only */
    const float* d_out;
    float* dq; float* dk; float* dv;
    float* dkeybias;
    float* delta;
    /* matrix-core arithmetic of the score / context products, as oe_gemm_args.precision:
     * 0 = fp32 MFMA (exact products), 1 = bf16 inputs, 3 = three-term bf16 split (2^-17 per product),
     * 6 = six terms on three exact bf16 pieces (fp32-MFMA-grade) */
    int precision;
    /* hint, forward: the (B, T1, T2) mask is zero above the diagonal (a decoder's self-attention): key blocks that lie wholly
     * above it are not visited.  Results are identical with or without the hint; ignored unless a full mask is given. */
    int causal;
} oe_attn_args;

int oe_attention_fwd(const oe_attn_args* args, void* stream);
int oe_attention_bwd(const oe_attn_args* args, void* stream);

/* Relative-position attention preparation (attention.py:185-200 without
 * rel_shift):  kp[b,t,h,:] = k[b,t,h,:] + p[t,h,:]  (dense (B,T,H,D)),
 * keybias[b,h,t] = scale*(u_h . k[b,t,h,:] + v_h . p[t,h,:]).
 * k: element (b,t,h,d) at k + b*k_bstride + t*k_rstride + h*D + d; p = linear_pos(pos_emb), row stride p_rstride. */
int oe_relpos_prepare(const float* k, long k_bstride, long k_rstride, const float* p, long p_rstride,
                      const float* u, const float* v, int B, int T, int H, int D, float scale, float* kp,
                      float* keybias, void* stream);
/* backward of the above: dk (k's strides) written, dp (T,H,D; row stride
 * dp_rstride) written, du/dv (H,D) accumulated atomically. */
int oe_relpos_backward(const float* dkp, const float* dkeybias, const float* k, long k_bstride, long k_rstride,
                       const float* p, long p_rstride, const float* u, const float* v, int B, int T, int H, int D,
                       float scale, float* dk, float* dp, long dp_rstride, float* du, float* dv, void* stream);

/* GLU over channels, a (rows, 2d) -> y (rows, d)  (convolution.py:104). */
int oe_glu_fwd(const float* a, long rows, int d, float* y, void* stream);
int oe_glu_bwd(const float* a, const float* dy, long rows, int d, float* da, void* stream);

/* out[i] = alpha * x[i] * dropmask(seed, i)/(1-p); rows (of `cols`) with
 * rowmask==0 -> 0.  Backward of the dropout/mask epilogues (torch.nn.Dropout in
 * encoder_layer.py:83,89,95,106 and decoder_layer.py:92,97,106). */
int oe_dropout_scale(const float* x, long n, int cols, float alpha, float p, unsigned long long seed,
                     const unsigned long long* seed_dev, const unsigned char* rowmask, float* out, void* stream);

/* Token embedding * sqrt(d) + positional table (decoder.py:144-147 +
 * embedding.py:59): out[r,:] = table[tok[r],:]*xscale + pe[r % L,:]; backward
 * accumulates into dtable atomically. tokens int64. */
int oe_embed_fwd(const long long* tokens, const float* table, const float* pe, long rows, int L, int d, int V,
                 float xscale, float* out, void* stream);
int oe_embed_bwd(const long long* tokens, const float* dout, long rows, int d, int V, float xscale, float* dtable,
                 void* stream);

/* in [A][B][C] -> out [A][C][B] (optionally accumulated): weight layout change
 * between the checkpoint layout (subsampling.py:79 OIHW, :113 channel-major
 * flatten) and the NHWC kernels. */
int oe_swap_last2(const float* in, long A, int Bd, int Cd, float* out, int accumulate, void* stream);
/* Helpers of the input gradient of a stride-2 3x3 NHWC convolution done as four stride-1 implicit GEMMs, one per parity
 * class of the input position (the backward of subsampling.py:110-116's second Conv2d):
 * oe_pad1_nhwc: dy (B, To, Fo, C) -> out (B, To+2, Fo+2, C) with a border of zeros (C % 4 == 0, 16-byte aligned);
 * oe_conv_dgrad_k3s2_weights: OIHW weight w[C][C][3][3] -> the four classes' B operands [ci][(window row, window col, co)]
 * back to back in out (9 C^2 floats; class (t1%2, f1%2) at offsets 0, 4C^2, 6C^2, 8C^2). */
int oe_pad1_nhwc(const float* dy, int B, int To, int Fo, int C, float* out, void* stream);
/* oe_pad1_nhwc straight into three bf16 planes of the padded tensor (plane n at planes + n * plane_stride elements, rows of C;
 * oe_split_planes' definition); out (the padded fp32 tensor) is optional - the parity-class GEMMs on pre-split operands read
 * the planes only. */
int oe_pad1_nhwc_planes(const float* dy, int B, int To, int Fo, int C, float* out, void* planes, long plane_stride, void* stream);
int oe_conv_dgrad_k3s2_weights(const float* w, int C, float* out, void* stream);

/* The joint loss of /root/reference/openeat/models/asr_model.py:150-157 and :196-198 as one launch on device scalars:
 *   att = loss_att * (1 - reverse_weight) + loss_att_r * reverse_weight   (loss_att_r NULL: att = loss_att)
 *   out = (ctc_weight * loss_ctc + (1 - ctc_weight) * att) * inv_accum    (loss_ctc NULL: out = att * inv_accum)
 * each product and sum rounded on its own in that order (the value torch's scalar ops produce); oe_loss_combine_bwd writes
 * the three gradients for an incoming scalar gradient g (d_ctc / d_att_r NULL where the term is absent). */
int oe_loss_combine(const float* loss_ctc, const float* loss_att, const float* loss_att_r, float ctc_weight, float reverse_weight,
                    float inv_accum, float* out, void* stream);
int oe_loss_combine_bwd(const float* g, float ctc_weight, float reverse_weight, float inv_accum, float* d_ctc, float* d_att,
                        float* d_att_r, void* stream);

/* out = a*(*a_dev)*x + b*y (y, a_dev may be NULL). */
int oe_axpby(const float* x, const float* y, long n, float a, float b, const float* a_dev, float* out, void* stream);

/* y = act(x) (swish.py:15-17) and out = dy * act'(pre). */
int oe_act_fwd(const float* x, long n, int act, float* y, void* stream);
int oe_act_grad(const float* dy, const float* pre, long n, int act, float* out, void* stream);

/* log_softmax over the last dim (ctc.py:56-64; asr_model.py:484-488). */
int oe_log_softmax(const float* x, long rows, int V, float* out, void* stream);
/* log_softmax(x)[row, idx_a[row]] (0 where idx_a is out of range) and, if out_b is given, log_softmax(x)[row, idx_b] for one
 * fixed column - the hypothesis' token and <eos> log-probabilities attention rescoring sums (asr_model.py:504-528) - without
 * writing the (rows, V) log-probability tensor. */
int oe_logprob_gather(const float* x, long rows, int V, const long long* idx_a, int idx_b, float* out_a, float* out_b, void* stream);
/* The decoders' token bookkeeping of a training step in one launch (asr_model.py:162-176 with common.py:61-132 and
 * mask.py:9-69): from the labels ys_pad (B, L) i32 (ignore_id where there is none) and their lengths (B) i32, at the fixed
 * width W = L + 1: ys_in = [sos, labels, eos...], ys_out = [labels, eos, ignore...] (both (B, W) i64), the same for the
 * reversed labels (r_ys_in / r_ys_out, or both NULL), and tgt_mask (B, W, W) u8 = (j < len + 1) && (j <= i). */
int oe_att_inputs(const int* ys_pad, const int* ys_lens, int B, int L, int sos, int eos, int ignore_id, long long* ys_in,
                  long long* ys_out, long long* r_ys_in, long long* r_ys_out, unsigned char* tgt_mask, void* stream);
/* Per-row top-k, sorted descending, of x (rows, V) or - log_softmax != 0 - of its row-wise log-softmax, fused
 * (asr_model.py:251, 358 `log_softmax(...).topk(beam_size)`; :258 `scores.topk`).  vals (rows, k) f32, idx (rows, k) i64;
 * ties go to the lowest index.  k <= V <= 40000. */
int oe_topk_rows(const float* x, long rows, int V, int k, int log_softmax, float* vals, long long* idx, void* stream);

/* Masked softmax (+ dropout) over the last dim of a MATERIALISED score tensor: the module-API method
 * MultiHeadedAttention.forward_attention (attention.py:65-97) - the training / decoding path never builds this tensor
 * (oe_attention_fwd).  scores (B,H,T1,T2) dense; mask bytes (b,i,j) at mask + b*mask_bstride + i*mask_rstride + j
 * (mask_rstride = 0: a (B,1,T2) key mask; NULL: no mask), 0 = masked: -inf before the softmax, 0 after it
 * (attention.py:85-88).  y = the softmax (kept for backward), out = y * dropout(drop_p, element index) (out may alias y
 * when drop_p == 0).  backward: dscores = y * (g - sum_j g_j y_j), g = dout * the same dropout mask; rows = B*H*T1. */
int oe_masked_softmax_fwd(const float* scores, const unsigned char* mask, long mask_bstride, long mask_rstride, int B, int H,
                          int T1, int T2, float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* y,
                          float* out, void* stream);
int oe_masked_softmax_bwd(const float* y, const float* dout, long rows, int T2, float drop_p, unsigned long long seed,
                          const unsigned long long* seed_dev, float* dscores, void* stream);

/* GlobalCMVN (modules/cmvn.py:43-45): y = (x - mean[f]) * istd[f]. */
int oe_global_cmvn(const float* x, const float* mean, const float* istd, long n, int F, float* y, void* stream);

/* Conv2d(1,C,3,stride 2)+ReLU of the subsampling front (subsampling.py:77-78):
 * x (B,T,F) -> y (B,T1,F1,C) NHWC, w (C,1,3,3).  wgrad: dy must already be
 * masked by y>0; dw/db accumulated atomically. */
int oe_conv1_fwd(const float* x, const float* w, const float* bias, int B, int T, int F, int C, float* y, void* stream);
/* the same, also writing y as three bf16 planes (oe_split_planes' layout: plane n at y_planes + n * plane_stride elements, rows of C)
 * for a precision-6 conv2 on pre-split operands; y_planes NULL = oe_conv1_fwd; y NULL (with y_planes): the planes are the only
 * copy written (oe_gemm_args.actgrad_bf16 reads the ReLU mask from plane 0) */
int oe_conv1_fwd_pl(const float* x, const float* w, const float* bias, int B, int T, int F, int C, float* y,
                    void* y_planes, long plane_stride, void* stream);
int oe_conv1_wgrad(const float* x, const float* dy, int B, int T, int F, int C, float* dw, float* db, void* stream);

/* Input gradient of Conv2d(C,C,3,stride 2) in gather form, fused with the
 * ReLU mask of the layer below (autograd of subsampling.py:78-79):
 * dcol (B*T2*F2, 9C) = dy @ W[co][kh][kw][ci]  ->  dx (B,T1,F1,C) * (y1 > 0). */
int oe_col2im_relu(const float* dcol, const float* y1, int B, int T1, int F1, int C, float* dx, void* stream);
/* the same for a KS x KS kernel with stride S (oe_col2im_relu = 3, 2) */
int oe_col2im_relu_ks(const float* dcol, const float* y1, int B, int T1, int F1, int C, int KS, int S, float* dx,
                      void* stream);

/* GLU + depthwise Conv1d(K, groups=d) of the Conformer conv module
 * (convolution.py:104-107): a (B,T,2d) -> y (B,T,d); w (d,1,K); causal => left
 * padding K-1 (convolution.py:40-47,92-93).  gpad (optional, [d]): value of the
 * GLU output on the virtual frames t < 0 - the reference pads BEFORE the
 * pointwise conv, so they carry GLU(pointwise bias); dgpad accumulates its
 * gradient.  backward: da (B,T,2d) written, dw/db accumulated atomically. */
int oe_dwconv_glu_fwd(const float* a, const float* w, const float* bias, const float* gpad, int B, int T, int d,
                      int K, int causal, float* y, void* stream);
/* oe_dwconv_glu_fwd plus the LayerNorm + activation that follows it in the Conformer convolution module
 * (/root/reference/openeat/modules/convolution.py:107-111) from the same launch: z (optional) = act(LN(y)), ln_stats = (mean, rstd)
 * per row as oe_layernorm_fwd writes them; bit-identical to oe_dwconv_glu_fwd followed by oe_layernorm_fwd. */
int oe_dwconv_glu_ln_fwd(const float* a, const float* w, const float* bias, const float* gpad, int B, int T, int d, int K, int causal,
                         float* y, const float* ln_gamma, const float* ln_beta, float ln_eps, int ln_act, float* z, float* ln_stats,
                         void* stream);
size_t oe_dwconv_glu_bwd_workspace_floats(int B, int T, int d, int K);
int oe_dwconv_glu_bwd(const float* a, const float* dy, const float* w, const float* gpad, int B, int T, int d, int K,
                      int causal, float* da, float* dw, float* db, float* dgpad, float* workspace, void* stream);

/* Label-smoothed KL loss + accuracy + gradient, fused
 * (label_smoothing_loss.py:58-91, common.py:135-157).  logits (rows, ldv) are
 * OVERWRITTEN by grad_scale/denom * (softmax - true_dist) when write_grad;
 * rows with target == ignore_id give 0.  denom = batch_size, or the number of
 * non-ignored rows when normalize_length.  out3 = {loss, #correct, #valid}.
 * target int64.  workspace: oe_lsm_workspace_bytes(rows). */
size_t oe_lsm_workspace_bytes(long rows);
int oe_lsm_loss_fused(float* logits, long ldv, long rows, int V, const long long* target, int ignore_id, float smoothing,
                      int normalize_length, float batch_size, float grad_scale, int write_grad, float* out3,
                      void* workspace, void* stream);

/* Global L2 norm of the flat gradient arena (clip_grad_norm_, executor.py:58). */
size_t oe_grad_norm_workspace_floats(void);
int oe_grad_norm(const float* g, long n, float* workspace, float* norm_out, void* stream);
/* clip + Adam over the flat arenas (torch.optim.Adam defaults; executor.py:59-61:
 * the step is skipped when the norm is not finite).  state: float[2] device,
 * state[0] = step count, state[1] = 0.  lr_dev (device scalar) overrides lr. */
int oe_adam_step(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float lr, float beta1,
                 float beta2, float eps, float max_norm, const float* total_norm, float* state, void* stream);

/* Kaldi log-mel filterbank on device, replacing the torchaudio.compliance.kaldi.fbank
 * call of dataset.py:93-100 (frame_length 25 ms, frame_shift 10 ms, snip_edges,
 * remove_dc_offset, preemphasis 0.97, povey window, power spectrum, 20 Hz..Nyquist
 * mel triangles, log with FLT_EPSILON floor) incl. the x 2^15 of dataset.py:75.
 *   wav (B, wav_stride) float; nsamples (B) valid samples per row or NULL (= wav_stride)
 *   out (B, Tmax, n_mel); frames beyond 1+(n-win)/hop are written as 0.
 *   window [win]; twiddle [256][2] = (cos, -sin)(2 pi k/512); mel filter m covers FFT bins
 *   mel_start[m] .. with weights mel_w[mel_off[m] .. mel_off[m+1]).
 *   cmvn_mean/istd (optional, [n_mel]): GlobalCMVN (modules/cmvn.py:43-45) fused in. */
int oe_fbank(const float* wav, const int* nsamples, int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
             float scale, float preemph, const float* window, const float* twiddle, const int* mel_start,
             const int* mel_off, const float* mel_w, float floor_eps, const float* cmvn_mean,
             const float* cmvn_istd, float* out, void* stream);
/* The same with kaldi's waveform dither (dataset.py:98, `dither=wav_dither`): N(0, dither^2) noise on every sample of every
 * frame's window (after the x 2^15), drawn from a counter-based generator keyed by (seed, frame, sample).  Distribution
 * parity with the reference (torch's global generator there), not bit parity.  dither = 0 is oe_fbank. */
int oe_fbank_dither(const float* wav, const int* nsamples, int B, long wav_stride, int Tmax, int win, int hop, int n_mel,
                    float scale, float preemph, const float* window, const float* twiddle, const int* mel_start,
                    const int* mel_off, const float* mel_w, float floor_eps, const float* cmvn_mean,
                    const float* cmvn_istd, float dither, unsigned long long seed, float* out, void* stream);
/* Per-utterance (x - mean_t)/std_t over each utterance's own nframes[b] frames, in place
 * (feature_processor.py:5-8: population std, no epsilon). */
int oe_utt_normalize(float* x, const int* nframes, int B, int Tmax, int F, void* stream);

/* SpecAugment masks (feature_processor.py:10-43) on the padded batch, in place: for utterance b the frames
 * [t_masks[b][k][0], t_masks[b][k][1]) and the bins [f_masks[b][k][0], f_masks[b][k][1]) become `value`
 * (SPEC_MASK = 0); only the utterance's own nframes[b] frames are touched.  The (start, end) pairs are drawn on
 * the host in the reference's order (openeat_amd/augment.py), which makes the result bit-identical for a given seed. */
int oe_spec_augment(float* x, const int* nframes, int B, int Tmax, int F, const int* t_masks, int nt, const int* f_masks,
                    int nf, float value, void* stream);
/* Spec-substitute (feature_processor.py:45-64): subs (B, ns, 3) = (start, end, pos); rows [start, end) are replaced
 * by rows [start - pos, end - pos), one substitution after the other; end - start <= max_rows. */
int oe_spec_substitute(float* x, int B, int Tmax, int F, const int* subs, int ns, int max_rows, void* stream);
/* Feature dither (dataset.py:197-201): x += (u - 0.5) * a, u ~ U[0,1) from Philox(seed, element); only each utterance's
 * own nframes[b] frames.  Distribution parity with the reference (which uses numpy's global generator), not bit parity. */
int oe_feature_dither(float* x, const int* nframes, int B, int Tmax, int F, float a, unsigned long long seed, void* stream);

/* Speed perturbation of a padded waveform batch (audio_processor.py:20-35: sox `speed s` + `rate sr`): utterance b is
 * read speed[b] times faster and resampled back to the same rate, out[b, i] = x_b(i * speed[b]) through a Hann-windowed
 * sinc (cutoff 0.95 * min(1, 1/s), 16 zero crossings each side, unit DC gain); speed 1 copies.  wav (B, ld_in) with
 * n_in[b] valid samples, out (B, ld_out) with n_out[b] valid samples (the host sets n_out[b] = floor(n_in[b] / speed[b]
 * + 0.5)), samples past n_out[b] up to Nmax_out are zeroed.  sox itself is outside the reference tree: distribution
 * parity, checked against a float64 restatement and scipy's polyphase resampler. */
int oe_speed_perturb(const float* wav, long ld_in, const int* n_in, const float* speed, int B, int Nmax_out, float* out,
                     long ld_out, const int* n_out, void* stream);

/* CTC prefix beam search on the device, one wavefront per utterance (asr_model.py:359-396; the host functions below are
 * the same algorithm on the CPU and serve as its checker).  topk_logp (B, Tmax, beam) f32 and topk_idx (B, Tmax, beam) i64
 * as oe_topk_rows writes them; lens (B) i32 valid frames per utterance or NULL; beam <= 16.  workspace:
 * oe_ctc_prefix_beam_workspace_bytes(B, Tmax, beam) bytes of device memory whose LAST 4-byte word the caller zeroes and
 * reads back after the stream has drained (non-zero: a prefix exceeded max_len).  Outputs, device memory: out_prefix
 * (B, beam, max_len) i32, out_len (B, beam) i32 (-1: fewer than `beam` prefixes exist), out_score (B, beam) f64. */
size_t oe_ctc_prefix_beam_workspace_bytes(int B, int Tmax, int beam);
int oe_ctc_prefix_beam(const float* topk_logp, const long long* topk_idx, int B, int Tmax, const int* lens, int beam, int max_len,
                       void* workspace, int* out_prefix, int* out_len, double* out_score, void* stream);
/* CTC prefix beam search, HOST code (all pointers are host pointers): the per-frame recursion of
 * asr_model.py:359-396 on the top-`beam` (log-prob, token) pairs of every frame (computed on the
 * device).  Doubles and insertion-ordered stable pruning as in the reference's Python, so the
 * n-best list and scores are identical.  out_prefix_host (beam, max_len) int32, out_len_host (beam)
 * (-1 = fewer than `beam` hypotheses), out_score_host (beam) = log_add(pb, pnb). */
int oe_ctc_prefix_beam_host(const float* topk_logp_host, const long long* topk_idx_host, int T, int beam,
                            int max_len, int* out_prefix_host, int* out_len_host, double* out_score_host);
/* The same for B utterances at once, spread over n_threads host threads (<= 0: all cores): inputs (B, Tmax, beam) with
 * lens[b] valid frames each; outputs (B, beam, max_len) / (B, beam) / (B, beam). */
int oe_ctc_prefix_beam_host_batch(const float* topk_logp_host, const long long* topk_idx_host, int B, int Tmax,
                                  const int* lens_host, int beam, int max_len, int* out_prefix_host, int* out_len_host,
                                  double* out_score_host, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* OPENEAT_HIP_H */
