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
 *     index m*n_cols+n) ; rowmask[m]==0 -> 0 ; v = residual[m,n] + beta*v ;
 *     C = v | C += v (accumulate) | atomicAdd (atomic_out, required when
 *     split_k > 1; C must hold the running value, e.g. zeros).
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
    float drop_p; unsigned long long seed;
    const unsigned char* rowmask;
    const float* residual; long ldr; float beta;
    int accumulate; int atomic_out;
    int conv_gather; int conv_t1, conv_f1, conv_t2, conv_f2, conv_c;
} oe_gemm_args;

int oe_gemm_f32(const oe_gemm_args* args, void* stream);

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
 * ------------------------------------------------------------------------- */
int oe_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int rows, int d,
                     const unsigned char* rowmask, float* y, float* stats, void* stream);
/* dx written; dgamma/dbeta ACCUMULATED atomically (caller zeroes them).
 * rowmask as in forward (masked rows: dx = 0, no dgamma/dbeta contribution). */
int oe_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats, int rows, int d,
                     const unsigned char* rowmask, float* dx, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------- *
 * CTC head: log_softmax + CTCLoss(reduction='sum', zero_infinity=True) and
 * its gradient w.r.t. the logits in one pass structure
 * (ctc.py:38-45: ctc_lo -> log_softmax(2) -> ctc_loss -> / B;  backward via
 * executor.py:56).  blank = 0.
 *   logits  (B, T, ldv>=V) batch-major, row stride ldv
 *   hlens   (B) valid frames, targets (B, Lmax) int32 (entries >= tlen ignored)
 *   out: nll (B) with infeasible -> 0, loss_sum[0] = sum_b nll_b,
 *        dlogits (B, T, ldv) = grad_scale * d loss_sum / d logits (may alias
 *        logits; padded frames / infeasible utterances exactly 0), or NULL.
 *   workspace: float[ oe_ctc_workspace_floats(B,T,Lmax) ].
 * ------------------------------------------------------------------------- */
size_t oe_ctc_workspace_floats(int B, int T, int Lmax);
int oe_ctc_loss_fused(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                      int Lmax, const int* tlens, float grad_scale, float* nll, float* loss_sum, float* dlogits,
                      float* workspace, void* stream);

/* CTC greedy search on device (asr_model.py:318-325 + common.py:187-196):
 * argmax over V per frame (lowest index wins ties, = topk(1)), frames >= hlen
 * become `eos`, repeats merged, blanks dropped.  out_tokens (B, T) int32,
 * padded with -1; out_lens (B). */
int oe_ctc_greedy(const float* logits, long ldv, int B, int T, int V, const int* hlens, int eos,
                  int* frame_best, int* out_tokens, int* out_lens, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENEAT_HIP_H */
