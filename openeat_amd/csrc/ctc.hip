// CTC head on device: softmax rows, alpha/beta recursion with wave shuffles,
// gradient w.r.t. the logits, and greedy search.
//
// Replaces  /root/reference/openeat/modules/ctc.py:38-45  (log_softmax ->
// torch.nn.CTCLoss(sum, zero_infinity) -> /B and its autograd) and
// /root/reference/openeat/models/asr_model.py:318-325 (greedy).
//
// Pass structure (HBM-bound; B*T*V*4 bytes = the logits):
//   k1 rows     : one wave per frame holds its row in registers: read logits ONCE, write
//                 softmax * scale ONCE (the dense part of the gradient, in place if asked),
//                 and emit lp[b,t,s] = log2-prob at the 2L+1 states
//   k2 alphabeta: one block per utterance, wave 0 runs alpha, wave 1 runs beta
//                 concurrently; neighbours s-1,s-2 come from lane shuffles; lp is
//                 prefetched a whole chunk of frames ahead (registers), so a step costs the
//                 shuffle + lse3 chain and not a global-memory round trip
//   k3 labels   : per frame, the <= 2L+1 classes that occur in the target get
//                 (softmax - sum_{s:l'_s=c} exp(alpha+beta-lp-ll)) * scale (states of one class are
//                 linked once per utterance by a third wave of k2); rows of infeasible
//                 utterances are zeroed (zero_infinity)
// Algorithmic bytes: 2*B*T*V*4 (+ 3 small state arrays), which is what k1 moves.
#include <stdlib.h>
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define NEG_INF (-INFINITY)
#define CTC_MAXQ 8          // states per lane in k1 / k3 (Sp <= 64 * 8)

// ------------------------------------------------------------------ k1 ------
// NV4 > 0: the row is NV4 float4 per lane (V <= 256 * NV4; rows 16-byte aligned and padded to a multiple of four floats);
// NV4 == 0: any V / alignment, the row is read twice (the second time from L2).
// dlogits is not __restrict__: it may alias logits (every load of a row is issued before its first store).
template <int NV4, bool WRITE>
__global__ __launch_bounds__(256) void ctc_rows_kernel(const float* logits, long ldv, long rows, int T, int V,
                                                        const int* __restrict__ hlens, const int* __restrict__ targets,
                                                        int Lmax, const int* __restrict__ tlens, int Sp, float scale,
                                                        const float* __restrict__ utt_weight, float* __restrict__ lp_out,
                                                        float* dlogits, int t0, int Tc, float2* __restrict__ rowstat) {
    // frames t0 .. t0 + Tc - 1 of every utterance (`rows` = B * Tc of them; the whole batch: t0 = 0, Tc = T)
    // rowstat (optional): (row maximum, 1 / sum exp(x - max)) of every live row, for ctc_dense_row
    const int lane = threadIdx.x & 63;
    const long rloc = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rloc >= rows) return;
    const int b = (int)(rloc / Tc), t = t0 + (int)(rloc % Tc);
    const long row = (long)b * T + t;
    const float* p = logits + row * ldv;
    float* g = WRITE ? dlogits + row * ldv : nullptr;
    const int nv = (V + 3) >> 2;
    if (t >= hlens[b]) {                      // padded frame: exact zeros, no statistics
        if (WRITE) {
            if (NV4 > 0) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int i = lane; i < nv; i += 64) reinterpret_cast<float4*>(g)[i] = z;
            } else {
                for (int i = lane; i < V; i += 64) g[i] = 0.f;
            }
        }
        return;
    }
    if (WRITE && utt_weight) scale *= utt_weight[b];
    const int L = min(tlens[b], Lmax);
    const int S = 2 * L + 1;
    // logits at the states' classes, loaded together with the row
    float lv[CTC_MAXQ];
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int st = lane + 64 * q;
        lv[q] = 0.f;
        if (st < S) lv[q] = p[(st & 1) ? targets[(long)b * Lmax + (st >> 1)] : 0];
    }
    float m, l2s;          // log-softmax in base 2 = (x - m) * log2(e) - log2(sum exp(x - m))
    if (NV4 > 0) {
        float4 v[NV4 > 0 ? NV4 : 1];
        const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i = lane + 64 * j;
            v[j] = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            if (i < nv) v[j] = p4[i];
        }
        m = NEG_INF;
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int e = (lane + 64 * j) * 4;        // columns past V (row padding) do not count
            if (e + 1 >= V) v[j].y = NEG_INF;
            if (e + 2 >= V) v[j].z = NEG_INF;
            if (e + 3 >= V) v[j].w = NEG_INF;
            m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            v[j].x = __expf(v[j].x - m); v[j].y = __expf(v[j].y - m);
            v[j].z = __expf(v[j].z - m); v[j].w = __expf(v[j].w - m);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        s = wave_sum(s);
        l2s = __builtin_amdgcn_logf(s);
        if (WRITE) {
            const float f = scale / s;
#pragma unroll
            for (int j = 0; j < NV4; ++j) {
                const int i = lane + 64 * j;
                if (i < nv) reinterpret_cast<float4*>(g)[i] = make_float4(v[j].x * f, v[j].y * f, v[j].z * f, v[j].w * f);
            }
        }
    } else {
        m = NEG_INF;
        for (int i = lane; i < V; i += 64) m = fmaxf(m, p[i]);
        m = wave_max(m);
        float s = 0.f;
        for (int i = lane; i < V; i += 64) s += __expf(p[i] - m);
        s = wave_sum(s);
        l2s = __builtin_amdgcn_logf(s);
        if (WRITE) {
            const float f = scale / s;
            for (int i = lane; i < V; i += 64) g[i] = __expf(p[i] - m) * f;
        }
    }
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int st = lane + 64 * q;
        if (st < S) lp_out[row * Sp + st] = (lv[q] - m) * 1.4426950408889634f - l2s;
    }
    if (rowstat && lane == 0) rowstat[row] = make_float2(m, __builtin_amdgcn_exp2f(-l2s));
}

// ---- the overlapped form (oe_ctc_loss_fused, ctc_pipe_mode 3 / 4) ---------------------------------------------------------
// One after the other the three passes cost rows + chain + labels (116 + 60 + 30 us at B = 64 x 16 s): the recursion is a serial
// chain on 64 waves and the label fix-up a latency-bound scatter, both with the chip idle around them.  Here the row statistics
// come first (from the logits: ctc_rows_kernel<.., false> with `rowstat`, a read-only pass; or from the per-32-column partials
// the projection GEMM's epilogue left: ctc_lp_kernel, a few microseconds), and then TWO launches of mixed blocks follow:
//   launch 2: one block per utterance runs alpha / beta / the label chains  +  dense-gradient blocks for frames [0, T/2)
//   launch 3: the label fix-up blocks of all frames                         +  dense-gradient blocks for frames [T/2, T)
// The special blocks come first in the grid (dispatched first), the dense blocks - pure streaming - fill the chip around them.
// No block waits for another: the dense blocks write every column EXCEPT the blank and the utterance's labels (a per-wave bitmask
// in LDS), the fix-up blocks write exactly those, so the two kinds never touch the same word and need no order between them
// (in place too: a dense wave may read a label column the fix-up has already overwritten, but it never uses or writes it).
// Rows of infeasible utterances are zeroed by a last small launch (ctc_final_kernel), which also sums the loss.
template <int NV4>
__device__ __forceinline__ void ctc_dense_row(const float* logits, long ldv, long rows, int T, int V, const int* __restrict__ hlens,
                                              const int* __restrict__ targets, int Lmax, const int* __restrict__ tlens, float scale,
                                              const float* __restrict__ utt_weight, const float2* __restrict__ rowstat, float* dlogits,
                                              int t0, int Tc, long rloc, unsigned* maskw) {
    constexpr int MW = NV4 * 8;                        // bitmask dwords: 256 * NV4 columns
    const int lane = threadIdx.x & 63;
    if (rloc >= rows) return;
    const int b = (int)(rloc / Tc), t = t0 + (int)(rloc % Tc);
    const long row = (long)b * T + t;
    const float* p = logits + row * ldv;
    float* g = dlogits + row * ldv;
    const int nv = (V + 3) >> 2;
    if (t >= hlens[b]) {                               // padded frame: exact zeros in every column
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lane; i < nv; i += 64) reinterpret_cast<float4*>(g)[i] = z;
        return;
    }
    float4 v[NV4];
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int i = lane + 64 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nv) v[j] = p4[i];
    }
    const float2 st = rowstat[row];
    if (utt_weight) scale *= utt_weight[b];
    // the columns the fix-up owns: blank and the utterance's labels
#pragma unroll
    for (int i = lane; i < MW; i += 64) maskw[i] = (i == 0) ? 1u : 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int L = min(tlens[b], Lmax);
    for (int i = lane; i < L; i += 64) {
        const int c = targets[(long)b * Lmax + i];
        if (c >= 0 && c < V) atomicOr(&maskw[c >> 5], 1u << (c & 31));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float f = scale * st.y;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            const int e = 4 * i;
            const unsigned bits = (maskw[e >> 5] >> (e & 31)) & 0xfu;
            const float4 o = make_float4(__expf(v[j].x - st.x) * f, __expf(v[j].y - st.x) * f, __expf(v[j].z - st.x) * f, __expf(v[j].w - st.x) * f);
            if (bits == 0u) {
                reinterpret_cast<float4*>(g)[i] = o;      // (columns past V inside the row padding get a value nobody reads, as in ctc_rows_kernel)
            } else {
                if (!(bits & 1u)) g[e] = o.x;
                if (!(bits & 2u)) g[e + 1] = o.y;
                if (!(bits & 4u)) g[e + 2] = o.z;
                if (!(bits & 8u)) g[e + 3] = o.w;
            }
        }
    }
}

// Row statistics from the projection GEMM's epilogue (oe_gemm_args.row_stats: per row and per group of 32 columns the pair
// (max, sum exp(x - max))) -> lp at the states and (row maximum, 1 / sum): one wave per frame, no pass over the logits.
__global__ __launch_bounds__(256) void ctc_lp_kernel(const float* __restrict__ logits, long ldv, long rows, int T, int V,
                                                      const int* __restrict__ hlens, const int* __restrict__ targets, int Lmax,
                                                      const int* __restrict__ tlens, int Sp, const float2* __restrict__ part, int groups,
                                                      float* __restrict__ lp_out, float2* __restrict__ rowstat) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / T), t = (int)(row % T);
    if (t >= hlens[b]) return;
    const float* p = logits + row * ldv;
    const int L = min(tlens[b], Lmax);
    const int S = 2 * L + 1;
    float lv[CTC_MAXQ];
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int st = lane + 64 * q;
        lv[q] = 0.f;
        if (st < S) lv[q] = p[(st & 1) ? targets[(long)b * Lmax + (st >> 1)] : 0];
    }
    float m = NEG_INF;
    const float2* pr = part + row * groups;
    float2 mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gidx = lane + 64 * k;
        mine[k] = make_float2(NEG_INF, 0.f);
        if (gidx < groups) mine[k] = pr[gidx];
        m = fmaxf(m, mine[k].x);
    }
    for (int gidx = lane + 256; gidx < groups; gidx += 64) m = fmaxf(m, pr[gidx].x);
    m = wave_max(m);
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (mine[k].y > 0.f) ssum += mine[k].y * __expf(mine[k].x - m);
    for (int gidx = lane + 256; gidx < groups; gidx += 64) { const float2 q2 = pr[gidx]; if (q2.y > 0.f) ssum += q2.y * __expf(q2.x - m); }
    ssum = wave_sum(ssum);
    const float l2s = __builtin_amdgcn_logf(ssum);
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int st = lane + 64 * q;
        if (st < S) lp_out[row * Sp + st] = (lv[q] - m) * 1.4426950408889634f - l2s;
    }
    if (lane == 0) rowstat[row] = make_float2(m, 1.f / ssum);
}

// ------------------------------------------------------------------ k2 ------
// The recursion is one dependent chain per frame on a lone wave, so its cost is the chain's length.  Kept short by:
//  * base-2 logs throughout (lp, alpha, beta, ll are log2 values in the workspace): exp / log are the bare v_exp_f32 /
//    v_log_f32, no scaling multiplies and none of logf's denormal handling (the argument of the log is in [1, 3]);
//  * a finite "log 0" (CTC_NEG) instead of -inf: max - max is 0 and never NaN, so there is no guard in the chain;
//    CTC_NEG + anything the recursion adds stays CTC_NEG in fp32 (ulp(1e30) = 7.6e22);
//  * neighbours through DPP wave shifts (one VALU move) instead of ds_bpermute round trips;
//  * the direction is a template parameter of the recursion, not a per-step branch.
#define CTC_NEG (-1.0e30f)
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453f

__device__ __forceinline__ float ctc_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float ctc_log2(float x) { return __builtin_amdgcn_logf(x); }
// lane i <- lane i-1 (lane 0 <- CTC_NEG) / lane i <- lane i+1 (lane 63 <- CTC_NEG)
__device__ __forceinline__ float wave_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, CTC_NEG), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, CTC_NEG), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// log2(2^a + 2^b + 2^c) + add
__device__ __forceinline__ float lse3_add(float a, float b, float c, float add) {
    const float m = fmaxf(a, fmaxf(b, c));
    return ctc_log2(ctc_exp2(a - m) + ctc_exp2(b - m) + ctc_exp2(c - m)) + (m + add);
}

// One direction of the recursion for utterance rows [base, base + Tb): NS states per lane; CH frames of lp are fetched
// at a time, one chunk ahead.  Returns this lane's final values in a[].
template <bool FWD, int NS, int CH, bool FULL>
__device__ __forceinline__ void ctc_chunk(int c0, int Tb, const long (&ostr)[NS], float (&a)[NS], const float (&cap)[NS],
                                          const float (&cur)[CH][NS], float* (&op)[NS]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (FULL || c0 + i < Tb) {       // wave-uniform; whole chunks carry no test
            // neighbour lane's nearest / second nearest state
            const float p1 = FWD ? wave_shr1(a[NS - 1]) : wave_shl1(a[0]);
            const float p2 = FWD ? (NS >= 2 ? wave_shr1(a[NS >= 2 ? NS - 2 : 0]) : wave_shr1(p1))
                                 : (NS >= 2 ? wave_shl1(a[NS >= 2 ? 1 : 0]) : wave_shl1(p1));
            float na[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                float n1, n2;
                if (FWD) {
                    n1 = (j >= 1) ? a[j >= 1 ? j - 1 : 0] : p1;
                    n2 = (j >= 2) ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
                } else {
                    n1 = (j + 1 < NS) ? a[j + 1 < NS ? j + 1 : 0] : p1;
                    n2 = (j + 2 < NS) ? a[j + 2 < NS ? j + 2 : 0] : (j + 1 < NS ? p1 : p2);
                }
                na[j] = lse3_add(a[j], n1, fminf(n2, cap[j]), cur[i][j]);
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                a[j] = na[j];
                op[j] += ostr[j];
                *op[j] = a[j];
            }
        }
    }
}

// One direction of the recursion for utterance rows [base, base + Tb): NS states per lane; CH frames of lp are fetched
// at a time, one chunk ahead.  Returns this lane's final values in a[].  dump: one float any lane may scribble on.
// k0, k1: the steps this call runs, [k0, k1) of the utterance's Tb (step k works on frame FWD ? k : Tb-1-k).  k0 = 0 starts
// the recursion (first column); k0 > 0 resumes it from the values step k0 - 1 left in out_rows (an earlier launch).
template <bool FWD, int NS, int CH>
__device__ __forceinline__ void ctc_recurse(int lane, int Tb, int S, int Sp, const int* __restrict__ tg, const float* __restrict__ lp_rows,
                                            float* __restrict__ out_rows, float* __restrict__ dump, float (&a)[NS], int k0, int k1) {
    const int s0 = lane * NS;
    float cap[NS];   // upper bound on the s-2 (alpha) / s+2 (beta) term: that transition exists only between different labels
    bool valid[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int s = s0 + j;
        valid[j] = s < S;
        bool skip = false;
        if (s < S && (s & 1)) {
            if (FWD) { if (s >= 2) skip = tg[s >> 1] != tg[(s - 2) >> 1]; }
            else { if (s + 2 < S) skip = tg[s >> 1] != tg[(s + 2) >> 1]; }
        }
        cap[j] = skip ? 3.0e38f : CTC_NEG;
    }
    // step k of the recursion works on frame  FWD ? k : Tb-1-k
    const long tstride = FWD ? (long)Sp : -(long)Sp;
    const float* lp0 = lp_rows + (long)(FWD ? 0 : Tb - 1) * Sp + s0;      // frame of step 0, this lane's first state
    // states past S store to the dump word with stride 0: the stores of a step need no predicate
    const int kfirst = k0 > 0 ? k0 - 1 : 0;              // the step whose values a[] holds before the loop
    float* op[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) op[j] = valid[j] ? out_rows + (long)(FWD ? 0 : Tb - 1) * Sp + kfirst * tstride + s0 + j : dump;
    long ostr[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) ostr[j] = valid[j] ? tstride : 0;
    if (k0 == 0) {
        // ---- first column
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int s = s0 + j;
            const bool start = FWD ? (s <= 1) : (s >= S - 2);
            a[j] = CTC_NEG;
            if (valid[j] && start) a[j] = lp0[j];
            *op[j] = a[j];
        }
    } else {
        // ---- resume: the previous launch's last column
#pragma unroll
        for (int j = 0; j < NS; ++j) a[j] = valid[j] ? *op[j] : CTC_NEG;
    }
    const int kb = kfirst + 1;                             // first step of the loop
    // ---- recursion, steps 1 .. Tb-1 in chunks of CH.  States past S carry lp = 0: they stay CTC_NEG (beta's flow is
    // from high s to low s, so they must), and their values go to the dump word.
    float cur[CH][NS], nxt[CH][NS];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            cur[i][j] = 0.f;
            if (kb + i < k1 && valid[j]) cur[i][j] = lp0[(kb + i) * tstride + j];
        }
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) asm volatile("" : "+v"(cur[i][j]));
    for (int c0 = kb; c0 < k1; c0 += CH) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                nxt[i][j] = 0.f;
                if (c0 + CH + i < k1 && valid[j]) nxt[i][j] = lp0[(c0 + CH + i) * tstride + j];
            }
        if (c0 + CH <= k1) ctc_chunk<FWD, NS, CH, true>(c0, k1, ostr, a, cap, cur, op);
        else ctc_chunk<FWD, NS, CH, false>(c0, k1, ostr, a, cap, cur, op);
        // the next chunk becomes current HERE: pinning the values makes the compiler wait for the prefetch once per
        // chunk.  Left to itself it waits lazily at each step's first use of a prefetched register, and with the steps'
        // stores in between that wait is vmcnt(0): every step then sits out its own store's acknowledgement
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                cur[i][j] = nxt[i][j];
                asm volatile("" : "+v"(cur[i][j]));
            }
    }
}

// One block per utterance: wave 0 runs alpha, wave 1 beta, and wave 2 meanwhile links the states that share a class (a
// label occurring more than once in the target), once per utterance: chain[b][s] for odd s = (next state with the same
// label, + 1; 0 = none) | (first of its label ? 1 << 16 : 0).  k3 follows these chains per frame instead of comparing
// labels (that comparison was O(S^2) per frame).
template <int NS, int CH>
__device__ __forceinline__ void ctc_alphabeta_block(int b, float* fin, int T, const int* __restrict__ hlens, const int* __restrict__ targets,
                                                    int Lmax, const int* __restrict__ tlens, int Sp,
                                                    const float* __restrict__ lp, float* __restrict__ alpha,
                                                    float* __restrict__ beta, float* __restrict__ ll_out,
                                                    float* __restrict__ nll_out, int* __restrict__ chain,
                                                    float* __restrict__ dump) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Tb = min(hlens[b], T);
    const int S = 2 * min(tlens[b], Lmax) + 1;
    const int* tg = targets + (long)b * Lmax;
    const long base = (long)b * T * Sp;
    if (threadIdx.x < 2) fin[threadIdx.x] = CTC_NEG;
    __syncthreads();
    if (wv == 2) {
        for (int s = 1 + 2 * lane; s < S; s += 128) {
            const int c = tg[s >> 1];
            int head = 1, nx = 0;
            for (int q = 1; q < s; q += 2) if (tg[q >> 1] == c) { head = 0; break; }
            for (int q = s + 2; q < S; q += 2) if (tg[q >> 1] == c) { nx = q + 1; break; }
            chain[(long)b * Sp + s] = nx | (head << 16);
        }
    } else if (Tb > 0 && wv < 2) {       // (a fourth wave - the mixed launch's 256-thread blocks - only keeps the barriers company)
        float a[NS];
        if (wv == 0) {
            ctc_recurse<true, NS, CH>(lane, Tb, S, Sp, tg, lp + base, alpha + base, dump, a, 0, Tb);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (lane * NS + j == S - 1) fin[0] = a[j];
                if (lane * NS + j == S - 2) fin[1] = a[j];
            }
        } else {
            ctc_recurse<false, NS, CH>(lane, Tb, S, Sp, tg, lp + base, beta + base, dump, a, 0, Tb);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // log2-likelihood; anything that still carries the CTC_NEG scale is an infeasible alignment:
        // zero_infinity makes it contribute 0 loss and 0 gradient
        float ll = NEG_INF;
        if (Tb > 0) {
            const float m = fmaxf(fin[0], fin[1]);
            const float v = m + ctc_log2(ctc_exp2(fin[0] - m) + ctc_exp2(fin[1] - m));
            if (v > 0.5f * CTC_NEG) ll = v;
        }
        ll_out[b] = ll;                                        // base 2, for k3
        nll_out[b] = (ll == NEG_INF) ? 0.f : -ll * CTC_LN2;
    }
}
template <int NS, int CH>
__global__ __launch_bounds__(192) void ctc_alphabeta_kernel(int T, const int* __restrict__ hlens, const int* __restrict__ targets,
                                                            int Lmax, const int* __restrict__ tlens, int Sp,
                                                            const float* __restrict__ lp, float* __restrict__ alpha,
                                                            float* __restrict__ beta, float* __restrict__ ll_out,
                                                            float* __restrict__ nll_out, int* __restrict__ chain,
                                                            float* __restrict__ dump) {
    __shared__ float fin[2];
    ctc_alphabeta_block<NS, CH>(blockIdx.x, fin, T, hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll_out, nll_out, chain, dump);
}
// overlapped form, launch 2: blocks 0 .. B-1 = the recursion of one utterance each (waves 0 - 2 of four), the rest = dense rows
template <int NV4, int NS, int CH>
__global__ __launch_bounds__(256) void ctc_chain_dense_kernel(int B, const float* logits, long ldv, int T, int V, const int* __restrict__ hlens,
                                                              const int* __restrict__ targets, int Lmax, const int* __restrict__ tlens,
                                                              int Sp, const float* __restrict__ lp, float* __restrict__ alpha,
                                                              float* __restrict__ beta, float* __restrict__ ll_out, float* __restrict__ nll_out,
                                                              int* __restrict__ chain, float* __restrict__ dump, float scale,
                                                              const float* __restrict__ utt_weight, const float2* __restrict__ rowstat,
                                                              float* dlogits, int t0, int Tc) {
    __shared__ float fin[2];
    __shared__ unsigned maskw[4][NV4 * 8];
    if ((int)blockIdx.x < B) {
        ctc_alphabeta_block<NS, CH>(blockIdx.x, fin, T, hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll_out, nll_out, chain, dump);
        return;
    }
    const int wv = threadIdx.x >> 6;
    ctc_dense_row<NV4>(logits, ldv, (long)B * Tc, T, V, hlens, targets, Lmax, tlens, scale, utt_weight, rowstat, dlogits, t0, Tc,
                       ((long)blockIdx.x - B) * 4 + wv, maskw[wv]);
}

// ---- the pipelined form (oe_ctc_loss_fused with T >= CTC_PIPE_MIN_T): the recursion in time chunks, one direction per launch.
// One wave per utterance runs its direction over the frames [t0, t1) of the chunk: alpha resumes from the column the
// previous chunk's launch left at frame t0 - 1, beta from frame t1; the launch that reaches the utterance's last frame
// (alpha) also publishes the log-likelihood.  The host queues alpha chunks 0, 1, .. on one stream and beta chunks NC-1, NC-2,
// .. on another, each behind the `rows` launch that produced its frames (ordinary stream events), so both chains run under
// the remaining `rows` traffic instead of after it.
template <bool FWD, int NS, int CH>
__global__ __launch_bounds__(64) void ctc_dir_kernel(int T, int t0, int t1, const int* __restrict__ hlens, const int* __restrict__ targets,
                                                     int Lmax, const int* __restrict__ tlens, int Sp, const float* __restrict__ lp,
                                                     float* __restrict__ state, float* __restrict__ ll_out, float* __restrict__ nll_out,
                                                     float* __restrict__ dump) {
    __shared__ float fin[2];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Tb = min(hlens[b], T);
    const int S = 2 * min(tlens[b], Lmax) + 1;
    const int* tg = targets + (long)b * Lmax;
    const long base = (long)b * T * Sp;
    if (FWD && Tb == 0 && t0 == 0 && lane == 0) { ll_out[b] = NEG_INF; nll_out[b] = 0.f; }
    const int f1 = min(t1, Tb);                       // frames [t0, f1) of this utterance lie in the chunk
    if (t0 >= f1) return;
    const int k0 = FWD ? t0 : Tb - f1, k1 = FWD ? f1 : Tb - t0;
    float a[NS];
    ctc_recurse<FWD, NS, CH>(lane, Tb, S, Sp, tg, lp + base, state + base, dump, a, k0, k1);
    if (FWD && f1 == Tb) {                            // the last frame: log2-likelihood (see ctc_alphabeta_kernel)
        if (lane < 2) fin[lane] = CTC_NEG;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (lane * NS + j == S - 1) fin[0] = a[j];
            if (lane * NS + j == S - 2) fin[1] = a[j];
        }
        __syncthreads();
        if (lane == 0) {
            const float m = fmaxf(fin[0], fin[1]);
            const float v = m + ctc_log2(ctc_exp2(fin[0] - m) + ctc_exp2(fin[1] - m));
            const float ll = (v > 0.5f * CTC_NEG) ? v : NEG_INF;
            ll_out[b] = ll;
            nll_out[b] = (ll == NEG_INF) ? 0.f : -ll * CTC_LN2;
        }
    }
}
// chain[b][s] for odd s (see ctc_alphabeta_kernel): one wave per utterance
__global__ __launch_bounds__(64) void ctc_chain_kernel(const int* __restrict__ targets, int Lmax, const int* __restrict__ tlens, int Sp,
                                                       int* __restrict__ chain) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int S = 2 * min(tlens[b], Lmax) + 1;
    const int* tg = targets + (long)b * Lmax;
    for (int s = 1 + 2 * lane; s < S; s += 128) {
        const int c = tg[s >> 1];
        int head = 1, nx = 0;
        for (int q = 1; q < s; q += 2) if (tg[q >> 1] == c) { head = 0; break; }
        for (int q = s + 2; q < S; q += 2) if (tg[q >> 1] == c) { nx = q + 1; break; }
        chain[(long)b * Sp + s] = nx | (head << 16);
    }
}

// ------------------------------------------------------------------ k3 ------
// One wave per frame (four per block).  k1 has written softmax * scale everywhere; the classes of the target get their
// occupation term here, and frames of an infeasible utterance (ll = -inf) are zeroed.  Block 0 also sums the loss.
// Sums run in a fixed order (blank: per-lane partials + shuffle tree; a label: along its chain): deterministic.
// ZERO_INF: rows of infeasible utterances are zeroed here (the three-launch form; the overlapped form leaves that to
// ctc_final_kernel, because its dense blocks write those rows at the same time)
template <bool ZERO_INF>
__device__ __forceinline__ void ctc_labels_block(float* sh, long blk, long rows, int B, int T, int V, long ldv, const int* __restrict__ hlens,
                                                 const int* __restrict__ targets, int Lmax, const int* __restrict__ tlens,
                                                 int Sp, const float* __restrict__ lp, const float* __restrict__ alpha,
                                                 const float* __restrict__ beta, const float* __restrict__ ll_in,
                                                 const int* __restrict__ chain, const float* __restrict__ nll,
                                                 float scale, const float* __restrict__ utt_weight,
                                                 float* __restrict__ dlogits, float* __restrict__ loss_sum, int t0, int Tc) {
    // frames t0 .. t0 + Tc - 1 of every utterance (`rows` = B * Tc; the whole batch: t0 = 0, Tc = T); sh: per wave gam[Sp]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (loss_sum && blk == 0 && wv == 0) {
        float s = 0.f;
        for (int i = lane; i < B; i += 64) s += utt_weight ? nll[i] * utt_weight[i] : nll[i];
        s = wave_sum(s);
        if (lane == 0) loss_sum[0] = s;
    }
    const long rloc = blk * 4 + wv;
    const bool inrange = rloc < rows;
    const int b = inrange ? (int)(rloc / Tc) : 0, t = inrange ? t0 + (int)(rloc % Tc) : 0;
    const long row = (long)b * T + t;
    const bool frame = inrange && t < hlens[b];
    const float ll = ll_in[b];
    float* g = dlogits + row * ldv;
    if (ZERO_INF && frame && ll == NEG_INF) {
        for (int i = lane; i < V; i += 64) g[i] = 0.f;
    }
    const bool live = frame && ll != NEG_INF;
    if (utt_weight) scale *= utt_weight[b];
    const int S = live ? 2 * min(tlens[b], Lmax) + 1 : 0;
    float* gam = sh + (size_t)wv * Sp;
    const float* lpr = lp + row * Sp;
    float blank = 0.f, own[CTC_MAXQ];
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int s = lane + 64 * q;
        own[q] = 0.f;
        if (s < S) {
            const long o = row * Sp + s;
            own[q] = lpr[s];
            const float gm = __builtin_amdgcn_exp2f(alpha[o] + beta[o] - own[q] - ll);
            gam[s] = gm;
            if (!(s & 1)) blank += gm;
        }
    }
    blank = wave_sum(blank);
    __syncthreads();
    if (live && lane == 0) g[0] = (__builtin_amdgcn_exp2f(own[0]) - blank) * scale;
    const int* ch = chain + (long)b * Sp;
#pragma unroll
    for (int q = 0; q < CTC_MAXQ; ++q) {
        const int s = lane + 64 * q;
        if (s < S && (s & 1)) {
            int info = ch[s];
            if (info >> 16) {
                float occ = gam[s];
                for (int nx = (info & 0xffff) - 1; nx >= 0; nx = (ch[nx] & 0xffff) - 1) occ += gam[nx];
                g[targets[(long)b * Lmax + (s >> 1)]] = (__builtin_amdgcn_exp2f(own[q]) - occ) * scale;
            }
        }
    }
}
__global__ __launch_bounds__(256) void ctc_labels_kernel(long rows, int B, int T, int V, long ldv, const int* __restrict__ hlens,
                                                          const int* __restrict__ targets, int Lmax, const int* __restrict__ tlens,
                                                          int Sp, const float* __restrict__ lp, const float* __restrict__ alpha,
                                                          const float* __restrict__ beta, const float* __restrict__ ll_in,
                                                          const int* __restrict__ chain, const float* __restrict__ nll,
                                                          float scale, const float* __restrict__ utt_weight,
                                                          float* __restrict__ dlogits, float* __restrict__ loss_sum, int t0, int Tc) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // per wave: gam[Sp]
    ctc_labels_block<true>(sh, blockIdx.x, rows, B, T, V, ldv, hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll_in, chain, nll, scale,
                           utt_weight, dlogits, loss_sum, t0, Tc);
}
// overlapped form, launch 3: blocks 0 .. nlab-1 = the label fix-up of four frames each (all frames), the rest = dense rows
template <int NV4>
__global__ __launch_bounds__(256) void ctc_labels_dense_kernel(long nlab, const float* logits, int B, int T, int V, long ldv,
                                                               const int* __restrict__ hlens, const int* __restrict__ targets, int Lmax,
                                                               const int* __restrict__ tlens, int Sp, const float* __restrict__ lp,
                                                               const float* __restrict__ alpha, const float* __restrict__ beta,
                                                               const float* __restrict__ ll_in, const int* __restrict__ chain,
                                                               const float* __restrict__ nll, float scale, const float* __restrict__ utt_weight,
                                                               const float2* __restrict__ rowstat, float* dlogits, int t0, int Tc) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // per wave: gam[Sp] (label blocks)
    __shared__ unsigned maskw[4][NV4 * 8];
    if ((long)blockIdx.x < nlab) {
        ctc_labels_block<false>(sh, blockIdx.x, (long)B * T, B, T, V, ldv, hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll_in, chain, nll,
                                scale, utt_weight, dlogits, nullptr, 0, T);
        return;
    }
    const int wv = threadIdx.x >> 6;
    ctc_dense_row<NV4>(logits, ldv, (long)B * Tc, T, V, hlens, targets, Lmax, tlens, scale, utt_weight, rowstat, dlogits, t0, Tc,
                       ((long)blockIdx.x - nlab) * 4 + wv, maskw[wv]);
}
// overlapped form, last launch: one block per utterance zeroes the rows of an infeasible utterance (zero_infinity; the live frames
// only - padded frames are zero already); block 0 also sums the loss
__global__ __launch_bounds__(256) void ctc_final_kernel(int B, int T, int V, long ldv, const int* __restrict__ hlens, const float* __restrict__ ll_in,
                                                         const float* __restrict__ nll, const float* __restrict__ utt_weight,
                                                         float* __restrict__ dlogits, float* __restrict__ loss_sum) {
    const int b = blockIdx.x;
    if (loss_sum && b == 0 && threadIdx.x < 64) {
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 64) s += utt_weight ? nll[i] * utt_weight[i] : nll[i];
        s = wave_sum(s);
        if (threadIdx.x == 0) loss_sum[0] = s;
    }
    if (ll_in[b] != NEG_INF) return;
    const int Tb = min(hlens[b], T);
    float* g = dlogits + (long)b * T * ldv;
    for (int t = 0; t < Tb; ++t)
        for (int i = threadIdx.x; i < V; i += 256) g[(long)t * ldv + i] = 0.f;
}

__global__ void ctc_sum_kernel(const float* __restrict__ nll, const float* __restrict__ utt_weight, int B, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += utt_weight ? nll[i] * utt_weight[i] : nll[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

// Streams and events of the pipelined form, one set per device, made on first use (a first use is never inside a graph
// capture: the engine runs an eager step before it captures).  Re-recording an event does not disturb waits already queued.
#define CTC_MAX_CHUNKS 8
#define OE_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { oe_set_error("oe_ctc_loss_fused: %s", hipGetErrorString(e_)); return (int)e_; } } while (0)
struct CtcPipe {
    hipStream_t sa = nullptr, sb = nullptr;
    hipEvent_t rows_done[CTC_MAX_CHUNKS], beta_done[CTC_MAX_CHUNKS], alpha_done;
    bool ok = false;
};
static CtcPipe* ctc_pipe() {
    static CtcPipe pipes[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    CtcPipe& p = pipes[dev];
    if (!p.ok) {
        if (hipStreamCreateWithFlags(&p.sa, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&p.sb, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < CTC_MAX_CHUNKS; ++i) {
            if (hipEventCreateWithFlags(&p.rows_done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&p.beta_done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&p.alpha_done, hipEventDisableTiming) != hipSuccess) return nullptr;
        p.ok = true;
    }
    return &p;
}

// OE_CTC_PIPE: 0 = rows, alpha/beta, labels one after the other (default); 1 / 2 = chunk pipeline on two streams where the batch
// is large enough / always; 3 / 4 = the overlapped form (mixed launches, ctc_dense_row) for large batches / always;
// OE_CTC_CHUNKS: time chunks of the chunk pipeline.  Both alternatives are bit-for-bit / tolerance-equal (tests run them) and both
// measured SLOWER than the default on MI355X (profiles/r03_experiments.md) - overlapped form at B = 64 x 16 s: statistics pass 86 us
// (a read-only sweep of the logits runs at 3.8 TB/s where the read + write `rows` pass moves 6.0), recursion beside dense rows 100 us
// (59 alone), fix-up beside dense rows 101 us: 294 us against 208; the streaming blocks take the memory system's latency up and the
// latency-bound blocks (the recursion's prefetch, the fix-up's dependent loads) stretch by what the overlap was to hide.
// Measured on MI355X (tools/ctc_graph_bench.py, profiles/r03_experiments.md): every cross-stream edge of the pipelined form
// costs more than the chain time it hides - from a HIP graph 216 us sequential against 335 / 352 / 384 / 436 us with 2 / 3 /
// 4 / 6 chunks at the north-star shape (eager launches: 209 against 310 with 4) - so it is not the default.
static int ctc_pipe_mode = getenv("OE_CTC_PIPE") ? atoi(getenv("OE_CTC_PIPE")) : 0;
static int ctc_pipe_chunks = getenv("OE_CTC_CHUNKS") ? atoi(getenv("OE_CTC_CHUNKS")) : 4;
extern "C" int oe_ctc_config(int pipe_mode, int chunks) {
    if (pipe_mode >= 0) ctc_pipe_mode = pipe_mode;
    if (chunks >= 2) ctc_pipe_chunks = chunks;
    return 0;
}

extern "C" size_t oe_ctc_workspace_floats(int B, int T, int Lmax) {
    size_t Sp = 2 * (size_t)Lmax + 1;
    return (size_t)B + 3 * (size_t)B * T * Sp + (size_t)B * Sp + 16 + 2 * (size_t)B * T + 2;      // .. + row statistics of the overlapped form
}

static int ctc_loss_impl(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                         int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll,
                         float* loss_sum, float* dlogits, float* workspace, const float* row_stats, int stats_groups, void* stream);
extern "C" int oe_ctc_loss_fused(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                                 int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll,
                                 float* loss_sum, float* dlogits, float* workspace, void* stream) {
    return ctc_loss_impl(logits, ldv, B, T, V, hlens, targets, Lmax, tlens, grad_scale, utt_weight, nll, loss_sum, dlogits, workspace, nullptr, 0, stream);
}
extern "C" int oe_ctc_loss_fused_stats(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                                       int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll,
                                       float* loss_sum, float* dlogits, float* workspace, const float* row_stats, int stats_groups, void* stream) {
    OE_REQUIRE(!row_stats || (stats_groups == (V + 31) / 32 && (((uintptr_t)row_stats) & 7) == 0),
               "oe_ctc_loss_fused_stats: row_stats must hold ceil(V / 32) = %d (max, sum) pairs per row, 8-byte aligned (got %d groups)", (V + 31) / 32, stats_groups);
    return ctc_loss_impl(logits, ldv, B, T, V, hlens, targets, Lmax, tlens, grad_scale, utt_weight, nll, loss_sum, dlogits, workspace, row_stats, stats_groups, stream);
}

static int ctc_loss_impl(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                         int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll,
                         float* loss_sum, float* dlogits, float* workspace, const float* row_stats, int stats_groups, void* stream) {
    OE_REQUIRE(logits && hlens && tlens && nll && workspace, "oe_ctc_loss_fused: null pointer");
    OE_REQUIRE(targets || Lmax == 0, "oe_ctc_loss_fused: null targets");
    OE_REQUIRE(B > 0 && T > 0 && V > 1 && Lmax >= 0 && ldv >= V, "oe_ctc_loss_fused: bad shape B=%d T=%d V=%d Lmax=%d ldv=%ld",
               B, T, V, Lmax, ldv);
    const int Sp = 2 * Lmax + 1;
    OE_REQUIRE(Sp <= 64 * CTC_MAXQ, "oe_ctc_loss_fused: target length %d exceeds the 255-label limit of the wave recursion", Lmax);
    hipStream_t st = (hipStream_t)stream;
    float* ll = workspace;
    float* lp = ll + B;
    float* alpha = lp + (size_t)B * T * Sp;
    float* beta = alpha + (size_t)B * T * Sp;
    int* chain = reinterpret_cast<int*>(beta + (size_t)B * T * Sp);
    float* dump = reinterpret_cast<float*>(chain + (size_t)B * Sp);      // the 16 spare floats at the end
    const long rows = (long)B * T;
    // register-resident rows need 16-byte aligned rows that are padded to whole float4s
    const bool vec = (((uintptr_t)logits & 15) == 0) && (!dlogits || ((uintptr_t)dlogits & 15) == 0) && (ldv % 4 == 0) &&
                     (ldv >= (((long)V + 3) & ~3L)) && V <= 256 * 32;
    const int nv4 = !vec ? 0 : V <= 256 * 4 ? 4 : V <= 256 * 8 ? 8 : V <= 256 * 16 ? 16 : 32;
    float2* rowstat = reinterpret_cast<float2*>((reinterpret_cast<uintptr_t>(dump + 16) + 7) & ~(uintptr_t)7);
    // ---- overlapped form (see ctc_dense_row): statistics, chains + first half of the dense rows, fix-up + second half, final
    const int pipe_mode0 = ctc_pipe_mode;
    if (dlogits && nv4 > 0 && (pipe_mode0 == 4 || (pipe_mode0 == 3 && (double)rows * V >= 2.0e7))) {
        if (row_stats) {
            hipLaunchKernelGGL(ctc_lp_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, st, logits, ldv, rows, T, V, hlens, targets, Lmax, tlens, Sp,
                               reinterpret_cast<const float2*>(row_stats), stats_groups, lp, rowstat);
        } else {
#define STATS(NV4) hipLaunchKernelGGL((ctc_rows_kernel<NV4, false>), dim3(oe_cdiv(rows, 4)), dim3(256), 0, st, logits, ldv, rows, T, V, hlens, targets, \
                                      Lmax, tlens, Sp, grad_scale, utt_weight, lp, nullptr, 0, T, rowstat)
            if (nv4 == 4) STATS(4); else if (nv4 == 8) STATS(8); else if (nv4 == 16) STATS(16); else STATS(32);
#undef STATS
        }
        OE_LAUNCH_CHECK("ctc_stats");
        const int T1 = T / 2;
        const long nd1 = oe_cdiv((long)B * T1, 4), nd2 = oe_cdiv((long)B * (T - T1), 4), nlab = oe_cdiv(rows, 4);
#define K2(NV4, NS, CH) hipLaunchKernelGGL((ctc_chain_dense_kernel<NV4, NS, CH>), dim3(B + nd1), dim3(256), 0, st, B, logits, ldv, T, V, hlens, targets, Lmax, \
                                           tlens, Sp, lp, alpha, beta, ll, nll, chain, dump, grad_scale, utt_weight, rowstat, dlogits, 0, T1)
#define K2S(NV4) do { if (Sp <= 64) K2(NV4, 1, 32); else if (Sp <= 128) K2(NV4, 2, 16); else if (Sp <= 256) K2(NV4, 4, 8); else K2(NV4, 8, 4); } while (0)
        if (nv4 == 4) K2S(4); else if (nv4 == 8) K2S(8); else if (nv4 == 16) K2S(16); else K2S(32);
#undef K2S
#undef K2
        OE_LAUNCH_CHECK("ctc_chain_dense");
#define K3(NV4) hipLaunchKernelGGL((ctc_labels_dense_kernel<NV4>), dim3(nlab + nd2), dim3(256), (size_t)4 * Sp * sizeof(float), st, nlab, logits, B, T, V, ldv, \
                                   hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll, chain, nll, grad_scale, utt_weight, rowstat, dlogits, T1, T - T1)
        if (nv4 == 4) K3(4); else if (nv4 == 8) K3(8); else if (nv4 == 16) K3(16); else K3(32);
#undef K3
        OE_LAUNCH_CHECK("ctc_labels_dense");
        hipLaunchKernelGGL(ctc_final_kernel, dim3(B), dim3(256), 0, st, B, T, V, ldv, hlens, ll, nll, utt_weight, dlogits, loss_sum);
        OE_LAUNCH_CHECK("ctc_final");
        return 0;
    }
#define ROWS(NV4, WR, T0, TC)                                                                                                  \
    hipLaunchKernelGGL((ctc_rows_kernel<NV4, WR>), dim3(oe_cdiv((long)B * (TC), 4)), dim3(256), 0, st, logits, ldv, (long)B * (TC), T, V, \
                       hlens, targets, Lmax, tlens, Sp, grad_scale, utt_weight, lp, dlogits, T0, TC, nullptr)
#define ROWS_W(NV4, T0, TC) do { if (dlogits) ROWS(NV4, true, T0, TC); else ROWS(NV4, false, T0, TC); } while (0)
    auto launch_rows = [&](int t0, int tc) {
        switch (nv4) {
            case 4: ROWS_W(4, t0, tc); break;
            case 8: ROWS_W(8, t0, tc); break;
            case 16: ROWS_W(16, t0, tc); break;
            case 32: ROWS_W(32, t0, tc); break;
            default: ROWS_W(0, t0, tc); break;
        }
    };
    auto launch_labels = [&](int t0, int tc, float* lsum) {
        hipLaunchKernelGGL(ctc_labels_kernel, dim3(oe_cdiv((long)B * tc, 4)), dim3(256), (size_t)4 * Sp * sizeof(float), st, (long)B * tc, B, T, V,
                           ldv, hlens, targets, Lmax, tlens, Sp, lp, alpha, beta, ll, chain, nll, grad_scale, utt_weight, dlogits, lsum, t0, tc);
    };
    // ---- pipelined form (opt-in, see ctc_pipe_mode): the recursion in time chunks under the remaining `rows` traffic
    // (ctc_dir_kernel) - rows 115 us, the two chains 60 us each, label fix-up 30 us at the 16 s north-star batch.
    const int pipe_mode = ctc_pipe_mode, pipe_chunks = ctc_pipe_chunks;
    CtcPipe* pp = nullptr;
    const int NC = min(max(pipe_chunks, 2), CTC_MAX_CHUNKS);
    if ((pipe_mode == 1 || pipe_mode == 2) && dlogits && (pipe_mode == 2 ? T >= 2 * NC : (T >= 32 * NC && (double)rows * V >= 1.5e7))) pp = ctc_pipe();
    if (pp) {
        const int Tq = oe_cdiv(T, NC);
        auto c0 = [&](int c) { return c * Tq; };
        auto cn = [&](int c) { return min(T, (c + 1) * Tq) - c * Tq; };
        hipLaunchKernelGGL(ctc_chain_kernel, dim3(B), dim3(64), 0, st, targets, Lmax, tlens, Sp, chain);
        // rows in the order first, last, second, second-to-last, ...: alpha needs the frames from the front, beta from the back
        int lo = 0, hi = NC - 1;
        for (int i = 0; i < NC; ++i) {
            const int c = (i & 1) ? hi-- : lo++;
            if (cn(c) > 0) launch_rows(c0(c), cn(c));
            OE_HIP(hipEventRecord(pp->rows_done[c], st));
        }
        OE_LAUNCH_CHECK("ctc_rows");
#define DIR(FWD, NS, CH, STR, C)                                                                                                   \
        hipLaunchKernelGGL((ctc_dir_kernel<FWD, NS, CH>), dim3(B), dim3(64), 0, STR, T, c0(C), c0(C) + cn(C), hlens, targets, Lmax, tlens, \
                           Sp, lp, (FWD) ? alpha : beta, ll, nll, dump + ((FWD) ? 0 : 8))
#define DIRS(FWD, STR, C) do { if (Sp <= 64) DIR(FWD, 1, 32, STR, C); else if (Sp <= 128) DIR(FWD, 2, 16, STR, C);                 \
                               else if (Sp <= 256) DIR(FWD, 4, 8, STR, C); else DIR(FWD, 8, 4, STR, C); } while (0)
        for (int c = 0; c < NC; ++c) {
            OE_HIP(hipStreamWaitEvent(pp->sa, pp->rows_done[c], 0));
            if (cn(c) > 0) DIRS(true, pp->sa, c);
        }
        OE_HIP(hipEventRecord(pp->alpha_done, pp->sa));
        for (int c = NC - 1; c >= 0; --c) {
            OE_HIP(hipStreamWaitEvent(pp->sb, pp->rows_done[c], 0));
            if (cn(c) > 0) DIRS(false, pp->sb, c);
            OE_HIP(hipEventRecord(pp->beta_done[c], pp->sb));
        }
#undef DIRS
#undef DIR
        OE_LAUNCH_CHECK("ctc_dir");
        // label fix-up per chunk once both directions have passed it (the log-likelihood comes with the end of alpha)
        OE_HIP(hipStreamWaitEvent(st, pp->alpha_done, 0));
        for (int c = NC - 1; c >= 0; --c) {
            OE_HIP(hipStreamWaitEvent(st, pp->beta_done[c], 0));
            if (cn(c) > 0) launch_labels(c0(c), cn(c), c == 0 ? loss_sum : nullptr);      // chunk 0 is last: every nll is final
        }
        OE_LAUNCH_CHECK("ctc_labels");
        return 0;
    }
    launch_rows(0, T);
#undef ROWS_W
#undef ROWS
    OE_LAUNCH_CHECK("ctc_rows");
#define AB(NS, CH) hipLaunchKernelGGL((ctc_alphabeta_kernel<NS, CH>), dim3(B), dim3(192), 0, st, T, hlens, targets, Lmax, tlens, \
                                      Sp, lp, alpha, beta, ll, nll, chain, dump)
    if (Sp <= 64) AB(1, 32); else if (Sp <= 128) AB(2, 16); else if (Sp <= 256) AB(4, 8); else AB(8, 4);
#undef AB
    OE_LAUNCH_CHECK("ctc_alphabeta");
    if (dlogits) {
        launch_labels(0, T, loss_sum);
        OE_LAUNCH_CHECK("ctc_labels");
    } else if (loss_sum) {
        hipLaunchKernelGGL(ctc_sum_kernel, dim3(1), dim3(64), 0, st, nll, utt_weight, B, loss_sum);
        OE_LAUNCH_CHECK("ctc_sum");
    }
    return 0;
}

// ------------------------------------------------------------- greedy -------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ldv, long rows, int V,
                                                           int* __restrict__ best) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * ldv;
    float bv = NEG_INF;
    int bi = 0x7fffffff;
    for (int i = lane; i < V; i += 64) {
        float v = p[i];
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = i; }   // strictly greater: lowest index kept inside a lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) best[row] = bi;
}

__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* __restrict__ best, int T, const int* __restrict__ hlens,
                                                           int eos, int* __restrict__ out_tokens, int* __restrict__ out_lens) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int hl = hlens[b];
    const int* row = best + (long)b * T;
    int* out = out_tokens + (long)b * T;
    int n = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int cur = 0, prev = -1;
        if (t < T) {
            cur = (t < hl) ? row[t] : eos;
            if (t > 0) prev = (t - 1 < hl) ? row[t - 1] : eos;
        }
        const bool keep = (t < T) && (cur != 0) && (cur != prev);
        const unsigned long long mask = __ballot(keep);
        const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) out[pos] = cur;
        n += __popcll(mask);
    }
    for (int t = n + lane; t < T; t += 64) out[t] = -1;
    if (lane == 0) out_lens[b] = n;
}

extern "C" int oe_ctc_greedy(const float* logits, long ldv, int B, int T, int V, const int* hlens, int eos,
                             int* frame_best, int* out_tokens, int* out_lens, void* stream) {
    OE_REQUIRE(logits && hlens && frame_best && out_tokens && out_lens, "oe_ctc_greedy: null pointer");
    OE_REQUIRE(B > 0 && T > 0 && V > 0 && ldv >= V, "oe_ctc_greedy: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, st, logits, ldv, rows, V, frame_best);
    OE_LAUNCH_CHECK("argmax_rows");
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, st, frame_best, T, hlens, eos, out_tokens, out_lens);
    OE_LAUNCH_CHECK("ctc_collapse");
    return 0;
}
