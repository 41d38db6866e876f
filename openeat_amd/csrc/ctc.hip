// CTC head on device: log-sum-exp per frame, alpha/beta recursion with wave
// shuffles, gradient w.r.t. the logits, and greedy search.
//
// Replaces  /root/reference/openeat/modules/ctc.py:38-45  (log_softmax ->
// torch.nn.CTCLoss(sum, zero_infinity) -> /B and its autograd) and
// /root/reference/openeat/models/asr_model.py:318-325 (greedy).
//
// Pass structure (HBM-bound; B*T*V*4 bytes = the logits):
//   k1 rowstats : read logits once           -> lse[b,t], lp[b,t,s] = logp at the 2L+1 states
//   k2 alphabeta: one block per utterance, wave 0 runs alpha, wave 1 runs beta
//                 concurrently; neighbours s-1,s-2 come from lane shuffles
//   k3 grad     : read logits once, write dlogits once:
//                 (softmax - sum_{s:l'_s=c} exp(alpha+beta-lp-ll)) * scale
// Algorithmic bytes: 2*B*T*V*4 (+ 3 small state arrays); k1's read is the one
// pass above that minimum (it disappears once the LSE is fused into the
// producing GEMM's epilogue).
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define NEG_INF (-INFINITY)

// ------------------------------------------------------------------ k1 ------
__global__ __launch_bounds__(256) void ctc_rowstats_kernel(const float* __restrict__ logits, long ldv, int B, int T, int V,
                                                            const int* __restrict__ hlens, const int* __restrict__ targets,
                                                            int Lmax, const int* __restrict__ tlens, int Sp,
                                                            float* __restrict__ lse_out, float* __restrict__ lp_out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), t = (int)(row % T);
    if (t >= hlens[b]) return;
    const float* p = logits + row * ldv;
    float m = NEG_INF, s = 0.f;
    const bool vec = ((((uintptr_t)p) & 15) == 0);
    int done = 0;
    if (vec) {
        const int nv = V >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = lane; i < nv; i += 64) {
            float4 v = p4[i];
            float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            float mn = fmaxf(m, mx);
            s = s * __expf(m - mn) + __expf(v.x - mn) + __expf(v.y - mn) + __expf(v.z - mn) + __expf(v.w - mn);
            m = mn;
        }
        done = nv << 2;
    }
    for (int i = done + lane; i < V; i += 64) {
        float x = p[i];
        float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    if (m == NEG_INF) s = 0.f;
    wave_lse(m, s);
    const float lse = m + __logf(s);
    if (lane == 0) lse_out[row] = lse;
    const int L = min(tlens[b], Lmax);
    const int S = 2 * L + 1;
    for (int st = lane; st < S; st += 64) {
        const int lab = (st & 1) ? targets[(long)b * Lmax + (st >> 1)] : 0;
        lp_out[row * Sp + st] = p[lab] - lse;
    }
}

// ------------------------------------------------------------------ k2 ------
__device__ __forceinline__ float lse3(float a, float b, float c) {
    float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

template <int NS>
__global__ __launch_bounds__(128) void ctc_alphabeta_kernel(int T, const int* __restrict__ hlens, const int* __restrict__ targets,
                                                            int Lmax, const int* __restrict__ tlens, int Sp,
                                                            const float* __restrict__ lp, float* __restrict__ alpha,
                                                            float* __restrict__ beta, float* __restrict__ ll_out,
                                                            float* __restrict__ nll_out) {
    __shared__ float fin[2];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const bool fwd = (threadIdx.x < 64);           // wave 0: alpha, wave 1: beta (wave-uniform)
    const int Tb = min(hlens[b], T);
    const int L = min(tlens[b], Lmax);
    const int S = 2 * L + 1;
    const int s0 = lane * NS;
    if (threadIdx.x < 2) fin[threadIdx.x] = NEG_INF;

    int lab[NS];
    bool skip[NS];   // may take the s-2 (alpha) / s+2 (beta) transition
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int s = s0 + j;
        lab[j] = (s < S && (s & 1)) ? targets[(long)b * Lmax + (s >> 1)] : 0;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int s = s0 + j;
        skip[j] = false;
        if (s < S && (s & 1)) {
            if (fwd) { if (s >= 2) skip[j] = lab[j] != targets[(long)b * Lmax + ((s - 2) >> 1)]; }
            else { if (s + 2 < S) skip[j] = lab[j] != targets[(long)b * Lmax + ((s + 2) >> 1)]; }
        }
    }
    __syncthreads();

    float* out = fwd ? alpha : beta;
    float a[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) a[j] = NEG_INF;

    if (Tb > 0) {
        const long base = (long)b * T;
        // ---- first column
        {
            const int t = fwd ? 0 : Tb - 1;
            const float* lpr = lp + (base + t) * Sp;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int s = s0 + j;
                const bool start = fwd ? (s <= 1) : (s >= S - 2);
                if (s < S && start) a[j] = lpr[s];
                if (s < S) out[(base + t) * Sp + s] = a[j];
            }
        }
        // ---- recursion; next row of lp is fetched one step ahead
        float cur[NS];
        if (Tb > 1) {
            const int t = fwd ? 1 : Tb - 2;
#pragma unroll
            for (int j = 0; j < NS; ++j) cur[j] = (s0 + j < S) ? lp[(base + t) * Sp + s0 + j] : NEG_INF;
        }
        for (int step = 1; step < Tb; ++step) {
            const int t = fwd ? step : Tb - 1 - step;
            float nxt[NS];
            if (step + 1 < Tb) {
                const int tn = fwd ? t + 1 : t - 1;
#pragma unroll
                for (int j = 0; j < NS; ++j) nxt[j] = (s0 + j < S) ? lp[(base + tn) * Sp + s0 + j] : NEG_INF;
            }
            float p1, p2;  // neighbour lane's nearest / second nearest state
            if (fwd) {
                p1 = __shfl_up(a[NS - 1], 1, 64);
                p2 = (NS >= 2) ? __shfl_up(a[NS >= 2 ? NS - 2 : 0], 1, 64) : __shfl_up(a[0], 2, 64);
                if (lane == 0) { p1 = NEG_INF; p2 = NEG_INF; }
                if (NS == 1 && lane == 1) p2 = NEG_INF;
            } else {
                p1 = __shfl_down(a[0], 1, 64);
                p2 = (NS >= 2) ? __shfl_down(a[NS >= 2 ? 1 : 0], 1, 64) : __shfl_down(a[0], 2, 64);
                if (lane == 63) { p1 = NEG_INF; p2 = NEG_INF; }
                if (NS == 1 && lane == 62) p2 = NEG_INF;
            }
            float na[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                float n1, n2;
                if (fwd) {
                    n1 = (j >= 1) ? a[j >= 1 ? j - 1 : 0] : p1;
                    n2 = (j >= 2) ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
                } else {
                    n1 = (j + 1 < NS) ? a[j + 1 < NS ? j + 1 : 0] : p1;
                    n2 = (j + 2 < NS) ? a[j + 2 < NS ? j + 2 : 0] : (j + 1 < NS ? p1 : p2);
                }
                if (!skip[j]) n2 = NEG_INF;
                na[j] = lse3(a[j], n1, n2) + cur[j];
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                a[j] = (s0 + j < S) ? na[j] : NEG_INF;
                if (s0 + j < S) out[(base + t) * Sp + s0 + j] = a[j];
                cur[j] = nxt[j];
            }
        }
        if (fwd) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (s0 + j == S - 1) fin[0] = a[j];
                if (S >= 2 && s0 + j == S - 2) fin[1] = a[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ll = NEG_INF;
        if (Tb > 0) {
            float m = fmaxf(fin[0], fin[1]);
            if (m != NEG_INF) ll = m + __logf(__expf(fin[0] - m) + __expf(fin[1] - m));
        }
        // zero_infinity: an infeasible alignment contributes 0 loss and 0 gradient
        ll_out[b] = ll;
        nll_out[b] = (ll == NEG_INF) ? 0.f : -ll;
    }
}

// ------------------------------------------------------------------ k3 ------
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, long ldv, int T, int V,
                                                        const int* __restrict__ hlens, const int* __restrict__ targets,
                                                        int Lmax, const int* __restrict__ tlens, int Sp,
                                                        const float* __restrict__ lse, const float* __restrict__ lp,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        const float* __restrict__ ll_in, float scale,
                                                        const float* __restrict__ utt_weight, float* dlogits) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // gam[S], then lab[S] (as int)
    const long row = blockIdx.x;
    const int b = (int)(row / T), t = (int)(row % T);
    if (utt_weight) scale *= utt_weight[b];
    const float* p = logits + row * ldv;
    float* g = dlogits + row * ldv;
    const float ll = ll_in[b];
    const bool live = (t < hlens[b]) && (ll != NEG_INF);
    const bool vec = ((((uintptr_t)p) & 15) == 0) && ((((uintptr_t)g) & 15) == 0);
    const int nv = vec ? (V >> 2) : 0;
    if (!live) {
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = threadIdx.x; i < nv; i += 256) reinterpret_cast<float4*>(g)[i] = z;
        for (int i = (nv << 2) + threadIdx.x; i < V; i += 256) g[i] = 0.f;
        return;
    }
    const float l = lse[row];
    // dense part: softmax * scale
    for (int i = threadIdx.x; i < nv; i += 256) {
        float4 v = reinterpret_cast<const float4*>(p)[i];
        v.x = __expf(v.x - l) * scale; v.y = __expf(v.y - l) * scale;
        v.z = __expf(v.z - l) * scale; v.w = __expf(v.w - l) * scale;
        reinterpret_cast<float4*>(g)[i] = v;
    }
    for (int i = (nv << 2) + threadIdx.x; i < V; i += 256) g[i] = __expf(p[i] - l) * scale;

    const int L = min(tlens[b], Lmax);
    const int S = 2 * L + 1;
    float* gam = sh;
    int* labs = reinterpret_cast<int*>(sh + S);
    for (int s = threadIdx.x; s < S; s += 256) {
        const float lps = lp[row * Sp + s];
        gam[s] = __expf(alpha[row * Sp + s] + beta[row * Sp + s] - lps - ll);
        labs[s] = (s & 1) ? targets[(long)b * Lmax + (s >> 1)] : 0;
    }
    __syncthreads();   // also orders the dense stores before the per-label stores below
    for (int s = threadIdx.x; s < S; s += 256) {
        const int c = labs[s];
        bool first = true;
        // blanks sit on even s, labels on odd s: only same-parity states can share a class
        for (int q = (s & 1); q < s; q += 2) if (labs[q] == c) { first = false; break; }
        if (!first) continue;
        float occ = 0.f;
        for (int q = s; q < S; q += 2) if (labs[q] == c) occ += gam[q];   // fixed order: deterministic
        g[c] = (__expf(lp[row * Sp + s]) - occ) * scale;
    }
}

__global__ void ctc_sum_kernel(const float* __restrict__ nll, const float* __restrict__ utt_weight, int B, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += utt_weight ? nll[i] * utt_weight[i] : nll[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

extern "C" size_t oe_ctc_workspace_floats(int B, int T, int Lmax) {
    size_t Sp = 2 * (size_t)Lmax + 1;
    return (size_t)B * T + (size_t)B + 3 * (size_t)B * T * Sp + 16;
}

extern "C" int oe_ctc_loss_fused(const float* logits, long ldv, int B, int T, int V, const int* hlens, const int* targets,
                                 int Lmax, const int* tlens, float grad_scale, const float* utt_weight, float* nll,
                                 float* loss_sum, float* dlogits, float* workspace, void* stream) {
    OE_REQUIRE(logits && hlens && tlens && nll && workspace, "oe_ctc_loss_fused: null pointer");
    OE_REQUIRE(targets || Lmax == 0, "oe_ctc_loss_fused: null targets");
    OE_REQUIRE(B > 0 && T > 0 && V > 1 && Lmax >= 0 && ldv >= V, "oe_ctc_loss_fused: bad shape B=%d T=%d V=%d Lmax=%d ldv=%ld",
               B, T, V, Lmax, ldv);
    const int Sp = 2 * Lmax + 1;
    OE_REQUIRE(Sp <= 64 * 8, "oe_ctc_loss_fused: target length %d exceeds the 255-label limit of the wave recursion", Lmax);
    hipStream_t st = (hipStream_t)stream;
    float* lse = workspace;
    float* ll = lse + (size_t)B * T;
    float* lp = ll + B;
    float* alpha = lp + (size_t)B * T * Sp;
    float* beta = alpha + (size_t)B * T * Sp;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(ctc_rowstats_kernel, dim3(oe_cdiv(rows, 4)), dim3(256), 0, st, logits, ldv, B, T, V, hlens, targets,
                       Lmax, tlens, Sp, lse, lp);
    OE_LAUNCH_CHECK("ctc_rowstats");
#define AB(NS) hipLaunchKernelGGL(ctc_alphabeta_kernel<NS>, dim3(B), dim3(128), 0, st, T, hlens, targets, Lmax, tlens, Sp, \
                                  lp, alpha, beta, ll, nll)
    if (Sp <= 64) AB(1); else if (Sp <= 128) AB(2); else if (Sp <= 256) AB(4); else AB(8);
#undef AB
    OE_LAUNCH_CHECK("ctc_alphabeta");
    if (loss_sum) {
        hipLaunchKernelGGL(ctc_sum_kernel, dim3(1), dim3(64), 0, st, nll, utt_weight, B, loss_sum);
        OE_LAUNCH_CHECK("ctc_sum");
    }
    if (dlogits) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)rows), dim3(256), (size_t)Sp * 8, st, logits, ldv, T, V, hlens,
                           targets, Lmax, tlens, Sp, lse, lp, alpha, beta, ll, grad_scale, utt_weight, dlogits);
        OE_LAUNCH_CHECK("ctc_grad");
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
