// CTC prefix beam search on the device, one wavefront per utterance
// (/root/reference/openeat/models/asr_model.py:359-396: the per-frame Python dict loop).
//
// Same arithmetic and the same ordering as the reference and as the host implementation (beam_host.cpp, which stays as
// the checker): python floats = doubles, log_add = max + log(sum exp(. - max)) accumulated in the order the reference
// visits (token j of the frame's top-k, then hypothesis h), pruning = stable sort by log_add(pb, pnb) descending, i.e.
// ties keep the dict's insertion order.
//
// What makes it parallel.  In one frame every entry of next_hyps receives a bounded, known set of updates:
//   * the entry of a current prefix n ("stay"): at most ONE update of pb (the blank, if it is in the top-k) and at most
//     TWO of pnb - the repeated last token without extension (from n itself) and the extension of n's parent m by n's
//     last token (from m, only if m is also in the beam) - both at the same j, so their order is the order of n and m in
//     the beam;
//   * the entry of a new prefix n + [s]: exactly one update.
// So the wave evaluates all beam x beam (token, hypothesis) pairs at once (lane = pair), routes the few extension
// candidates that land on an existing prefix to that prefix's lane through LDS, applies each entry's updates in the
// reference's order, stamps every entry with the sequence number of its first touch (2 (j |beam| + h) + {0: the
// hypothesis' own entry, 1: its extension}), and picks the best `beam` entries by (score descending, stamp ascending) in
// `beam` rounds of a wave-wide arg-max.  Prefixes are identified by (64-bit polynomial hash of the token sequence,
// length) instead of the host version's trie - equal prefixes always meet, distinct ones collide with probability
// ~2^-64 per pair - and their tokens are recovered at the end by walking per-frame back-pointers (slot of the parent,
// token appended) kept in a workspace.
// Differences from the host version: exp / log are the device's double-precision routines (<= 1 ulp, not glibc's), so a
// score can differ in its last bits.
#include <math.h>
#include "oe_common.h"
#include "../../include/openeat_hip.h"

#define PB_MAXBEAM 16
#define PB_MAXC ((PB_MAXBEAM * PB_MAXBEAM + 63) / 64)          // (token, hypothesis) pairs per lane
#define PB_HASH_MUL 0x9E3779B97F4A7C15ull

__device__ __forceinline__ double pb_neg() { return -__builtin_huge_val(); }
// log_add (common.py:198-206): max + log(sum of exp(. - max)) in argument order.  exp(0) = 1 and exp(-inf) = 0 exactly, so the
// term of the maximum and the terms that are -inf are written down instead of computed: the same sum bit for bit, with one
// double-precision exp left in the common case instead of two or three (they were most of a frame's 15 us).
__device__ __forceinline__ double pb_term(double x, double m) {
    return x == m ? 1.0 : (x == pb_neg() ? 0.0 : exp(x - m));
}
__device__ __forceinline__ double pb_log_add2(double a, double b) {
    const double ninf = pb_neg();
    if (a == ninf && b == ninf) return ninf;
    const double m = fmax(a, b);
    return m + log(pb_term(a, m) + pb_term(b, m));
}
__device__ __forceinline__ double pb_log_add3(double a, double b, double c) {
    const double ninf = pb_neg();
    if (a == ninf && b == ninf && c == ninf) return ninf;
    const double m = fmax(a, fmax(b, c));
    return m + log(pb_term(a, m) + pb_term(b, m) + pb_term(c, m));
}
__device__ __forceinline__ double pb_shfl_xor(double v, int o) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, o, 64);
    hi = __shfl_xor(hi, o, 64);
    return __hiloint2double(hi, lo);
}

struct PbHyp {          // one entry of next_hyps
    unsigned long long key;
    double pb, pnb, score;
    int len, last, parent, tok, order;
};

__global__ __launch_bounds__(64) void ctc_prefix_beam_kernel(const float* __restrict__ topk_logp, const long long* __restrict__ topk_idx,
                                                             int Tmax, const int* __restrict__ lens, int beam, int max_len,
                                                             int* __restrict__ hist, int* __restrict__ out_prefix,
                                                             int* __restrict__ out_len, double* __restrict__ out_score,
                                                             int* __restrict__ status) {
    __shared__ unsigned long long cur_key[PB_MAXBEAM];
    __shared__ double cur_pb[PB_MAXBEAM], cur_pnb[PB_MAXBEAM];
    __shared__ int cur_len[PB_MAXBEAM], cur_last[PB_MAXBEAM];
    __shared__ double tk_ps[PB_MAXBEAM];
    __shared__ int tk_s[PB_MAXBEAM];
    __shared__ int con_has[PB_MAXBEAM], con_from[PB_MAXBEAM], con_three[PB_MAXBEAM], con_order[PB_MAXBEAM];
    __shared__ double con_a[PB_MAXBEAM], con_b[PB_MAXBEAM];
    __shared__ unsigned long long nx_key[PB_MAXBEAM];
    __shared__ double nx_pb[PB_MAXBEAM], nx_pnb[PB_MAXBEAM];
    __shared__ int nx_len[PB_MAXBEAM], nx_last[PB_MAXBEAM];

    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = min(lens ? lens[b] : Tmax, Tmax);
    const double NEG = pb_neg();
    int ncur = 1;
    if (lane == 0) { cur_key[0] = 0; cur_pb[0] = 0.0; cur_pnb[0] = NEG; cur_len[0] = 0; cur_last[0] = -1; }
    __syncthreads();
    int* hist_b = hist + (long)b * Tmax * beam * 2;

    for (int t = 0; t < T; ++t) {
        if (lane < beam) {
            const long o = ((long)b * Tmax + t) * beam + lane;
            tk_ps[lane] = (double)topk_logp[o];
            tk_s[lane] = (int)topk_idx[o];
        }
        if (lane < ncur) con_has[lane] = 0;
        __syncthreads();

        // ---- extension candidates: pair p = (j, h), lanes p, p + 64, ...
        PbHyp cand[PB_MAXC + 1];
        bool alive[PB_MAXC + 1];
        const int npairs = beam * ncur;
#pragma unroll
        for (int c = 0; c < PB_MAXC; ++c) {
            const int p = lane + 64 * c;
            alive[c] = false;
            if (p < npairs) {
                const int j = p / ncur, h = p - j * ncur;
                const int s = tk_s[j];
                if (s != 0) {
                    const double ps = tk_ps[j];
                    const unsigned long long key = cur_key[h] * PB_HASH_MUL + (unsigned long long)(s + 1);
                    const int len = cur_len[h] + 1;
                    const bool rep = (s == cur_last[h]);
                    const double a = cur_pb[h] + ps, bb = cur_pnb[h] + ps;
                    int hit = -1;
                    for (int n = 0; n < ncur; ++n)
                        if (cur_key[n] == key && cur_len[n] == len) hit = n;
                    if (hit >= 0) {                    // lands on a prefix that is already in the beam: that prefix's lane applies it
                        con_has[hit] = 1; con_from[hit] = h; con_three[hit] = rep ? 0 : 1; con_a[hit] = a; con_b[hit] = bb;
                        con_order[hit] = 2 * p + 1;
                    } else {
                        alive[c] = true;
                        cand[c].key = key; cand[c].len = len; cand[c].last = s; cand[c].parent = h; cand[c].tok = s;
                        cand[c].pb = NEG;
                        cand[c].pnb = rep ? a : pb_log_add3(NEG, a, bb);
                        cand[c].order = 2 * p + 1;
                    }
                }
            }
        }
        __syncthreads();

        // ---- the current prefixes' own entries: lane n
        alive[PB_MAXC] = false;
        if (lane < ncur) {
            const int n = lane;
            const double pb = cur_pb[n], pnb = cur_pnb[n];
            const int last = cur_last[n];
            int j0 = -1, jr = -1;
            for (int j = 0; j < beam; ++j) {
                if (tk_s[j] == 0 && j0 < 0) j0 = j;
                if (tk_s[j] == last && last > 0 && jr < 0) jr = j;
            }
            double npb = NEG, npnb = NEG;
            int order = 0x7fffffff;
            bool touched = false;
            if (j0 >= 0) {
                npb = pb_log_add3(NEG, pb + tk_ps[j0], pnb + tk_ps[j0]);
                order = min(order, 2 * (j0 * ncur + n));
                touched = true;
            }
            const bool rep = jr >= 0, con = con_has[n] != 0;
            const bool con_first = con && (!rep || con_from[n] < n);
            if (con && con_first) npnb = con_three[n] ? pb_log_add3(npnb, con_a[n], con_b[n]) : pb_log_add2(npnb, con_a[n]);
            if (rep) { npnb = pb_log_add2(npnb, pnb + tk_ps[jr]); order = min(order, 2 * (jr * ncur + n)); touched = true; }
            if (con && !con_first) npnb = con_three[n] ? pb_log_add3(npnb, con_a[n], con_b[n]) : pb_log_add2(npnb, con_a[n]);
            if (con) { order = min(order, con_order[n]); touched = true; }
            if (touched) {
                alive[PB_MAXC] = true;
                PbHyp& e = cand[PB_MAXC];
                e.key = cur_key[n]; e.len = cur_len[n]; e.last = last; e.parent = n; e.tok = -1; e.pb = npb; e.pnb = npnb; e.order = order;
            }
        }
#pragma unroll
        for (int c = 0; c <= PB_MAXC; ++c)
            if (alive[c]) cand[c].score = pb_log_add2(cand[c].pb, cand[c].pnb);

        // ---- the best `beam` entries by (score descending, first touch ascending)
        int nsel = 0;
        for (int r = 0; r < beam; ++r) {
            double bs = NEG;
            int bo = 0x7fffffff, bc = -1;
#pragma unroll
            for (int c = 0; c <= PB_MAXC; ++c)
                if (alive[c] && (bc < 0 || cand[c].score > bs || (cand[c].score == bs && cand[c].order < bo))) { bs = cand[c].score; bo = cand[c].order; bc = c; }
            double ws = bs;
            int wo = bo;                                      // lanes without a candidate carry (NEG, INT_MAX)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double os = pb_shfl_xor(ws, o);
                const int oo = __shfl_xor(wo, o, 64);
                if (oo != 0x7fffffff && (wo == 0x7fffffff || os > ws || (os == ws && oo < wo))) { ws = os; wo = oo; }
            }
            if (wo == 0x7fffffff) break;                      // nothing left (wave-uniform)
            if (bc >= 0 && bo == wo) {                        // stamps are unique: this lane holds the winner
#pragma unroll
                for (int c = 0; c <= PB_MAXC; ++c)
                    if (c == bc) {
                        nx_key[r] = cand[c].key; nx_pb[r] = cand[c].pb; nx_pnb[r] = cand[c].pnb; nx_len[r] = cand[c].len; nx_last[r] = cand[c].last;
                        hist_b[((long)t * beam + r) * 2] = cand[c].parent;
                        hist_b[((long)t * beam + r) * 2 + 1] = cand[c].tok;
                        alive[c] = false;
                    }
            }
            nsel = r + 1;
        }
        __syncthreads();
        if (lane < nsel) { cur_key[lane] = nx_key[lane]; cur_pb[lane] = nx_pb[lane]; cur_pnb[lane] = nx_pnb[lane]; cur_len[lane] = nx_len[lane]; cur_last[lane] = nx_last[lane]; }
        ncur = nsel;
        __syncthreads();
    }

    // ---- results: scores, lengths, tokens by walking the back-pointers
    if (lane < beam) {
        const long o = (long)b * beam + lane;
        if (lane < ncur) {
            const int len = cur_len[lane];
            out_score[o] = pb_log_add2(cur_pb[lane], cur_pnb[lane]);
            out_len[o] = len;
            if (len > max_len) { atomicExch(status, 1); return; }
            int slot = lane, pos = len - 1;
            for (int t = T - 1; t >= 0 && pos >= 0; --t) {
                const int parent = hist_b[((long)t * beam + slot) * 2], tok = hist_b[((long)t * beam + slot) * 2 + 1];
                if (tok >= 0) out_prefix[o * max_len + pos--] = tok;
                slot = parent;
            }
        } else {
            out_score[o] = NEG;
            out_len[o] = -1;
        }
    }
}

extern "C" size_t oe_ctc_prefix_beam_workspace_bytes(int B, int Tmax, int beam) {
    return ((size_t)B * (size_t)max(Tmax, 1) * (size_t)beam * 2 + 1) * sizeof(int);
}

extern "C" int oe_ctc_prefix_beam(const float* topk_logp, const long long* topk_idx, int B, int Tmax, const int* lens, int beam,
                                  int max_len, void* workspace, int* out_prefix, int* out_len, double* out_score, void* stream) {
    OE_REQUIRE(topk_logp && topk_idx && workspace && out_prefix && out_len && out_score, "oe_ctc_prefix_beam: null pointer");
    OE_REQUIRE(B > 0 && Tmax >= 0 && max_len >= 0, "oe_ctc_prefix_beam: bad shape B=%d Tmax=%d max_len=%d", B, Tmax, max_len);
    OE_REQUIRE(beam >= 1 && beam <= PB_MAXBEAM, "oe_ctc_prefix_beam: beam must be 1..%d (got %d)", PB_MAXBEAM, beam);
    int* hist = (int*)workspace;
    int* status = hist + (size_t)B * (size_t)max(Tmax, 1) * (size_t)beam * 2;     // the caller zeroes this word and reads it back
    hipLaunchKernelGGL(ctc_prefix_beam_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, topk_logp, topk_idx, Tmax, lens, beam, max_len,
                       hist, out_prefix, out_len, out_score, status);
    OE_LAUNCH_CHECK("ctc_prefix_beam");
    return 0;
}
