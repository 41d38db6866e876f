// CTC prefix beam search on the host (native replacement of the Python dict loop of
// /root/reference/openeat/models/asr_model.py:359-396).  Input: per frame the top-`beam` CTC
// log-probabilities and token ids (computed on the GPU); output: the `beam` best prefixes with
// their scores.  Semantics reproduced exactly:
//   * python floats = IEEE doubles; log_add = a_max + log(sum exp(a - a_max)) (common.py:198-206),
//     summed in argument order;
//   * next_hyps is an insertion-ordered dict; the pruning `sorted(..., reverse=True)[:beam]` is a
//     stable sort, i.e. ties keep insertion order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "../../include/openeat_hip.h"

extern "C" void oe_set_error(const char* fmt, ...);

namespace {
const double NEG = -std::numeric_limits<double>::infinity();

inline double log_add3(double a, double b, double c) {
    if (a == NEG && b == NEG && c == NEG) return NEG;
    const double m = std::max(a, std::max(b, c));
    return m + std::log(std::exp(a - m) + std::exp(b - m) + std::exp(c - m));
}
inline double log_add2(double a, double b) {
    if (a == NEG && b == NEG) return NEG;
    const double m = std::max(a, b);
    return m + std::log(std::exp(a - m) + std::exp(b - m));
}

struct Hyp {
    std::vector<int> prefix;
    double pb, pnb;
};
}  // namespace

extern "C" int oe_ctc_prefix_beam_host(const float* topk_logp_host, const long long* topk_idx_host, int T, int beam,
                                       int max_len, int* out_prefix_host, int* out_len_host, double* out_score_host) {
    if (!topk_logp_host || !topk_idx_host || !out_prefix_host || !out_len_host || !out_score_host || T < 0 || beam <= 0 ||
        max_len < 0) {
        oe_set_error("oe_ctc_prefix_beam_host: bad arguments");
        return -1;
    }
    std::vector<Hyp> cur(1);
    cur[0].pb = 0.0;
    cur[0].pnb = NEG;
    std::vector<Hyp> nxt;
    std::map<std::vector<int>, int> index;                 // prefix -> position in nxt (insertion order kept by nxt)
    auto slot = [&](const std::vector<int>& p) -> Hyp& {
        auto it = index.find(p);
        if (it != index.end()) return nxt[it->second];
        index.emplace(p, (int)nxt.size());
        nxt.push_back(Hyp{p, NEG, NEG});
        return nxt.back();
    };
    std::vector<int> ext;
    for (int t = 0; t < T; ++t) {
        nxt.clear();
        index.clear();
        for (int j = 0; j < beam; ++j) {
            const int s = (int)topk_idx_host[(long)t * beam + j];
            const double ps = (double)topk_logp_host[(long)t * beam + j];
            for (size_t h = 0; h < cur.size(); ++h) {
                // copy what we need: slot() may reallocate nxt, but cur is a separate vector
                const std::vector<int>& prefix = cur[h].prefix;
                const double pb = cur[h].pb, pnb = cur[h].pnb;
                const int last = prefix.empty() ? -1 : prefix.back();
                if (s == 0) {
                    Hyp& e = slot(prefix);
                    e.pb = log_add3(e.pb, pb + ps, pnb + ps);
                } else if (s == last) {
                    {
                        Hyp& e = slot(prefix);
                        e.pnb = log_add2(e.pnb, pnb + ps);
                    }
                    ext = prefix;
                    ext.push_back(s);
                    Hyp& e2 = slot(ext);
                    e2.pnb = log_add2(e2.pnb, pb + ps);
                } else {
                    ext = prefix;
                    ext.push_back(s);
                    Hyp& e2 = slot(ext);
                    e2.pnb = log_add3(e2.pnb, pb + ps, pnb + ps);
                }
            }
        }
        // sorted(items, key=log_add(pb, pnb), reverse=True): CPython implements reverse=True as
        // reverse / stable ascending sort / reverse, which keeps ties in their ORIGINAL order
        std::vector<double> key(nxt.size());
        for (size_t i = 0; i < nxt.size(); ++i) key[i] = log_add2(nxt[i].pb, nxt[i].pnb);
        std::vector<int> order(nxt.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b]; });
        const size_t keep = std::min<size_t>(beam, order.size());
        std::vector<Hyp> pruned;
        pruned.reserve(keep);
        for (size_t i = 0; i < keep; ++i) pruned.push_back(std::move(nxt[order[i]]));
        cur.swap(pruned);
    }
    for (int i = 0; i < beam; ++i) {
        if (i < (int)cur.size()) {
            const int n = (int)cur[i].prefix.size();
            if (n > max_len) {
                oe_set_error("oe_ctc_prefix_beam_host: prefix longer than max_len=%d", max_len);
                return -1;
            }
            out_len_host[i] = n;
            if (n) std::memcpy(out_prefix_host + (long)i * max_len, cur[i].prefix.data(), (size_t)n * sizeof(int));
            out_score_host[i] = log_add2(cur[i].pb, cur[i].pnb);
        } else {
            out_len_host[i] = -1;
            out_score_host[i] = NEG;
        }
    }
    return 0;
}
