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
#include <atomic>
#include <thread>
#include <unordered_map>
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

// Prefixes live in a trie: a node is (parent, token); two prefixes are equal iff they are the same node, so the
// per-frame dictionary of the reference (keyed by the token tuple) becomes a map keyed by the node id and every
// candidate update is O(1) instead of a comparison of two token sequences.
struct Trie {
    std::vector<int> parent, token, length;
    std::unordered_map<uint64_t, int> child;
    Trie() { parent.push_back(-1); token.push_back(-1); length.push_back(0); }
    int extend(int node, int tok) {
        const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)tok;
        auto it = child.find(key);
        if (it != child.end()) return it->second;
        const int id = (int)parent.size();
        parent.push_back(node); token.push_back(tok); length.push_back(length[node] + 1);
        child.emplace(key, id);
        return id;
    }
};

struct Hyp {
    int node;
    double pb, pnb;
};

// one utterance; returns 0 or -1 (prefix longer than max_len)
int beam_one(const float* topk_logp, const long long* topk_idx, int T, int beam, int max_len, int* out_prefix, int* out_len,
             double* out_score) {
    Trie trie;
    std::vector<Hyp> cur(1, Hyp{0, 0.0, NEG}), nxt, pruned;
    std::unordered_map<int, int> index;                    // node -> position in nxt (nxt keeps the insertion order)
    auto slot = [&](int node) -> Hyp& {
        auto it = index.find(node);
        if (it != index.end()) return nxt[it->second];
        index.emplace(node, (int)nxt.size());
        nxt.push_back(Hyp{node, NEG, NEG});
        return nxt.back();
    };
    std::vector<double> key;
    std::vector<int> order;
    for (int t = 0; t < T; ++t) {
        nxt.clear();
        index.clear();
        for (int j = 0; j < beam; ++j) {
            const int s = (int)topk_idx[(long)t * beam + j];
            const double ps = (double)topk_logp[(long)t * beam + j];
            for (size_t h = 0; h < cur.size(); ++h) {
                const int node = cur[h].node;
                const double pb = cur[h].pb, pnb = cur[h].pnb;
                const int last = trie.token[node];          // -1 for the empty prefix
                if (s == 0) {
                    Hyp& e = slot(node);
                    e.pb = log_add3(e.pb, pb + ps, pnb + ps);
                } else if (s == last) {
                    {
                        Hyp& e = slot(node);
                        e.pnb = log_add2(e.pnb, pnb + ps);
                    }
                    Hyp& e2 = slot(trie.extend(node, s));
                    e2.pnb = log_add2(e2.pnb, pb + ps);
                } else {
                    Hyp& e2 = slot(trie.extend(node, s));
                    e2.pnb = log_add3(e2.pnb, pb + ps, pnb + ps);
                }
            }
        }
        // sorted(items, key=log_add(pb, pnb), reverse=True): CPython implements reverse=True as
        // reverse / stable ascending sort / reverse, which keeps ties in their ORIGINAL order
        key.resize(nxt.size());
        order.resize(nxt.size());
        for (size_t i = 0; i < nxt.size(); ++i) { key[i] = log_add2(nxt[i].pb, nxt[i].pnb); order[i] = (int)i; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b]; });
        const size_t keep = std::min<size_t>(beam, order.size());
        pruned.clear();
        for (size_t i = 0; i < keep; ++i) pruned.push_back(nxt[order[i]]);
        cur.swap(pruned);
    }
    for (int i = 0; i < beam; ++i) {
        if (i < (int)cur.size()) {
            const int n = trie.length[cur[i].node];
            if (n > max_len) return -1;
            out_len[i] = n;
            int node = cur[i].node;
            for (int k = n - 1; k >= 0; --k) { out_prefix[(long)i * max_len + k] = trie.token[node]; node = trie.parent[node]; }
            out_score[i] = log_add2(cur[i].pb, cur[i].pnb);
        } else {
            out_len[i] = -1;
            out_score[i] = NEG;
        }
    }
    return 0;
}
}  // namespace

extern "C" int oe_ctc_prefix_beam_host(const float* topk_logp_host, const long long* topk_idx_host, int T, int beam,
                                       int max_len, int* out_prefix_host, int* out_len_host, double* out_score_host) {
    if (!topk_logp_host || !topk_idx_host || !out_prefix_host || !out_len_host || !out_score_host || T < 0 || beam <= 0 ||
        max_len < 0) {
        oe_set_error("oe_ctc_prefix_beam_host: bad arguments");
        return -1;
    }
    if (beam_one(topk_logp_host, topk_idx_host, T, beam, max_len, out_prefix_host, out_len_host, out_score_host)) {
        oe_set_error("oe_ctc_prefix_beam_host: prefix longer than max_len=%d", max_len);
        return -1;
    }
    return 0;
}

// B utterances at once, spread over host threads (utterances are independent: asr_model.py:444 handles one per call).
extern "C" int oe_ctc_prefix_beam_host_batch(const float* topk_logp_host, const long long* topk_idx_host, int B, int Tmax,
                                             const int* lens_host, int beam, int max_len, int* out_prefix_host,
                                             int* out_len_host, double* out_score_host, int n_threads) {
    if (!topk_logp_host || !topk_idx_host || !lens_host || !out_prefix_host || !out_len_host || !out_score_host || B <= 0 ||
        Tmax < 0 || beam <= 0 || max_len < 0) {
        oe_set_error("oe_ctc_prefix_beam_host_batch: bad arguments");
        return -1;
    }
    for (int b = 0; b < B; ++b)
        if (lens_host[b] < 0 || lens_host[b] > Tmax) { oe_set_error("oe_ctc_prefix_beam_host_batch: lens[%d] out of range", b); return -1; }
    if (n_threads <= 0) n_threads = std::min(16, (int)std::thread::hardware_concurrency());   // a one-GPU share of the host
    n_threads = std::max(1, std::min(n_threads, std::min(B, 64)));
    std::atomic<int> next(0), failed(0);
    auto work = [&]() {
        for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
            const long in_off = (long)b * Tmax * beam;
            if (beam_one(topk_logp_host + in_off, topk_idx_host + in_off, lens_host[b], beam, max_len,
                         out_prefix_host + (long)b * beam * max_len, out_len_host + (long)b * beam, out_score_host + (long)b * beam))
                failed.store(1);
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < n_threads; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (failed.load()) { oe_set_error("oe_ctc_prefix_beam_host_batch: prefix longer than max_len=%d", max_len); return -1; }
    return 0;
}
