// Logits processors and token sampling (see sampling.h).
#include "sampling.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <random>

namespace kjarni {

void apply_repetition_penalty(std::vector<float>& logits, const std::vector<uint32_t>& tokens, float penalty)
{
    if (penalty == 1.0f) return;
    for (uint32_t t : tokens) {  // once per occurrence, like the reference
        if (t >= logits.size()) continue;
        const float s = logits[t];
        logits[t] = s < 0.0f ? s * penalty : s / penalty;
    }
}

void apply_no_repeat_ngram(std::vector<float>& logits, const std::vector<uint32_t>& tokens, size_t n)
{
    if (n == 0 || tokens.size() + 1 < n) return;
    const size_t len = tokens.size();
    for (size_t i = 0; i + n <= len; ++i) {
        if (!std::equal(tokens.begin() + (ptrdiff_t)i, tokens.begin() + (ptrdiff_t)(i + n - 1), tokens.end() - (ptrdiff_t)(n - 1))) continue;
        const uint32_t banned = tokens[i + n - 1];
        if (banned < logits.size()) logits[banned] = -std::numeric_limits<float>::infinity();
    }
}

uint32_t argmax_last(const std::vector<float>& logits)
{
    size_t best = 0;
    for (size_t i = 1; i < logits.size(); ++i)
        if (logits[i] >= logits[best]) best = i;
    return (uint32_t)best;
}

namespace {

// softmax_inplace over the survivors, summed in ascending id order (masked entries would add exactly 0).
void softmax(const std::vector<float>& vals, std::vector<float>& probs)
{
    probs.resize(vals.size());
    if (vals.empty()) return;
    float mx = -std::numeric_limits<float>::infinity();
    for (float v : vals) mx = std::max(mx, v);
    float sum = 0.0f;
    for (size_t i = 0; i < vals.size(); ++i) {
        probs[i] = std::exp(vals[i] - mx);
        sum += probs[i];
    }
    if (sum > 0.0f) {
        const float scale = 1.0f / sum;
        for (float& p : probs) p *= scale;
    }
}

// Positions of `vals` in the reference's sort order (value descending, stable = ascending position), produced
// lazily: `enough(prefix)` is called with ever longer exact prefixes of that order until it returns true or the
// prefix is the whole array.
std::vector<uint32_t> sorted_prefix(const std::vector<float>& vals, const std::function<bool(const std::vector<uint32_t>&)>& enough)
{
    const size_t n = vals.size();
    std::vector<uint32_t> cand;
    if (n == 0) return cand;
    float mx = -std::numeric_limits<float>::infinity();
    for (float v : vals) mx = std::max(mx, v);
    const auto by_value = [&](uint32_t a, uint32_t b) { return vals[a] > vals[b] || (vals[a] == vals[b] && a < b); };
    float tau = 10.0f;
    for (;;) {
        const bool all = !(tau < 1e30f) || !std::isfinite(mx);
        const float floor = mx - tau;
        cand.clear();
        for (size_t i = 0; i < n; ++i)
            if (all || vals[i] >= floor) cand.push_back((uint32_t)i);
        std::sort(cand.begin(), cand.end(), by_value);
        if (all || cand.size() == n || enough(cand)) return cand;
        tau *= 4.0f;
    }
}

}  // namespace

void sampling_distribution(const std::vector<float>& logits, const SamplingParams& p, std::vector<uint32_t>& ids, std::vector<float>& probs)
{
    const size_t vocab = logits.size();
    ids.resize(vocab);
    for (size_t i = 0; i < vocab; ++i) ids[i] = (uint32_t)i;
    std::vector<float> vals(logits);
    auto keep = [&](std::vector<uint32_t>& positions) {  // survivors by position, back to ascending id order
        std::sort(positions.begin(), positions.end());
        std::vector<uint32_t> nid(positions.size());
        std::vector<float> nval(positions.size());
        for (size_t i = 0; i < positions.size(); ++i) {
            nid[i] = ids[positions[i]];
            nval[i] = vals[positions[i]];
        }
        ids.swap(nid);
        vals.swap(nval);
    };

    if (p.top_k >= 0 && (size_t)p.top_k < vals.size()) {  // top_k_filtering
        const size_t k = (size_t)p.top_k;
        std::vector<uint32_t> order = sorted_prefix(vals, [&](const std::vector<uint32_t>& c) { return c.size() >= k; });
        order.resize(std::min(order.size(), k));
        keep(order);
    }
    if (p.top_p >= 0.0f && !vals.empty()) {  // top_p_filtering: keep through the first token that pushes the mass past p
        softmax(vals, probs);
        size_t cut = 0;
        bool found = false;
        auto scan = [&](const std::vector<uint32_t>& c) {
            float cumulative = 0.0f;
            for (size_t i = 0; i < c.size(); ++i) {
                cumulative += probs[c[i]];
                if (cumulative > p.top_p) {
                    cut = i;
                    found = true;
                    return true;
                }
            }
            return false;
        };
        std::vector<uint32_t> order = sorted_prefix(vals, scan);
        if (!found) scan(order);
        if (found) {
            order.resize(cut + 1);
            keep(order);
        }
    }
    if (p.min_p >= 0.0f && !vals.empty()) {  // min_p_filtering
        softmax(vals, probs);
        float max_prob = 0.0f;
        for (float q : probs) max_prob = std::max(max_prob, q);
        const float cutoff = max_prob * p.min_p;
        std::vector<uint32_t> pos;
        for (size_t i = 0; i < probs.size(); ++i)
            if (!(probs[i] < cutoff)) pos.push_back((uint32_t)i);
        keep(pos);
    }
    const float temp = p.temperature < 1e-5f ? 1.0f : p.temperature;
    for (float& v : vals) v /= temp;
    softmax(vals, probs);
}

uint32_t sample_from_distribution(const std::vector<uint32_t>& ids, const std::vector<float>& probs, float uniform, size_t vocab)
{
    if (vocab == 0) return 0;
    if (0.0f >= uniform) return 0;  // index 0 already satisfies `cumulative >= uniform` whatever its probability
    float cumulative = 0.0f;
    for (size_t i = 0; i < ids.size(); ++i) {
        cumulative += probs[i];
        if (cumulative >= uniform) return ids[i];
    }
    return (uint32_t)(vocab - 1);
}

UniformRng::UniformRng()
{
    std::random_device rd;
    reseed(((uint64_t)rd() << 32) ^ rd());
}

void UniformRng::reseed(uint64_t seed)
{
    for (auto& s : s_) {  // splitmix64
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        s = z ^ (z >> 31);
    }
}

float UniformRng::next()
{
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    const uint64_t result = rotl(s_[1] * 5, 7) * 9;
    const uint64_t t = s_[1] << 17;
    s_[2] ^= s_[0];
    s_[3] ^= s_[1];
    s_[1] ^= s_[2];
    s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = rotl(s_[3], 45);
    return (float)(result >> 40) * (1.0f / 16777216.0f);
}

}  // namespace kjarni
