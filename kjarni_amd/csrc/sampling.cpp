// Logits processors and token sampling (see sampling.h).
#include "sampling.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <random>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kjarni {

void apply_repetition_penalty(float* logits, size_t vocab, const std::vector<uint32_t>& tokens, float penalty)
{
    if (penalty == 1.0f) return;
    for (uint32_t t : tokens) {  // once per occurrence, like the reference
        if (t >= vocab) continue;
        const float s = logits[t];
        logits[t] = s < 0.0f ? s * penalty : s / penalty;
    }
}

void apply_no_repeat_ngram(float* logits, size_t vocab, const std::vector<uint32_t>& tokens, size_t n)
{
    if (n == 0 || tokens.size() + 1 < n) return;
    const size_t len = tokens.size();
    for (size_t i = 0; i + n <= len; ++i) {
        if (!std::equal(tokens.begin() + (ptrdiff_t)i, tokens.begin() + (ptrdiff_t)(i + n - 1), tokens.end() - (ptrdiff_t)(n - 1))) continue;
        const uint32_t banned = tokens[i + n - 1];
        if (banned < vocab) logits[banned] = -std::numeric_limits<float>::infinity();
    }
}

uint32_t argmax_last(const float* logits, size_t vocab)
{
    size_t best = 0;
    for (size_t i = 1; i < vocab; ++i)
        if (logits[i] >= logits[best]) best = i;
    return (uint32_t)best;
}

namespace {

constexpr float kNegInf = -std::numeric_limits<float>::infinity();

// ---- full-vocabulary passes --------------------------------------------------------------------------
// The reference runs softmax_inplace over the whole vocabulary up to three times per token (scalar exp, running sum).
// Here the whole-vocabulary work is one max pass and one exp + sum pass, 8 lanes wide when the CPU has AVX2 (expf by
// range reduction + degree-6 polynomial, < 2 ulp; 8 interleaved partial sums).  A top-p cut-off hangs on the rounding of
// that 10^5-term f32 sum, so whenever the crossing is within rounding of p the sum is redone in the reference's index
// order and the decision repeated.

float max_scalar(const float* v, size_t n)
{
    float mx = kNegInf;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, v[i]);
    return mx;
}

void exp_scalar(const float* v, size_t n, float mx, float inv_temp, float* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = std::exp((v[i] - mx) * inv_temp);
}

#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) float max_avx2(const float* v, size_t n)
{
    __m256 m = _mm256_set1_ps(kNegInf);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) m = _mm256_max_ps(m, _mm256_loadu_ps(v + i));
    float lanes[8];
    _mm256_storeu_ps(lanes, m);
    float mx = kNegInf;
    for (float x : lanes) mx = std::max(mx, x);
    for (; i < n; ++i) mx = std::max(mx, v[i]);
    return mx;
}

__attribute__((target("avx2,fma"))) inline __m256 exp_avx2(__m256 x)
{
    // exp(x) for x <= 0: n = round(x / ln 2), r = x - n ln 2 (two-step), e^r by a degree-6 polynomial, scale by 2^n.
    const __m256 lo = _mm256_set1_ps(-87.33654f);
    const __m256 under = _mm256_cmp_ps(x, lo, _CMP_LT_OQ);
    x = _mm256_max_ps(x, lo);
    const __m256 nf = _mm256_round_ps(_mm256_mul_ps(x, _mm256_set1_ps(1.44269504088896341f)), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    __m256 r = _mm256_fnmadd_ps(nf, _mm256_set1_ps(0.693359375f), x);
    r = _mm256_fnmadd_ps(nf, _mm256_set1_ps(-2.12194440e-4f), r);
    __m256 p = _mm256_set1_ps(1.9875691500e-4f);
    p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(1.3981999507e-3f));
    p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(8.3334519073e-3f));
    p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(4.1665795894e-2f));
    p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(1.6666665459e-1f));
    p = _mm256_fmadd_ps(p, r, _mm256_set1_ps(5.0000001201e-1f));
    p = _mm256_fmadd_ps(p, _mm256_mul_ps(r, r), _mm256_add_ps(r, _mm256_set1_ps(1.0f)));
    const __m256i e = _mm256_slli_epi32(_mm256_add_epi32(_mm256_cvtps_epi32(nf), _mm256_set1_epi32(127)), 23);
    return _mm256_andnot_ps(under, _mm256_mul_ps(p, _mm256_castsi256_ps(e)));
}

__attribute__((target("avx2,fma"))) float exp_all_avx2(const float* v, size_t n, float mx, float inv_temp, float* out)
{
    const __m256 vm = _mm256_set1_ps(mx), vt = _mm256_set1_ps(inv_temp);
    __m256 acc = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 e = exp_avx2(_mm256_mul_ps(_mm256_sub_ps(_mm256_loadu_ps(v + i), vm), vt));
        _mm256_storeu_ps(out + i, e);
        acc = _mm256_add_ps(acc, e);
    }
    float lanes[8];
    _mm256_storeu_ps(lanes, acc);
    float sum = ((lanes[0] + lanes[1]) + (lanes[2] + lanes[3])) + ((lanes[4] + lanes[5]) + (lanes[6] + lanes[7]));
    for (; i < n; ++i) {
        out[i] = std::exp((v[i] - mx) * inv_temp);
        sum += out[i];
    }
    return sum;
}

const bool kHaveAvx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
#else
const bool kHaveAvx2 = false;
#endif

constexpr size_t kVectorFrom = 4096;  // below this the scalar loops (bit-identical to the reference's order) are used

float vec_max(const float* v, size_t n)
{
#if defined(__x86_64__)
    if (kHaveAvx2 && n >= kVectorFrom) return max_avx2(v, n);
#endif
    return max_scalar(v, n);
}

float running_sum(const float* e, size_t n)  // the reference's summation order
{
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum += e[i];
    return sum;
}

// out[i] = exp((v[i] - mx) * inv_temp).  Returns the sum of out: `exact` tells whether it is the running sum in index order
// (scalar path) or 8 interleaved partial sums (AVX2 path; differs from the running sum by rounding order, ~1e-5 relative
// on a 10^5-term sum -- callers that compare a cumulative mass with a threshold re-sum in order when the decision is close).
float vec_exp_sum(const float* v, size_t n, float mx, float inv_temp, float* out, bool& exact)
{
#if defined(__x86_64__)
    if (kHaveAvx2 && n >= kVectorFrom) {
        exact = false;
        return exp_all_avx2(v, n, mx, inv_temp, out);
    }
#endif
    exp_scalar(v, n, mx, inv_temp, out);
    exact = true;
    return running_sum(out, n);
}

// softmax_inplace over a (small) survivor list, summed in ascending id order exactly as the reference does.
void softmax(const std::vector<float>& vals, std::vector<float>& probs)
{
    probs.resize(vals.size());
    if (vals.empty()) return;
    float mx = kNegInf;
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

// First threshold for sorted_prefix from a 1/8-wide histogram of (max - value): the distance below the maximum that
// holds `count` values, or `mass` of the sum of `weights` (the exp values), plus one bin of margin.
float tau_for_count(const float* vals, size_t n, float mx, size_t count)
{
    constexpr int kBins = 512;  // covers 64 below the maximum
    std::vector<uint32_t> hist(kBins + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        const float d = (mx - vals[i]) * 8.0f;
        hist[d < (float)kBins ? (int)d : kBins]++;
    }
    size_t seen = 0;
    for (int b = 0; b < kBins; ++b) {
        seen += hist[b];
        if (seen >= count) return (float)(b + 2) / 8.0f;
    }
    return 1e31f;
}

float tau_for_mass(const float* vals, const float* weights, size_t n, float mx, double mass)
{
    constexpr int kBins = 512;
    std::vector<double> hist(kBins + 1, 0.0);
    for (size_t i = 0; i < n; ++i) {
        const float d = (mx - vals[i]) * 8.0f;
        hist[d < (float)kBins ? (int)d : kBins] += weights[i];
    }
    double seen = 0.0;
    for (int b = 0; b < kBins; ++b) {
        seen += hist[b];
        if (seen > mass) return (float)(b + 2) / 8.0f;
    }
    return 1e31f;
}

// Positions of `vals` in the reference's sort order (value descending, stable = ascending position), produced
// lazily: `enough(prefix)` is called with ever longer exact prefixes of that order until it returns true or the
// prefix is the whole array.  tau0: first threshold below the maximum.
std::vector<uint32_t> sorted_prefix(const float* vals, size_t n, float mx, float tau0,
                                    const std::function<bool(const std::vector<uint32_t>&)>& enough)
{
    std::vector<uint32_t> cand;
    if (n == 0) return cand;
    const auto by_value = [&](uint32_t a, uint32_t b) { return vals[a] > vals[b] || (vals[a] == vals[b] && a < b); };
    float tau = tau0;
    float upper = std::numeric_limits<float>::infinity();  // values >= upper are already in `cand`, sorted
    for (;;) {
        const bool all = !(tau < 1e30f) || !std::isfinite(mx);
        const float floor = all ? kNegInf : mx - tau;
        // the next band [floor, upper): everything in it sorts after everything already collected
        const size_t before = cand.size();
        for (size_t i = 0; i < n; ++i) {
            const float v = vals[i];
            if ((all ? !(v >= upper) : (v >= floor && v < upper))) cand.push_back((uint32_t)i);
        }
        std::sort(cand.begin() + (ptrdiff_t)before, cand.end(), by_value);
        if (all || cand.size() == n || enough(cand)) return cand;
        upper = floor;
        tau = tau < 64.0f ? tau + 2.0f : tau * 2.0f;
    }
}


// sample_token's filters over a VIEW of the vocabulary: either all of it (`complete`: n = vocab, position = token id), or
// the candidate list the device cut out of it (llm_kernels.hip, sample_candidates): every token whose logit is >= `floor`,
// in ascending id order, with the maximum `mx` and the sum of exp(logit - mx) over the WHOLE vocabulary (`sum`, summed
// in tree order on the device: never `exact`) supplied alongside.  Everything outside the list sorts after everything
// inside it, so the list's own sort order is an exact prefix of the reference's; a filter that needs more than the list
// holds, or whose decision hangs on the rounding order of that sum, returns false and the caller fetches the logits.
struct VocabView {
    const float* vals;       // logits of the view's positions
    size_t n;
    const uint32_t* tokens;  // null: position == token id
    bool complete;
    float mx;                // maximum over the whole vocabulary
    float floor;             // candidates: every token outside the view has a logit < floor
    float sum;               // candidates: sum of exp(logit - mx) over the whole vocabulary (not in index order)
};

bool sampling_distribution_view(const VocabView& view, size_t vocab, const SamplingParams& p, std::vector<uint32_t>& ids,
                                std::vector<float>& probs)
{
    // Survivors: everything (`all`, values read straight from the view) until a filter shrinks the set to (ids, vals).
    const float* logits = view.vals;
    const size_t n_view = view.n;
    bool all = true, have_exps = false;
    float exps_sum = 0.0f;
    std::vector<float> vals, exps;
    ids.clear();
    probs.clear();
    if (vocab == 0) return true;
    const float mx_all = view.mx;
    auto token_of = [&](uint32_t pos) { return view.tokens ? view.tokens[pos] : pos; };
    auto shrink_from_all = [&](std::vector<uint32_t>& positions) {  // positions index the view
        std::sort(positions.begin(), positions.end());
        ids.resize(positions.size());
        vals.resize(positions.size());
        for (size_t i = 0; i < positions.size(); ++i) {
            ids[i] = token_of(positions[i]);
            vals[i] = logits[positions[i]];
        }
        all = false;
    };
    auto shrink = [&](std::vector<uint32_t>& positions) {  // positions index (ids, vals)
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
    // exp(logit - mx) over the view and the sum over the whole vocabulary
    auto view_exps = [&](bool& exact) {
        exps.resize(n_view);
        if (view.complete) return vec_exp_sum(logits, n_view, mx_all, 1.0f, exps.data(), exact);
        exp_scalar(logits, n_view, mx_all, 1.0f, exps.data());
        exact = false;
        return view.sum;
    };

    if (p.top_k >= 0 && (size_t)p.top_k < vocab) {  // top_k_filtering
        const size_t k = (size_t)p.top_k;
        if (!view.complete && n_view < k) return false;
        std::vector<uint32_t> order = sorted_prefix(logits, n_view, mx_all, tau_for_count(logits, n_view, mx_all, k),
                                                    [&](const std::vector<uint32_t>& c) { return c.size() >= k; });
        order.resize(std::min(order.size(), k));
        shrink_from_all(order);
    }
    if (p.top_p >= 0.0f) {  // top_p_filtering: keep through the first token that pushes the mass past p
        size_t cut = 0;
        bool found = false;
        if (all) {
            bool exact = false;
            float sum = view_exps(exact);
            have_exps = true;
            exps_sum = sum;
            std::vector<uint32_t> order;
            for (int attempt = 0; attempt < 2; ++attempt) {
                const float scale = sum > 0.0f ? 1.0f / sum : 1.0f;
                bool ambiguous = false;
                found = false;
                auto scan = [&](const std::vector<uint32_t>& c) {
                    float cumulative = 0.0f;
                    for (size_t i = 0; i < c.size(); ++i) {
                        const float before = cumulative;
                        cumulative += exps[c[i]] * scale;
                        if (cumulative > p.top_p) {
                            cut = i;
                            found = true;
                            // with an out-of-order sum the crossing is only trusted when it is not within rounding of p
                            const float eps = 2e-4f * cumulative;
                            if (!exact && (cumulative - p.top_p <= eps || p.top_p - before <= eps)) ambiguous = true;
                            return true;
                        }
                    }
                    return false;
                };
                if (attempt == 1) {
                    scan(order);  // the same sorted candidates, now against the in-order sum
                } else {
                    // a peaked distribution crosses p within a few units of the maximum: one cheap threshold pass; otherwise
                    // the histogram picks the threshold
                    order = sorted_prefix(logits, n_view, mx_all, 6.0f, [&](const std::vector<uint32_t>& c) { return scan(c) || true; });
                }
                if (!found) {
                    const float tau0 = tau_for_mass(logits, exps.data(), n_view, mx_all, (double)p.top_p * (double)sum);
                    order = sorted_prefix(logits, n_view, mx_all, tau0, scan);
                    if (!found) {
                        scan(order);
                        if (!found && !view.complete) return false;  // the crossing lies beyond the candidates
                        if (!found && !exact && order.size() == n_view) ambiguous = true;  // total mass within rounding of p
                    }
                }
                if (!ambiguous) break;
                if (!view.complete) return false;  // the in-order sum needs every logit
                sum = running_sum(exps.data(), n_view);
                exps_sum = sum;
                exact = true;
            }
            if (found) {
                order.resize(cut + 1);
                shrink_from_all(order);
            }
        } else if (!vals.empty()) {
            softmax(vals, probs);
            float mx = kNegInf;
            for (float v : vals) mx = std::max(mx, v);
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
            std::vector<uint32_t> order = sorted_prefix(vals.data(), vals.size(), mx, 4.0f, scan);
            if (!found) scan(order);
            if (found) {
                order.resize(cut + 1);
                shrink(order);
            }
        }
    }
    if (p.min_p >= 0.0f) {  // min_p_filtering
        if (all) {
            float sum = exps_sum;
            if (!have_exps) {
                bool exact = false;
                sum = view_exps(exact);
            }
            const float scale = sum > 0.0f ? 1.0f / sum : 1.0f;
            const float max_prob = std::max(0.0f, 1.0f * scale);  // exp(0) * scale
            const float cutoff = max_prob * p.min_p;
            // prob >= cutoff  <=>  logit >= max + ln(min_p) up to rounding: gather with slack, decide exactly
            const float slack = p.min_p > 0.0f ? -std::log(p.min_p) + 1e-3f : std::numeric_limits<float>::infinity();
            if (!view.complete && !(view.floor <= mx_all - slack)) return false;  // survivors may lie outside the candidates
            std::vector<uint32_t> pos;
            for (size_t i = 0; i < n_view; ++i)
                if (!(logits[i] < mx_all - slack) && !(exps[i] * scale < cutoff)) pos.push_back((uint32_t)i);
            if (pos.size() < vocab) shrink_from_all(pos);
        } else if (!vals.empty()) {
            softmax(vals, probs);
            float max_prob = 0.0f;
            for (float q : probs) max_prob = std::max(max_prob, q);
            const float cutoff = max_prob * p.min_p;
            std::vector<uint32_t> pos;
            for (size_t i = 0; i < probs.size(); ++i)
                if (!(probs[i] < cutoff)) pos.push_back((uint32_t)i);
            shrink(pos);
        }
    }
    const float temp = p.temperature < 1e-5f ? 1.0f : p.temperature;
    if (all) {  // temperature only: the whole vocabulary is the distribution
        if (!view.complete) return false;
        ids.resize(vocab);
        for (size_t i = 0; i < vocab; ++i) ids[i] = (uint32_t)i;
        probs.resize(vocab);
        // (x / t) - max(x / t) == (x - max) / t for t > 0 up to rounding
        bool exact = false;
        const float sum = vec_exp_sum(logits, vocab, mx_all, 1.0f / temp, probs.data(), exact);
        if (sum > 0.0f) {
            const float scale = 1.0f / sum;
            for (float& q : probs) q *= scale;
        }
        return true;
    }
    for (float& v : vals) v /= temp;
    softmax(vals, probs);
    return true;
}

}  // namespace

void sampling_distribution(const float* logits, size_t vocab, const SamplingParams& p, std::vector<uint32_t>& ids, std::vector<float>& probs)
{
    ids.clear();
    probs.clear();
    if (vocab == 0) return;
    const VocabView view{logits, vocab, nullptr, true, vec_max(logits, vocab), kNegInf, 0.0f};
    (void)sampling_distribution_view(view, vocab, p, ids, probs);
}

bool sampling_distribution_candidates(const uint32_t* cand_ids, const float* cand_vals, size_t n_cand, float mx, float floor, float sum,
                                      size_t vocab, const SamplingParams& p, std::vector<uint32_t>& ids, std::vector<float>& probs)
{
    ids.clear();
    probs.clear();
    if (vocab == 0) return true;
    if (n_cand == 0 || !std::isfinite(mx) || !(sum > 0.0f)) return false;
    if (n_cand >= vocab) return false;  // nothing was cut: the plain path is the same work
    // ascending token id = the vocabulary's position order (the reference's sort is stable in it)
    std::vector<uint32_t> order(n_cand);
    for (size_t i = 0; i < n_cand; ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cand_ids[a] < cand_ids[b]; });
    std::vector<uint32_t> tok(n_cand);
    std::vector<float> val(n_cand);
    for (size_t i = 0; i < n_cand; ++i) {
        tok[i] = cand_ids[order[i]];
        val[i] = cand_vals[order[i]];
    }
    const VocabView view{val.data(), n_cand, tok.data(), false, mx, floor, sum};
    return sampling_distribution_view(view, vocab, p, ids, probs);
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
