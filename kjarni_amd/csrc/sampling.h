// Logits processors and token sampling of the decoder generation loop (host side).
//
//   apply_repetition_penalty_mut / apply_no_repeat_ngram   crates/kjarni-transformers/src/common/sampling.rs:8-57
//   sample_token (top-k -> top-p -> min-p -> temperature -> softmax -> draw)   sampling.rs:81-114, 131-184
//   softmax_inplace                                        crates/kjarni-transformers/src/activations.rs:223-242
//
// The reference sorts the whole vocabulary twice per token; the filters here produce the same survivor set from a
// thresholded candidate list (everything outside the list sorts after everything inside it), so a 128k vocabulary
// costs one exp pass instead of two full sorts.
#pragma once
#include <cstdint>
#include <vector>

namespace kjarni {

struct SamplingParams {
    float temperature = 1.0f;
    int64_t top_k = -1;   // < 0: not set
    float top_p = -1.0f;  // < 0: not set
    float min_p = -1.0f;  // < 0: not set
};

void apply_repetition_penalty(float* logits, size_t vocab, const std::vector<uint32_t>& tokens, float penalty);
void apply_no_repeat_ngram(float* logits, size_t vocab, const std::vector<uint32_t>& tokens, size_t ngram);
inline void apply_repetition_penalty(std::vector<float>& logits, const std::vector<uint32_t>& tokens, float penalty)
{
    apply_repetition_penalty(logits.data(), logits.size(), tokens, penalty);
}
inline void apply_no_repeat_ngram(std::vector<float>& logits, const std::vector<uint32_t>& tokens, size_t ngram)
{
    apply_no_repeat_ngram(logits.data(), logits.size(), tokens, ngram);
}

// Greedy: the LAST maximum (Iterator::max_by keeps the later of equal elements).
uint32_t argmax_last(const float* logits, size_t vocab);
inline uint32_t argmax_last(const std::vector<float>& logits) { return argmax_last(logits.data(), logits.size()); }

// The distribution sample_token draws from: surviving token ids in ascending order and their probabilities.
void sampling_distribution(const float* logits, size_t vocab, const SamplingParams& p, std::vector<uint32_t>& ids, std::vector<float>& probs);
inline void sampling_distribution(const std::vector<float>& logits, const SamplingParams& p, std::vector<uint32_t>& ids,
                                  std::vector<float>& probs)
{
    sampling_distribution(logits.data(), logits.size(), p, ids, probs);
}
// The same distribution from what the device hands over instead of the logits (llm_kernels.hip, launch_sample_candidates):
// every token with a logit >= floor (ids / values in any order), the maximum mx and the sum of exp(logit - mx) over the
// whole vocabulary.  False: these candidates do not decide it (a filter reaches past them, or a top-p crossing is within
// the rounding of the device's sum) -- fetch the logits and call sampling_distribution().
bool sampling_distribution_candidates(const uint32_t* cand_ids, const float* cand_vals, size_t n_cand, float mx, float floor, float sum,
                                      size_t vocab, const SamplingParams& p, std::vector<uint32_t>& ids, std::vector<float>& probs);
// sample_from_probs over the full vocabulary: first index whose running sum reaches `uniform`, else vocab - 1.
uint32_t sample_from_distribution(const std::vector<uint32_t>& ids, const std::vector<float>& probs, float uniform, size_t vocab);

// rand::Rng::gen::<f32>() analogue: 24 random bits in [0, 1).  xoshiro256**, seeded from the OS or explicitly.
class UniformRng {
public:
    UniformRng();
    explicit UniformRng(uint64_t seed) { reseed(seed); }
    void reseed(uint64_t seed);
    float next();

private:
    uint64_t s_[4];
};

}  // namespace kjarni
