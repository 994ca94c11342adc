// WhisperModel: log-mel front end, convolutional stem, pre-norm encoder and the cross-attention decoder
// of a Whisper checkpoint resident in HBM, plus the greedy transcription loop.
//
//   WhisperConfig / tensor names   crates/kjarni-models/src/models/whisper/config.rs:11-190
//   front end                      crates/kjarni-transformers/src/audio/mel.rs:44-391
//   encoder                        cpu/encoder_decoder/cpu_encoder.rs:193-262 (hidden-state input, pre-norm, final norm)
//   decoder                        cpu/encoder_decoder/cpu_decoder.rs:399-516
//   greedy loop, prompt, filtering crates/kjarni-models/src/models/whisper/transcriber.rs:122-460
#pragma once
#include "device_arena.h"
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "encoder.h"

namespace kjarni {

struct WhisperConfig {
    int d_model = 0, encoder_layers = 0, decoder_layers = 0, heads = 0, encoder_ffn = 0, decoder_ffn = 0;
    int vocab = 0, max_source_positions = 1500, max_target_positions = 448, num_mel_bins = 80;
    uint32_t eos_token_id = 50257;
    bool scale_embedding = false;
    static WhisperConfig from_json(const std::string& text);
};

// Decode side of the byte-level BPE tokenizer (`tokenizers` ByteLevel decoder) + token lookup.
class ByteLevelVocab {
public:
    void load(const std::string& tokenizer_json_path);
    // Tokenizer::decode(ids, skip_special_tokens): tokens joined, mapped back to bytes, lossy UTF-8.
    std::string decode(const std::vector<uint32_t>& ids, bool skip_special) const;
    bool token_to_id(const std::string& token, uint32_t& id) const;
    size_t size() const { return id_to_token_.size(); }

private:
    std::vector<std::string> id_to_token_;
    std::vector<uint8_t> has_token_, special_;
    std::unordered_map<std::string, uint32_t> token_to_id_;
};

class WhisperModel {
public:
    static constexpr int kChunkSamples = 480000, kFrames = 3000, kNfft = 400, kHop = 160;
    static constexpr uint32_t kSot = 50258, kEot = 50257, kTranscribe = 50359, kTranslate = 50360,
                              kNoTimestamps = 50363, kTimestampBegin = 50364, kFirstSpecial = 50257;

    static std::unique_ptr<WhisperModel> load(const std::string& dir, int device);
    ~WhisperModel();
    WhisperModel(const WhisperModel&) = delete;
    WhisperModel& operator=(const WhisperModel&) = delete;

    const WhisperConfig& config() const { return cfg_; }
    const ByteLevelVocab& vocab() const { return vocab_; }
    int device() const { return device_; }
    uint32_t eos_token_id() const { return eos_; }

    // compute_mel_spectrogram with MelConfig::whisper(): host samples (any length) -> log-mel on the device;
    // mel_out (host, [n_mels, 3000] as the reference lays it out) may be null.
    void log_mel(const float* samples, int64_t n_samples, float* mel_out);
    // AudioConvFrontend::forward + encoder on a HOST mel [n_mels, frames] (frames even); encode_mel of transcriber.rs:122-141.
    void encode_mel(const float* mel, int frames);
    // log_mel + encode on the device-resident mel.
    void encode_audio(const float* samples, int64_t n_samples);
    int encoder_frames() const { return enc_frames_; }
    void encoder_output(float* out) const;  // [enc_frames, d_model] to the host

    // Decoder over the current encoder output.  begin_decode() projects the cross-attention K/V and clears
    // the self-attention cache; forward() runs `n` new tokens (n <= 8) and returns the last row's logits
    // (device pointer, valid until the next call).
    void begin_decode();
    const float* forward(const uint32_t* ids, int n);
    int cache_len() const { return cache_len_; }
    void last_hidden(float* out, int rows) const;  // final-normed hidden states of the last forward() call
    void logits_to_host(float* out) const;
    uint32_t pick_token(bool timestamps);

    // decode_chunk (transcriber.rs:144-240) up to the generated ids; on_token returning false stops early.
    std::vector<uint32_t> greedy(const std::vector<uint32_t>& prompt, bool timestamps, size_t max_tokens,
                                 const std::function<bool(uint32_t)>& on_token);

    // Several 30-second chunks decoded in lock step ("lanes"): the chunks of a long recording are independent
    // (transcriber.rs:85-120 cuts them up front, every chunk starts from the same prompt), and a decoder step is bound by
    // kernel launches, not by the rows it carries -- so up to kMaxLanes chunks share each launch.  begin_decode_lane(l)
    // projects the current encoder output into lane l's cross-attention K/V; greedy_lanes then runs all lanes to their
    // EOS and returns, per lane, exactly what greedy() would have returned for that chunk alone.
    static constexpr int kMaxLanes = 8;
    void begin_decode_lane(int lane);
    std::vector<std::vector<uint32_t>> greedy_lanes(int lanes, const std::vector<uint32_t>& prompt, bool timestamps, size_t max_tokens,
                                                    const std::function<bool()>& keep_going);

private:
    WhisperModel() = default;
    float* upload(const std::vector<float>& host);
    float* dalloc(size_t floats);
    void conv_and_encode(const float* mel_t, int ld_mel, int frames);  // device mel, time-major
    // One decoder pass over n new tokens whose ids are on the device.  device_pos: position / key count / cache
    // row are read from dpos_ on the device (graph-capturable: no launch parameter depends on the step).
    void decoder_pass(const uint32_t* ids_dev, int n, bool device_pos);
    void enqueue_pick(bool timestamps, bool record);
    hipGraphExec_t step_graph(bool timestamps);
    void ensure_lanes();
    void decoder_pass_lanes(int lanes, bool device_pos);
    hipGraphExec_t lane_step_graph(bool timestamps, int lanes);

    struct EncLayer {
        float *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
    };
    struct DecLayer {
        float *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b;                               // self attention (Q|K|V fused)
        float *cq, *cbq, *ckv, *cbkv, *co, *cbo, *ln2_g, *ln2_b;                    // cross attention
        float *w1, *b1, *w2, *b2, *ln3_g, *ln3_b;                                   // feed forward
        float *self_k, *self_v, *cross_kv;                                          // caches
    };

    WhisperConfig cfg_;
    ByteLevelVocab vocab_;
    int device_ = 0;
    uint32_t eos_ = 50257;
    DeviceArena arena_;   // every device buffer of the model
    size_t weight_bytes_ = 0;

    // front end
    int k_dft_ = 0, n_dft_ = 0, k_mel_ = 0, n_mel_pad_ = 0, k_conv1_ = 0;
    float *window_ = nullptr, *w_dft_ = nullptr, *w_mel_ = nullptr;
    float *conv1_w_ = nullptr, *conv1_b_ = nullptr, *conv2_w_ = nullptr, *conv2_b_ = nullptr, *enc_pos_ = nullptr;
    int enc_pos_rows_ = 0;
    float *enc_ln_g_ = nullptr, *enc_ln_b_ = nullptr;
    std::vector<EncLayer> enc_;
    // decoder
    float *tok_emb_ = nullptr, *dec_pos_ = nullptr, *dec_ln_g_ = nullptr, *dec_ln_b_ = nullptr, *lm_head_ = nullptr;
    std::vector<DecLayer> dec_;

    // workspace
    float *audio_ = nullptr, *frames_ = nullptr, *dft_ = nullptr, *power_ = nullptr, *melraw_ = nullptr, *mel_t_ = nullptr;
    uint32_t* max_scratch_ = nullptr;
    float *cols_ = nullptr, *conv1_out_ = nullptr, *hidden_ = nullptr, *normed_ = nullptr, *qkv_ = nullptr, *ctx_ = nullptr,
          *mid_ = nullptr;
    uint32_t* ones_ = nullptr;
    size_t audio_cap_ = 0;
    int max_frames_ = 0, enc_frames_ = 0;
    // decoder workspace
    float *dh_ = nullptr, *dn_ = nullptr, *dq_ = nullptr, *dctx_ = nullptr, *dmid_ = nullptr, *dlast_ = nullptr, *logits_ = nullptr;
    uint32_t* dids_ = nullptr;
    int32_t *dtoken_ = nullptr, *dhist_ = nullptr;
    int *dpos_ = nullptr, *dcount_ = nullptr;
    float* att_scratch_ = nullptr;
    float* gemm_scratch_ = nullptr;  // partial tiles of the encoder's mid-size GEMM route (gemm.hip)
    size_t gemm_scratch_floats_ = 0;
    int cache_len_ = 0, cache_cap_ = 0, last_rows_ = 0, hist_cap_ = 0;
    hipStream_t stream_ = nullptr;
    hipGraphExec_t graphs_[2] = {nullptr, nullptr};
    // lanes (allocated on first use): caches interleaved by lane ([position][lane][H]) so that a step's K / V rows of all
    // lanes are consecutive cache rows, cross K/V lane-major
    std::vector<float*> lane_self_k_, lane_self_v_, lane_cross_kv_;
    float* lane_logits_ = nullptr;
    int32_t *lane_tokens_ = nullptr, *lane_hist_ = nullptr;
    int *lane_counts_ = nullptr, *drow_ = nullptr;
    unsigned long long* dbest_ = nullptr;  // per-lane argmax keys of the two-stage pick
    hipGraphExec_t lane_graphs_[2][kMaxLanes + 1] = {};
    static constexpr int kSelfSplits = 16, kCrossSplits = 12;
};

}  // namespace kjarni
