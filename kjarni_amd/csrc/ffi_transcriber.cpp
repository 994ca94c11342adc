// kjarni_transcriber_*: the Whisper transcription API as a C ABI.
//
// The reference's kjarni-ffi crate has no transcriber group (crates/kjarni-ffi/src/ holds callback, chat,
// classifier, embedder, error, indexer, reranker, searcher); these entry points mirror the Rust API of
// crates/kjarni/src/transcriber/{builder,model,types,validation}.rs in the style of the other groups and
// reuse the token-callback types the FFI already declares (crates/kjarni-ffi/src/callback.rs:36-47).
//
// Host logic here (integer / byte work the reference also runs on the CPU): WAV decoding
// (crates/kjarni-transformers/src/audio/loader.rs:125-300), resampling and the chunk loop
// (crates/kjarni/src/transcriber/model.rs:91-176, 333-356), prompt tokens, timestamp parsing and stitching
// (crates/kjarni-models/src/models/whisper/transcriber.rs:85-455).  Mel, encoder and decoder run on the GPU.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

#include "host_util.h"
#include "../../include/kjarni_hip.h"
#include "ffi_common.h"
#include "registry.h"
#include "unicode.h"
#include "whisper.h"

using namespace kjarni;

namespace {

std::string lower_ascii(std::string s)
{
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

struct TranscriberFailure : std::runtime_error {
    KjarniErrorCode code;
    TranscriberFailure(KjarniErrorCode c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- WAV (hound 3.5 semantics as load_wav_reader uses it) -------------------------------------------
struct Wav {
    std::vector<float> samples;  // interleaved
    uint32_t rate = 0;
    uint16_t channels = 0;
};

Wav parse_wav(const std::string& data)
{
    auto u16 = [&](size_t p) { return (uint16_t)((uint8_t)data[p] | ((uint8_t)data[p + 1] << 8)); };
    auto u32 = [&](size_t p) { return (uint32_t)u16(p) | ((uint32_t)u16(p + 2) << 16); };
    if (data.size() < 12 || data.compare(0, 4, "RIFF") != 0 || data.compare(8, 4, "WAVE") != 0)
        throw std::runtime_error("Failed to open WAV file: no RIFF tag found");
    size_t pos = 12;
    bool have_fmt = false;
    uint16_t tag = 0, channels = 0, bits = 0;
    uint32_t rate = 0;
    const char* pcm = nullptr;
    size_t pcm_len = 0;
    while (pos + 8 <= data.size()) {
        const std::string id = data.substr(pos, 4);
        const size_t size = u32(pos + 4);
        const size_t body = pos + 8;
        const size_t avail = std::min(size, data.size() - body);
        if (id == "fmt ") {
            if (avail < 16) throw std::runtime_error("Failed to open WAV file: invalid fmt chunk size");
            tag = u16(body);
            channels = u16(body + 2);
            rate = u32(body + 4);
            bits = u16(body + 14);
            if (tag == 0xFFFE && avail >= 26) tag = u16(body + 24);  // WAVE_FORMAT_EXTENSIBLE: sub-format
            have_fmt = true;
        } else if (id == "data") {
            pcm = data.data() + body;
            pcm_len = avail;
            break;  // hound reads samples from the first data chunk
        }
        pos = body + size + (size & 1);
    }
    if (!have_fmt || !pcm) throw std::runtime_error("Failed to open WAV file: missing fmt or data chunk");
    if (channels == 0) throw std::runtime_error("Failed to open WAV file: file contains zero channels");
    Wav w;
    w.rate = rate;
    w.channels = channels;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(pcm);
    if (tag == 3) {
        if (bits != 32) throw std::runtime_error("Failed to read float samples: unsupported float width");
        w.samples.resize(pcm_len / 4);
        std::memcpy(w.samples.data(), p, w.samples.size() * 4);
    } else if (tag == 1) {
        if (bits != 8 && bits != 16 && bits != 24 && bits != 32) throw std::runtime_error("Unsupported bit depth: " + std::to_string(bits));
        const float max_value = (float)(1u << (bits - 1));  // loader.rs:136
        switch (bits) {
        case 8:
            w.samples.resize(pcm_len);
            for (size_t i = 0; i < pcm_len; ++i) w.samples[i] = (float)((int)p[i] - 128) / max_value;
            break;
        case 16:
            w.samples.resize(pcm_len / 2);
            for (size_t i = 0; i < w.samples.size(); ++i) w.samples[i] = (float)(int16_t)(p[2 * i] | (p[2 * i + 1] << 8)) / max_value;
            break;
        case 24:
            w.samples.resize(pcm_len / 3);
            for (size_t i = 0; i < w.samples.size(); ++i) {
                int32_t v = p[3 * i] | (p[3 * i + 1] << 8) | (p[3 * i + 2] << 16);
                if (v & 0x800000) v -= 1 << 24;
                w.samples[i] = (float)v / max_value;
            }
            break;
        case 32:
            w.samples.resize(pcm_len / 4);
            for (size_t i = 0; i < w.samples.size(); ++i) {
                int32_t v;
                std::memcpy(&v, p + 4 * i, 4);
                w.samples[i] = (float)v / max_value;
            }
            break;
        default: throw std::runtime_error("Unsupported bit depth: " + std::to_string(bits));
        }
    } else {
        throw std::runtime_error("Failed to open WAV file: unsupported format tag " + std::to_string(tag));
    }
    return w;
}

// loader.rs:223-252 / transcriber/model.rs:333-356
std::vector<float> resample_linear(const std::vector<float>& s, uint32_t from, uint32_t to)
{
    if (from == to || s.empty()) return s;
    const double ratio = (double)to / (double)from;
    const size_t out_len = (size_t)std::ceil((double)s.size() * ratio);
    std::vector<float> out(out_len);
    for (size_t i = 0; i < out_len; ++i) {
        const double src = (double)i / ratio;
        const size_t lo = (size_t)std::floor(src);
        if (lo >= s.size()) {
            out[i] = 0.0f;
            continue;
        }
        const size_t hi = std::min(lo + 1, s.size() - 1);
        const float frac = (float)(src - (double)lo);
        out[i] = s[lo] + (s[hi] - s[lo]) * frac;
    }
    return out;
}

// load_audio with the Transcriber's loader config (model.rs:291-299): mono, 16 kHz, no normalisation.
std::vector<float> load_audio_16k_mono(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to open WAV file: " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    Wav w = parse_wav(ss.str());
    std::vector<float> mono;
    if (w.channels > 1) {  // convert_to_mono (loader.rs:214-220): chunk sum / channels, a short tail chunk included
        const size_t ch = w.channels;
        mono.reserve(w.samples.size() / ch + 1);
        for (size_t i = 0; i < w.samples.size(); i += ch) {
            float sum = 0.0f;
            for (size_t c = i; c < std::min(i + ch, w.samples.size()); ++c) sum += w.samples[c];
            mono.push_back(sum / (float)ch);
        }
    } else {
        mono = std::move(w.samples);
    }
    return resample_linear(mono, w.rate, 16000);
}

struct Segment {
    float start, end;
    std::string text;
};
struct ChunkResult {
    std::vector<Segment> segments;
    std::string text;
};

bool is_blank(const std::string& s)  // str::trim().is_empty()
{
    std::vector<uint32_t> cps;
    if (!unicode::decode_utf8(s.data(), s.size(), cps)) return false;
    for (uint32_t cp : cps)
        if (!unicode::is_whitespace(cp)) return false;
    return true;
}

// transcriber.rs:340-409
std::vector<Segment> parse_timestamp_segments(const std::vector<uint32_t>& ids, const ByteLevelVocab& vocab, float offset)
{
    std::vector<Segment> segs;
    bool have_start = false;
    float start = 0.0f;
    std::vector<uint32_t> cur;
    for (uint32_t id : ids) {
        if (id >= WhisperModel::kTimestampBegin) {
            const float t = (float)(id - WhisperModel::kTimestampBegin) * 0.02f + offset;
            if (!have_start) {
                have_start = true;
                start = t;
            } else {
                const std::string text = vocab.decode(cur, true);
                if (!is_blank(text)) segs.push_back({start, t, text});
                start = t;
                cur.clear();
            }
        } else if (id < WhisperModel::kFirstSpecial) {
            cur.push_back(id);
        }
    }
    if (have_start && !cur.empty()) {
        const std::string text = vocab.decode(cur, true);
        if (!is_blank(text)) segs.push_back({start, start + 30.0f, text});
    }
    return segs;
}

// transcriber.rs:301-338
ChunkResult finalize_chunk(const std::vector<uint32_t>& ids, const ByteLevelVocab& vocab, bool timestamps, float offset)
{
    ChunkResult r;
    if (timestamps) {
        r.segments = parse_timestamp_segments(ids, vocab, offset);
        for (const Segment& s : r.segments) r.text += s.text;
        return r;
    }
    std::vector<uint32_t> text_ids;
    for (uint32_t id : ids)
        if (id < WhisperModel::kFirstSpecial) text_ids.push_back(id);
    r.text = vocab.decode(text_ids, true);
    r.segments.push_back({offset, offset + 30.0f, r.text});
    return r;
}

bool is_chunk_boundary(float t)  // transcriber.rs:452-455
{
    const float rem = std::fmod(t, 30.0f);
    return rem < 0.02f || (30.0f - rem) < 0.02f;
}

}  // namespace

struct KjarniTranscriber {
    std::unique_ptr<WhisperModel> model;
    std::string model_name, language;
    bool has_language = false, translate = false, timestamps = false, quiet = false;
    size_t max_tokens = 448;
    std::mutex mu;

    void report(KjarniTranscriptionProgressFn cb, void* user, KjarniTranscriptionStage stage, size_t cur, size_t total,
                const std::string* msg) const
    {
        if (cb) {
            KjarniTranscriptionProgress p;
            p.stage = stage;
            p.current = cur;
            p.total = total;
            p.message = msg ? msg->c_str() : nullptr;
            cb(p, user);
        } else if (!quiet) {  // model.rs:319-347
            static const char* names[] = {"Loading audio", "Encoding", "Decoding", "Stitching"};
            if (stage == KJARNI_TRANSCRIPTION_LOADING_AUDIO) {
                if (msg) std::fprintf(stderr, "Loading audio: %s\n", msg->c_str());
            } else if (stage == KJARNI_TRANSCRIPTION_STITCHING) {
                std::fprintf(stderr, "\n");
            } else if (total > 0) {
                std::fprintf(stderr, "\r  %s [%zu/%zu]%s%s", names[stage], cur + 1, total, msg ? " " : "", msg ? msg->c_str() : "");
            }
        }
    }

    std::vector<uint32_t> prompt() const  // transcriber.rs:273-299
    {
        std::vector<uint32_t> t{WhisperModel::kSot};
        const std::string lang = has_language ? lower_ascii(language) : "en";
        uint32_t id = 50259;
        if (!model->vocab().token_to_id("<|" + lang + "|>", id)) id = 50259;
        t.push_back(id);
        t.push_back(translate ? WhisperModel::kTranslate : WhisperModel::kTranscribe);
        if (!timestamps) t.push_back(WhisperModel::kNoTimestamps);
        return t;
    }

    // transcribe_audio_inner (crates/kjarni/src/transcriber/model.rs:108-176) / stream_audio (:196-262)
    void run(const std::vector<float>& samples, float duration_secs, KjarniTranscriptionProgressFn progress, void* puser,
             KjarniTokenCallbackFn on_token, void* tuser, const KjarniCancelToken* cancel, KjarniTranscription* out)
    {
        std::lock_guard<std::mutex> lock(mu);
        const size_t chunk = (size_t)WhisperModel::kChunkSamples;
        const size_t total_chunks = samples.empty() ? 0 : (samples.size() <= chunk ? 1 : (samples.size() + chunk - 1) / chunk);
        std::vector<ChunkResult> results;
        std::vector<float> buf(chunk);
        const std::vector<uint32_t> prompt_ids = prompt();
        bool stopped = false;
        // Without a per-token callback the chunks of a long recording are decoded several at a time in lock step
        // (WhisperModel::greedy_lanes): same tokens per chunk, one set of kernel launches per step for all of them.
        int lanes = WhisperModel::kMaxLanes;
        if (const char* v = std::getenv("KJARNI_HIP_WHISPER_LANES")) lanes = std::max(1, std::min(WhisperModel::kMaxLanes, std::atoi(v)));
        const bool batched = !on_token && total_chunks > 1 && lanes > 1;
        auto load_chunk = [&](size_t i) {
            const size_t begin = i * chunk, n = std::min(chunk, samples.size() - begin);
            std::fill(buf.begin(), buf.end(), 0.0f);  // chunk_audio: zero-padded to 30 s (transcriber.rs:87-119)
            std::memcpy(buf.data(), samples.data() + begin, n * sizeof(float));
        };
        for (size_t i0 = 0; batched && i0 < total_chunks && !stopped; i0 += (size_t)lanes) {
            const size_t nb = std::min((size_t)lanes, total_chunks - i0);
            for (size_t j = 0; j < nb; ++j) {  // the reference's event order (encoding i, decoding i), chunk by chunk
                const size_t i = i0 + j;
                if (kjarni_cancel_token_is_cancelled(cancel)) throw TranscriberFailure(KJARNI_ERROR_CANCELLED, "Transcription cancelled");
                const std::string msg = "Chunk " + std::to_string(i + 1) + "/" + std::to_string(total_chunks);
                report(progress, puser, KJARNI_TRANSCRIPTION_ENCODING, i, total_chunks, &msg);
                load_chunk(i);
                model->encode_audio(buf.data(), (int64_t)chunk);
                model->begin_decode_lane((int)j);
                report(progress, puser, KJARNI_TRANSCRIPTION_DECODING, i, total_chunks, &msg);
            }
            const std::vector<std::vector<uint32_t>> ids =
                model->greedy_lanes((int)nb, prompt_ids, timestamps, max_tokens, [&] { return !kjarni_cancel_token_is_cancelled(cancel); });
            if (kjarni_cancel_token_is_cancelled(cancel)) throw TranscriberFailure(KJARNI_ERROR_CANCELLED, "Transcription cancelled");
            for (size_t j = 0; j < nb; ++j) results.push_back(finalize_chunk(ids[j], model->vocab(), timestamps, (float)(i0 + j) * 30.0f));
        }
        for (size_t i = 0; !batched && i < total_chunks && !stopped; ++i) {
            if (kjarni_cancel_token_is_cancelled(cancel)) throw TranscriberFailure(KJARNI_ERROR_CANCELLED, "Transcription cancelled");
            const float offset = (float)i * 30.0f;
            const std::string msg = "Chunk " + std::to_string(i + 1) + "/" + std::to_string(total_chunks);
            report(progress, puser, KJARNI_TRANSCRIPTION_ENCODING, i, total_chunks, &msg);
            load_chunk(i);
            model->encode_audio(buf.data(), (int64_t)chunk);
            report(progress, puser, KJARNI_TRANSCRIPTION_DECODING, i, total_chunks, &msg);
            const ByteLevelVocab& vocab = model->vocab();
            const std::vector<uint32_t> ids = model->greedy(prompt_ids, timestamps, max_tokens, [&](uint32_t id) {
                if (kjarni_cancel_token_is_cancelled(cancel)) {
                    stopped = true;
                    return false;
                }
                if (!on_token) return true;
                const std::string text = vocab.decode({id}, false);
                KjarniToken tok;
                tok.text = text.c_str();
                tok.token_id = id;
                tok.is_special = id >= WhisperModel::kFirstSpecial;
                if (!on_token(tok, tuser)) {
                    stopped = true;  // the receiver went away: the stream ends
                    return false;
                }
                return true;
            });
            results.push_back(finalize_chunk(ids, vocab, timestamps, offset));
        }
        if (kjarni_cancel_token_is_cancelled(cancel)) throw TranscriberFailure(KJARNI_ERROR_CANCELLED, "Transcription cancelled");
        report(progress, puser, KJARNI_TRANSCRIPTION_STITCHING, 0, 0, nullptr);
        // stitch_segments + merge_boundary_segments (transcriber.rs:412-449)
        std::string text;
        std::vector<Segment> merged;
        for (const ChunkResult& r : results) {
            text += r.text;
            for (const Segment& s : r.segments) {
                if (!merged.empty() && std::fabs(merged.back().end - s.start) < 0.02f && is_chunk_boundary(merged.back().end)) {
                    merged.back().end = s.end;
                    merged.back().text += s.text;
                } else {
                    merged.push_back(s);
                }
            }
        }
        KjarniTranscription t;
        std::memset(&t, 0, sizeof t);
        t.text = dup_cstr(text);
        t.language = dup_cstr(has_language ? language : "en");
        t.duration_secs = duration_secs;
        if (!merged.empty()) {
            t.segments = static_cast<KjarniTranscriptionSegment*>(std::calloc(merged.size(), sizeof(KjarniTranscriptionSegment)));
            if (!t.segments) throw std::bad_alloc();
            for (size_t i = 0; i < merged.size(); ++i) {
                t.segments[i].start = merged[i].start;
                t.segments[i].end = merged[i].end;
                t.segments[i].text = dup_cstr(merged[i].text);
            }
            t.num_segments = merged.size();
        }
        *out = t;
    }
};

namespace {

template <class F>
KjarniErrorCode transcriber_guarded(F&& fn)
{
    try {
        fn();
        return KJARNI_OK;
    } catch (const TranscriberFailure& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const std::exception& e) {
        set_last_error(std::string("Transcription failed: ") + e.what());
        return KJARNI_ERROR_INFERENCE_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_INFERENCE_FAILED;
    }
}

// builder.rs:145-159 plus the sizes this library can also serve from a local directory
const char* resolve_repo(const std::string& id)
{
    const std::string s = lower_ascii(id);
    if (s == "whisper-small" || s == "whisper_small" || s == "small") return "openai/whisper-small";
    if (s == "whisper-large-v3" || s == "whisper_large_v3" || s == "large-v3" || s == "large") return "openai/whisper-large-v3";
    if (s == "whisper-tiny" || s == "tiny") return "openai/whisper-tiny";
    if (s == "whisper-base" || s == "base") return "openai/whisper-base";
    if (s == "whisper-medium" || s == "medium") return "openai/whisper-medium";
    return nullptr;
}

std::string canonical_name(const std::string& id)
{
    const char* repo = resolve_repo(id);
    if (!repo) return id;
    return std::string(repo).substr(std::strlen("openai/"));
}

}  // namespace

KJARNI_EXPORT KjarniTranscriberConfig kjarni_transcriber_config_default(void)
{
    KjarniTranscriberConfig c;
    std::memset(&c, 0, sizeof c);
    c.device = KJARNI_DEVICE_CPU;
    c.task = KJARNI_TASK_TRANSCRIBE;
    c.timestamps = 0;
    c.max_tokens_per_chunk = 448;
    c.quiet = 0;
    return c;
}

KJARNI_EXPORT KjarniErrorCode kjarni_transcriber_new(const KjarniTranscriberConfig* config, KjarniTranscriber** out)
{
    if (!out) return KJARNI_ERROR_NULL_POINTER;
    const KjarniTranscriberConfig dflt = kjarni_transcriber_config_default();
    const KjarniTranscriberConfig& c = config ? *config : dflt;
    for (const char* s : {c.cache_dir, c.model_name, c.model_path, c.language})
        if (s && !valid_utf8(s)) return KJARNI_ERROR_INVALID_UTF8;
    try {
        auto h = std::make_unique<KjarniTranscriber>();
        const std::string id = c.model_name ? c.model_name : "whisper-small";
        // validate_config (validation.rs:37-67)
        if (c.language) {
            const std::string lang = c.language;
            if (lang.empty()) throw InvalidConfig("Invalid config: Language code cannot be empty");
            if (lang.size() > 10) throw InvalidConfig("Invalid config: Language code too long: '" + lang + "'");
            h->language = lang;
            h->has_language = true;
        }
        const size_t max_tokens = c.max_tokens_per_chunk;
        if (max_tokens == 0) throw InvalidConfig("Invalid config: max_tokens_per_chunk must be > 0");
        if (max_tokens > 4096)
            throw InvalidConfig("Invalid config: max_tokens_per_chunk too large: " + std::to_string(max_tokens) + " (max 4096)");
        h->max_tokens = max_tokens;
        h->translate = c.task == KJARNI_TASK_TRANSLATE;
        h->timestamps = c.timestamps != 0;
        h->quiet = c.quiet != 0;
        std::string dir;
        if (c.model_path) {
            dir = c.model_path;
            h->model_name = c.model_name ? canonical_name(id) : dir;
        } else {
            const char* repo = resolve_repo(id);
            if (!repo)
                throw InvalidConfig("Invalid config: Unknown model: '" + id + "'. Try 'whisper-small' or 'whisper-large-v3'.");
            std::string repo_dir = repo;
            std::replace(repo_dir.begin(), repo_dir.end(), '/', '_');
            dir = (c.cache_dir ? std::string(c.cache_dir) : default_cache_dir()) + "/" + repo_dir;
            h->model_name = canonical_name(id);
        }
        struct stat sb;
        for (const char* f : {"config.json", "tokenizer.json", "model.safetensors"})
            if (::stat((dir + "/" + f).c_str(), &sb) != 0 &&
                !(std::string(f) == "model.safetensors" && ::stat((dir + "/model.safetensors.index.json").c_str(), &sb) == 0))
                throw ModelNotFound("Model load failed: " + dir + "/" + f + " is missing (models are not downloaded)");
        int device = 0;
        if (const char* e = std::getenv("KJARNI_HIP_DEVICE")) device = std::atoi(e);
        h->model = WhisperModel::load(dir, device);
        *out = h.release();
        return KJARNI_OK;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const ModelNotFound& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_MODEL_NOT_FOUND;
    } catch (const InvalidConfig& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;
    } catch (const std::exception& e) {
        set_last_error(std::string("Model load failed: ") + e.what());
        return KJARNI_ERROR_LOAD_FAILED;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_LOAD_FAILED;
    }
}

KJARNI_EXPORT void kjarni_transcriber_free(KjarniTranscriber* t) { delete t; }

KJARNI_EXPORT void kjarni_transcription_free(const KjarniTranscription* t)
{
    if (!t) return;
    std::free(t->text);
    std::free(t->language);
    if (t->segments) {
        for (size_t i = 0; i < t->num_segments; ++i) std::free(t->segments[i].text);
        std::free(t->segments);
    }
}

KJARNI_EXPORT KjarniErrorCode kjarni_transcriber_transcribe_audio_with_callbacks(
    KjarniTranscriber* t, const float* samples, size_t num_samples, uint32_t sample_rate, KjarniTranscriptionProgressFn progress,
    void* progress_user_data, KjarniTokenCallbackFn on_token, void* token_user_data, const KjarniCancelToken* cancel_token,
    KjarniTranscription* out)
{
    if (!t || !out || (num_samples && !samples)) return KJARNI_ERROR_NULL_POINTER;
    std::memset(out, 0, sizeof *out);
    return transcriber_guarded([&] {
        if (sample_rate == 0) throw TranscriberFailure(KJARNI_ERROR_INVALID_CONFIG, "Invalid config: sample_rate must be > 0");
        std::vector<float> s(samples, samples + num_samples);
        if (sample_rate != 16000) s = resample_linear(s, sample_rate, 16000);
        const float duration = (float)s.size() / 16000.0f;
        t->run(s, duration, progress, progress_user_data, on_token, token_user_data, cancel_token, out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_transcriber_transcribe_audio(KjarniTranscriber* t, const float* samples, size_t num_samples,
                                                                  uint32_t sample_rate, KjarniTranscription* out)
{
    return kjarni_transcriber_transcribe_audio_with_callbacks(t, samples, num_samples, sample_rate, nullptr, nullptr, nullptr, nullptr,
                                                              nullptr, out);
}

KJARNI_EXPORT KjarniErrorCode kjarni_transcriber_transcribe_file_with_callbacks(
    KjarniTranscriber* t, const char* path, KjarniTranscriptionProgressFn progress, void* progress_user_data,
    KjarniTokenCallbackFn on_token, void* token_user_data, const KjarniCancelToken* cancel_token, KjarniTranscription* out)
{
    if (!t || !path || !out) return KJARNI_ERROR_NULL_POINTER;
    std::memset(out, 0, sizeof *out);
    if (!valid_utf8(path)) return KJARNI_ERROR_INVALID_UTF8;
    return transcriber_guarded([&] {
        // validate_audio_path (validation.rs:11-34)
        const std::string p = path;
        struct stat sb;
        if (::stat(p.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode))
            throw TranscriberFailure(KJARNI_ERROR_MODEL_NOT_FOUND, "Invalid audio path: " + p);
        const size_t slash = p.rfind('/');
        const std::string name = slash == std::string::npos ? p : p.substr(slash + 1);
        const size_t dot = name.rfind('.');
        const std::string ext = (dot == std::string::npos || dot == 0) ? "" : lower_ascii(name.substr(dot + 1));
        if (ext != "wav" && ext != "mp3" && ext != "flac" && ext != "ogg")
            throw TranscriberFailure(KJARNI_ERROR_INVALID_CONFIG, "Unsupported audio format: " + ext);
        t->report(progress, progress_user_data, KJARNI_TRANSCRIPTION_LOADING_AUDIO, 0, 0, &p);
        if (ext != "wav")  // loader.rs:84-96 without the symphonia feature
            throw TranscriberFailure(KJARNI_ERROR_INFERENCE_FAILED,
                                     "Audio load failed: Format '" + ext + "' requires the 'symphonia' feature. Only WAV is supported by default.");
        std::vector<float> s;
        try {
            s = load_audio_16k_mono(p);
        } catch (const std::exception& e) {
            throw TranscriberFailure(KJARNI_ERROR_INFERENCE_FAILED, std::string("Audio load failed: ") + e.what());
        }
        const float duration = (float)s.size() / 16000.0f;
        t->run(s, duration, progress, progress_user_data, on_token, token_user_data, cancel_token, out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_transcriber_transcribe_file(KjarniTranscriber* t, const char* path, KjarniTranscription* out)
{
    return kjarni_transcriber_transcribe_file_with_callbacks(t, path, nullptr, nullptr, nullptr, nullptr, nullptr, out);
}

KJARNI_EXPORT size_t kjarni_transcriber_model_name(const KjarniTranscriber* t, char* buf, size_t buf_len)
{
    if (!t) return 0;
    const size_t required = t->model_name.size() + 1;
    if (!buf || buf_len == 0) return required;
    const size_t n = std::min(t->model_name.size(), buf_len - 1);
    std::memcpy(buf, t->model_name.data(), n);
    buf[n] = '\0';
    return required;
}

// ---- one stage at a time (kjarni_hip.h; parity tests and benchmarks) -------------------------------

struct KjarniHipWhisper {
    std::unique_ptr<WhisperModel> model;
    std::mutex mu;
};

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_load(const char* model_dir, int32_t device, KjarniHipWhisper** out)
{
    if (!model_dir || !out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        auto h = std::make_unique<KjarniHipWhisper>();
        h->model = WhisperModel::load(model_dir, device);
        *out = h.release();
    });
}

KJARNI_EXPORT void kjarni_hip_whisper_free(KjarniHipWhisper* w) { delete w; }

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_dims(const KjarniHipWhisper* w, int32_t* d_model, int32_t* n_mels, int32_t* vocab,
                                                      int32_t* encoder_frames)
{
    if (!w) return KJARNI_ERROR_NULL_POINTER;
    if (d_model) *d_model = w->model->config().d_model;
    if (n_mels) *n_mels = w->model->config().num_mel_bins;
    if (vocab) *vocab = w->model->config().vocab;
    if (encoder_frames) *encoder_frames = w->model->encoder_frames();
    return KJARNI_OK;
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_log_mel(KjarniHipWhisper* w, const float* samples, size_t num_samples, float* mel_out)
{
    if (!w || !samples || !mel_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        w->model->log_mel(samples, (int64_t)num_samples, mel_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_encode_mel(KjarniHipWhisper* w, const float* mel, int32_t frames, float* hidden_out)
{
    if (!w || !mel) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        w->model->encode_mel(mel, frames);
        if (hidden_out) w->model->encoder_output(hidden_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_encode_audio(KjarniHipWhisper* w, const float* samples, size_t num_samples,
                                                              float* hidden_out)
{
    if (!w || !samples) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        w->model->encode_audio(samples, (int64_t)num_samples);
        if (hidden_out) w->model->encoder_output(hidden_out);
        else hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_decode_begin(KjarniHipWhisper* w)
{
    if (!w) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        w->model->begin_decode();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_decode_forward(KjarniHipWhisper* w, const uint32_t* ids, int32_t n, float* hidden_out,
                                                                float* logits_out)
{
    if (!w || !ids) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        w->model->forward(ids, n);
        if (hidden_out) w->model->last_hidden(hidden_out, n);
        if (logits_out) w->model->logits_to_host(logits_out);
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_greedy(KjarniHipWhisper* w, const uint32_t* prompt, int32_t n_prompt, int32_t timestamps,
                                                        size_t max_tokens, uint32_t* ids_out, size_t capacity, size_t* n_out)
{
    if (!w || !prompt || !ids_out || !n_out) return KJARNI_ERROR_NULL_POINTER;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::lock_guard<std::mutex> lock(w->mu);
        const std::vector<uint32_t> ids =
            w->model->greedy(std::vector<uint32_t>(prompt, prompt + n_prompt), timestamps != 0, max_tokens, nullptr);
        *n_out = ids.size();
        std::memcpy(ids_out, ids.data(), std::min(capacity, ids.size()) * sizeof(uint32_t));
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_hip_whisper_decode_text(const KjarniHipWhisper* w, const uint32_t* ids, size_t n, int32_t skip_special,
                                                             char** out)
{
    if (!w || !out || (n && !ids)) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_UNKNOWN, [&] {
        *out = dup_cstr(w->model->vocab().decode(std::vector<uint32_t>(ids, ids + n), skip_special != 0));
    });
}

// ---- host-only pieces ------------------------------------------------------------------------------

KJARNI_EXPORT KjarniErrorCode kjarni_audio_load_wav(const char* path, KjarniFloatArray* out, uint32_t* original_sample_rate)
{
    if (!path || !out) return KJARNI_ERROR_NULL_POINTER;
    out->data = nullptr;
    out->len = 0;
    return guarded(KJARNI_ERROR_INFERENCE_FAILED, [&] {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("Failed to open WAV file: ") + path);
        std::ostringstream ss;
        ss << f.rdbuf();
        if (original_sample_rate) *original_sample_rate = parse_wav(ss.str()).rate;
        const std::vector<float> s = load_audio_16k_mono(path);
        if (s.empty()) return;
        out->data = static_cast<float*>(std::malloc(s.size() * sizeof(float)));
        if (!out->data) throw std::bad_alloc();
        std::memcpy(out->data, s.data(), s.size() * sizeof(float));
        out->len = s.size();
    });
}

KJARNI_EXPORT KjarniErrorCode kjarni_bytelevel_decode(const char* tokenizer_json_path, const uint32_t* ids, size_t n, int32_t skip_special,
                                                      char** out)
{
    if (!tokenizer_json_path || !out || (n && !ids)) return KJARNI_ERROR_NULL_POINTER;
    *out = nullptr;
    return guarded(KJARNI_ERROR_LOAD_FAILED, [&] {
        ByteLevelVocab v;
        v.load(tokenizer_json_path);
        *out = dup_cstr(v.decode(std::vector<uint32_t>(ids, ids + n), skip_special != 0));
    });
}
