#include "host_util.h"
#include "whisper.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "json.h"
#include "safetensors.h"
#include "unicode.h"
#include "whisper_kernels.h"

namespace kjarni {

namespace {

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// GPT-2 bytes_to_unicode (the `tokenizers` ByteLevel alphabet): code point -> byte.
const std::unordered_map<uint32_t, uint8_t>& char_to_byte()
{
    static const std::unordered_map<uint32_t, uint8_t> map = [] {
        std::unordered_map<uint32_t, uint8_t> m;
        bool direct[256] = {};
        for (int b = 33; b <= 126; ++b) direct[b] = true;
        for (int b = 161; b <= 172; ++b) direct[b] = true;
        for (int b = 174; b <= 255; ++b) direct[b] = true;
        uint32_t next = 256;
        for (int b = 0; b < 256; ++b) {
            if (direct[b]) m[(uint32_t)b] = (uint8_t)b;
            else m[next++] = (uint8_t)b;
        }
        return m;
    }();
    return map;
}

// String::from_utf8_lossy: every maximal invalid subpart becomes U+FFFD.
std::string utf8_lossy(const std::string& in)
{
    std::string out;
    const size_t n = in.size();
    size_t i = 0;
    auto cont = [&](size_t p, uint8_t lo, uint8_t hi) { return p < n && (uint8_t)in[p] >= lo && (uint8_t)in[p] <= hi; };
    while (i < n) {
        const uint8_t b = (uint8_t)in[i];
        size_t len = 0;
        if (b < 0x80) {
            out.push_back((char)b);
            ++i;
            continue;
        }
        size_t ok = 1;  // bytes of a valid prefix consumed so far
        if (b >= 0xC2 && b <= 0xDF) {
            len = 2;
            if (cont(i + 1, 0x80, 0xBF)) ok = 2;
        } else if (b >= 0xE0 && b <= 0xEF) {
            len = 3;
            const uint8_t lo = b == 0xE0 ? 0xA0 : 0x80, hi = b == 0xED ? 0x9F : 0xBF;
            if (cont(i + 1, lo, hi)) ok = cont(i + 2, 0x80, 0xBF) ? 3 : 2;
        } else if (b >= 0xF0 && b <= 0xF4) {
            len = 4;
            const uint8_t lo = b == 0xF0 ? 0x90 : 0x80, hi = b == 0xF4 ? 0x8F : 0xBF;
            if (cont(i + 1, lo, hi)) ok = cont(i + 2, 0x80, 0xBF) ? (cont(i + 3, 0x80, 0xBF) ? 4 : 3) : 2;
        }
        if (len != 0 && ok == len) {
            out.append(in, i, len);
            i += len;
        } else {
            out += "\xEF\xBF\xBD";
            i += (len == 0) ? 1 : ok;
        }
    }
    return out;
}

}  // namespace

WhisperConfig WhisperConfig::from_json(const std::string& text)
{
    const Json j = Json::parse(text);
    WhisperConfig c;
    auto req = [&](const char* k) {
        const Json* v = j.find(k);
        if (!v || !v->is_number()) throw std::runtime_error(std::string("config.json: missing field `") + k + "`");
        return (int)v->as_int();
    };
    c.d_model = req("d_model");
    c.encoder_layers = req("encoder_layers");
    c.decoder_layers = req("decoder_layers");
    c.heads = req("encoder_attention_heads");
    c.encoder_ffn = req("encoder_ffn_dim");
    c.decoder_ffn = req("decoder_ffn_dim");
    c.vocab = req("vocab_size");
    c.max_source_positions = req("max_source_positions");
    c.max_target_positions = req("max_target_positions");
    c.num_mel_bins = req("num_mel_bins");
    c.eos_token_id = (uint32_t)req("eos_token_id");
    c.scale_embedding = j.get_bool("scale_embedding", false);
    if (c.d_model <= 0 || c.heads <= 0 || c.d_model % c.heads != 0) throw std::runtime_error("config.json: bad d_model / heads");
    return c;
}

// ---- tokenizer (decode side) ----------------------------------------------------------------------

void ByteLevelVocab::load(const std::string& path)
{
    const Json j = Json::parse(slurp(path));
    const Json* model = j.find("model");
    const Json* vocab = model ? model->find("vocab") : nullptr;
    if (!vocab || !vocab->is_object()) throw std::runtime_error(path + ": no model.vocab");
    auto put = [&](uint32_t id, const std::string& tok, bool special) {
        if (id > (1u << 24)) throw std::runtime_error(path + ": token id " + std::to_string(id) + " is out of range");
        if (id >= id_to_token_.size()) {
            id_to_token_.resize(id + 1);
            has_token_.resize(id + 1, 0);
            special_.resize(id + 1, 0);
        }
        id_to_token_[id] = tok;
        has_token_[id] = 1;
        special_[id] = special ? 1 : 0;
        token_to_id_[tok] = id;
    };
    for (const auto& kv : vocab->obj) put((uint32_t)kv.second.as_int(), kv.first, false);
    if (const Json* added = j.find("added_tokens"); added && added->is_array())
        for (const Json& a : added->arr) {
            const Json* id = a.find("id");
            const Json* content = a.find("content");
            if (!id || !content) continue;
            put((uint32_t)id->as_int(), content->as_string(), a.get_bool("special", false));
        }
}

bool ByteLevelVocab::token_to_id(const std::string& token, uint32_t& id) const
{
    auto it = token_to_id_.find(token);
    if (it == token_to_id_.end()) return false;
    id = it->second;
    return true;
}

std::string ByteLevelVocab::decode(const std::vector<uint32_t>& ids, bool skip_special) const
{
    const auto& c2b = char_to_byte();
    std::string bytes;
    for (uint32_t id : ids) {
        if (id >= id_to_token_.size() || !has_token_[id]) continue;  // unknown ids decode to nothing
        if (skip_special && special_[id]) continue;
        const std::string& tok = id_to_token_[id];
        std::vector<uint32_t> cps;
        std::string mapped;
        bool ok = unicode::decode_utf8(tok.data(), tok.size(), cps);
        if (ok)
            for (uint32_t cp : cps) {
                auto it = c2b.find(cp);
                if (it == c2b.end()) {
                    ok = false;
                    break;
                }
                mapped.push_back((char)it->second);
            }
        bytes += ok ? mapped : tok;  // a token with a character outside the alphabet keeps its own bytes
    }
    return utf8_lossy(bytes);
}

// ---- model ----------------------------------------------------------------------------------------

float* WhisperModel::upload(const std::vector<float>& host)
{
    float* d = dalloc(host.size());
    if (!host.empty()) hip_check(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(weights)");
    weight_bytes_ += host.size() * sizeof(float);
    return d;
}

float* WhisperModel::dalloc(size_t floats)
{
    return static_cast<float*>(arena_.alloc(std::max<size_t>(floats, 4) * sizeof(float)));   // (device_arena.h: blocks, not one hipMalloc per tensor)
}

WhisperModel::~WhisperModel()
{
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (hipGraphExec_t g : graphs_)
        if (g) (void)hipGraphExecDestroy(g);
    if (stream_) (void)hipStreamDestroy(stream_);
    arena_.release();
}

std::unique_ptr<WhisperModel> WhisperModel::load(const std::string& dir, int device)
{
    if (visible_device_count() <= device) throw GpuUnavailable("no usable HIP device " + std::to_string(device));
    std::unique_ptr<WhisperModel> m(new WhisperModel());
    m->device_ = device;
    hip_check(hipSetDevice(device), "hipSetDevice");
    m->cfg_ = WhisperConfig::from_json(slurp(dir + "/config.json"));
    const WhisperConfig& c = m->cfg_;
    const int H = c.d_model, d = H / c.heads;
    if ((d & 3) || 256 % d != 0) throw std::runtime_error("unsupported head dimension " + std::to_string(d));
    m->vocab_.load(dir + "/tokenizer.json");
    m->eos_ = c.eos_token_id;
    {
        std::ifstream g(dir + "/generation_config.json");  // HFGenerationConfig::load_or_default: eos may be a list
        if (g) {
            std::ostringstream ss;
            ss << g.rdbuf();
            try {
                const Json gj = Json::parse(ss.str());
                if (const Json* e = gj.find("eos_token_id")) {
                    if (e->is_number()) m->eos_ = (uint32_t)e->as_int();
                    else if (e->is_array() && !e->arr.empty() && e->arr[0].is_number()) m->eos_ = (uint32_t)e->arr[0].as_int();
                }
            } catch (const std::exception&) {
            }
        }
    }
    SafeTensors st;
    st.open_dir(dir);
    std::vector<float> buf;
    auto get = [&](const std::string& name, std::vector<int64_t> want) {
        const std::vector<int64_t> shape = st.read_f32(name, buf);
        if (!want.empty() && shape != want) throw std::runtime_error("tensor " + name + " has an unexpected shape");
        return shape;
    };
    auto up = [&](const std::string& name, std::vector<int64_t> want) {
        get(name, want);
        return m->upload(buf);
    };
    auto up_opt_bias = [&](const std::string& name, int n) {  // with_optional_bias
        if (st.contains(name)) return up(name, {n});
        return m->upload(std::vector<float>((size_t)n, 0.0f));
    };

    // front end tables, built as the reference builds them (f32 arithmetic, libm cos/sin/log/exp)
    {
        std::vector<float> win(kNfft);
        for (int i = 0; i < kNfft; ++i) win[i] = 0.5f * (1.0f - std::cos(2.0f * (float)M_PI * (float)i / (float)kNfft));
        m->window_ = m->upload(win);
        const int bins = kNfft / 2 + 1;
        m->k_dft_ = round_up(kNfft, 32);
        m->n_dft_ = 512;  // cos rows at 0.., sin rows at 256..
        std::vector<float> w((size_t)m->n_dft_ * m->k_dft_, 0.0f);
        for (int k = 0; k < bins; ++k)
            for (int i = 0; i < kNfft; ++i) {
                const float angle = -2.0f * (float)M_PI * (float)(k * i) / (float)kNfft;
                w[(size_t)k * m->k_dft_ + i] = std::cos(angle);
                w[(size_t)(256 + k) * m->k_dft_ + i] = std::sin(angle);
            }
        m->w_dft_ = m->upload(w);
        // mel.rs:163-233
        const int n_mels = c.num_mel_bins;
        m->k_mel_ = round_up(bins, 32);
        m->n_mel_pad_ = round_up(n_mels, 128);
        const float sr = 16000.0f, fmin = 0.0f, fmax = 8000.0f;
        const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp;
        const float logstep = std::log(6.4f) / 27.0f;
        auto hz_to_mel = [&](float hz) { return hz < min_log_hz ? hz / f_sp : min_log_mel + std::log(hz / min_log_hz) / logstep; };
        auto mel_to_hz = [&](float mel) { return mel < min_log_mel ? mel * f_sp : min_log_hz * std::exp(logstep * (mel - min_log_mel)); };
        const float mel_min = hz_to_mel(fmin), mel_max = hz_to_mel(fmax);
        std::vector<float> mel_f((size_t)n_mels + 2);
        for (int i = 0; i <= n_mels + 1; ++i) mel_f[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
        std::vector<float> fb((size_t)m->n_mel_pad_ * m->k_mel_, 0.0f);
        for (int i = 0; i < n_mels; ++i) {
            const float enorm = 2.0f / (mel_f[i + 2] - mel_f[i]);
            for (int k = 0; k < bins; ++k) {
                const float freq = sr * (float)k / (float)kNfft;
                const float lower = (freq - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
                const float upper = (mel_f[i + 2] - freq) / (mel_f[i + 2] - mel_f[i + 1]);
                float v = std::max(0.0f, std::min(lower, upper));
                v *= enorm;
                fb[(size_t)i * m->k_mel_ + k] = v;
            }
        }
        m->w_mel_ = m->upload(fb);
    }

    // convolutional stem: weights re-laid as [out, k*C + c] (im2col order), K padded to a multiple of 32
    {
        const int C = c.num_mel_bins;
        get("model.encoder.conv1.weight", {H, C, 3});
        m->k_conv1_ = round_up(3 * C, 32);
        std::vector<float> w((size_t)H * m->k_conv1_, 0.0f);
        for (int o = 0; o < H; ++o)
            for (int ch = 0; ch < C; ++ch)
                for (int k = 0; k < 3; ++k) w[(size_t)o * m->k_conv1_ + k * C + ch] = buf[((size_t)o * C + ch) * 3 + k];
        m->conv1_w_ = m->upload(w);
        m->conv1_b_ = up("model.encoder.conv1.bias", {H});
        get("model.encoder.conv2.weight", {H, H, 3});
        const int k2 = round_up(3 * H, 32);
        std::vector<float> w2((size_t)H * k2, 0.0f);
        for (int o = 0; o < H; ++o)
            for (int ch = 0; ch < H; ++ch)
                for (int k = 0; k < 3; ++k) w2[(size_t)o * k2 + k * H + ch] = buf[((size_t)o * H + ch) * 3 + k];
        m->conv2_w_ = m->upload(w2);
        m->conv2_b_ = up("model.encoder.conv2.bias", {H});
        if (st.contains("model.encoder.embed_positions.weight")) {
            const auto shape = get("model.encoder.embed_positions.weight", {});
            if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("encoder position table has an unexpected shape");
            m->enc_pos_rows_ = (int)shape[0];
            m->enc_pos_ = m->upload(buf);
        } else {  // mel.rs:289-293: sinusoidal fallback
            const int rows = c.max_source_positions;
            std::vector<float> e((size_t)rows * H, 0.0f);
            for (int p = 0; p < rows; ++p)
                for (int i = 0; i < H / 2; ++i) {
                    const float angle = (float)p / std::pow(10000.0f, 2.0f * (float)i / (float)H);
                    e[(size_t)p * H + 2 * i] = std::sin(angle);
                    e[(size_t)p * H + 2 * i + 1] = std::cos(angle);
                }
            m->enc_pos_rows_ = rows;
            m->enc_pos_ = m->upload(e);
        }
    }

    auto fuse = [&](const std::vector<std::pair<std::string, std::string>>& parts, int in_dim, float*& w_out, float*& b_out) {
        std::vector<float> w, b;
        for (const auto& p : parts) {
            get(p.first, {H, in_dim});
            w.insert(w.end(), buf.begin(), buf.end());
            if (st.contains(p.second)) {
                get(p.second, {H});
                b.insert(b.end(), buf.begin(), buf.end());
            } else {
                b.insert(b.end(), (size_t)H, 0.0f);
            }
        }
        w_out = m->upload(w);
        b_out = m->upload(b);
    };

    m->enc_.resize((size_t)c.encoder_layers);
    for (int i = 0; i < c.encoder_layers; ++i) {
        const std::string p = "model.encoder.layers." + std::to_string(i);
        EncLayer& L = m->enc_[(size_t)i];
        fuse({{p + ".self_attn.q_proj.weight", p + ".self_attn.q_proj.bias"},
              {p + ".self_attn.k_proj.weight", p + ".self_attn.k_proj.bias"},
              {p + ".self_attn.v_proj.weight", p + ".self_attn.v_proj.bias"}},
             H, L.wqkv, L.bqkv);
        L.wo = up(p + ".self_attn.out_proj.weight", {H, H});
        L.bo = up_opt_bias(p + ".self_attn.out_proj.bias", H);
        L.ln1_g = up(p + ".self_attn_layer_norm.weight", {H});
        L.ln1_b = up(p + ".self_attn_layer_norm.bias", {H});
        L.w1 = up(p + ".fc1.weight", {c.encoder_ffn, H});
        L.b1 = up_opt_bias(p + ".fc1.bias", c.encoder_ffn);
        L.w2 = up(p + ".fc2.weight", {H, c.encoder_ffn});
        L.b2 = up_opt_bias(p + ".fc2.bias", H);
        L.ln2_g = up(p + ".final_layer_norm.weight", {H});
        L.ln2_b = up(p + ".final_layer_norm.bias", {H});
    }
    m->enc_ln_g_ = up("model.encoder.layer_norm.weight", {H});
    m->enc_ln_b_ = up("model.encoder.layer_norm.bias", {H});

    m->tok_emb_ = up("model.decoder.embed_tokens.weight", {c.vocab, H});
    m->lm_head_ = st.contains("proj_out.weight") ? up("proj_out.weight", {c.vocab, H}) : m->tok_emb_;
    {
        const auto shape = get("model.decoder.embed_positions.weight", {});
        if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("decoder position table has an unexpected shape");
        m->cfg_.max_target_positions = (int)shape[0];
        m->dec_pos_ = m->upload(buf);
    }
    m->dec_ln_g_ = up("model.decoder.layer_norm.weight", {H});
    m->dec_ln_b_ = up("model.decoder.layer_norm.bias", {H});
    m->max_frames_ = kFrames;
    m->cache_cap_ = 4 + 4096 + 8;
    m->dec_.resize((size_t)c.decoder_layers);
    for (int i = 0; i < c.decoder_layers; ++i) {
        const std::string p = "model.decoder.layers." + std::to_string(i);
        DecLayer& L = m->dec_[(size_t)i];
        fuse({{p + ".self_attn.q_proj.weight", p + ".self_attn.q_proj.bias"},
              {p + ".self_attn.k_proj.weight", p + ".self_attn.k_proj.bias"},
              {p + ".self_attn.v_proj.weight", p + ".self_attn.v_proj.bias"}},
             H, L.wqkv, L.bqkv);
        L.wo = up(p + ".self_attn.out_proj.weight", {H, H});
        L.bo = up_opt_bias(p + ".self_attn.out_proj.bias", H);
        L.ln1_g = up(p + ".self_attn_layer_norm.weight", {H});
        L.ln1_b = up(p + ".self_attn_layer_norm.bias", {H});
        L.cq = up(p + ".encoder_attn.q_proj.weight", {H, H});
        L.cbq = up_opt_bias(p + ".encoder_attn.q_proj.bias", H);
        fuse({{p + ".encoder_attn.k_proj.weight", p + ".encoder_attn.k_proj.bias"},
              {p + ".encoder_attn.v_proj.weight", p + ".encoder_attn.v_proj.bias"}},
             H, L.ckv, L.cbkv);
        L.co = up(p + ".encoder_attn.out_proj.weight", {H, H});
        L.cbo = up_opt_bias(p + ".encoder_attn.out_proj.bias", H);
        L.ln2_g = up(p + ".encoder_attn_layer_norm.weight", {H});
        L.ln2_b = up(p + ".encoder_attn_layer_norm.bias", {H});
        L.w1 = up(p + ".fc1.weight", {c.decoder_ffn, H});
        L.b1 = up_opt_bias(p + ".fc1.bias", c.decoder_ffn);
        L.w2 = up(p + ".fc2.weight", {H, c.decoder_ffn});
        L.b2 = up_opt_bias(p + ".fc2.bias", H);
        L.ln3_g = up(p + ".final_layer_norm.weight", {H});
        L.ln3_b = up(p + ".final_layer_norm.bias", {H});
        L.self_k = m->dalloc((size_t)m->cache_cap_ * H);
        L.self_v = m->dalloc((size_t)m->cache_cap_ * H);
        L.cross_kv = m->dalloc((size_t)(kFrames / 2) * 2 * H);
    }

    // workspace for one 30-second chunk
    const int F = kFrames, T2 = F / 2, I = std::max(c.encoder_ffn, c.decoder_ffn);
    const int kcols = std::max(m->k_conv1_, round_up(3 * H, 32));
    m->frames_ = m->dalloc((size_t)F * m->k_dft_);
    m->dft_ = m->dalloc((size_t)F * m->n_dft_);
    m->power_ = m->dalloc((size_t)F * m->k_mel_);
    m->melraw_ = m->dalloc((size_t)F * m->n_mel_pad_);
    m->mel_t_ = m->dalloc((size_t)F * c.num_mel_bins);
    m->max_scratch_ = reinterpret_cast<uint32_t*>(m->dalloc(4));
    m->cols_ = m->dalloc((size_t)F * kcols);
    m->conv1_out_ = m->dalloc((size_t)F * H);
    m->hidden_ = m->dalloc((size_t)T2 * H);
    m->normed_ = m->dalloc((size_t)T2 * H);
    m->qkv_ = m->dalloc((size_t)T2 * 3 * H);
    m->ctx_ = m->dalloc((size_t)T2 * H);
    m->mid_ = m->dalloc((size_t)T2 * I);
    m->ones_ = reinterpret_cast<uint32_t*>(m->dalloc((size_t)T2));
    {
        std::vector<uint32_t> ones((size_t)T2, 1u);
        hip_check(hipMemcpy(m->ones_, ones.data(), ones.size() * 4, hipMemcpyHostToDevice), "hipMemcpy(ones)");
    }
    m->dh_ = m->dalloc(8 * (size_t)H);
    m->dn_ = m->dalloc(8 * (size_t)H);
    m->dq_ = m->dalloc(8 * (size_t)H);
    m->dctx_ = m->dalloc(8 * (size_t)H);
    m->dlast_ = m->dalloc(8 * (size_t)H);
    m->dmid_ = m->dalloc(8 * (size_t)I);
    m->logits_ = m->dalloc((size_t)c.vocab);
    m->dids_ = reinterpret_cast<uint32_t*>(m->dalloc(8));
    m->dtoken_ = reinterpret_cast<int32_t*>(m->dalloc(4));
    m->hist_cap_ = 4096 + 16;
    m->dhist_ = reinterpret_cast<int32_t*>(m->dalloc((size_t)m->hist_cap_));
    m->dpos_ = reinterpret_cast<int*>(m->dalloc(4));
    m->dcount_ = reinterpret_cast<int*>(m->dalloc(4));
    m->att_scratch_ = m->dalloc(decode_attention_scratch_floats(8, c.heads, d, std::max(kSelfSplits, kCrossSplits)));
    m->gemm_scratch_floats_ = gemm_scratch_floats(8192, c.d_model);
    m->gemm_scratch_ = m->dalloc(m->gemm_scratch_floats_);
    m->dbest_ = reinterpret_cast<unsigned long long*>(m->dalloc(2 * (size_t)kMaxLanes));
    hip_check(hipMemset(m->dbest_, 0, sizeof(unsigned long long) * kMaxLanes), "memset(pick scratch)");
    hip_check(hipStreamCreateWithFlags(&m->stream_, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize(load)");
#ifdef KJARNI_TUNING
    if (const char* v = std::getenv("KJARNI_HIP_GEMV_ROWS")) set_gemv_rows_variant(std::atoi(v));  // kernel A/B measurements
#endif
    return m;
}

void WhisperModel::log_mel(const float* samples, int64_t n, float* mel_out)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (n <= 0) throw std::runtime_error("empty audio");
    if ((size_t)n > audio_cap_) {
        audio_ = dalloc((size_t)n);
        audio_cap_ = (size_t)n;
    }
    hipStream_t s = stream_;
    const int bins = kNfft / 2 + 1, n_mels = cfg_.num_mel_bins;
    hip_check(hipMemcpyAsync(audio_, samples, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s), "H2D audio");
    hip_check(launch_mel_frames(audio_, n, window_, kNfft, kHop, kFrames, k_dft_, frames_, s), "mel_frames");
    hip_check(launch_gemm(frames_, k_dft_, w_dft_, nullptr, nullptr, 0, dft_, n_dft_, kFrames, n_dft_, k_dft_, EPI_BIAS, s), "dft gemm");
    hip_check(launch_mel_power(dft_, n_dft_, 256, bins, k_mel_, kFrames, power_, s), "mel_power");
    hip_check(launch_gemm(power_, k_mel_, w_mel_, nullptr, nullptr, 0, melraw_, n_mel_pad_, kFrames, n_mel_pad_, k_mel_, EPI_BIAS, s),
              "mel gemm");
    hip_check(launch_mel_log_normalize(melraw_, n_mel_pad_, n_mels, kFrames, max_scratch_, n_mels, mel_t_, s), "log-mel");
    if (mel_out) {
        std::vector<float> t((size_t)kFrames * n_mels);
        hip_check(hipMemcpyAsync(t.data(), mel_t_, t.size() * sizeof(float), hipMemcpyDeviceToHost, s), "D2H mel");
        hip_check(hipStreamSynchronize(s), "sync");
        for (int f = 0; f < kFrames; ++f)
            for (int m = 0; m < n_mels; ++m) mel_out[(size_t)m * kFrames + f] = t[(size_t)f * n_mels + m];
    }
}

void WhisperModel::conv_and_encode(const float* mel_t, int ld_mel, int frames)
{
    hipStream_t s = stream_;
    const int H = cfg_.d_model, C = cfg_.num_mel_bins, heads = cfg_.heads, d = H / heads, I = cfg_.encoder_ffn;
    const int T1 = frames, T2 = (frames + 2 - 3) / 2 + 1;
    const int k2 = round_up(3 * H, 32);
    const GemmScratch sc{gemm_scratch_, gemm_scratch_floats_};  // 1 500 / 3 000 rows: the 64 x 64-tile route of gemm.hip
    if (frames > max_frames_ || frames < 1) throw std::runtime_error("mel has too many frames for the workspace");
    // conv1 (stride 1) + tanh-GELU, conv2 (stride 2) + tanh-GELU: mel.rs:303-311
    hip_check(launch_im2col3(mel_t, ld_mel, frames, C, 1, 1, T1, k_conv1_, cols_, s), "im2col conv1");
    hip_check(launch_gemm(cols_, k_conv1_, conv1_w_, conv1_b_, nullptr, 0, conv1_out_, H, T1, H, k_conv1_, EPI_BIAS_GELU_NEW, s, sc), "conv1");
    hip_check(launch_im2col3(conv1_out_, H, T1, H, 2, 1, T2, k2, cols_, s), "im2col conv2");
    hip_check(launch_gemm(cols_, k2, conv2_w_, conv2_b_, nullptr, 0, hidden_, H, T2, H, k2, EPI_BIAS_GELU_NEW, s, sc), "conv2");
    hip_check(launch_add_rows(hidden_, T2, H, T2, enc_pos_, enc_pos_rows_, s), "positions");
    // pre-norm encoder layers, mask of ones (transcriber.rs:134-138; encoder_layer.rs:195-212)
    for (const EncLayer& L : enc_) {
        hip_check(launch_layernorm(hidden_, L.ln1_g, L.ln1_b, 1e-5f, T2, H, normed_, s), "ln1");
        hip_check(launch_gemm(normed_, H, L.wqkv, L.bqkv, nullptr, 0, qkv_, 3 * H, T2, 3 * H, H, EPI_BIAS, s, sc), "qkv");
        hip_check(launch_attention(qkv_, ones_, 1, T2, heads, d, -1e9f, ctx_, s), "attention");
        hip_check(launch_gemm(ctx_, H, L.wo, L.bo, hidden_, H, hidden_, H, T2, H, H, EPI_BIAS_RESIDUAL, s, sc), "out proj");
        hip_check(launch_layernorm(hidden_, L.ln2_g, L.ln2_b, 1e-5f, T2, H, normed_, s), "ln2");
        hip_check(launch_gemm(normed_, H, L.w1, L.b1, nullptr, 0, mid_, I, T2, I, H, EPI_BIAS_GELU, s, sc), "fc1");
        hip_check(launch_gemm(mid_, I, L.w2, L.b2, hidden_, H, hidden_, H, T2, H, I, EPI_BIAS_RESIDUAL, s, sc), "fc2");
    }
    hip_check(launch_layernorm(hidden_, enc_ln_g_, enc_ln_b_, 1e-5f, T2, H, hidden_, s), "final ln");
    enc_frames_ = T2;
}

void WhisperModel::encode_mel(const float* mel, int frames)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    const int C = cfg_.num_mel_bins;
    if (frames > max_frames_ || frames < 1) throw std::runtime_error("mel has too many frames for the workspace");
    std::vector<float> t((size_t)frames * C);
    for (int m = 0; m < C; ++m)
        for (int f = 0; f < frames; ++f) t[(size_t)f * C + m] = mel[(size_t)m * frames + f];
    hip_check(hipMemcpyAsync(mel_t_, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D mel");
    hip_check(hipStreamSynchronize(stream_), "sync");  // `t` is pageable and about to go out of scope
    conv_and_encode(mel_t_, C, frames);
}

void WhisperModel::encode_audio(const float* samples, int64_t n)
{
    log_mel(samples, n, nullptr);
    conv_and_encode(mel_t_, cfg_.num_mel_bins, kFrames);
}

void WhisperModel::encoder_output(float* out) const
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(stream_), "sync");
    hip_check(hipMemcpy(out, hidden_, (size_t)enc_frames_ * cfg_.d_model * sizeof(float), hipMemcpyDeviceToHost), "D2H encoder");
}

void WhisperModel::begin_decode()
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (enc_frames_ <= 0) throw std::runtime_error("no encoder output to decode from");
    const int H = cfg_.d_model;
    for (const DecLayer& L : dec_)  // precompute_cross_attention_kv (cpu_decoder.rs:420-430): K | V per encoder frame
        hip_check(launch_gemm(hidden_, H, L.ckv, L.cbkv, nullptr, 0, L.cross_kv, 2 * H, enc_frames_, 2 * H, H, EPI_BIAS, stream_), "cross kv");
    cache_len_ = 0;
    hip_check(hipMemsetAsync(dpos_, 0, sizeof(int), stream_), "reset pos");
    hip_check(hipMemsetAsync(dcount_, 0, sizeof(int), stream_), "reset count");
}

void WhisperModel::decoder_pass(const uint32_t* ids_dev, int n, bool device_pos)
{
    hipStream_t s = stream_;
    const int H = cfg_.d_model, heads = cfg_.heads, d = H / heads, I = cfg_.decoder_ffn;
    const int* pos_ptr = device_pos ? dpos_ : nullptr;
    // One token: the first projection builds the embedded row itself and the vocabulary head folds the final LayerNorm in (two
    // launches fewer per step; the same arithmetic, element for element) -- where the one-row kernel takes the row (512 / 2048 floats).
    GemvArgs probe;
    probe.rows = 1; probe.k = H; probe.gamma = dec_ln_g_; probe.beta = dec_ln_b_; probe.W = lm_head_; probe.embed_ids = ids_dev;
    bool fold = n == 1 && gemv_rows_takes_row_extras(probe) && H == 512;
    bool fold_head = fold;
#ifdef KJARNI_TUNING
    static const char* no_fold = std::getenv("KJARNI_HIP_WHISPER_NO_FOLD");  // measurements: "1" neither, "embed" / "head" not that one
    if (no_fold && no_fold[0] == '1') fold = fold_head = false;
    if (no_fold && no_fold[0] == 'e') fold = false;
    if (no_fold && no_fold[0] == 'h') fold_head = false;
#endif
    if (!fold)
        hip_check(launch_decoder_embed(ids_dev, n, H, cfg_.vocab, tok_emb_, dec_pos_, cfg_.max_target_positions, cache_len_, pos_ptr,
                                       cfg_.scale_embedding ? 1 : 0, dh_, s), "decoder embed");
    bool first_layer = true;
    auto gemv = [&](const float* X, int64_t ldx, const float* g, const float* b, const float* W, const float* bias, const float* R,
                    int n_out, int k, float* Y, GemmEpilogue epi, const char* what) {
        GemvArgs a;
        a.X = X; a.ldx = ldx; a.rows = n; a.gamma = g; a.beta = b; a.eps = 1e-5f; a.W = W; a.bias = bias; a.R = R; a.ldr = H;
        a.n_out = n_out; a.k = k; a.Y0 = Y; a.ldy0 = n_out; a.epi = epi;
        hip_check(launch_gemv_rows(a, s), what);
    };
    for (const DecLayer& L : dec_) {
        // pre-norm layer (decoder_cross_attn_layer.rs:123-151); every LayerNorm rides on the projection after it
        {
            GemvArgs a;  // LN1 + Q | K | V: Q to scratch, K / V rows straight into the cache
            a.X = dh_; a.ldx = H; a.rows = n; a.gamma = L.ln1_g; a.beta = L.ln1_b; a.eps = 1e-5f; a.W = L.wqkv; a.bias = L.bqkv;
            a.n_out = 3 * H; a.k = H; a.seg = H; a.Y0 = dq_; a.ldy0 = H; a.Y1 = L.self_k; a.Y2 = L.self_v; a.ldy12 = H;
            a.row_off = cache_len_; a.row_off_ptr = pos_ptr; a.epi = EPI_BIAS;
            if (fold && first_layer) {
                a.embed_ids = ids_dev; a.embed_word = tok_emb_; a.embed_pos_table = dec_pos_; a.embed_vocab = cfg_.vocab;
                a.embed_max_pos = cfg_.max_target_positions; a.embed_pos = cache_len_; a.embed_pos_ptr = pos_ptr;
                a.embed_scale = cfg_.scale_embedding ? std::sqrt((float)H) : 1.0f;
                a.x_raw_out = dh_;
            }
            first_layer = false;
            hip_check(launch_gemv_rows(a, s), "ln1 + qkv");
        }
        // one token: the output projections merge the attention's per-split slabs themselves (no combine launches)
        const bool merge_self = n == 1 && gemv_row_att_supported(H, kSelfSplits, d), merge_cross = n == 1 && gemv_row_att_supported(H, kCrossSplits, d);
        hip_check(launch_decode_attention(dq_, H, n, L.self_k, H, L.self_v, H, cache_len_ + n, pos_ptr, cache_cap_, heads, d, cache_len_,
                                          kSelfSplits, att_scratch_, merge_self ? nullptr : dctx_, H, s), "self attention");
        if (merge_self)
            hip_check(launch_gemv_row_att(att_scratch_, kSelfSplits, d, L.wo, L.bo, dh_, H, H, dh_, s), "self out");
        else
            gemv(dctx_, H, nullptr, nullptr, L.wo, L.bo, dh_, H, H, dh_, EPI_BIAS_RESIDUAL, "self out");
        gemv(dh_, H, L.ln2_g, L.ln2_b, L.cq, L.cbq, nullptr, H, H, dq_, EPI_BIAS, "ln2 + cross q");
        hip_check(launch_decode_attention(dq_, H, n, L.cross_kv, 2 * H, L.cross_kv + H, 2 * H, enc_frames_, nullptr, enc_frames_, heads, d, -1,
                                          kCrossSplits, att_scratch_, merge_cross ? nullptr : dctx_, H, s), "cross attention");
        if (merge_cross)
            hip_check(launch_gemv_row_att(att_scratch_, kCrossSplits, d, L.co, L.cbo, dh_, H, H, dh_, s), "cross out");
        else
            gemv(dctx_, H, nullptr, nullptr, L.co, L.cbo, dh_, H, H, dh_, EPI_BIAS_RESIDUAL, "cross out");
        gemv(dh_, H, L.ln3_g, L.ln3_b, L.w1, L.b1, nullptr, I, H, dmid_, EPI_BIAS_GELU, "ln3 + fc1");
        gemv(dmid_, I, nullptr, nullptr, L.w2, L.b2, dh_, H, I, dh_, EPI_BIAS_RESIDUAL, "fc2");
    }
    GemvArgs a;
    a.ldx = H; a.rows = 1; a.W = lm_head_; a.n_out = cfg_.vocab; a.k = H; a.Y0 = logits_;
    a.ldy0 = cfg_.vocab; a.epi = EPI_BIAS;
    if (fold_head) {
        a.X = dh_; a.gamma = dec_ln_g_; a.beta = dec_ln_b_; a.eps = 1e-5f; a.x_norm_out = dlast_;
    } else {
        hip_check(launch_layernorm(dh_, dec_ln_g_, dec_ln_b_, 1e-5f, n, H, dlast_, s), "final ln");
        a.X = dlast_ + (size_t)(n - 1) * H;
    }
    hip_check(launch_gemv_rows(a, s), "lm head");
}

const float* WhisperModel::forward(const uint32_t* ids, int n)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (n < 1 || n > 8) throw std::runtime_error("decoder forward takes 1..8 tokens");
    if (cache_len_ + n > cache_cap_) throw std::runtime_error("decoder cache is full");
    hip_check(hipMemcpyAsync(dids_, ids, (size_t)n * 4, hipMemcpyHostToDevice, stream_), "H2D ids");
    decoder_pass(dids_, n, false);
    cache_len_ += n;
    last_rows_ = n;
    hip_check(hipMemcpyAsync(dpos_, &cache_len_, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D pos");
    hip_check(hipStreamSynchronize(stream_), "sync");  // ids / cache_len_ are host stack values
    return logits_;
}

void WhisperModel::last_hidden(float* out, int rows) const
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    rows = std::min(rows, last_rows_);
    hip_check(hipStreamSynchronize(stream_), "sync");
    hip_check(hipMemcpy(out, dlast_, (size_t)rows * cfg_.d_model * sizeof(float), hipMemcpyDeviceToHost), "D2H hidden");
}

void WhisperModel::logits_to_host(float* out) const
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(stream_), "sync");
    hip_check(hipMemcpy(out, logits_, (size_t)cfg_.vocab * sizeof(float), hipMemcpyDeviceToHost), "D2H logits");
}

void WhisperModel::enqueue_pick(bool timestamps, bool record)
{
    hip_check(launch_pick_token(logits_, cfg_.vocab, (int)kFirstSpecial, (int)eos_, (int)kTimestampBegin, timestamps ? 1 : 0, dtoken_,
                                record ? dhist_ : nullptr, record ? dcount_ : nullptr, record ? dpos_ : nullptr, stream_, 1, 0, nullptr,
                                record ? dbest_ : nullptr),
              "pick token");
}

uint32_t WhisperModel::pick_token(bool timestamps)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    enqueue_pick(timestamps, false);
    int32_t t = 0;
    hip_check(hipMemcpyAsync(&t, dtoken_, 4, hipMemcpyDeviceToHost, stream_), "D2H token");
    hip_check(hipStreamSynchronize(stream_), "sync");
    return (uint32_t)t;
}

// One greedy step with nothing step-dependent in the launch parameters: input token = dtoken_, position /
// key count / cache row = *dpos_; the chosen token is appended to dhist_ and the counters advance on the
// device.  Captured once per `timestamps` setting and replayed.
hipGraphExec_t WhisperModel::step_graph(bool timestamps)
{
    hipGraphExec_t& exec = graphs_[timestamps ? 1 : 0];
    if (exec) return exec;
    hipGraph_t graph = nullptr;
    hip_check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "begin capture");
    try {
        decoder_pass(reinterpret_cast<const uint32_t*>(dtoken_), 1, true);
        enqueue_pick(timestamps, true);
    } catch (...) {
        (void)hipStreamEndCapture(stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    hip_check(hipStreamEndCapture(stream_, &graph), "end capture");
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    hip_check(e, "graph instantiate");
    return exec;
}

std::vector<uint32_t> WhisperModel::greedy(const std::vector<uint32_t>& prompt, bool timestamps, size_t max_tokens,
                                           const std::function<bool(uint32_t)>& on_token)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (prompt.empty() || prompt.size() > 8) throw std::runtime_error("prompt must hold 1..8 tokens");
    if (max_tokens + 1 > (size_t)hist_cap_) throw std::runtime_error("max_tokens exceeds the decoder's capacity");
    begin_decode();
    forward(prompt.data(), (int)prompt.size());
    // first token from the last prompt position (transcriber.rs:186-190); recorded on the device as token 0.
    // The prompt's forward() already stored the position; pick advances it again, so set it back afterwards.
    enqueue_pick(timestamps, true);
    hip_check(hipMemcpyAsync(dpos_, &cache_len_, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D pos");
    std::vector<uint32_t> out;
    std::vector<int32_t> hist((size_t)hist_cap_);
    size_t seen = 0;
    bool done = false;
    auto drain = [&](size_t produced) {  // hand newly produced tokens to the caller in order
        for (; seen < produced && !done; ++seen) {
            const uint32_t tok = (uint32_t)hist[seen];
            out.push_back(tok);
            if (tok == eos_) {
                done = true;
            } else if (on_token && !on_token(tok)) {
                done = true;
            } else if (out.size() == max_tokens + 1) {
                done = true;
            }
        }
    };
    hip_check(hipMemcpyAsync(hist.data(), dhist_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_), "D2H token");
    hip_check(hipStreamSynchronize(stream_), "sync");
    drain(1);
    hipGraphExec_t exec = done ? nullptr : step_graph(timestamps);
    size_t produced = 1;
    // The loop of transcriber.rs:200-237, several steps per host round trip: tokens past an EOS / a stop are
    // computed but never reported.
    const size_t burst = on_token ? 4 : 16;
    while (!done) {
        const size_t steps = std::min(burst, max_tokens + 1 - produced);
        for (size_t i = 0; i < steps; ++i) hip_check(hipGraphLaunch(exec, stream_), "graph launch");
        hip_check(hipMemcpyAsync(hist.data() + produced, dhist_ + produced, steps * sizeof(int32_t), hipMemcpyDeviceToHost, stream_),
                  "D2H tokens");
        hip_check(hipStreamSynchronize(stream_), "sync");
        produced += steps;
        cache_len_ += (int)steps;
        drain(produced);
    }
    return out;
}

// ---- lanes: several chunks decoded in lock step --------------------------------------------------------

void WhisperModel::ensure_lanes()
{
    if (lane_logits_) return;
    const int H = cfg_.d_model;
    const size_t B = (size_t)kMaxLanes;
    for (size_t l = 0; l < dec_.size(); ++l) {
        lane_self_k_.push_back(dalloc((size_t)cache_cap_ * B * H));
        lane_self_v_.push_back(dalloc((size_t)cache_cap_ * B * H));
        lane_cross_kv_.push_back(dalloc(B * (size_t)max_frames_ / 2 * 2 * H));
    }
    lane_logits_ = dalloc(B * (size_t)cfg_.vocab);
    lane_tokens_ = reinterpret_cast<int32_t*>(dalloc(B));
    lane_hist_ = reinterpret_cast<int32_t*>(dalloc(B * (size_t)hist_cap_));
    lane_counts_ = reinterpret_cast<int*>(dalloc(B));
    drow_ = reinterpret_cast<int*>(dalloc(4));
}

void WhisperModel::begin_decode_lane(int lane)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (lane < 0 || lane >= kMaxLanes) throw std::runtime_error("lane out of range");
    if (enc_frames_ <= 0) throw std::runtime_error("no encoder output to decode from");
    ensure_lanes();
    const int H = cfg_.d_model;
    for (size_t l = 0; l < dec_.size(); ++l)
        hip_check(launch_gemm(hidden_, H, dec_[l].ckv, dec_[l].cbkv, nullptr, 0, lane_cross_kv_[l] + (size_t)lane * enc_frames_ * 2 * H, 2 * H,
                              enc_frames_, 2 * H, H, EPI_BIAS, stream_), "cross kv");
}

// decoder_pass for `lanes` independent sequences at the same position: row r of every buffer is lane r.
void WhisperModel::decoder_pass_lanes(int lanes, bool device_pos)
{
    hipStream_t s = stream_;
    const int H = cfg_.d_model, heads = cfg_.heads, d = H / heads, I = cfg_.decoder_ffn, n = lanes;
    const int* pos_ptr = device_pos ? dpos_ : nullptr;
    const int* row_ptr = device_pos ? drow_ : nullptr;
    hip_check(launch_decoder_embed(reinterpret_cast<const uint32_t*>(lane_tokens_), n, H, cfg_.vocab, tok_emb_, dec_pos_, cfg_.max_target_positions,
                                   cache_len_, pos_ptr, cfg_.scale_embedding ? 1 : 0, dh_, s, 1), "decoder embed");
    auto gemv = [&](const float* X, int64_t ldx, const float* g, const float* b, const float* W, const float* bias, const float* R,
                    int n_out, int k, float* Y, GemmEpilogue epi, const char* what) {
        GemvArgs a;
        a.X = X; a.ldx = ldx; a.rows = n; a.gamma = g; a.beta = b; a.eps = 1e-5f; a.W = W; a.bias = bias; a.R = R; a.ldr = H;
        a.n_out = n_out; a.k = k; a.Y0 = Y; a.ldy0 = n_out; a.epi = epi;
        hip_check(launch_gemv_rows(a, s), what);
    };
    for (size_t l = 0; l < dec_.size(); ++l) {
        const DecLayer& L = dec_[l];
        {
            GemvArgs a;  // LN1 + Q | K | V: the lanes' K / V rows are cache rows position * lanes + lane
            a.X = dh_; a.ldx = H; a.rows = n; a.gamma = L.ln1_g; a.beta = L.ln1_b; a.eps = 1e-5f; a.W = L.wqkv; a.bias = L.bqkv;
            a.n_out = 3 * H; a.k = H; a.seg = H; a.Y0 = dq_; a.ldy0 = H; a.Y1 = lane_self_k_[l]; a.Y2 = lane_self_v_[l]; a.ldy12 = H;
            a.row_off = cache_len_ * lanes; a.row_off_ptr = row_ptr; a.epi = EPI_BIAS;
            hip_check(launch_gemv_rows(a, s), "ln1 + qkv");
        }
        hip_check(launch_decode_attention(dq_, H, n, lane_self_k_[l], (int64_t)lanes * H, lane_self_v_[l], (int64_t)lanes * H, cache_len_ + 1, pos_ptr,
                                          cache_cap_, heads, d, -1, kSelfSplits, att_scratch_, dctx_, H, s, 1, H, H, 1), "self attention");
        gemv(dctx_, H, nullptr, nullptr, L.wo, L.bo, dh_, H, H, dh_, EPI_BIAS_RESIDUAL, "self out");
        gemv(dh_, H, L.ln2_g, L.ln2_b, L.cq, L.cbq, nullptr, H, H, dq_, EPI_BIAS, "ln2 + cross q");
        const int64_t lane_kv = (int64_t)enc_frames_ * 2 * H;
        hip_check(launch_decode_attention(dq_, H, n, lane_cross_kv_[l], 2 * H, lane_cross_kv_[l] + H, 2 * H, enc_frames_, nullptr, enc_frames_, heads, d,
                                          -1, kCrossSplits, att_scratch_, dctx_, H, s, 1, lane_kv, lane_kv, 1), "cross attention");
        gemv(dctx_, H, nullptr, nullptr, L.co, L.cbo, dh_, H, H, dh_, EPI_BIAS_RESIDUAL, "cross out");
        gemv(dh_, H, L.ln3_g, L.ln3_b, L.w1, L.b1, nullptr, I, H, dmid_, EPI_BIAS_GELU, "ln3 + fc1");
        gemv(dmid_, I, nullptr, nullptr, L.w2, L.b2, dh_, H, I, dh_, EPI_BIAS_RESIDUAL, "fc2");
    }
    hip_check(launch_layernorm(dh_, dec_ln_g_, dec_ln_b_, 1e-5f, n, H, dlast_, s), "final ln");
    GemvArgs a;
    a.X = dlast_; a.ldx = H; a.rows = n; a.W = lm_head_; a.n_out = cfg_.vocab; a.k = H; a.Y0 = lane_logits_; a.ldy0 = cfg_.vocab; a.epi = EPI_BIAS;
    hip_check(launch_gemv_rows(a, s), "lm head");
}

hipGraphExec_t WhisperModel::lane_step_graph(bool timestamps, int lanes)
{
    hipGraphExec_t& exec = lane_graphs_[timestamps ? 1 : 0][lanes];
    if (exec) return exec;
    hipGraph_t graph = nullptr;
    hip_check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "begin capture");
    try {
        decoder_pass_lanes(lanes, true);
        hip_check(launch_pick_token(lane_logits_, cfg_.vocab, (int)kFirstSpecial, (int)eos_, (int)kTimestampBegin, timestamps ? 1 : 0, lane_tokens_,
                                    lane_hist_, lane_counts_, dpos_, stream_, lanes, hist_cap_, drow_, dbest_), "pick token");
    } catch (...) {
        (void)hipStreamEndCapture(stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    hip_check(hipStreamEndCapture(stream_, &graph), "end capture");
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    hip_check(e, "graph instantiate");
    return exec;
}

std::vector<std::vector<uint32_t>> WhisperModel::greedy_lanes(int lanes, const std::vector<uint32_t>& prompt, bool timestamps, size_t max_tokens,
                                                              const std::function<bool()>& keep_going)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (lanes < 1 || lanes > kMaxLanes) throw std::runtime_error("lanes must be 1..8");
    if (prompt.empty() || prompt.size() > 8) throw std::runtime_error("prompt must hold 1..8 tokens");
    if (max_tokens + 1 > (size_t)hist_cap_) throw std::runtime_error("max_tokens exceeds the decoder's capacity");
    ensure_lanes();
    // The prompt, one position at a time, the same token in every lane (decode_chunk feeds it as one block: same rows,
    // same arithmetic per row).
    cache_len_ = 0;
    std::vector<int32_t> same((size_t)lanes);
    for (uint32_t t : prompt) {
        std::fill(same.begin(), same.end(), (int32_t)t);
        hip_check(hipMemcpyAsync(lane_tokens_, same.data(), same.size() * 4, hipMemcpyHostToDevice, stream_), "H2D prompt token");
        decoder_pass_lanes(lanes, false);
        cache_len_ += 1;
        hip_check(hipStreamSynchronize(stream_), "sync");
    }
    hip_check(hipMemsetAsync(lane_counts_, 0, (size_t)kMaxLanes * sizeof(int), stream_), "reset counts");
    // first generated token of every lane from the last prompt position; pick advances the counters, so they are set after it
    hip_check(launch_pick_token(lane_logits_, cfg_.vocab, (int)kFirstSpecial, (int)eos_, (int)kTimestampBegin, timestamps ? 1 : 0, lane_tokens_,
                                lane_hist_, lane_counts_, dpos_, stream_, lanes, hist_cap_, drow_, dbest_), "pick token");
    const int row0 = cache_len_ * lanes;
    hip_check(hipMemcpyAsync(dpos_, &cache_len_, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D pos");
    hip_check(hipMemcpyAsync(drow_, &row0, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D row");

    std::vector<std::vector<uint32_t>> out((size_t)lanes);
    std::vector<std::vector<int32_t>> hist((size_t)lanes, std::vector<int32_t>((size_t)hist_cap_));
    std::vector<uint8_t> done((size_t)lanes, 0);
    std::vector<size_t> seen((size_t)lanes, 0);
    auto all_done = [&] { return std::all_of(done.begin(), done.end(), [](uint8_t d) { return d != 0; }); };
    auto fetch = [&](size_t from, size_t count) {
        for (int l = 0; l < lanes; ++l)
            hip_check(hipMemcpyAsync(hist[(size_t)l].data() + from, lane_hist_ + (size_t)l * hist_cap_ + from, count * sizeof(int32_t),
                                     hipMemcpyDeviceToHost, stream_), "D2H tokens");
        hip_check(hipStreamSynchronize(stream_), "sync");
    };
    auto drain = [&](size_t produced) {
        for (int l = 0; l < lanes; ++l)
            for (; seen[(size_t)l] < produced && !done[(size_t)l]; ++seen[(size_t)l]) {
                const uint32_t tok = (uint32_t)hist[(size_t)l][seen[(size_t)l]];
                out[(size_t)l].push_back(tok);
                if (tok == eos_ || out[(size_t)l].size() == max_tokens + 1) done[(size_t)l] = 1;
            }
    };
    fetch(0, 1);
    drain(1);
    size_t produced = 1;
    hipGraphExec_t exec = all_done() ? nullptr : lane_step_graph(timestamps, lanes);
    while (!all_done()) {
        if (keep_going && !keep_going()) break;
        const size_t steps = std::min<size_t>(16, max_tokens + 1 - produced);
        if (steps == 0) break;
        for (size_t i = 0; i < steps; ++i) hip_check(hipGraphLaunch(exec, stream_), "graph launch");
        fetch(produced, steps);
        produced += steps;
        cache_len_ += (int)steps;
        drain(produced);
    }
    return out;
}

}  // namespace kjarni
