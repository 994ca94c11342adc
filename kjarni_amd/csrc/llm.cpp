#include "host_util.h"
#include "llm.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "json.h"
#include "kernels.h"
#include "llm_kernels.h"
#include "safetensors.h"
#include "whisper_kernels.h"

namespace kjarni {

namespace {

uint16_t f32_to_bf16(float v)  // round to nearest even
{
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

LlmConfig LlmConfig::from_json(const std::string& text)
{
    const Json j = Json::parse(text);
    LlmConfig c;
    auto req = [&](const char* k) {
        const Json* v = j.find(k);
        if (!v || !v->is_number()) throw std::runtime_error(std::string("config.json: missing field `") + k + "`");
        return (int)v->as_int();
    };
    c.model_type = j.get_string("model_type", "llama");
    if (c.model_type != "llama" && c.model_type != "qwen2" && c.model_type != "mistral")
        throw std::runtime_error("unsupported decoder model_type '" + c.model_type + "' (llama, qwen2 and mistral are)");
    c.hidden = req("hidden_size");
    c.layers = req("num_hidden_layers");
    c.heads = req("num_attention_heads");
    c.kv_heads = (int)j.get_int("num_key_value_heads", c.heads);
    c.inter = req("intermediate_size");
    c.vocab = req("vocab_size");
    c.max_pos = req("max_position_embeddings");
    c.head_dim = (int)j.get_int("head_dim", c.hidden / c.heads);
    // llama/config.rs:137-152, qwen/config.rs:70-76, mistral/config.rs:54-56 + :168 defaults.  Mistral runs on the Llama
    // decoder (mistral/model.rs:56-62); its sliding_window field is never read by the reference.
    const bool llama = c.model_type == "llama", mistral = c.model_type == "mistral";
    c.eps = (float)j.get_double("rms_norm_eps", (llama || mistral) ? 1e-5 : 1e-6);
    c.rope_theta = (float)j.get_double("rope_theta", llama ? 500000.0 : (mistral ? 10000.0 : 1000000.0));
    c.tie_embeddings = j.get_bool("tie_word_embeddings", llama);
    if (const Json* rs = j.find("rope_scaling"); rs && rs->is_object()) {
        c.has_rope_scaling = true;
        c.rope_type = rs->get_string("rope_type", rs->get_string("type", ""));
        c.rope_factor = (float)rs->get_double("factor", 1.0);
        c.rope_low = (float)rs->get_double("low_freq_factor", 1.0);
        c.rope_high = (float)rs->get_double("high_freq_factor", 4.0);
        c.rope_original_max = (int)rs->get_int("original_max_position_embeddings", 8192);
    }
    if (const Json* e = j.find("eos_token_id")) {
        if (e->is_number()) c.eos_ids.push_back((uint32_t)e->as_int());
        else if (e->is_array())
            for (const Json& x : e->arr)
                if (x.is_number()) c.eos_ids.push_back((uint32_t)x.as_int());
    }
    if (const Json* b = j.find("bos_token_id"); b && b->is_number()) {
        c.has_bos = true;
        c.bos_id = (uint32_t)b->as_int();
    }
    if (c.heads <= 0 || c.kv_heads <= 0 || c.heads % c.kv_heads != 0 || c.head_dim * c.heads != c.hidden)
        throw std::runtime_error("config.json: unsupported head geometry");
    return c;
}

float* LlmModel::dalloc(size_t floats)
{
    return static_cast<float*>(arena_.alloc(std::max<size_t>(floats, 4) * sizeof(float)));   // (device_arena.h: blocks, not one hipMalloc per tensor)
}

float* LlmModel::upload_f32(const std::vector<float>& host)
{
    float* d = dalloc(host.size());
    if (!host.empty()) hip_check(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(weights)");
    weight_bytes_ += host.size() * sizeof(float);
    return d;
}

void* LlmModel::upload_weight(const std::vector<float>& host)
{
    if (!bf16_) return upload_f32(host);
    std::vector<uint16_t> h(host.size());
    for (size_t i = 0; i < host.size(); ++i) h[i] = f32_to_bf16(host[i]);  // exact when the file already held bf16
    void* d = dalloc((host.size() + 1) / 2);
    if (!h.empty()) hip_check(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice), "hipMemcpy(weights)");
    weight_bytes_ += h.size() * 2;
    return d;
}

LlmModel::~LlmModel()
{
    if (host_logits_) (void)hipHostFree(host_logits_);
    if (samp_host_) (void)hipHostFree(samp_host_);
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (graph_) (void)hipGraphExecDestroy(graph_);
    if (stream_) (void)hipStreamDestroy(stream_);
    arena_.release();
}

std::unique_ptr<LlmModel> LlmModel::load(const std::string& dir, int device, int weights, int max_context)
{
    if (visible_device_count() <= device) throw GpuUnavailable("no usable HIP device " + std::to_string(device));
    std::unique_ptr<LlmModel> m(new LlmModel());
    m->device_ = device;
    hip_check(hipSetDevice(device), "hipSetDevice");
    m->cfg_ = LlmConfig::from_json(slurp(dir + "/config.json"));
    const LlmConfig& c = m->cfg_;
    const int H = c.hidden, d = c.head_dim, kv = c.kv_heads * d;
    if ((d & 3) || d > 128 || 256 % (d / 4) != 0 || (H & 7) || (c.inter & 7)) throw std::runtime_error("unsupported decoder geometry");
    SafeTensors st;
    st.open_dir(dir);
    m->bf16_ = weights == 2 || (weights == 0 && st.get("model.layers.0.self_attn.q_proj.weight").dtype == "BF16");
    std::vector<float> buf, tmp;
    auto get = [&](const std::string& name, std::vector<int64_t> want) {
        const std::vector<int64_t> shape = st.read_f32(name, buf);
        if (shape != want) throw std::runtime_error("tensor " + name + " has an unexpected shape");
    };
    m->layers_.resize((size_t)c.layers);
    m->cache_cap_ = std::min(max_context > 0 ? max_context : c.max_pos, c.max_pos);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "model.layers." + std::to_string(i);
        Layer& L = m->layers_[(size_t)i];
        std::vector<float> w, b;
        bool any_bias = false;
        for (const auto& nm : {std::make_pair(std::string("q_proj"), H), std::make_pair(std::string("k_proj"), kv),
                               std::make_pair(std::string("v_proj"), kv)}) {
            get(p + ".self_attn." + nm.first + ".weight", {nm.second, H});
            w.insert(w.end(), buf.begin(), buf.end());
            if (st.contains(p + ".self_attn." + nm.first + ".bias")) {  // Qwen2 (qwen/config.rs:228-234)
                st.read_f32(p + ".self_attn." + nm.first + ".bias", tmp);
                b.insert(b.end(), tmp.begin(), tmp.end());
                any_bias = true;
            } else {
                b.insert(b.end(), (size_t)nm.second, 0.0f);
            }
        }
        L.wqkv = m->upload_weight(w);
        L.bqkv = any_bias ? m->upload_f32(b) : nullptr;
        get(p + ".self_attn.o_proj.weight", {H, H});
        L.wo = m->upload_weight(buf);
        get(p + ".mlp.gate_proj.weight", {c.inter, H});
        L.gate = m->upload_weight(buf);
        get(p + ".mlp.up_proj.weight", {c.inter, H});
        L.up = m->upload_weight(buf);
        get(p + ".mlp.down_proj.weight", {H, c.inter});
        L.down = m->upload_weight(buf);
        get(p + ".input_layernorm.weight", {H});
        L.ln1 = m->upload_f32(buf);
        get(p + ".post_attention_layernorm.weight", {H});
        L.ln2 = m->upload_f32(buf);
        L.k_cache = m->dalloc((size_t)m->cache_cap_ * kv);
        L.v_cache = m->dalloc((size_t)m->cache_cap_ * kv);
    }
    get("model.embed_tokens.weight", {c.vocab, H});
    m->embed_ = m->upload_weight(buf);
    if (c.tie_embeddings || !st.contains("lm_head.weight")) {
        m->lm_head_ = m->embed_;
    } else {
        get("lm_head.weight", {c.vocab, H});
        m->lm_head_ = m->upload_weight(buf);
    }
    get("model.norm.weight", {H});
    m->final_norm_ = m->upload_f32(buf);

    // RoPE tables as the reference builds them (rope/mod.rs:62-130), [cache_cap, d/2]
    {
        const int half = d / 2;
        std::vector<float> inv((size_t)half);
        for (int i = 0; i < half; ++i) inv[(size_t)i] = 1.0f / std::pow(c.rope_theta, (float)(2 * i) / (float)d);
        if (c.has_rope_scaling && c.rope_type == "llama3") {
            const float low_wl = (float)c.rope_original_max / c.rope_low, high_wl = (float)c.rope_original_max / c.rope_high;
            for (int i = 0; i < half; ++i) {
                const float base = inv[(size_t)i];
                const float wl = 2.0f * (float)M_PI / base;
                if (wl < high_wl) continue;
                if (wl > low_wl) {
                    inv[(size_t)i] = base / c.rope_factor;
                } else {
                    const float smooth = ((float)c.rope_original_max / wl - c.rope_low) / (c.rope_high - c.rope_low);
                    inv[(size_t)i] = base / ((1.0f - smooth) * c.rope_factor + smooth);
                }
            }
        }
        std::vector<float> cs((size_t)m->cache_cap_ * half), sn((size_t)m->cache_cap_ * half);
        for (int p = 0; p < m->cache_cap_; ++p)
            for (int i = 0; i < half; ++i) {
                const float angle = (float)p * inv[(size_t)i];
                cs[(size_t)p * half + i] = std::cos(angle);
                sn[(size_t)p * half + i] = std::sin(angle);
            }
        m->cos_ = m->upload_f32(cs);
        m->sin_ = m->upload_f32(sn);
    }
    // key ranges per head: up to 512 keys each (128 of them are one register-held pass of the attention kernel); few enough
    // that the output projection can merge the slabs itself
    m->splits_ = std::max(1, std::min(64, (m->cache_cap_ + 511) / 512));
#ifdef KJARNI_TUNING
    if (const char* v = std::getenv("KJARNI_HIP_LLM_SPLITS")) m->splits_ = std::max(1, std::atoi(v));  // measurements
#endif
    while ((m->cache_cap_ + m->splits_ - 1) / m->splits_ > 512) ++m->splits_;
    m->h_ = m->dalloc(8 * (size_t)H);
    m->q_ = m->dalloc(8 * (size_t)H);
    m->ctx_ = m->dalloc(8 * (size_t)H);
    m->last_ = m->dalloc(8 * (size_t)H);
    m->mid_ = m->dalloc(8 * (size_t)c.inter);
    m->logits_ = m->dalloc((size_t)c.vocab);
    m->att_scratch_ = m->dalloc(decode_attention_scratch_floats(8, c.heads, d, m->splits_));
    m->ids_ = reinterpret_cast<uint32_t*>(m->dalloc(8));
    m->token_ = reinterpret_cast<int32_t*>(m->dalloc(4));
    m->hist_cap_ = m->cache_cap_ + 16;
    m->hist_ = reinterpret_cast<int32_t*>(m->dalloc((size_t)m->hist_cap_));
    m->pos_ = reinterpret_cast<int*>(m->dalloc(4));
    m->count_ = reinterpret_cast<int*>(m->dalloc(4));
    m->best_ = reinterpret_cast<unsigned long long*>(m->dalloc(4));
    hip_check(hipMemset(m->best_, 0, 8), "memset");
    hip_check(hipStreamCreateWithFlags(&m->stream_, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize(load)");
    return m;
}

void LlmModel::reset()
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    cache_len_ = 0;
    hip_check(hipMemsetAsync(pos_, 0, sizeof(int), stream_), "reset pos");
    hip_check(hipMemsetAsync(count_, 0, sizeof(int), stream_), "reset count");
}

void LlmModel::pass(const uint32_t* ids_dev, int n, bool device_pos)
{
    hipStream_t s = stream_;
    const LlmConfig& c = cfg_;
    const int H = c.hidden, d = c.head_dim, kv = c.kv_heads * d, I = c.inter;
    const int* pp = device_pos ? pos_ : nullptr;
    // one token: the first layer's projection gathers the embedding row itself (one launch fewer per step)
    bool embed_in_qkv = n == 1 && H <= 8192 && !layers_.empty() && llm_qkv_rope_embeds(H, layers_[0].ln1, layers_[0].wqkv, embed_);
#ifdef KJARNI_TUNING
    static const bool no_fold = std::getenv("KJARNI_HIP_LLM_NO_FOLD") != nullptr;  // measurements: the embedding gather as its own launch
    if (no_fold) embed_in_qkv = false;
#endif
    if (!embed_in_qkv) hip_check(launch_llm_embed(ids_dev, n, H, c.vocab, embed_, bf16_ ? 1 : 0, h_, s), "embed");
    bool first_layer = true;
    for (const Layer& L : layers_) {
        if (n == 1 && H <= 8192) {  // decode step: norm + projection + rotation in one launch
            const bool emb = embed_in_qkv && first_layer;
            hip_check(launch_llm_qkv_rope(h_, L.ln1, c.eps, L.wqkv, bf16_ ? 1 : 0, L.bqkv, H, c.heads, c.kv_heads, d, cos_, sin_, q_,
                                          L.k_cache, L.v_cache, cache_len_, pp, s, emb ? ids_dev : nullptr, emb ? embed_ : nullptr,
                                          c.vocab, emb ? h_ : nullptr), "norm + qkv + rope");
            first_layer = false;
        } else {
        LlmGemvArgs a;  // RMSNorm + Q | K | V (decoder_attention.rs:61-82): K / V rows land in the cache
        a.X = h_; a.ldx = H; a.rows = n; a.gamma = L.ln1; a.eps = c.eps; a.W = L.wqkv; a.bf16 = bf16_; a.bias = L.bqkv;
        a.n_out = H + 2 * kv; a.k = H; a.seg_q = H; a.seg_kv = kv; a.Y0 = q_; a.ldy0 = H; a.Y1 = L.k_cache; a.Y2 = L.v_cache; a.ldy12 = kv;
        a.row_off = cache_len_; a.row_off_ptr = pp;
        hip_check(launch_llm_gemv(a, s), "norm + qkv");
        hip_check(launch_rope(q_, H, n, c.heads, d, cos_, sin_, cache_len_, pp, 0, s), "rope q");
        hip_check(launch_rope(L.k_cache, kv, n, c.kv_heads, d, cos_, sin_, cache_len_, pp, 1, s), "rope k");
        }
        // one token: the output projection merges the attention's per-split slabs itself (no combine launch)
        const bool merge_in_proj = n == 1 && c.heads * d == H && llm_gemv_merges_attention(H, splits_, d);
        hip_check(launch_decode_attention(q_, H, n, L.k_cache, kv, L.v_cache, kv, cache_len_ + n, pp, cache_cap_, c.heads, d, cache_len_,
                                          splits_, att_scratch_, merge_in_proj ? nullptr : ctx_, H, s, c.heads / c.kv_heads), "attention");
        LlmGemvArgs o;
        o.X = ctx_; o.ldx = H; o.rows = n; o.W = L.wo; o.bf16 = bf16_; o.R = h_; o.ldr = H; o.n_out = H; o.k = H; o.Y0 = h_; o.ldy0 = H;
        if (merge_in_proj) {
            o.X = att_scratch_; o.att_splits = splits_; o.att_head_dim = d;
        }
        hip_check(launch_llm_gemv(o, s), "o proj");
#ifdef KJARNI_TUNING
        static const bool touch = std::getenv("KJARNI_HIP_LLM_TOUCH") != nullptr;  // measurements: weights already in the memory-side cache
        if (touch && n == 1) {
            const size_t wb = bf16_ ? 2 : 4;
            hip_check(launch_touch(L.gate, (size_t)I * H * wb, reinterpret_cast<unsigned*>(att_scratch_), s), "touch");
            hip_check(launch_touch(L.up, (size_t)I * H * wb, reinterpret_cast<unsigned*>(att_scratch_), s), "touch");
            hip_check(launch_touch(L.down, (size_t)I * H * wb, reinterpret_cast<unsigned*>(att_scratch_), s), "touch");
        }
#endif
        LlmGemvArgs g;  // RMSNorm + SwiGLU (swiglu.rs:32-57)
        g.X = h_; g.ldx = H; g.rows = n; g.gamma = L.ln2; g.eps = c.eps; g.W = L.gate; g.W2 = L.up; g.bf16 = bf16_; g.swiglu = 1;
        g.n_out = I; g.k = H; g.Y0 = mid_; g.ldy0 = I;
        hip_check(launch_llm_gemv(g, s), "norm + gate/up");
        LlmGemvArgs dn;
        dn.X = mid_; dn.ldx = I; dn.rows = n; dn.W = L.down; dn.bf16 = bf16_; dn.R = h_; dn.ldr = H; dn.n_out = H; dn.k = I; dn.Y0 = h_; dn.ldy0 = H;
        hip_check(launch_llm_gemv(dn, s), "down proj");
    }
    LlmGemvArgs lm;
    lm.ldx = H; lm.rows = 1; lm.W = lm_head_; lm.bf16 = bf16_; lm.n_out = c.vocab; lm.k = H;
    lm.Y0 = logits_; lm.ldy0 = c.vocab;
    if (n == 1 && llm_gemv_streams(H, lm_head_, nullptr)) {  // one token: the head normalises the row itself (and stores it)
        lm.X = h_; lm.gamma = final_norm_; lm.eps = c.eps; lm.norm_out = last_;
    } else {
        hip_check(launch_rmsnorm(h_, final_norm_, c.eps, n, H, last_, s), "final norm");
        lm.X = last_ + (size_t)(n - 1) * H;
    }
    hip_check(launch_llm_gemv(lm, s), "lm head");
}

// Prompt rows through the fp32 matrix cores (prefill_gemm_kernel) instead of 8-row GEMV passes: per layer RMSNorm ->
// Q, K, V projections (K and V rows land in the cache) -> RoPE -> causal attention over the cache -> o-proj + residual
// -> RMSNorm -> gate, up -> silu(gate) * up -> down-proj + residual; same formulas as pass().  After the last layer the
// final norm runs on the last (n - 1) % 8 + 1 rows (what last_hidden() exposes) and the lm head on the last row.
void LlmModel::prefill_rows(const uint32_t* ids_host, int n)
{
    hipStream_t s = stream_;
    const LlmConfig& c = cfg_;
    const int H = c.hidden, d = c.head_dim, kv = c.kv_heads * d, I = c.inter;
    const int wb = bf16_ ? 1 : 0;
    constexpr int kChunk = 2048;
#ifdef KJARNI_TUNING
    static const int kTileRows = [] {
        const char* v = std::getenv("KJARNI_HIP_LLM_TILE_ROWS");  // measurements
        return v ? std::atoi(v) : 512;
    }();
#else
    constexpr int kTileRows = 512;  // rows from which a projection may take the encoder's 128 x 128-tile f32 GEMM (if its tiles fill the chip)
#endif
    if (!ph_) {
        prefill_cap_ = kChunk;
        const size_t P = (size_t)prefill_cap_;
        ph_ = dalloc(P * H);
        pn_ = dalloc(P * H);
        pq_ = dalloc(P * H);
        pctx_ = dalloc(P * H);
        pg_ = dalloc(P * I);
        pu_ = dalloc(P * I);
        pids_ = reinterpret_cast<uint32_t*>(dalloc(P));
        psplit_ = dalloc(prefill_gemm_scratch_floats(prefill_cap_, std::max(I, H)));
        if (bf16_) pw32_ = dalloc(std::max((size_t)(H + 2 * kv) * H, (size_t)I * H));
    }
    const size_t wsz = bf16_ ? 2 : 4;
    auto at = [&](const void* w, size_t elems) { return static_cast<const void*>(static_cast<const char*>(w) + elems * wsz); };
    for (int done = 0; done < n; done += prefill_cap_) {
        const int m = std::min(prefill_cap_, n - done);
        hip_check(hipMemcpyAsync(pids_, ids_host + done, (size_t)m * 4, hipMemcpyHostToDevice, s), "H2D ids");
        hip_check(launch_llm_embed(pids_, m, H, c.vocab, embed_, wb, ph_, s), "embed");
        // Y[m, N] = A W^T (+ bias) (+ R), or with `gate`: gate = silu(gate) * (A W^T).  Blocks of >= kTileRows rows run the
        // encoder's 128 x 128-tile f32 GEMM (gemm.hip), bf16 weights on an f32 copy made just before (100 MB moved per 69 GFLOP
        // at 2 048 rows).  Measured on the 1B shape: 2 048 rows 46.2 -> 42.6 ms, 1 792 rows 40.4 -> 39.1 ms, 1 536 rows 33.1 -> 36.1 ms
        // (the 2 048-wide projections are then 192 tiles on 256 CUs): hence kTileRows.
        // per projection: the 128 x 128 tiles when they number at least one per CU (m / 128 x N / 128 >= 208), from kTileRows rows
        const bool tile_shapes = H % 128 == 0 && I % 128 == 0 && kv % 128 == 0 && (!bf16_ || pw32_);
        auto proj = [&](const float* Ain, int lda, const void* W, const float* bias, const float* R, float* Y, int ldy, int N, int K,
                        float* gate, const char* what) {
#ifdef KJARNI_TUNING
            static const int min_tiles_env = [] { const char* v = std::getenv("KJARNI_HIP_LLM_MIN_TILES"); return v ? std::atoi(v) : 0; }();
            const int min_tiles = min_tiles_env > 0 ? min_tiles_env : 208;
#else
            // measured on the 1B shape: 192 tiles (1 536 rows x 2 048 columns) are faster on the 64 x 64 kernel, 224 on the tiles -- with
            // f32 weights (both kernels on the f32 matrix cores) and again with bf16 weights (both on the bf16 matrix cores: 768 /
            // 1 024 tokens 8.1 / 9.8 ms at 208 against 9.9 / 11.2 at 96 and 8.8 / 11.2 with no tiles at all)
            constexpr int min_tiles = 208;
#endif
            const bool tiles = tile_shapes && m >= kTileRows && (int64_t)((m + 127) / 128) * (N / 128) >= min_tiles;
            if (!tiles) {
                hip_check(launch_prefill_gemm(Ain, lda, W, wb, bias, R, ldy, Y, ldy, m, N, K, s, psplit_, gate), what);
                return;
            }
            ++tile_gemm_calls_;
            // bf16 weights: the bf16 matrix cores take them as they are, the f32 activations as three exact bf16 pieces (the same
            // products as the f32 GEMM on a widened copy: gemm_split.hip) -- 2 048-token prompt 33.8 -> 18.1 ms, no 100 MB copy
#ifdef KJARNI_TUNING
            static const bool widen = [] { const char* v = std::getenv("KJARNI_HIP_LLM_WIDEN"); return v && v[0] == '1'; }();
#else
            constexpr bool widen = false;
#endif
            if (bf16_ && !widen && K % 64 == 0) {
                if (gate)
                    hip_check(launch_gemm_bf16_weights(Ain, lda, W, bias, gate, ldy, gate, ldy, m, N, K, EPI_BIAS_MUL_SILU, s), what);
                else
                    hip_check(launch_gemm_bf16_weights(Ain, lda, W, bias, R, ldy, Y, ldy, m, N, K, R ? EPI_BIAS_RESIDUAL : EPI_BIAS, s), what);
                return;
            }
            const float* W32 = static_cast<const float*>(W);
            if (bf16_) {
                hip_check(launch_widen_bf16(W, pw32_, (size_t)N * K, s), "widen");
                W32 = pw32_;
            }
            if (gate)
                hip_check(launch_gemm(Ain, lda, W32, bias, gate, ldy, gate, ldy, m, N, K, EPI_BIAS_MUL_SILU, s), what);
            else
                hip_check(launch_gemm(Ain, lda, W32, bias, R, ldy, Y, ldy, m, N, K, R ? EPI_BIAS_RESIDUAL : EPI_BIAS, s), what);
        };
        for (const Layer& L : layers_) {
            float* k_rows = L.k_cache + (size_t)cache_len_ * kv;
            float* v_rows = L.v_cache + (size_t)cache_len_ * kv;
            hip_check(launch_rmsnorm(ph_, L.ln1, c.eps, m, H, pn_, s), "rmsnorm 1");
            proj(pn_, H, L.wqkv, L.bqkv, nullptr, pq_, H, H, H, nullptr, "q proj");
            proj(pn_, H, at(L.wqkv, (size_t)H * H), L.bqkv ? L.bqkv + H : nullptr, nullptr, k_rows, kv, kv, H, nullptr, "k proj");
            proj(pn_, H, at(L.wqkv, (size_t)(H + kv) * H), L.bqkv ? L.bqkv + H + kv : nullptr, nullptr, v_rows, kv, kv, H, nullptr, "v proj");
            hip_check(launch_rope(pq_, H, m, c.heads, d, cos_, sin_, cache_len_, nullptr, 0, s), "rope q");
            hip_check(launch_rope(L.k_cache, kv, m, c.kv_heads, d, cos_, sin_, cache_len_, nullptr, 1, s), "rope k");
            if (prefill_attention_supported(d)) {
                hip_check(launch_prefill_attention(pq_, H, m, L.k_cache, kv, L.v_cache, kv, cache_len_, c.heads, d, c.heads / c.kv_heads, pctx_, H, s),
                          "attention");
            } else {
                for (int r = 0; r < m; r += 8) {  // 8 query rows at a time against everything cached up to them
                    const int rows = std::min(8, m - r);
                    hip_check(launch_decode_attention(pq_ + (size_t)r * H, H, rows, L.k_cache, kv, L.v_cache, kv, cache_len_ + r + rows, nullptr,
                                                      cache_cap_, c.heads, d, cache_len_ + r, splits_, att_scratch_, pctx_ + (size_t)r * H, H, s,
                                                      c.heads / c.kv_heads), "attention");
                }
            }
            proj(pctx_, H, L.wo, nullptr, ph_, ph_, H, H, H, nullptr, "o proj");
            hip_check(launch_rmsnorm(ph_, L.ln2, c.eps, m, H, pn_, s), "rmsnorm 2");
            proj(pn_, H, L.gate, nullptr, nullptr, pg_, I, I, H, nullptr, "gate");
            proj(pn_, H, L.up, nullptr, nullptr, pu_, I, I, H, pg_, "up + swiglu");
            proj(pg_, I, L.down, nullptr, ph_, ph_, H, H, I, nullptr, "down proj");
        }
        cache_len_ += m;
        if (done + m == n) {
            const int rows = (n - 1) % 8 + 1;
            hip_check(launch_rmsnorm(ph_ + (size_t)(m - rows) * H, final_norm_, c.eps, rows, H, last_, s), "final norm");
            LlmGemvArgs lm;
            lm.X = last_ + (size_t)(rows - 1) * H; lm.ldx = H; lm.rows = 1; lm.W = lm_head_; lm.bf16 = bf16_; lm.n_out = c.vocab; lm.k = H;
            lm.Y0 = logits_; lm.ldy0 = c.vocab;
            hip_check(launch_llm_gemv(lm, s), "lm head");
            last_rows_ = rows;
        }
        hip_check(hipStreamSynchronize(s), "sync");  // pids_ and the activations are reused by the next chunk
    }
}

void LlmModel::forward(const uint32_t* ids, int n)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (n < 1) throw std::runtime_error("forward needs at least one token");
    if (cache_len_ + n > cache_cap_) throw std::runtime_error("context is full");
#ifdef KJARNI_TUNING
    static const int kMinGemmRows = [] {
        const char* v = std::getenv("KJARNI_HIP_LLM_PREFILL_MIN");  // measurements: rows from which the matrix-core route is used
        return v ? std::atoi(v) : 24;
    }();
#else
    constexpr int kMinGemmRows = 24;  // rows from which the matrix-core route is used
#endif
    const int kvd = cfg_.kv_heads * cfg_.head_dim;
    if (n >= kMinGemmRows && cfg_.hidden % 32 == 0 && cfg_.inter % 32 == 0 && kvd % 4 == 0 && cfg_.head_dim % 2 == 0) {
        prefill_rows(ids, n);
    } else {
        for (int i = 0; i < n; i += 8) {
            const int m = std::min(8, n - i);
            hip_check(hipMemcpyAsync(ids_, ids + i, (size_t)m * 4, hipMemcpyHostToDevice, stream_), "H2D ids");
            pass(ids_, m, false);
            cache_len_ += m;
            last_rows_ = m;
            hip_check(hipStreamSynchronize(stream_), "sync");  // ids_ is reused by the next block
        }
    }
    hip_check(hipMemcpyAsync(pos_, &cache_len_, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D pos");
    hip_check(hipStreamSynchronize(stream_), "sync");
}

void LlmModel::last_hidden(float* out, int rows) const
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(stream_), "sync");
    hip_check(hipMemcpy(out, last_, (size_t)std::min(rows, last_rows_) * cfg_.hidden * sizeof(float), hipMemcpyDeviceToHost), "D2H hidden");
}

void LlmModel::logits_to_host(float* out) const
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipStreamSynchronize(stream_), "sync");
    hip_check(hipMemcpy(out, logits_, (size_t)cfg_.vocab * sizeof(float), hipMemcpyDeviceToHost), "D2H logits");
}

void LlmModel::enqueue_argmax(bool record)
{
    hip_check(launch_argmax(logits_, cfg_.vocab, best_, token_, record ? hist_ : nullptr, record ? count_ : nullptr,
                            record ? pos_ : nullptr, stream_), "argmax");
}

uint32_t LlmModel::argmax()
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    enqueue_argmax(false);
    int32_t t = 0;
    hip_check(hipMemcpyAsync(&t, token_, 4, hipMemcpyDeviceToHost, stream_), "D2H token");
    hip_check(hipStreamSynchronize(stream_), "sync");
    return (uint32_t)t;
}

hipGraphExec_t LlmModel::step_graph()
{
    if (graph_) return graph_;
    hipGraph_t graph = nullptr;
    hip_check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "begin capture");
    try {
        pass(reinterpret_cast<const uint32_t*>(token_), 1, true);
        enqueue_argmax(true);
    } catch (...) {
        (void)hipStreamEndCapture(stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    hip_check(hipStreamEndCapture(stream_, &graph), "end capture");
    const hipError_t e = hipGraphInstantiate(&graph_, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    hip_check(e, "graph instantiate");
    return graph_;
}

std::vector<uint32_t> LlmModel::generate(const std::vector<uint32_t>& prompt, size_t max_new_tokens, float repetition_penalty,
                                         int no_repeat_ngram, const std::function<bool(uint32_t)>& on_token)
{
    GenerateOptions o;
    o.max_new_tokens = max_new_tokens;
    o.repetition_penalty = repetition_penalty;
    o.no_repeat_ngram = no_repeat_ngram;
    return generate(prompt, o, on_token);
}

std::vector<uint32_t> LlmModel::generate(const std::vector<uint32_t>& prompt, const GenerateOptions& opt,
                                         const std::function<bool(uint32_t)>& on_token)
{
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (prompt.empty()) throw std::runtime_error("cannot generate from empty prompt");
    if ((int)prompt.size() > cache_cap_) throw std::runtime_error("prompt does not fit the context");
    if (opt.sample && !opt.uniform) throw std::runtime_error("sampling needs a uniform source");
    reset();
    forward(prompt.data(), (int)prompt.size());
    std::vector<uint32_t> out, all(prompt);
    const std::vector<uint32_t>& stops = opt.stop_ids.empty() ? cfg_.eos_ids : opt.stop_ids;
    const auto is_stop = [&](uint32_t t) { return std::find(stops.begin(), stops.end(), t) != stops.end(); };
    // generator.rs:243-246 and 309-317: stop at the model's context and at max_len = prompt + max_new_tokens | max_length.
    const size_t max_len = opt.max_len ? opt.max_len : prompt.size() + opt.max_new_tokens;
    const size_t context_limit = std::min((size_t)cache_cap_, max_len);
    const size_t max_new_tokens = opt.max_new_tokens;
    const float repetition_penalty = opt.repetition_penalty;
    const int no_repeat_ngram = opt.no_repeat_ngram;

    if ((opt.sample || repetition_penalty != 1.0f || no_repeat_ngram > 0) && device_sampling_) {
        // Logits processors and sampling (generator.rs:331-343), with everything that is O(vocab) on the device
        // (llm_kernels.hip): the processors edit the logits where the vocabulary head left them; for a sampled token three
        // small launches cut the vocabulary down to the candidates within reach of the filters and sum the exponentials,
        // and the host receives a 32-byte header + a few hundred (token, logit) pairs instead of 4 x vocab bytes.  It
        // finishes top-k / top-p / min-p / temperature / the draw on them exactly as the reference does on the full array
        // (sampling.cpp); when the candidates cannot decide (rare: a crossing within the rounding of the device's sum, a
        // nearly flat distribution) it fetches the logits -- already processed -- and runs the full-array path.
        const size_t vocab = (size_t)cfg_.vocab;
        const size_t out_bytes = sizeof(SampleHeader) + (size_t)kCandCap * sizeof(SampleCandidate);
        if (!samp_scratch_) {
            samp_scratch_ = dalloc((sample_scratch_bytes() + 3) / 4);
            hip_check(hipMemset(samp_scratch_, 0, sample_scratch_bytes()), "memset");
            samp_out_ = reinterpret_cast<uint8_t*>(dalloc((out_bytes + 3) / 4));
            hip_check(hipMemset(samp_out_, 0, out_bytes), "memset");
            hip_check(hipHostMalloc((void**)&samp_host_, out_bytes, hipHostMallocDefault), "hipHostMalloc");
            samp_tokens_ = reinterpret_cast<int32_t*>(dalloc((size_t)cache_cap_ + 16));
            samp_distinct_ = reinterpret_cast<int32_t*>(dalloc((size_t)cache_cap_ + 16));
            samp_counts_ = reinterpret_cast<int*>(dalloc(vocab));
            samp_ndistinct_ = reinterpret_cast<int*>(dalloc(4));
        }
        if (opt.sample && !host_logits_)
            hip_check(hipHostMalloc((void**)&host_logits_, vocab * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
        SampleHeader* header_dev = reinterpret_cast<SampleHeader*>(samp_out_);
        SampleCandidate* cand_dev = reinterpret_cast<SampleCandidate*>(samp_out_ + sizeof(SampleHeader));
        const SampleHeader* header = reinterpret_cast<const SampleHeader*>(samp_host_);
        const SampleCandidate* cand = reinterpret_cast<const SampleCandidate*>(samp_host_ + sizeof(SampleHeader));
        const bool processors = repetition_penalty != 1.0f || no_repeat_ngram > 0;
        if (processors) {  // the history so far = the prompt
            hip_check(hipMemsetAsync(samp_counts_, 0, vocab * sizeof(int), stream_), "memset counts");
            hip_check(hipMemsetAsync(samp_ndistinct_, 0, sizeof(int), stream_), "memset");
            hip_check(hipMemcpyAsync(samp_tokens_, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_), "H2D history");
            hip_check(launch_token_counts(samp_tokens_, (int)all.size(), (int)vocab, samp_counts_, samp_distinct_, samp_ndistinct_, stream_),
                      "token counts");
            hip_check(hipStreamSynchronize(stream_), "sync");  // (`all` grows below: the copy must have read it)
        }
        std::vector<float> probs, cvals;
        std::vector<uint32_t> ids, cids;
        hipGraphExec_t exec = nullptr;
        int skip_candidates = 0;
        for (size_t step = 0; step < max_new_tokens; ++step) {
            if (all.size() >= context_limit) break;
            if (processors)
                hip_check(launch_logits_processors(logits_, (int)vocab, samp_tokens_, (int)all.size(), samp_counts_, samp_distinct_,
                                                   samp_ndistinct_, repetition_penalty, no_repeat_ngram, stream_), "logits processors");
            uint32_t next;
            if (opt.sample) {
                // A distribution the candidates could not decide (nearly flat: top-p reaches through most of the vocabulary) rarely
                // becomes decidable on the next token: after a miss the cut is not attempted for a few tokens.
                const bool attempt = skip_candidates == 0;
                if (!attempt) --skip_candidates;
                const size_t first = sizeof(SampleHeader) + (size_t)kCandFirst * sizeof(SampleCandidate);
                if (attempt) {
                    hip_check(launch_sample_candidates(logits_, (int)vocab, opt.sampling.top_k, opt.sampling.top_p, opt.sampling.min_p,
                                                       samp_scratch_, header_dev, cand_dev, kCandCap, stream_), "sample candidates");
                    hip_check(hipMemcpyAsync(samp_host_, samp_out_, first, hipMemcpyDeviceToHost, stream_), "D2H candidates");
                    hip_check(hipStreamSynchronize(stream_), "sync");
                }
                bool decided = false;
                if (attempt && !header->overflow && header->count <= (uint32_t)kCandCap) {
                    const size_t n = header->count;
                    if (n > (size_t)kCandFirst) {
                        hip_check(hipMemcpyAsync(samp_host_ + first, samp_out_ + first, (n - kCandFirst) * sizeof(SampleCandidate),
                                                 hipMemcpyDeviceToHost, stream_), "D2H candidates");
                        hip_check(hipStreamSynchronize(stream_), "sync");
                    }
                    cids.resize(n);
                    cvals.resize(n);
                    for (size_t i = 0; i < n; ++i) {
                        cids[i] = cand[i].token;
                        cvals[i] = cand[i].logit;
                    }
                    decided = sampling_distribution_candidates(cids.data(), cvals.data(), n, header->mx, header->floor, header->sum, vocab,
                                                               opt.sampling, ids, probs);
                }
                if (decided) {
                    ++tokens_from_candidates_;
                } else {
                    ++tokens_from_logits_;
                    if (attempt) skip_candidates = 8;
                    hip_check(hipMemcpyAsync(host_logits_, logits_, vocab * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H logits");
                    hip_check(hipStreamSynchronize(stream_), "sync");
                    sampling_distribution(host_logits_, vocab, opt.sampling, ids, probs);  // (the processors already ran, on the device)
                }
                next = sample_from_distribution(ids, probs, opt.uniform(), vocab);
            } else {  // greedy on processed logits: the device's argmax (last maximum wins), four bytes back
                enqueue_argmax(false);
                int32_t t = 0;
                hip_check(hipMemcpyAsync(&t, token_, sizeof(t), hipMemcpyDeviceToHost, stream_), "D2H token");
                hip_check(hipStreamSynchronize(stream_), "sync");
                next = (uint32_t)t;
                ++tokens_from_candidates_;
            }
            if (is_stop(next)) break;
            all.push_back(next);
            out.push_back(next);
            if (on_token && !on_token(next)) break;
            if (all.size() >= context_limit) break;
            if (!exec) exec = step_graph();
            const int32_t tok = (int32_t)next;
            hip_check(hipMemcpyAsync(token_, &tok, sizeof(tok), hipMemcpyHostToDevice, stream_), "H2D token");
            if (processors) {
                int32_t* slot = samp_tokens_ + (all.size() - 1);
                hip_check(hipMemcpyAsync(slot, &tok, sizeof(tok), hipMemcpyHostToDevice, stream_), "H2D history");
                hip_check(launch_token_counts(slot, 1, (int)vocab, samp_counts_, samp_distinct_, samp_ndistinct_, stream_), "token counts");
            }
            hip_check(hipGraphLaunch(exec, stream_), "graph launch");
            cache_len_ += 1;
        }
        return out;
    }

    if (opt.sample || repetition_penalty != 1.0f || no_repeat_ngram > 0) {
        // (device sampling switched off: the checker path of the tests)
        // Logits processors and sampling work on the host copy of the logits (generator.rs:331-343): one pass per token.
        // The logits land in a pinned host buffer (one async copy per token at full PCIe rate).
        if (!host_logits_) hip_check(hipHostMalloc((void**)&host_logits_, (size_t)cfg_.vocab * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
        float* lg = host_logits_;
        const size_t vocab = (size_t)cfg_.vocab;
        std::vector<float> probs;
        std::vector<uint32_t> ids;
        hipGraphExec_t exec = nullptr;
        for (size_t step = 0; step < max_new_tokens; ++step) {
            if (all.size() >= context_limit) break;
            hip_check(hipMemcpyAsync(lg, logits_, vocab * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H logits");
            hip_check(hipStreamSynchronize(stream_), "sync");
            apply_repetition_penalty(lg, vocab, all, repetition_penalty);
            if (no_repeat_ngram > 0) apply_no_repeat_ngram(lg, vocab, all, (size_t)no_repeat_ngram);
            uint32_t next;
            if (opt.sample) {
                sampling_distribution(lg, vocab, opt.sampling, ids, probs);
                next = sample_from_distribution(ids, probs, opt.uniform(), vocab);
            } else {
                next = argmax_last(lg, vocab);
            }
            if (is_stop(next)) break;
            all.push_back(next);
            out.push_back(next);
            if (on_token && !on_token(next)) break;
            if (all.size() >= context_limit) break;
            // The step itself is the replayed graph of the greedy loop: it reads its input token from token_ (overwritten
            // here with the host's choice; the graph's own argmax result is ignored) and advances position and key count
            // on the device.
            if (!exec) exec = step_graph();
            const int32_t tok = (int32_t)next;
            hip_check(hipMemcpyAsync(token_, &tok, sizeof(tok), hipMemcpyHostToDevice, stream_), "H2D token");
            hip_check(hipGraphLaunch(exec, stream_), "graph launch");
            cache_len_ += 1;
        }
        return out;
    }

    // Plain greedy: token, position and key count stay on the device; one graph replay per token, the host
    // looks every few steps.  Tokens computed past a stop token / a stop request are discarded.
    enqueue_argmax(true);
    hip_check(hipMemcpyAsync(pos_, &cache_len_, sizeof(int), hipMemcpyHostToDevice, stream_), "H2D pos");
    std::vector<int32_t> hist((size_t)hist_cap_);
    size_t produced = 0, seen = 0;
    bool done = max_new_tokens == 0;
    auto drain = [&](size_t upto) {
        for (; seen < upto && !done; ++seen) {
            const uint32_t tok = (uint32_t)hist[seen];
            if (all.size() >= context_limit || is_stop(tok)) {
                done = true;
                break;
            }
            all.push_back(tok);
            out.push_back(tok);
            if ((on_token && !on_token(tok)) || out.size() >= max_new_tokens) done = true;
        }
    };
    hip_check(hipMemcpyAsync(hist.data(), hist_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_), "D2H token");
    hip_check(hipStreamSynchronize(stream_), "sync");
    produced = 1;
    drain(1);
    hipGraphExec_t exec = done ? nullptr : step_graph();
    const size_t burst = on_token ? 4 : 16;
    while (!done) {
        size_t steps = std::min(burst, max_new_tokens - out.size());
        steps = std::min(steps, (size_t)cache_cap_ - (size_t)cache_len_);
        if (steps == 0) break;
#ifdef KJARNI_TUNING
        static const bool eager = std::getenv("KJARNI_HIP_LLM_EAGER") != nullptr;  // measurements: the same launches, not replayed
        if (eager) {
            for (size_t i = 0; i < steps; ++i) {
                pass(reinterpret_cast<const uint32_t*>(token_), 1, true);
                enqueue_argmax(true);
            }
        } else
#endif
        for (size_t i = 0; i < steps; ++i) hip_check(hipGraphLaunch(exec, stream_), "graph launch");
        hip_check(hipMemcpyAsync(hist.data() + produced, hist_ + produced, steps * sizeof(int32_t), hipMemcpyDeviceToHost, stream_), "D2H tokens");
        hip_check(hipStreamSynchronize(stream_), "sync");
        produced += steps;
        cache_len_ += (int)steps;
        drain(produced);
    }
    return out;
}

}  // namespace kjarni
