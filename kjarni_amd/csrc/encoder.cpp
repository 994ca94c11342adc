#include "encoder.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "json.h"
#include "safetensors.h"

namespace kjarni {

void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        (void)hipGetLastError();
        throw HipError(msg);
    }
}

int visible_device_count()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static std::string read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

EncoderModel::~EncoderModel()
{
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (void* p : allocs_) (void)hipFree(p);
    for (hipEvent_t e : prof_pool_) (void)hipEventDestroy(e);
    for (const PendingEvent& pe : prof_pending_) {
        (void)hipEventDestroy(pe.start);
        (void)hipEventDestroy(pe.stop);
    }
    for (auto& w : ws_all_) {
        for (void* p : {(void*)w->hidden, (void*)w->qkv, (void*)w->ctx, (void*)w->mid, (void*)w->feat, (void*)w->split, w->stage,
                        (void*)w->tok_src, (void*)w->cu, (void*)w->lens})
            if (p) (void)hipFree(p);
        if (w->pin) (void)hipHostFree(w->pin);
        if (w->lens_host) (void)hipHostFree(w->lens_host);
        if (w->done) (void)hipEventDestroy(w->done);
        if (w->stream) (void)hipStreamDestroy(w->stream);
    }
}

float* EncoderModel::upload(const std::vector<float>& host)
{
    float* d = nullptr;
    const size_t bytes = host.size() * sizeof(float);
    hip_check(hipMalloc((void**)&d, bytes ? bytes : 4), "hipMalloc(weights)");
    allocs_.push_back(d);
    if (bytes) hip_check(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy(weights)");
    weight_bytes_ += bytes;
    return d;
}

namespace {

struct Names {
    std::string emb, q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_g, ln1_b, w1, b1, w2, b2, ln2_g, ln2_b;
    std::string qkv_w;              // one fused [3H,H] tensor instead of q_w / k_w / v_w
    std::string wg;                 // SwiGLU gate
    std::string emb_ln_g, emb_ln_b; // embedding LayerNorm when it is not <emb>LayerNorm.*
};

std::string fmt_layer(const std::string& pattern, int i)
{
    std::string s = pattern;
    const size_t p = s.find("{}");
    if (p != std::string::npos) s.replace(p, 2, std::to_string(i));
    return s;
}

// Tensor-name layouts: crates/kjarni-models/src/models/sentence_encoder/configs.rs
// :218-366 (BERT, plain and "bert."-prefixed) and :638-687 (DistilBERT).
Names bert_names(const std::string& pre)
{
    Names n;
    n.emb = pre + "embeddings.";
    const std::string l = pre + "encoder.layer.{}.";
    n.q_w = l + "attention.self.query.weight";
    n.q_b = l + "attention.self.query.bias";
    n.k_w = l + "attention.self.key.weight";
    n.k_b = l + "attention.self.key.bias";
    n.v_w = l + "attention.self.value.weight";
    n.v_b = l + "attention.self.value.bias";
    n.o_w = l + "attention.output.dense.weight";
    n.o_b = l + "attention.output.dense.bias";
    n.ln1_g = l + "attention.output.LayerNorm.weight";
    n.ln1_b = l + "attention.output.LayerNorm.bias";
    n.w1 = l + "intermediate.dense.weight";
    n.b1 = l + "intermediate.dense.bias";
    n.w2 = l + "output.dense.weight";
    n.b2 = l + "output.dense.bias";
    n.ln2_g = l + "output.LayerNorm.weight";
    n.ln2_b = l + "output.LayerNorm.bias";
    return n;
}

// MPNet: sentence_encoder/configs.rs:426-462 (the relative attention bias of the checkpoint is not read there either).
Names mpnet_names()
{
    Names n;
    n.emb = "embeddings.";
    const std::string l = "encoder.layer.{}.";
    n.q_w = l + "attention.attn.q.weight";
    n.q_b = l + "attention.attn.q.bias";
    n.k_w = l + "attention.attn.k.weight";
    n.k_b = l + "attention.attn.k.bias";
    n.v_w = l + "attention.attn.v.weight";
    n.v_b = l + "attention.attn.v.bias";
    n.o_w = l + "attention.attn.o.weight";
    n.o_b = l + "attention.attn.o.bias";
    n.ln1_g = l + "attention.LayerNorm.weight";
    n.ln1_b = l + "attention.LayerNorm.bias";
    n.w1 = l + "intermediate.dense.weight";
    n.b1 = l + "intermediate.dense.bias";
    n.w2 = l + "output.dense.weight";
    n.b2 = l + "output.dense.bias";
    n.ln2_g = l + "output.LayerNorm.weight";
    n.ln2_b = l + "output.LayerNorm.bias";
    return n;
}

Names distilbert_names(const std::string& pre)
{
    Names n;
    n.emb = pre + "embeddings.";
    const std::string l = pre + "transformer.layer.{}.";
    n.q_w = l + "attention.q_lin.weight";
    n.q_b = l + "attention.q_lin.bias";
    n.k_w = l + "attention.k_lin.weight";
    n.k_b = l + "attention.k_lin.bias";
    n.v_w = l + "attention.v_lin.weight";
    n.v_b = l + "attention.v_lin.bias";
    n.o_w = l + "attention.out_lin.weight";
    n.o_b = l + "attention.out_lin.bias";
    n.ln1_g = l + "sa_layer_norm.weight";
    n.ln1_b = l + "sa_layer_norm.bias";
    n.w1 = l + "ffn.lin1.weight";
    n.b1 = l + "ffn.lin1.bias";
    n.w2 = l + "ffn.lin2.weight";
    n.b2 = l + "ffn.lin2.bias";
    n.ln2_g = l + "output_layer_norm.weight";
    n.ln2_b = l + "output_layer_norm.bias";
    return n;
}

// Nomic (sentence_encoder/configs.rs:219-275): fused Wqkv, no biases anywhere, SwiGLU with fc11 = gate and fc12 = up.
Names nomic_names()
{
    Names n;
    n.emb = "embeddings.";
    n.emb_ln_g = "emb_ln.weight";
    n.emb_ln_b = "emb_ln.bias";
    const std::string l = "encoder.layers.{}.";
    n.qkv_w = l + "attn.Wqkv.weight";
    n.o_w = l + "attn.out_proj.weight";
    n.ln1_g = l + "norm1.weight";
    n.ln1_b = l + "norm1.bias";
    n.wg = l + "mlp.fc11.weight";
    n.w1 = l + "mlp.fc12.weight";
    n.w2 = l + "mlp.fc2.weight";
    n.ln2_g = l + "norm2.weight";
    n.ln2_b = l + "norm2.bias";
    return n;
}

void expect_shape(const std::vector<int64_t>& got, std::initializer_list<int64_t> want,
                  const std::string& name)
{
    std::vector<int64_t> w(want);
    if (got != w) {
        std::string m = "tensor " + name + " has unexpected shape [";
        for (size_t i = 0; i < got.size(); ++i) m += (i ? "," : "") + std::to_string(got[i]);
        m += "], expected [";
        for (size_t i = 0; i < w.size(); ++i) m += (i ? "," : "") + std::to_string(w[i]);
        throw std::runtime_error(m + "]");
    }
}

}  // namespace

std::unique_ptr<EncoderModel> EncoderModel::load(const std::string& dir, int device)
{
    const int ndev = visible_device_count();
    if (ndev <= 0)
        throw GpuUnavailable("no HIP device is visible: this library runs the encoder on an AMD GPU only");
    if (device < 0 || device >= ndev)
        throw GpuUnavailable("HIP device " + std::to_string(device) + " requested but only " +
                             std::to_string(ndev) + " visible");

    std::unique_ptr<EncoderModel> m(new EncoderModel());
    m->device_ = device;
    // (opt-in: a combined call's bits depend on who else happened to call -- off unless asked for)
    if (const char* e = std::getenv("KJARNI_HIP_COMBINE")) m->combining_ = e[0] != '\0' && !(e[0] == '0' && e[1] == '\0');
    if (const char* e = std::getenv("KJARNI_HIP_TWO_LANES")) m->two_lanes_ = !(e[0] == '0' && e[1] == '\0');
    EncoderConfig& c = m->cfg_;
    c.config_json = read_file(dir + "/config.json");
    Json cfg = Json::parse(c.config_json);
    SafeTensors st;
    st.open_dir(dir);

    c.model_type = cfg.get_string("model_type", "bert");
    Names names;
    if (c.model_type == "distilbert") {
        const std::string pre = st.contains("distilbert.embeddings.word_embeddings.weight") ? "distilbert." : "";
        names = distilbert_names(pre);
        c.hidden = (int)cfg.get_int("dim", 768);
        c.layers = (int)cfg.get_int("n_layers", 6);
        c.heads = (int)cfg.get_int("n_heads", 12);
        c.inter = (int)cfg.get_int("hidden_dim", 4 * c.hidden);
        c.eps = 1e-12f;                // configs.rs:620 (DistilBERT norm_eps is fixed)
        c.ffn_act = EPI_BIAS_GELU;     // configs.rs:621
    } else if (c.model_type == "roberta" || c.model_type == "distilroberta") {
        // sequence_classifier/configs.rs:149-280: BERT's layer layout under "roberta.", positions start at 2.
        const std::string pre = st.contains("roberta.embeddings.word_embeddings.weight") ? "roberta." : "";
        names = bert_names(pre);
        c.hidden = (int)cfg.get_int("hidden_size", 0);
        c.layers = (int)cfg.get_int("num_hidden_layers", 0);
        c.heads = (int)cfg.get_int("num_attention_heads", 0);
        c.inter = (int)cfg.get_int("intermediate_size", 0);
        c.eps = (float)cfg.get_double("layer_norm_eps", 1e-5);
        c.pos_offset = 2;  // extra_pos_embeddings (configs.rs:223)
        const std::string act = cfg.get_string("hidden_act", "gelu");
        c.ffn_act = act == "gelu_new" ? EPI_BIAS_GELU_NEW : (act == "relu" ? EPI_BIAS_RELU : EPI_BIAS_GELU);  // configs.rs:217-222
    } else if (c.model_type == "mpnet") {
        names = mpnet_names();
        c.hidden = (int)cfg.get_int("hidden_size", 0);
        c.layers = (int)cfg.get_int("num_hidden_layers", 0);
        c.heads = (int)cfg.get_int("num_attention_heads", 0);
        c.inter = (int)cfg.get_int("intermediate_size", 0);
        c.eps = (float)cfg.get_double("layer_norm_eps", 1e-5);
        c.pos_offset = 2;                 // sentence_encoder/configs.rs:416
        c.ffn_act = EPI_BIAS_GELU_NEW;    // configs.rs:410: the tanh form whatever hidden_act says
    } else if (c.model_type == "nomic_bert") {
        // BertConfig with its serde aliases (sentence_encoder/configs.rs:15-27, 43-44, 140-149)
        auto key = [&](const char* a, const char* b) { return (int)cfg.get_int(a, cfg.get_int(b, 0)); };
        names = nomic_names();
        c.hidden = key("hidden_size", "n_embd");
        c.layers = key("num_hidden_layers", "n_layer");
        c.heads = key("num_attention_heads", "n_head");
        c.inter = key("intermediate_size", "n_inner");
        if (c.inter <= 0) c.inter = 4 * c.hidden;
        if (!cfg.find("layer_norm_eps") && !cfg.find("layer_norm_epsilon"))
            throw std::runtime_error("config.json: missing field `layer_norm_eps`");
        c.eps = (float)cfg.get_double("layer_norm_eps", cfg.get_double("layer_norm_epsilon", 1e-12));
        c.gated_ffn = true;  // the layout always names a gate (configs.rs:243-246)
        c.max_pos = key("n_positions", "max_position_embeddings");
        if (c.max_pos <= 0) c.max_pos = 512;
        // configs.rs:175-183: either rotary key switches RoPE on; theta defaults to 10000
        if (cfg.find("rotary_emb_fraction") || cfg.find("rotary_emb_base") || cfg.find("rotary_embedding_fraction") ||
            cfg.find("rotary_embedding_base"))
            c.rope_theta = (float)cfg.get_double("rotary_emb_base", cfg.get_double("rotary_embedding_base", 10000.0));
    } else {
        // "Default to BertConfig for BERT-like models" (sentence_encoder/model.rs:50-53, sequence_classifier/mod.rs:60-63):
        // plain BERT and every other model_type with BERT's tensor names, e.g. bge-m3's "xlm-roberta" (positions from 0 here,
        // as the reference reads it).
        const std::string pre = st.contains("bert.embeddings.word_embeddings.weight") ? "bert." : "";
        names = bert_names(pre);
        c.hidden = (int)cfg.get_int("hidden_size", 0);
        c.layers = (int)cfg.get_int("num_hidden_layers", 0);
        c.heads = (int)cfg.get_int("num_attention_heads", 0);
        c.inter = (int)cfg.get_int("intermediate_size", 0);
        if (c.inter <= 0) c.inter = 4 * c.hidden;  // configs.rs:160-167
        c.eps = (float)cfg.get_double("layer_norm_eps", 1e-12);
        // configs.rs:194-200: "gelu" -> erf GELU, "gelu_new" -> tanh GELU, "relu"; anything else erf GELU
        std::string act = cfg.get_string("hidden_act", cfg.get_string("activation_function", "gelu"));
        if (act == "gelu_new") c.ffn_act = EPI_BIAS_GELU_NEW;
        else if (act == "relu") c.ffn_act = EPI_BIAS_RELU;
        else if (act == "swiglu") throw std::runtime_error("activation 'swiglu' needs model_type 'nomic_bert' (a gate weight)");
        else c.ffn_act = EPI_BIAS_GELU;
    }
    if (c.hidden <= 0 || c.layers <= 0 || c.heads <= 0 || c.hidden % c.heads != 0)
        throw std::runtime_error("invalid encoder dimensions in config.json");

    hip_check(hipSetDevice(device), "hipSetDevice");

    std::vector<float> buf, buf2, buf3;
    const int H = c.hidden, I = c.inter;
    auto shape = st.read_f32(names.emb + "word_embeddings.weight", buf);
    if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("word_embeddings has wrong shape");
    c.vocab = (int)shape[0];
    m->word_ = m->upload(buf);
    if (c.model_type != "nomic_bert") {  // Nomic's layout has position_embedding: None (configs.rs:254)
        shape = st.read_f32(names.emb + "position_embeddings.weight", buf);
        if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("position_embeddings has wrong shape");
        c.max_pos = (int)shape[0];
        m->pos_ = m->upload(buf);
    }
    if (c.rope_theta > 0.0f) {
        // RoPE::new(head_dim, max_seq_len, theta): rope/mod.rs:57-62 (inverse frequencies), :96-116 (caches)
        const int d = H / c.heads, half = d / 2;
        if (d % 2) throw std::runtime_error("RoPE needs an even head dimension");
        std::vector<float> inv((size_t)half), cs((size_t)c.max_pos * d), sn((size_t)c.max_pos * d);
        for (int i = 0; i < half; ++i) inv[(size_t)i] = 1.0f / std::pow(c.rope_theta, (float)(2 * i) / (float)d);
        for (int p = 0; p < c.max_pos; ++p)
            for (int i = 0; i < half; ++i) {
                const float angle = (float)p * inv[(size_t)i];
                const float cv = std::cos(angle), sv = std::sin(angle);
                cs[(size_t)p * d + i] = cs[(size_t)p * d + i + half] = cv;
                sn[(size_t)p * d + i] = sn[(size_t)p * d + i + half] = sv;
            }
        m->rope_cos_ = m->upload(cs);
        m->rope_sin_ = m->upload(sn);
    }
    const bool typed_family = c.model_type != "distilbert" && c.model_type != "mpnet";
    if (typed_family && st.contains(names.emb + "token_type_embeddings.weight")) {
        shape = st.read_f32(names.emb + "token_type_embeddings.weight", buf);
        if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("token_type_embeddings has wrong shape");
        c.type_vocab = (int)shape[0];
        m->type_ = m->upload(buf);
    }
    const std::string eg = names.emb_ln_g.empty() ? names.emb + "LayerNorm.weight" : names.emb_ln_g;
    const std::string eb = names.emb_ln_b.empty() ? names.emb + "LayerNorm.bias" : names.emb_ln_b;
    expect_shape(st.read_f32(eg, buf), {H}, eg);
    m->emb_ln_g_ = m->upload(buf);
    expect_shape(st.read_f32(eb, buf), {H}, eb);
    m->emb_ln_b_ = m->upload(buf);

    m->layers_.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        DeviceLayer& L = m->layers_[i];
        // Fused [3H,H] QKV weight (cpu/encoder/qkv_projection.rs:30-41).
        std::vector<float> wqkv((size_t)3 * H * H), bqkv((size_t)3 * H);
        const std::string* wn[3] = {&names.q_w, &names.k_w, &names.v_w};
        const std::string* bn[3] = {&names.q_b, &names.k_b, &names.v_b};
        if (!names.qkv_w.empty()) {
            // the checkpoint already holds Q | K | V rows in one tensor (transformer_encoder.rs:75-83); no bias
            const std::string w_name = fmt_layer(names.qkv_w, i);
            expect_shape(st.read_f32(w_name, wqkv), {3 * H, H}, w_name);
        }
        for (int p = 0; p < 3 && names.qkv_w.empty(); ++p) {
            const std::string w_name = fmt_layer(*wn[p], i), b_name = fmt_layer(*bn[p], i);
            expect_shape(st.read_f32(w_name, buf), {H, H}, w_name);
            std::memcpy(wqkv.data() + (size_t)p * H * H, buf.data(), sizeof(float) * H * H);
            if (st.contains(b_name)) {
                expect_shape(st.read_f32(b_name, buf), {H}, b_name);
                std::memcpy(bqkv.data() + (size_t)p * H, buf.data(), sizeof(float) * H);
            } else {
                std::fill(bqkv.begin() + (size_t)p * H, bqkv.begin() + (size_t)(p + 1) * H, 0.0f);
            }
        }
        L.wqkv = m->upload(wqkv);
        L.bqkv = names.qkv_w.empty() ? m->upload(bqkv) : nullptr;
        auto up = [&](const std::string& pattern, std::initializer_list<int64_t> want) {
            const std::string name = fmt_layer(pattern, i);
            expect_shape(st.read_f32(name, buf), want, name);
            return m->upload(buf);
        };
        auto up_bias = [&](const std::string& pattern, int n) -> float* {
            if (pattern.empty()) return nullptr;  // the layout has no such bias
            const std::string name = fmt_layer(pattern, i);
            if (!st.contains(name)) {
                buf.assign((size_t)n, 0.0f);
                return m->upload(buf);
            }
            expect_shape(st.read_f32(name, buf), {n}, name);
            return m->upload(buf);
        };
        L.wo = up(names.o_w, {H, H});
        L.bo = up_bias(names.o_b, H);
        L.ln1_g = up(names.ln1_g, {H});
        L.ln1_b = up(names.ln1_b, {H});
        if (!names.wg.empty()) L.wg = up(names.wg, {I, H});
        L.w1 = up(names.w1, {I, H});
        L.b1 = up_bias(names.b1, I);
        L.w2 = up(names.w2, {H, I});
        L.b2 = up_bias(names.b2, H);
        L.ln2_g = up(names.ln2_g, {H});
        L.ln2_b = up(names.ln2_b, {H});
    }

    // Classification head auto-detection: cpu/encoder/classifier.rs:103-202.
    std::string dense_w, dense_b, cls_w, cls_b;
    if (st.contains("classification_head.dense.weight")) {
        dense_w = "classification_head.dense.weight"; dense_b = "classification_head.dense.bias";
        cls_w = "classification_head.out_proj.weight"; cls_b = "classification_head.out_proj.bias";
        c.head_kind = 1;
    } else if (st.contains("classifier.dense.weight")) {
        dense_w = "classifier.dense.weight"; dense_b = "classifier.dense.bias";
        cls_w = "classifier.out_proj.weight"; cls_b = "classifier.out_proj.bias";
        c.head_kind = 1;
    } else if (st.contains("pre_classifier.weight")) {
        dense_w = "pre_classifier.weight"; dense_b = "pre_classifier.bias";
        cls_w = "classifier.weight"; cls_b = "classifier.bias";
        c.head_kind = 2;
    } else if (st.contains("bert.pooler.dense.weight") && st.contains("classifier.weight")) {
        dense_w = "bert.pooler.dense.weight"; dense_b = "bert.pooler.dense.bias";
        cls_w = "classifier.weight"; cls_b = "classifier.bias";
        c.head_kind = 1;
    } else if (st.contains("classifier.weight")) {
        cls_w = "classifier.weight"; cls_b = "classifier.bias";
        c.head_kind = 3;
    }
    if (c.head_kind != 0) {
        if (!dense_w.empty()) {
            expect_shape(st.read_f32(dense_w, buf), {H, H}, dense_w);
            m->head_dense_w_ = m->upload(buf);
            if (st.contains(dense_b)) st.read_f32(dense_b, buf); else buf.assign((size_t)H, 0.0f);
            m->head_dense_b_ = m->upload(buf);
        }
        shape = st.read_f32(cls_w, buf);
        if (shape.size() != 2 || shape[1] != H) throw std::runtime_error("classifier weight has wrong shape");
        c.num_labels = (int)shape[0];
        m->head_cls_w_ = m->upload(buf);
        if (st.contains(cls_b)) st.read_f32(cls_b, buf); else buf.assign((size_t)c.num_labels, 0.0f);
        m->head_cls_b_ = m->upload(buf);
        // id2label in id order
        if (const Json* id2 = cfg.find("id2label")) {
            if (id2->is_object()) {
                c.labels.assign((size_t)c.num_labels, std::string());
                for (const auto& kv : id2->obj) {
                    const long id = std::strtol(kv.first.c_str(), nullptr, 10);
                    if (id >= 0 && id < c.num_labels && kv.second.is_string()) c.labels[(size_t)id] = kv.second.str;
                }
            }
        }
    }
    return m;
}

int64_t EncoderModel::sentences_per_chunk(int seq) const
{
    int64_t n = chunk_tokens_ / std::max(seq, 1);
    return n < 1 ? 1 : n;
}

// ---- workspace pool ---------------------------------------------------------------------------------

EncoderModel::Lease::Lease(EncoderModel& m, hipStream_t stream, bool own_stream) : m_(m), ws_(nullptr), stream_(stream)
{
    hip_check(hipSetDevice(m.device_), "hipSetDevice");
    {
        std::unique_lock<std::mutex> lock(m.ws_mu_);
        for (;;) {
            if (!m.ws_free_.empty()) {
                ws_ = m.ws_free_.back();
                m.ws_free_.pop_back();
                break;
            }
            if ((int)m.ws_all_.size() < kMaxWorkspaces) {
                m.ws_all_.push_back(std::make_unique<Workspace>());
                ws_ = m.ws_all_.back().get();
                break;
            }
            m.ws_cv_.wait(lock);
        }
    }
    try {
        if (!ws_->stream) hip_check(hipStreamCreateWithFlags(&ws_->stream, hipStreamNonBlocking), "hipStreamCreate");
        if (!ws_->done) hip_check(hipEventCreateWithFlags(&ws_->done, hipEventDisableTiming), "hipEventCreate");
        if (own_stream) stream_ = ws_->stream;
        // The previous user's launches may still be running on another stream: order this stream behind them.
        if (ws_->done_pending && ws_->done_stream != stream_)
            hip_check(hipStreamWaitEvent(stream_, ws_->done, 0), "hipStreamWaitEvent");
    } catch (...) {
        std::lock_guard<std::mutex> lock(m_.ws_mu_);
        m_.ws_free_.push_back(ws_);
        m_.ws_cv_.notify_one();
        throw;
    }
}

EncoderModel::Lease::~Lease()
{
    if (hipEventRecord(ws_->done, stream_) == hipSuccess) {
        ws_->done_pending = true;
        ws_->done_stream = stream_;
    } else {
        // Cannot order the next user behind this call's launches: drain them here instead.
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream_);
        ws_->done_pending = false;
    }
    std::lock_guard<std::mutex> lock(m_.ws_mu_);
    m_.ws_free_.push_back(ws_);
    m_.ws_cv_.notify_one();
}

// Grows the activation buffers.  The recorded capacity drops to 0 BEFORE anything is freed, so a failed
// hipMalloc (an oversized batch) leaves a workspace that the next, smaller call re-allocates instead of
// launching kernels on null buffers.
void EncoderModel::reserve(Workspace& ws, int64_t tokens, int64_t sentences)
{
    if (tokens <= ws.tokens && sentences <= ws.sentences) return;
    // Earlier launches on this workspace may still be reading the buffers.
    if (ws.done_pending) {
        hip_check(hipEventSynchronize(ws.done), "hipEventSynchronize(workspace)");
        ws.done_pending = false;
    }
    const size_t H = (size_t)cfg_.hidden, I = (size_t)cfg_.inter;
    // the per-token buffers and the per-sentence buffer grow independently (a ragged call on packed rows holds more
    // sentences per chunk than a padded one of the same token count)
    if (tokens > ws.tokens) {
        ws.tokens = 0;
        ws.split_floats = 0;
        for (float** p : {&ws.hidden, &ws.qkv, &ws.ctx, &ws.mid, &ws.split})
            if (*p) {
                (void)hipFree(*p);
                *p = nullptr;
            }
        if (ws.tok_src) {
            (void)hipFree(ws.tok_src);
            ws.tok_src = nullptr;
        }
        const size_t T = (size_t)tokens;
        hip_check(hipMalloc((void**)&ws.hidden, T * H * 4), "hipMalloc(ws_hidden)");
        hip_check(hipMalloc((void**)&ws.qkv, T * 3 * H * 4), "hipMalloc(ws_qkv)");
        hip_check(hipMalloc((void**)&ws.ctx, T * H * 4), "hipMalloc(ws_ctx)");
        hip_check(hipMalloc((void**)&ws.mid, T * I * 4), "hipMalloc(ws_mid)");
        hip_check(hipMalloc((void**)&ws.tok_src, T * sizeof(int32_t)), "hipMalloc(ws_tok_src)");
        const size_t split_floats = gemm_scratch_floats(tokens, cfg_.hidden);
        hip_check(hipMalloc((void**)&ws.split, split_floats * 4), "hipMalloc(ws_split)");
        ws.split_floats = split_floats;
        ws.tokens = tokens;
    }
    if (sentences > ws.sentences) {
        ws.sentences = 0;
        if (ws.feat) {
            (void)hipFree(ws.feat);
            ws.feat = nullptr;
        }
        hip_check(hipMalloc((void**)&ws.feat, (size_t)sentences * H * 4), "hipMalloc(ws_feat)");
        ws.sentences = sentences;
    }
}

void* EncoderModel::reserve_stage(Workspace& ws, size_t bytes)
{
    if (bytes > ws.stage_bytes) {
        if (ws.done_pending) {
            hip_check(hipEventSynchronize(ws.done), "hipEventSynchronize(stage)");
            ws.done_pending = false;
        }
        ws.stage_bytes = 0;
        if (ws.stage) (void)hipFree(ws.stage);
        ws.stage = nullptr;
        hip_check(hipMalloc(&ws.stage, bytes), "hipMalloc(stage)");
        ws.stage_bytes = bytes;
    }
    return ws.stage;
}

namespace {
const char* const kKindNames[KK_COUNT] = {"embed_layernorm", "gemm_qkv", "attention", "gemm_out_proj", "layernorm",
                                          "gemm_fc1", "gemm_fc2", "pool", "head", "rope"};
}

void EncoderModel::profile_begin(uint32_t kinds_mask)
{
    DeviceGuard guard;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::lock_guard<std::mutex> lock(prof_mu_);
    prof_mask_ = kinds_mask;
    for (const PendingEvent& pe : prof_pending_) {
        prof_pool_.push_back(pe.start);
        prof_pool_.push_back(pe.stop);
    }
    prof_pending_.clear();
    const char* act_sym = cfg_.ffn_act == EPI_BIAS_GELU ? "gemm_nt_f32_mfma<EPI_BIAS_GELU>"
                          : cfg_.ffn_act == EPI_BIAS_GELU_NEW ? "gemm_nt_f32_mfma<EPI_BIAS_GELU_NEW>"
                                                              : "gemm_nt_f32_mfma<EPI_BIAS_RELU>";
    const char* res_sym = fuse_layernorm() ? "gemm_nt_f32_mfma_ln" : "gemm_nt_f32_mfma<EPI_BIAS_RESIDUAL>";
    const char* syms[KK_COUNT] = {"embed_layernorm_kernel", "gemm_nt_f32_mfma<EPI_BIAS>", "attention_kernel",
                                  res_sym, "layernorm_kernel", act_sym, res_sym, "pool_kernel", "head",
                                  "rope_qk_kernel"};
    for (int k = 0; k < KK_COUNT; ++k) {
        prof_stats_[k] = KernelStat();
        prof_stats_[k].kind = kKindNames[k];
        prof_stats_[k].symbol = syms[k];
    }
    prof_on_ = true;
}

hipEvent_t EncoderModel::prof_start(int kind, hipStream_t stream, double flops, double bytes)
{
    if (!prof_on_) return nullptr;  // racy read is fine: the profiler is a single-caller tool
    std::lock_guard<std::mutex> lock(prof_mu_);
    if (!prof_on_ || !((prof_mask_ >> kind) & 1u)) return nullptr;
    auto get = [&]() {
        hipEvent_t e;
        if (!prof_pool_.empty()) {
            e = prof_pool_.back();
            prof_pool_.pop_back();
        } else {
            hip_check(hipEventCreate(&e), "hipEventCreate");
        }
        return e;
    };
    PendingEvent pe{kind, get(), get()};
    hip_check(hipEventRecord(pe.start, stream), "hipEventRecord");
    prof_pending_.push_back(pe);
    prof_stats_[kind].launches += 1;
    prof_stats_[kind].flops += flops;
    prof_stats_[kind].bytes += bytes;
    return pe.stop;
}

void EncoderModel::prof_stop(hipEvent_t stop, hipStream_t stream)
{
    if (stop) hip_check(hipEventRecord(stop, stream), "hipEventRecord");
}

std::vector<KernelStat> EncoderModel::profile_end()
{
    DeviceGuard guard;
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    std::lock_guard<std::mutex> lock(prof_mu_);
    for (const PendingEvent& pe : prof_pending_) {
        float ms = 0.0f;
        hip_check(hipEventElapsedTime(&ms, pe.start, pe.stop), "hipEventElapsedTime");
        prof_stats_[pe.kind].total_ms += ms;
        prof_pool_.push_back(pe.start);
        prof_pool_.push_back(pe.stop);
    }
    prof_pending_.clear();
    prof_on_ = false;
    return std::vector<KernelStat>(prof_stats_, prof_stats_ + KK_COUNT);
}

// embed -> embed_norm -> layers (post-norm):
//   h1 = LN1(x + Attn(x)); y = LN2(h1 + FFN(h1))
// (cpu/encoder/encoder_layer.rs:113-179 / 216-232, transformer_encoder.rs:335-368;
// no final norm, :300-302).  `hidden` is both the residual stream and the output.
void EncoderModel::forward_chunk(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                                 int64_t batch, int seq, float mask_value, float* hidden,
                                 hipStream_t stream, const PackView* pack)
{
    const int H = cfg_.hidden, I = cfg_.inter;
    // rows of every activation buffer: all tokens of the padded layout, or the kept tokens only (packed rows)
    const int64_t T = pack ? pack->tokens : batch * seq;
    const int32_t* cu = pack ? pack->cu : nullptr;
    const int32_t* tok_src = pack ? pack->tok_src : nullptr;
    const double Td = (double)T, Hd = H, Id = I;
    // Algorithmic work per launch (2*M*N*K per GEMM; attention = QK^T + PV), and
    // algorithmic bytes (each operand read once, each output written once).
    const double f_qkv = 2.0 * Td * 3 * Hd * Hd, b_qkv = 4.0 * (Td * Hd + 3 * Hd * Hd + Td * 3 * Hd);
    const double f_att = 4.0 * (pack ? pack->sum_len_sq : (double)batch * seq * seq) * Hd, b_att = 4.0 * (Td * 3 * Hd + Td * Hd);
    const double f_out = 2.0 * Td * Hd * Hd, b_out = 4.0 * (Td * Hd * 3 + Hd * Hd);
    const double f_fc1 = 2.0 * Td * Hd * Id, b_fc1 = 4.0 * (Td * Hd + Hd * Id + Td * Id);
    const double f_fc2 = 2.0 * Td * Hd * Id, b_fc2 = 4.0 * (Td * Id + Hd * Id + 2 * Td * Hd);
    const double b_ln = 8.0 * Td * Hd;
    // Up to gemm_few_rows_max(H) tokens the projections take the few-rows kernel (K split over the waves of a workgroup,
    // gemm.hip) and LayerNorm stays its own small launch (after FC2: the reduce of its K slices); beyond that the residual
    // projections carry it in their epilogue.
    const GemmScratch sc{ws.split, ws.split_floats};
    const int64_t few = gemm_few_rows_max(H);
    const bool fused_ln = T > few && (fuse_layernorm() || (gemm_mid_layernorm_supported(T, H, H) && gemm_mid_layernorm_supported(T, H, I)));

    hipEvent_t pe = prof_start(KK_EMBED_LN, stream, 0.0, 4.0 * (2 * Td + 2 * Td * Hd));
    hip_check(launch_embed_layernorm(ids, type_ids, word_, pos_, type_, emb_ln_g_, emb_ln_b_, cfg_.eps, T,
                                     seq, H, cfg_.vocab, cfg_.max_pos, cfg_.type_vocab, cfg_.pos_offset,
                                     0, hidden, stream, tok_src),
              "embed_layernorm");
    prof_stop(pe, stream);
    // hidden = LN(A W^T + b + hidden), in place: a workgroup owns whole rows, reads its residual rows before it
    // writes them, and no other workgroup touches those rows.
    auto residual_ln = [&](int kind, const float* A, int K, const float* W, const float* b, const float* g,
                           const float* beta, double flops, double bytes, const char* what) {
        // (a handful of rows: only the long-K projection takes the fused route -- K slices + a LayerNorm reduce)
        if (fused_ln || (T <= few && gemm_mid_layernorm_supported(T, H, K))) {
            hipEvent_t e = prof_start(kind, stream, flops, bytes);
            hip_check(launch_gemm_residual_layernorm(A, K, W, b, hidden, H, g, beta, cfg_.eps, hidden, H, T, H, K,
                                                     stream, sc), what);
            prof_stop(e, stream);
            return;
        }
        hipEvent_t e = prof_start(kind, stream, flops, bytes);
        hip_check(launch_gemm(A, K, W, b, hidden, H, hidden, H, T, H, K, EPI_BIAS_RESIDUAL, stream, sc), what);
        prof_stop(e, stream);
        e = prof_start(KK_LAYERNORM, stream, 0.0, b_ln);
        hip_check(launch_layernorm(hidden, g, beta, cfg_.eps, T, H, hidden, stream), "layernorm");
        prof_stop(e, stream);
    };
    for (const DeviceLayer& L : layers_) {
        pe = prof_start(KK_GEMM_QKV, stream, f_qkv, b_qkv);
        hip_check(launch_gemm(hidden, H, L.wqkv, L.bqkv, nullptr, 0, ws.qkv, 3 * H, T, 3 * H, H, EPI_BIAS, stream, sc),
                  "gemm(qkv)");
        prof_stop(pe, stream);
        if (rope_cos_) {
            pe = prof_start(KK_ROPE, stream, 0.0, 4.0 * (4 * Td * Hd));
            hip_check(launch_rope_qk(ws.qkv, rope_cos_, rope_sin_, T, seq, cfg_.heads, H / cfg_.heads, stream, tok_src), "rope");
            prof_stop(pe, stream);
        }
        pe = prof_start(KK_ATTENTION, stream, f_att, b_att);
        hip_check(launch_attention(ws.qkv, pack ? nullptr : mask, batch, pack ? pack->max_len : seq, cfg_.heads, H / cfg_.heads,
                                   mask_value, ws.ctx, stream, cu),
                  "attention");
        prof_stop(pe, stream);
        residual_ln(KK_GEMM_OUT, ws.ctx, H, L.wo, L.bo, L.ln1_g, L.ln1_b, f_out, b_out, "gemm(out_proj)");
        if (L.wg) {
            // SwiGLU (cpu/feedforward/swiglu.rs:40-50): the gate projection lands in ws.mid, the up projection's
            // epilogue multiplies it by silu(gate) in place (read then written by the same thread).
            pe = prof_start(KK_GEMM_FC1, stream, f_fc1, b_fc1);
            hip_check(launch_gemm(hidden, H, L.wg, nullptr, nullptr, 0, ws.mid, I, T, I, H, EPI_BIAS, stream, sc), "gemm(gate)");
            prof_stop(pe, stream);
            pe = prof_start(KK_GEMM_FC1, stream, f_fc1, b_fc1 + 4.0 * Td * Id);
            hip_check(launch_gemm(hidden, H, L.w1, L.b1, ws.mid, I, ws.mid, I, T, I, H, EPI_BIAS_MUL_SILU, stream, sc),
                      "gemm(up * silu(gate))");
            prof_stop(pe, stream);
        } else {
            pe = prof_start(KK_GEMM_FC1, stream, f_fc1, b_fc1);
            hip_check(launch_gemm(hidden, H, L.w1, L.b1, nullptr, 0, ws.mid, I, T, I, H, cfg_.ffn_act, stream, sc),
                      "gemm(fc1)");
            prof_stop(pe, stream);
        }
        residual_ln(KK_GEMM_FC2, ws.mid, I, L.w2, L.b2, L.ln2_g, L.ln2_b, f_fc2, b_fc2, "gemm(fc2)");
    }
}

bool EncoderModel::fuse_layernorm() const
{
    return gemm_residual_layernorm_supported(cfg_.hidden, cfg_.hidden) &&
           gemm_residual_layernorm_supported(cfg_.hidden, cfg_.inter);
}

void EncoderModel::hidden_states(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                                 int64_t batch, int seq, float mask_value, float* out, hipStream_t stream)
{
    if (batch <= 0 || seq <= 0) return;
    Lease lease(*this, stream, false);
    const int64_t per = sentences_per_chunk(seq);
    reserve(lease.ws(), std::min(per, batch) * seq, std::min(per, batch));
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int64_t nb = std::min(per, batch - b0);
        forward_chunk(lease.ws(), ids + b0 * seq, mask ? mask + b0 * seq : nullptr,
                      type_ids ? type_ids + b0 * seq : nullptr, nb, seq, mask_value,
                      out + b0 * seq * (int64_t)cfg_.hidden, stream);
    }
}

// Ragged batches.  The reference pads every sentence of a call to the longest (BatchLongest, pipeline/encoder/
// loader.rs:98-115) and computes all rows; a padded token's hidden state is observable through neither embed (pooling
// skips it, pooling/mod.rs:11-33) nor logits (token 0 only), and as a KEY it contributes exactly 0 to every kept
// query's softmax (utils/masks.rs:4-36: score overwritten by -1e9 / -inf).  So embed / logits run the layers over the
// kept tokens only; hidden_states (which returns padded rows too) keeps the padded layout.  Not packed: calls without
// padding, a sentence whose token 0 is masked (CLS / the all-masked row rules need the padded rows), mask values other
// than 0 / 1.
void EncoderModel::plan_packing(Workspace& ws, const uint32_t* mask_dev, const uint32_t* mask_host, int64_t batch, int seq,
                                hipStream_t stream, PackPlan& plan)
{
    plan.packed = false;
    const int mode = packing_;
    if (mode == 0 || !mask_dev || batch <= 0 || seq <= 1) return;
    std::vector<uint32_t> lens_vec;
    const uint32_t* lens = nullptr;
    if (mask_host) {
        lens_vec.resize((size_t)batch);
        for (int64_t b = 0; b < batch; ++b) {
            const uint32_t* row = mask_host + b * seq;
            uint32_t n = 0, over = 0;
            for (int s = 0; s < seq; ++s) {
                n += row[s] != 0u;
                over |= row[s];
            }
            lens_vec[(size_t)b] = n | ((over > 1u || row[0] == 0u) ? 0x80000000u : 0u);
        }
        lens = lens_vec.data();
    } else {
        // The mask lives on the device (packing mode 2 only: the call then blocks on `stream`): one small kernel + a read-back
        // of 4 bytes per sentence into pinned memory.  A single sentence is not worth the round trip (a lone sentence is
        // normally as long as its padded length); a capturing stream cannot be synchronised, so it takes the padded layout.
        if (mode < 2 || batch == 1) return;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        if (cap != hipStreamCaptureStatusNone) return;
        // (pinned buffer: lens [batch] on the way down, then cu [batch + chunks] on the way up -- 2 * batch + 64 words cover both)
        const size_t want = 2 * (size_t)batch + 64;
        if (want > ws.lens_cap) {
            if (ws.done_pending) {
                hip_check(hipEventSynchronize(ws.done), "hipEventSynchronize(lens)");
                ws.done_pending = false;
            }
            ws.lens_cap = 0;
            if (ws.lens) (void)hipFree(ws.lens);
            if (ws.lens_host) (void)hipHostFree(ws.lens_host);
            ws.lens = nullptr;
            ws.lens_host = nullptr;
            hip_check(hipMalloc((void**)&ws.lens, (size_t)batch * sizeof(uint32_t)), "hipMalloc(lens)");
            hip_check(hipHostMalloc((void**)&ws.lens_host, want * sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc(lens)");
            ws.lens_cap = want;
        }
        hip_check(launch_mask_lengths(mask_dev, batch, seq, ws.lens, stream), "mask_lengths");
        hip_check(hipMemcpyAsync(ws.lens_host, ws.lens, (size_t)batch * sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "D2H lens");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(lens)");
        lens_vec.assign(ws.lens_host, ws.lens_host + batch);  // (the pinned buffer is reused for cu below)
        lens = lens_vec.data();
    }
    int64_t total = 0;
    for (int64_t b = 0; b < batch; ++b) {
        if (lens[b] & 0x80000000u) return;
        total += lens[b];
    }
    if (total == batch * seq) return;  // nothing is padded

    // chunks: maximal runs of whole sentences within chunk_tokens_ packed rows (a row index must also fit 31 bits)
    const int64_t cap = std::min<int64_t>(std::max<int64_t>(chunk_tokens_, seq), ((int64_t)1 << 31) - seq - 1);  // (cu is 32-bit)
    const int64_t max_nb = std::max<int64_t>(1, ((int64_t)1 << 31) / seq - 1);
    plan.cu.reserve((size_t)batch + 64);
    for (int64_t b = 0; b < batch;) {
        PackPlan::Chunk c{b, 0, 0, 0, 0.0, plan.cu.size()};
        plan.cu.push_back(0);
        while (b < batch && c.nb < max_nb && c.tokens + (int64_t)lens[(size_t)b] <= cap) {
            const int l = (int)lens[(size_t)b];
            c.tokens += l;
            c.max_len = std::max(c.max_len, l);
            c.sum_len_sq += (double)l * l;
            plan.cu.push_back((int32_t)c.tokens);
            ++c.nb;
            ++b;
        }
        plan.max_tokens = std::max(plan.max_tokens, c.tokens);
        plan.max_sentences = std::max(plan.max_sentences, c.nb);
        plan.chunks.push_back(c);
    }
    if (plan.cu.size() > ws.cu_ints) {
        if (ws.done_pending) {
            hip_check(hipEventSynchronize(ws.done), "hipEventSynchronize(cu)");
            ws.done_pending = false;
        }
        ws.cu_ints = 0;
        if (ws.cu) (void)hipFree(ws.cu);
        ws.cu = nullptr;
        hip_check(hipMalloc((void**)&ws.cu, plan.cu.size() * sizeof(int32_t)), "hipMalloc(cu)");
        ws.cu_ints = plan.cu.size();
    }
    if (!mask_host && plan.cu.size() <= ws.lens_cap) {
        // device-pointer call: the prefix sums go up from the workspace's pinned buffer, which outlives the call (the workspace is
        // not leased again before this stream has passed its `done` event) -- no second synchronisation
        std::memcpy(ws.lens_host, plan.cu.data(), plan.cu.size() * sizeof(int32_t));
        hip_check(hipMemcpyAsync(ws.cu, ws.lens_host, plan.cu.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream), "H2D cu");
    } else {
        hip_check(hipMemcpyAsync(ws.cu, plan.cu.data(), plan.cu.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream), "H2D cu");
        // (device-pointer entry points return without synchronising: the copy must have read plan.cu before it goes away)
        if (!mask_host) hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(cu)");
    }
    plan.packed = true;
}

void EncoderModel::embed_on(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                            int64_t batch, int seq, PoolMode pool, bool normalize, float mask_value, float* out,
                            hipStream_t stream, const PackPlan& plan)
{
    if (plan.packed) {
        reserve(ws, plan.max_tokens, plan.max_sentences);
        for (const PackPlan::Chunk& c : plan.chunks) {
            const PackView pv{ws.cu + c.cu_off, ws.tok_src, c.tokens, c.max_len, c.sum_len_sq};
            hip_check(launch_pack_index(mask + c.b0 * seq, pv.cu, c.nb, seq, ws.tok_src, stream), "pack_index");
            forward_chunk(ws, ids + c.b0 * seq, nullptr, type_ids ? type_ids + c.b0 * seq : nullptr, c.nb, seq, mask_value,
                          ws.hidden, stream, &pv);
            hipEvent_t pe = prof_start(KK_POOL, stream, 0.0, 4.0 * ((double)c.tokens * cfg_.hidden + (double)c.nb * cfg_.hidden));
            hip_check(launch_pool(ws.hidden, nullptr, c.nb, seq, cfg_.hidden, pool, normalize ? 1 : 0,
                                  out + c.b0 * (int64_t)cfg_.hidden, stream, pv.cu),
                      "pool");
            prof_stop(pe, stream);
        }
        return;
    }
    const int64_t per = sentences_per_chunk(seq);
    reserve(ws, std::min(per, batch) * seq, std::min(per, batch));
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int64_t nb = std::min(per, batch - b0);
        const uint32_t* m = mask ? mask + b0 * seq : nullptr;
        forward_chunk(ws, ids + b0 * seq, m, type_ids ? type_ids + b0 * seq : nullptr, nb, seq, mask_value,
                      ws.hidden, stream);
        hipEvent_t pe = prof_start(KK_POOL, stream, 0.0, 4.0 * ((double)nb * seq * cfg_.hidden + (double)nb * cfg_.hidden));
        hip_check(launch_pool(ws.hidden, m, nb, seq, cfg_.hidden, pool, normalize ? 1 : 0,
                              out + b0 * (int64_t)cfg_.hidden, stream),
                  "pool");
        prof_stop(pe, stream);
    }
}

void EncoderModel::logits_on(Workspace& ws, const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                             int64_t batch, int seq, float mask_value, float* out, hipStream_t stream, const PackPlan& plan)
{
    if (cfg_.head_kind == 0) throw std::runtime_error("model has no classification head");
    const int H = cfg_.hidden;
    if (plan.packed) {
        reserve(ws, plan.max_tokens, plan.max_sentences);
        for (const PackPlan::Chunk& c : plan.chunks) {
            const PackView pv{ws.cu + c.cu_off, ws.tok_src, c.tokens, c.max_len, c.sum_len_sq};
            hip_check(launch_pack_index(mask + c.b0 * seq, pv.cu, c.nb, seq, ws.tok_src, stream), "pack_index");
            forward_chunk(ws, ids + c.b0 * seq, nullptr, type_ids ? type_ids + c.b0 * seq : nullptr, c.nb, seq, mask_value,
                          ws.hidden, stream, &pv);
            // CLS rows (the first row of every sentence: token 0 is kept, or the call would not be packed) gathered into
            // ws.ctx, which the layers are done with; then the head as in the padded layout.
            hip_check(launch_pool(ws.hidden, nullptr, c.nb, seq, H, POOL_CLS, 0, ws.ctx, stream, pv.cu), "gather CLS rows");
            const float* feat = ws.ctx;
            if (cfg_.head_kind == 1 || cfg_.head_kind == 2) {
                hip_check(launch_gemm(ws.ctx, H, head_dense_w_, head_dense_b_, nullptr, 0, ws.feat, H, c.nb, H, H,
                                      cfg_.head_kind == 1 ? EPI_BIAS_TANH : EPI_BIAS_RELU, stream),
                          "gemm(head dense)");
                feat = ws.feat;
            }
            hip_check(launch_small_linear(feat, H, head_cls_w_, head_cls_b_, c.nb, H, cfg_.num_labels,
                                          out + c.b0 * cfg_.num_labels, stream),
                      "classifier");
        }
        return;
    }
    const int64_t per = sentences_per_chunk(seq);
    reserve(ws, std::min(per, batch) * seq, std::min(per, batch));
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int64_t nb = std::min(per, batch - b0);
        forward_chunk(ws, ids + b0 * seq, mask ? mask + b0 * seq : nullptr, type_ids ? type_ids + b0 * seq : nullptr,
                      nb, seq, mask_value, ws.hidden, stream);
        // CLS rows are read in place: row stride seq*H (cpu/encoder/classifier.rs:219).
        const float* feat = ws.hidden;
        int64_t ld = (int64_t)seq * H;
        if (cfg_.head_kind == 1 || cfg_.head_kind == 2) {
            hip_check(launch_gemm(ws.hidden, ld, head_dense_w_, head_dense_b_, nullptr, 0, ws.feat, H, nb, H, H,
                                  cfg_.head_kind == 1 ? EPI_BIAS_TANH : EPI_BIAS_RELU, stream),
                      "gemm(head dense)");
            feat = ws.feat;
            ld = H;
        }
        hip_check(launch_small_linear(feat, ld, head_cls_w_, head_cls_b_, nb, H, cfg_.num_labels,
                                      out + b0 * cfg_.num_labels, stream),
                  "classifier");
    }
}

void EncoderModel::embed(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                         int seq, PoolMode pool, bool normalize, float mask_value, float* out,
                         hipStream_t stream)
{
    if (batch <= 0 || seq <= 0) return;
    Lease lease(*this, stream, false);
    // (the activation buffers of a padded chunk first: an oversized call fails here, before any kernel has read its pointers)
    reserve(lease.ws(), std::min(sentences_per_chunk(seq), batch) * seq, std::min(sentences_per_chunk(seq), batch));
    PackPlan plan;
    plan_packing(lease.ws(), mask, nullptr, batch, seq, stream, plan);
    embed_on(lease.ws(), ids, mask, type_ids, batch, seq, pool, normalize, mask_value, out, stream, plan);
}

void EncoderModel::logits(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                          int seq, float mask_value, float* out, hipStream_t stream)
{
    if (batch <= 0 || seq <= 0) return;
    if (cfg_.head_kind == 0) throw std::runtime_error("model has no classification head");
    Lease lease(*this, stream, false);
    reserve(lease.ws(), std::min(sentences_per_chunk(seq), batch) * seq, std::min(sentences_per_chunk(seq), batch));
    PackPlan plan;
    plan_packing(lease.ws(), mask, nullptr, batch, seq, stream, plan);
    logits_on(lease.ws(), ids, mask, type_ids, batch, seq, mask_value, out, stream, plan);
}

// ---- host-pointer entry points ------------------------------------------------------------------------
// One lease covers staging, compute and the copy back, all on the workspace's own stream; only that stream
// is synchronised, so calls from other host threads (other workspaces, other streams) keep running.
template <class F>
void EncoderModel::run_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                            int seq, size_t out_floats, float* out, bool out_written_once, F&& body)
{
    if (batch <= 0 || seq <= 0) return;
    Lease lease(*this, nullptr, true);
    Workspace& ws = lease.ws();
    hipStream_t st = lease.stream();
    const size_t tok_bytes = (size_t)batch * (size_t)seq * sizeof(uint32_t);
    const size_t in_bytes = tok_bytes * (type_ids ? 3 : 2);
    const size_t out_off = (in_bytes + 255) & ~(size_t)255;
    uint8_t* base = static_cast<uint8_t*>(reserve_stage(ws, out_off + out_floats * sizeof(float)));
    uint32_t* ids_d = reinterpret_cast<uint32_t*>(base);
    uint32_t* mask_d = reinterpret_cast<uint32_t*>(base + tok_bytes);
    uint32_t* type_d = type_ids ? reinterpret_cast<uint32_t*>(base + 2 * tok_bytes) : nullptr;
    float* out_d = reinterpret_cast<float*>(base + out_off);
    // Small calls (one sentence is ~45 dependent launches of ~5 us: every copy command is a few percent of the call): ids |
    // mask | types gathered in pinned host memory and sent as one copy; the output leaves through the device mapping of the
    // same pinned buffer -- the last kernel writes it across the bus itself, so there is no copy command behind it.
    constexpr size_t kPinnedStageBytes = 256 * 1024;
    const size_t out_bytes = out_floats * sizeof(float);
    if (out_off + out_bytes <= kPinnedStageBytes) {
        if (!ws.pin) {
            void* h = nullptr;
            void* d = nullptr;
            if (hipHostMalloc(&h, kPinnedStageBytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                ws.pin = static_cast<uint8_t*>(h);
                ws.pin_dev = static_cast<uint8_t*>(d);
            } else {
                (void)hipGetLastError();
                if (h) (void)hipHostFree(h);
            }
        }
        if (ws.pin) {
            std::memcpy(ws.pin, ids, tok_bytes);
            std::memcpy(ws.pin + tok_bytes, mask, tok_bytes);
            if (type_ids) std::memcpy(ws.pin + 2 * tok_bytes, type_ids, tok_bytes);
            hip_check(hipMemcpyAsync(base, ws.pin, in_bytes, hipMemcpyHostToDevice, st), "H2D ids | mask | types");
            if (out_written_once) {  // (pooled embeddings, logits: written by one kernel, never read back)
                body(ws, ids_d, mask_d, type_d, reinterpret_cast<float*>(ws.pin_dev + out_off), st);
                hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
                std::memcpy(out, ws.pin + out_off, out_bytes);
            } else {             // (hidden states: the output buffer is the residual stream of every layer)
                body(ws, ids_d, mask_d, type_d, out_d, st);
                hip_check(hipMemcpyAsync(out, out_d, out_bytes, hipMemcpyDeviceToHost, st), "D2H output");
                hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
            }
            return;
        }
    }
    hip_check(hipMemcpyAsync(ids_d, ids, tok_bytes, hipMemcpyHostToDevice, st), "H2D ids");
    hip_check(hipMemcpyAsync(mask_d, mask, tok_bytes, hipMemcpyHostToDevice, st), "H2D mask");
    if (type_ids) hip_check(hipMemcpyAsync(type_d, type_ids, tok_bytes, hipMemcpyHostToDevice, st), "H2D type ids");
    body(ws, ids_d, mask_d, type_d, out_d, st);
    hip_check(hipMemcpyAsync(out, out_d, out_floats * sizeof(float), hipMemcpyDeviceToHost, st), "D2H output");
    hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
}

void EncoderModel::hidden_states_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids,
                                      int64_t batch, int seq, float mask_value, float* out)
{
    run_host(ids, mask, type_ids, batch, seq, (size_t)batch * seq * cfg_.hidden, out, false,
             [&](Workspace& ws, uint32_t* i, uint32_t* k, uint32_t* t, float* o, hipStream_t st) {
                 const int64_t per = sentences_per_chunk(seq);
                 reserve(ws, std::min(per, batch) * seq, std::min(per, batch));
                 for (int64_t b0 = 0; b0 < batch; b0 += per) {
                     const int64_t nb = std::min(per, batch - b0);
                     forward_chunk(ws, i + b0 * seq, k + b0 * seq, t ? t + b0 * seq : nullptr, nb, seq, mask_value,
                                   o + b0 * seq * (int64_t)cfg_.hidden, st);
                 }
             });
}

// ---- lanes -----------------------------------------------------------------------------------------------
// A call of a few thousand tokens (the reference's default batch of 32) is ~43 kernels of 15-50 us, each of which pays a
// launch gap, a prologue and an output burst that nothing overlaps: the chip idles a third of the time.
// Sentences are independent, so such a call runs as two or three parts (cut where the kept tokens halve / third) on as many
// workspaces and streams, the other parts enqueued by persistent helper threads: the parts drift apart by a fraction of a kernel
// and each one's idle phases fall under the others' matrix work.  Measured in the library (tools/mid_probe.py, sentences x 128
// tokens; whole / two / three parts, ms): 24: 0.93 / 0.83 / -; 32: 1.02 / 0.89 / -; 36: 1.16 / 1.16 / 1.10; 48: 1.57 / 1.41 / 1.36;
// 56: 1.62 / 1.58 / 1.56; 64: 1.85 / 1.77 / 1.77; 72: 2.35 / - / 1.99; 96: 2.68 / - / 2.47; 100: 2.73 / - / 2.66; 112: 2.95 / - / 2.84;
// 128: 3.11 / - / 3.30 (left whole, below); 160: 4.54 / - / 3.93; 192: 4.88 / - / 4.72; a 6 x 768 model at 72 .. 192 sentences
// - 3 .. - 6 % (160: + 3 %).  Below ~2 300 tokens one launch sequence is faster; above 24 576 three parts no longer fit the mid-size
// route.
// Up to 8 192 tokens the whole call would take the mid-size projection route as its parts do, where a row's result does not
// depend on the rows beside it: the split is invisible in the results, and a call that finds the helpers busy simply runs unsplit.
// Above that the whole call would run on the large-batch tiles (another summation order): there the split is part of what the call
// computes -- it is made whenever lanes are on, and a part whose helper is busy runs on the caller's thread after the caller's own.
namespace {
constexpr int64_t kTwoLaneMinTokens = 2304, kThreeLaneMinTokens = 4608, kLaneMaxTokens = 24576;
}

EncoderModel::Lane::~Lane()
{
    {
        std::lock_guard<std::mutex> lock(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
}

void EncoderModel::Lane::loop()
{
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
        cv_.wait(lock, [&] { return has_task_ || stop_; });
        if (stop_) return;
        std::function<void()> t = std::move(task_);
        lock.unlock();
        std::exception_ptr err;
        try {
            t();
        } catch (...) {
            err = std::current_exception();
        }
        lock.lock();
        error_ = err;
        has_task_ = false;
        cv_.notify_all();
    }
}

bool EncoderModel::Lane::try_begin(std::function<void()> task)
{
    std::lock_guard<std::mutex> lock(mu_);
    if (busy_ || stop_) return false;
    if (!thread_.joinable()) thread_ = std::thread([this] { loop(); });
    busy_ = true;
    has_task_ = true;
    error_ = nullptr;
    task_ = std::move(task);
    cv_.notify_all();
    return true;
}

void EncoderModel::Lane::wait()
{
    std::unique_lock<std::mutex> lock(mu_);
    cv_.wait(lock, [&] { return !has_task_; });
    const std::exception_ptr err = error_;
    error_ = nullptr;
    busy_ = false;
    lock.unlock();
    if (err) std::rethrow_exception(err);
}

// part(b0, nb): runs sentences [b0, b0 + nb) of the call.  Returns false when the call does not split (the caller runs it whole).
template <class F>
bool EncoderModel::run_two_lanes(const uint32_t* mask, int64_t batch, int seq, F&& part)
{
    if (!two_lanes_ || batch < 4 || batch * seq < kTwoLaneMinTokens) return false;
    // kept tokens (what the packed layout computes on) and where they halve / third
    // (plan_packing's own test: ONE row that cannot be packed -- mask values above 1, or a row whose first token is masked --
    // puts the whole call on the padded layout.  A part without that row would still pack, so the parts would not compute what
    // the unsplit call computes and the result would depend on whether a helper was free: such a call does not split.)
    std::vector<int64_t> upto((size_t)batch + 1, 0);
    bool packable = packing_ >= 1 && seq > 1;
    for (int64_t b = 0; b < batch; ++b) {
        int64_t n = 0;
        uint32_t over = 0;
        const uint32_t* row = mask + b * seq;
        for (int s = 0; s < seq; ++s) {
            n += row[s] != 0u;
            over |= row[s];
        }
        if (over > 1u || row[0] == 0u) packable = false;
        upto[(size_t)b + 1] = upto[(size_t)b] + n;
    }
    if (packing_ >= 1 && !packable) return false;
    if (!packable)
        for (int64_t b = 0; b < batch; ++b) upto[(size_t)b + 1] = (b + 1) * (int64_t)seq;  // padded layout: every row computes seq tokens
    const int64_t total = upto[(size_t)batch];
    static const int64_t lane_max_tokens = [] {  // (measurements: KJARNI_HIP_LANES_MAX_TOKENS)
        const char* e = std::getenv("KJARNI_HIP_LANES_MAX_TOKENS");
        return e && std::atoll(e) > 0 ? (int64_t)std::atoll(e) : kLaneMaxTokens;
    }();
    if (total < kTwoLaneMinTokens || total > lane_max_tokens) return false;
    // (where the whole call is already at its best: with the fused LayerNorm tiles -- 64 whole rows each -- 14 337 .. 16 384 rows
    // are 225 .. 256 tiles, one per CU: 128 x 128 tokens 3.11 ms whole against 3.30 in three parts)
    if (fuse_layernorm() && total > 14336 && total <= 16384) return false;
    // (8 192; fewer in the f32-on-bf16 mode, whose large tiles start earlier)
    const int64_t kSameRouteMaxTokens = gemm_mid_route_max_rows();
    // every part on the kernels a mid-size call takes (the small-call attention kernel sums in another order)
    auto cuts_ok = [&](const int64_t* cut, int parts) {
        for (int i = 0; i < parts; ++i) {
            const int64_t nb = cut[i + 1] - cut[i];
            if (nb < 1 || (seq <= 128 && nb * cfg_.heads <= attention_small_call_items())) return false;
            const int64_t tokens = upto[(size_t)cut[i + 1]] - upto[(size_t)cut[i]];
            if (tokens > kSameRouteMaxTokens || tokens < 1024) return false;  // (a mid-size call itself: neither the few-rows nor the large-batch kernels)
        }
        return true;
    };
    auto cut_into = [&](int parts, int64_t* cut) {
        cut[0] = 0;
        for (int i = 1; i < parts; ++i) {
            int64_t c = cut[i - 1] + 1;
            while (c < batch - (parts - i) && upto[(size_t)c] * parts < total * i) ++c;
            cut[i] = c;
        }
        cut[parts] = batch;
    };
    int64_t cut[4];
    int parts = 0;
    static const int max_parts = [] {  // (measurements: KJARNI_HIP_LANES_MAX=2 keeps calls of up to 8 192 tokens in two parts)
        const char* e = std::getenv("KJARNI_HIP_LANES_MAX");
        return e && std::atoi(e) == 2 ? 2 : 3;
    }();
    if (total >= kThreeLaneMinTokens && batch >= 6 && (max_parts >= 3 || total > kSameRouteMaxTokens)) {
        cut_into(3, cut);
        if (cuts_ok(cut, 3)) parts = 3;
    }
    if (parts == 0 && total <= kSameRouteMaxTokens) {
        cut_into(2, cut);
        if (cuts_ok(cut, 2)) parts = 2;
    }
    if (parts == 0) return false;
    const bool must = total > kSameRouteMaxTokens;  // (the parts' routes are not the whole call's: the split defines the result)
    bool on_helper[2] = {false, false};
    for (int i = 1; i < parts; ++i) {
        const int64_t b0 = cut[i], nb = cut[i + 1] - cut[i];
        on_helper[i - 1] = lane_[i - 1].try_begin([&part, b0, nb] { part(b0, nb); });
        if (!on_helper[i - 1] && i == 1 && !must) return false;  // (nothing started yet: run unsplit, the same bits)
    }
    std::exception_ptr err;
    auto run_here = [&](int i) {
        try {
            part(cut[i], cut[i + 1] - cut[i]);
        } catch (...) {
            if (!err) err = std::current_exception();
        }
    };
    run_here(0);
    for (int i = 1; i < parts; ++i)
        if (!on_helper[i - 1]) run_here(i);  // (its helper was busy with another call)
    for (int i = 1; i < parts; ++i) {
        if (!on_helper[i - 1]) continue;
        try {
            lane_[i - 1].wait();
        } catch (...) {
            if (!err) err = std::current_exception();
        }
    }
    if (err) std::rethrow_exception(err);
    return true;
}

void EncoderModel::embed_host_now(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                                  int seq, PoolMode pool, bool normalize, float mask_value, float* out)
{
    const int H = cfg_.hidden;
    auto one = [&](int64_t b0, int64_t nb) {
        const uint32_t *i0 = ids + b0 * seq, *m0 = mask + b0 * seq, *t0 = type_ids ? type_ids + b0 * seq : nullptr;
        run_host(i0, m0, t0, nb, seq, (size_t)nb * H, out + b0 * H, true,
                 [&, m0, nb](Workspace& ws, uint32_t* i, uint32_t* k, uint32_t* t, float* o, hipStream_t st) {
                     PackPlan plan;  // (lives until run_host has synchronised the stream)
                     plan_packing(ws, k, m0, nb, seq, st, plan);
                     embed_on(ws, i, k, t, nb, seq, pool, normalize, mask_value, o, st, plan);
                 });
    };
    if (run_two_lanes(mask, batch, seq, one)) return;
    run_host(ids, mask, type_ids, batch, seq, (size_t)batch * cfg_.hidden, out, true,
             [&](Workspace& ws, uint32_t* i, uint32_t* k, uint32_t* t, float* o, hipStream_t st) {
                 PackPlan plan;  // (lives until run_host has synchronised the stream)
                 plan_packing(ws, k, mask, batch, seq, st, plan);
                 embed_on(ws, i, k, t, batch, seq, pool, normalize, mask_value, o, st, plan);
             });
}

void EncoderModel::logits_host_now(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                                   int seq, float mask_value, float* out)
{
    if (cfg_.head_kind == 0) throw std::runtime_error("model has no classification head");
    const int NL = cfg_.num_labels;
    auto one = [&](int64_t b0, int64_t nb) {
        const uint32_t *i0 = ids + b0 * seq, *m0 = mask + b0 * seq, *t0 = type_ids ? type_ids + b0 * seq : nullptr;
        run_host(i0, m0, t0, nb, seq, (size_t)nb * NL, out + b0 * NL, true,
                 [&, m0, nb](Workspace& ws, uint32_t* i, uint32_t* k, uint32_t* t, float* o, hipStream_t st) {
                     PackPlan plan;
                     plan_packing(ws, k, m0, nb, seq, st, plan);
                     logits_on(ws, i, k, t, nb, seq, mask_value, o, st, plan);
                 });
    };
    if (run_two_lanes(mask, batch, seq, one)) return;
    run_host(ids, mask, type_ids, batch, seq, (size_t)batch * cfg_.num_labels, out, true,
             [&](Workspace& ws, uint32_t* i, uint32_t* k, uint32_t* t, float* o, hipStream_t st) {
                 PackPlan plan;
                 plan_packing(ws, k, mask, batch, seq, st, plan);
                 logits_on(ws, i, k, t, batch, seq, mask_value, o, st, plan);
             });
}

// ---- call combining -------------------------------------------------------------------------------------
// A small call (a sentence to embed or classify, a handful) is ~45 dependent launches of ~5 us on a fraction of the chip, and
// the reference lets any number of threads call one handle at once (kjarni-ffi/src/lib.rs:25-32).  Threads that do so share
// the launches: a call queues itself; if no leader is at work it becomes one and runs -- alone when nothing else is queued,
// which is the uncontended case and costs one uncontended lock more than before; calls that arrive while a leader's batch is
// on the device wait, and the next leader (the first of them to wake) takes every compatible queued call along as ONE batch of
// packed rows (sentences of different lengths padded to the longest: the packed layout drops the padding again).  Up to
// two batches are in flight at once (one stages and tokenises while the other computes).
namespace {
constexpr int64_t kCombineMaxCallRows = 8, kCombineMaxCallTokens = 1024;  // what counts as a small call
constexpr int64_t kCombineMaxBatchRows = 256, kCombineMaxBatchTokens = 16384;
constexpr int kCombineLeadersDefault = 2;
// (measurements: KJARNI_HIP_COMBINE_LEADERS = batches in flight at once)
int combine_leaders()
{
    static const int n = [] {
        const char* e = std::getenv("KJARNI_HIP_COMBINE_LEADERS");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1 && v <= 4 ? v : kCombineLeadersDefault;
    }();
    return n;
}
}  // namespace

void EncoderModel::run_combined(const std::vector<CombineReq*>& reqs)
{
    const CombineReq& f = *reqs.front();
    if (reqs.size() == 1) {
        if (f.kind == 0) embed_host_now(f.ids, f.mask, f.type_ids, f.batch, f.seq, (PoolMode)f.pool, f.normalize, f.mask_value, f.out);
        else logits_host_now(f.ids, f.mask, f.type_ids, f.batch, f.seq, f.mask_value, f.out);
        return;
    }
    int64_t rows = 0;
    int seq = 0;
    for (const CombineReq* r : reqs) {
        rows += r->batch;
        seq = std::max(seq, r->seq);
    }
    const size_t n = (size_t)rows * (size_t)seq;
    std::vector<uint32_t> ids(n, 0u), mask(n, 0u), types(f.type_ids ? n : 0, 0u);
    int64_t at = 0;
    for (const CombineReq* r : reqs)
        for (int64_t b = 0; b < r->batch; ++b, ++at) {
            std::memcpy(&ids[(size_t)at * seq], r->ids + b * r->seq, (size_t)r->seq * 4);
            std::memcpy(&mask[(size_t)at * seq], r->mask + b * r->seq, (size_t)r->seq * 4);
            if (f.type_ids) std::memcpy(&types[(size_t)at * seq], r->type_ids + b * r->seq, (size_t)r->seq * 4);
        }
    std::vector<float> out((size_t)rows * f.out_per_row);
    if (f.kind == 0)
        embed_host_now(ids.data(), mask.data(), f.type_ids ? types.data() : nullptr, rows, seq, (PoolMode)f.pool, f.normalize, f.mask_value,
                       out.data());
    else
        logits_host_now(ids.data(), mask.data(), f.type_ids ? types.data() : nullptr, rows, seq, f.mask_value, out.data());
    at = 0;
    for (const CombineReq* r : reqs) {
        std::memcpy(r->out, &out[(size_t)at * f.out_per_row], (size_t)r->batch * f.out_per_row * sizeof(float));
        at += r->batch;
    }
}

void EncoderModel::submit_small(CombineReq& req)
{
    std::unique_lock<std::mutex> lock(combine_mu_);
    combine_queue_.push_back(&req);
    while (!req.done) {
        // A call whose own request is already in some leader's batch only waits for it: leading another batch now would put
        // other callers' work in front of its own return.
        if (req.taken || combine_leaders_ >= combine_leaders() || combine_queue_.empty()) {
            combine_cv_.wait(lock);
            continue;
        }
        // lead: the oldest queued call and every compatible one behind it, within the batch bounds (rows, and rows x the
        // LONGEST sequence taken so far: what the combined batch is padded to before the packed layout drops the padding)
        ++combine_leaders_;
        std::vector<CombineReq*> take;
        int64_t rows = 0;
        int max_seq = 0;
        for (auto it = combine_queue_.begin(); it != combine_queue_.end();) {
            CombineReq* r = *it;
            const int seq = std::max(max_seq, r->seq);
            if (take.empty() || (r->compatible(*take.front()) && rows + r->batch <= kCombineMaxBatchRows &&
                                 (rows + r->batch) * seq <= kCombineMaxBatchTokens)) {
                take.push_back(r);
                r->taken = true;
                rows += r->batch;
                max_seq = seq;
                it = combine_queue_.erase(it);
            } else {
                ++it;
            }
        }
        lock.unlock();
        std::vector<std::exception_ptr> errs(take.size());
        try {
            run_combined(take);
        } catch (...) {
            // The shared forward failed: every rider runs again ALONE, so that only the caller whose input (or whose own
            // forward) fails sees an error -- as on the reference, where calls do not know of each other.
            for (size_t i = 0; i < take.size(); ++i) {
                try {
                    run_combined(std::vector<CombineReq*>{take[i]});
                } catch (...) {
                    errs[i] = std::current_exception();
                }
            }
        }
        lock.lock();
        for (size_t i = 0; i < take.size(); ++i) {
            take[i]->error = errs[i];
            take[i]->done = true;
        }
        --combine_leaders_;
        combine_cv_.notify_all();
    }
    lock.unlock();
    if (req.error) std::rethrow_exception(req.error);
}

void EncoderModel::embed_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                              int seq, PoolMode pool, bool normalize, float mask_value, float* out)
{
    if (batch <= 0 || seq <= 0) return;
    if (combining_ && batch <= kCombineMaxCallRows && batch * seq <= kCombineMaxCallTokens) {
        CombineReq r{0, ids, mask, type_ids, batch, seq, (int)pool, normalize, mask_value, out, (size_t)cfg_.hidden, false, false, nullptr};
        submit_small(r);
        return;
    }
    embed_host_now(ids, mask, type_ids, batch, seq, pool, normalize, mask_value, out);
}

void EncoderModel::logits_host(const uint32_t* ids, const uint32_t* mask, const uint32_t* type_ids, int64_t batch,
                               int seq, float mask_value, float* out)
{
    if (cfg_.head_kind == 0) throw std::runtime_error("model has no classification head");
    if (batch <= 0 || seq <= 0) return;
    if (combining_ && batch <= kCombineMaxCallRows && batch * seq <= kCombineMaxCallTokens) {
        CombineReq r{1, ids, mask, type_ids, batch, seq, 0, false, mask_value, out, (size_t)cfg_.num_labels, false, false, nullptr};
        submit_small(r);
        return;
    }
    logits_host_now(ids, mask, type_ids, batch, seq, mask_value, out);
}

}  // namespace kjarni
