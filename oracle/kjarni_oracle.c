/*
 * kjarni_oracle.c -- CPU restatement of the Kjarni encoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (kjarni_amd/, the C-ABI
 * library) may include, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * reported CPU baseline -- never as the thing measured as "value" or shipped.
 *
 * Parity status: PINNED against the reference's model-free golden tests
 * (tests/test_oracle_goldens.py): encoder-layer post/pre-norm goldens
 * (crates/kjarni-transformers/src/cpu/encoder/encoder_layer.rs:349-448,
 * 694-780), FFN golden (cpu/feedforward/standard_new.rs:155-191), pooling
 * goldens (cpu/encoder/traits.rs:796-895, pooling/mod.rs:70-153), LayerNorm
 * (cpu/normalization/layer_norm.rs:228-307), activation scalars
 * (activations.rs:312-329), cosine / VectorStore (kjarni-search/src/vector.rs
 * :169-433).  The reference itself (Rust, needs cargo + crates.io) cannot be
 * built or imported in this environment, so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it restates.  All paths below
 * are relative to /root/reference/crates/.
 *
 * Third-party arithmetic restated here (not vendored in the reference):
 *   - faer 0.20 matmul (attention QK^T / PV): restated as a plain k-ordered
 *     f32 dot product; pinned by the encoder-layer goldens at 1e-4.
 *   - libm 0.2 erff/tanhf/expf: C libm's erff/tanhf/expf (both are
 *     correctly-rounded-to-~1ulp ports of the same FreeBSD msun sources).
 *   - ndarray 0.16 mean_axis/var_axis (alloc-path LayerNorm): plain
 *     sequential f32 sums (population variance, ddof = 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Activations: kjarni-transformers/src/activations.rs                        */
/* ------------------------------------------------------------------------- */

/* activations.rs:15-17 */
static const float KO_SQRT_2_INV = 0.7071067811865475f;
static const float KO_SQRT_2_OVER_PI = 0.7978845608f;
static const float KO_GELU_COEFF = 0.044715f;

enum { KO_ACT_GELU = 0, KO_ACT_GELU_NEW = 1, KO_ACT_RELU = 2, KO_ACT_TANH = 3, KO_ACT_NONE = 4 };

/* activations.rs:56-59 gelu_scalar */
KO_API float ko_gelu(float x) { return 0.5f * x * (1.0f + erff(x * KO_SQRT_2_INV)); }

/* activations.rs:62-66 gelu_new_scalar */
KO_API float ko_gelu_new(float x)
{
    float x3 = x * x * x;
    float inner = KO_SQRT_2_OVER_PI * (x + KO_GELU_COEFF * x3);
    return 0.5f * x * (1.0f + tanhf(inner));
}

/* activations.rs:69-71 */
KO_API float ko_relu(float x) { return x > 0.0f ? x : 0.0f; }

static inline float ko_act(float x, int act)
{
    switch (act) {
    case KO_ACT_GELU: return ko_gelu(x);
    case KO_ACT_GELU_NEW: return ko_gelu_new(x);
    case KO_ACT_RELU: return ko_relu(x);
    case KO_ACT_TANH: return tanhf(x);
    default: return x;
    }
}

/* The same scalars over an array (activations.rs:94-107 apply_activation_slice maps them over a slice); for tests that check
 * millions of activations at once. */
KO_API void ko_activation_array(float *x, int64_t n, int act)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = ko_act(x[i], act);
}

/* activations.rs:223-242 softmax_inplace: max, exp(x-max), sum, scale by 1/sum if sum > 0 */
KO_API void ko_softmax_row(float *row, int n)
{
    if (n <= 0) return;
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = (row[i] > mx) ? row[i] : mx; /* f32::max: NaN-ignoring; inputs here are never NaN before masking */
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        row[i] = expf(row[i] - mx);
        sum += row[i];
    }
    if (sum > 0.0f) {
        float scale = 1.0f / sum;
        for (int i = 0; i < n; ++i) row[i] *= scale;
    }
}

/* ------------------------------------------------------------------------- */
/* LayerNorm: cpu/normalization/layer_norm.rs                                 */
/* ------------------------------------------------------------------------- */

/* layer_norm.rs:96-131 forward_2d_noalloc_scalar (also the math of the ndarray
 * alloc path :203-215: population variance, eps inside the sqrt). */
KO_API void ko_layer_norm(const float *x, const float *gamma, const float *beta, float eps,
                          int64_t rows, int hidden, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < rows; ++t) {
        const float *in = x + t * hidden;
        float *o = out + t * hidden;
        float sum = 0.0f;
        for (int i = 0; i < hidden; ++i) sum += in[i];
        float mean = sum / (float)hidden;
        float var = 0.0f;
        for (int i = 0; i < hidden; ++i) {
            float d = in[i] - mean;
            var += d * d;
        }
        float inv_std = 1.0f / sqrtf(var / (float)hidden + eps);
        for (int i = 0; i < hidden; ++i) o[i] = (in[i] - mean) * inv_std * gamma[i] + beta[i];
    }
}

/* ------------------------------------------------------------------------- */
/* Linear: linear_layer/linear_layer.rs:160-282, cpu/ops/matmul.rs:571-686    */
/* y[m,n] = sum_k x[m,k] * w[n,k] + b[n]; W is [out,in] row-major (HF layout)  */
/* ------------------------------------------------------------------------- */

/* Plain restatement: k-ordered scalar dot product, bias added once at the end
 * (cpu/kernels/x86/f32.rs:96-106). */
KO_API void ko_linear(const float *x, const float *w, const float *bias, int64_t m, int k, int n,
                      float *y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
        const float *xr = x + i * k;
        for (int j = 0; j < n; ++j) {
            const float *wr = w + (int64_t)j * k;
            float s = 0.0f;
            for (int p = 0; p < k; ++p) s += xr[p] * wr[p];
            y[i * n + j] = s + (bias ? bias[j] : 0.0f);
        }
    }
}

__attribute__((target("avx2,fma"))) static inline float ko_hsum256(__m256 v)
{
    __m128 hi = _mm256_extractf128_ps(v, 1);
    __m128 lo = _mm256_castps256_ps128(v);
    __m128 s = _mm_add_ps(hi, lo);
    __m128 h64 = _mm_movehl_ps(s, s);
    __m128 s64 = _mm_add_ps(s, h64);
    __m128 h32 = _mm_shuffle_ps(s64, s64, 1);
    return _mm_cvtss_f32(_mm_add_ss(s64, h32));
}

/* cpu/kernels/x86/f32.rs:8-127 matmul_block_4x3_f32: 4 tokens x 3 outputs,
 * 8-wide FMA over K, horizontal sum, scalar tail, bias once. */
__attribute__((target("avx2,fma"))) static void ko_block_4x3(float *out, int64_t out_stride,
                                                             const float *a, const float *b, int k,
                                                             const float *bias)
{
    __m256 c[4][3];
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 3; ++j) c[r][j] = _mm256_setzero_ps();
    int p = 0;
    for (; p + 8 <= k; p += 8) {
        __m256 a0 = _mm256_loadu_ps(a + p), a1 = _mm256_loadu_ps(a + k + p);
        __m256 a2 = _mm256_loadu_ps(a + 2 * k + p), a3 = _mm256_loadu_ps(a + 3 * k + p);
        for (int j = 0; j < 3; ++j) {
            __m256 wv = _mm256_loadu_ps(b + (int64_t)j * k + p);
            c[0][j] = _mm256_fmadd_ps(a0, wv, c[0][j]);
            c[1][j] = _mm256_fmadd_ps(a1, wv, c[1][j]);
            c[2][j] = _mm256_fmadd_ps(a2, wv, c[2][j]);
            c[3][j] = _mm256_fmadd_ps(a3, wv, c[3][j]);
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 3; ++j) {
            float s = ko_hsum256(c[r][j]);
            for (int q = p; q < k; ++q) s += a[r * k + q] * b[(int64_t)j * k + q];
            if (bias) s += bias[j];
            out[r * out_stride + j] = s;
        }
}

/* cpu/ops/matmul.rs:571-686 matmul_2d_cpu_f32_batched: rayon over 64-token
 * row blocks; 4x3 register tile; scalar paths for n%3 and tokens%4 tails.
 * This is the variant the timed CPU baseline uses ("port" of the reference's
 * blocking); ko_linear above is the plain form.  Both are checked equal to
 * 1e-5 in tests/test_oracle_goldens.py. */
__attribute__((target("avx2,fma"))) KO_API void ko_linear_blocked(const float *x, const float *w,
                                                                  const float *bias, int64_t m,
                                                                  int k, int n, float *y)
{
    const int64_t BLOCK = 64;
    int64_t nblocks = (m + BLOCK - 1) / BLOCK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t blk = 0; blk < nblocks; ++blk) {
        int64_t t0 = blk * BLOCK;
        int64_t ntok = (m - t0 < BLOCK) ? (m - t0) : BLOCK;
        const float *in_block = x + t0 * k;
        float *out_block = y + t0 * n;
        int64_t t = 0;
        for (; t + 4 <= ntok; t += 4) {
            const float *in = in_block + t * k;
            int j = 0;
            for (; j + 3 <= n; j += 3)
                ko_block_4x3(out_block + t * n + j, n, in, w + (int64_t)j * k, k,
                             bias ? bias + j : NULL);
            for (; j < n; ++j) {
                const float *wr = w + (int64_t)j * k;
                float bv = bias ? bias[j] : 0.0f;
                for (int r = 0; r < 4; ++r) {
                    float s = 0.0f;
                    for (int p = 0; p < k; ++p) s += in[r * k + p] * wr[p];
                    out_block[(t + r) * n + j] = s + bv;
                }
            }
        }
        for (; t < ntok; ++t) {
            const float *in = in_block + t * k;
            for (int j = 0; j < n; ++j) {
                const float *wr = w + (int64_t)j * k;
                float s = 0.0f;
                for (int p = 0; p < k; ++p) s += in[p] * wr[p];
                out_block[t * n + j] = s + (bias ? bias[j] : 0.0f);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Embeddings: cpu/embeddings/mod.rs:181-295                                   */
/* ------------------------------------------------------------------------- */

/* Embeddings::forward: h = W_word[id] (ids >= vocab leave zeros, :232-236),
 * optional * sqrt(hidden) (:193-196), + P[offset + s] for s < max_pos - offset
 * (:198-212), + T[type] (:214-223; when type_ids == NULL row 0 is added to
 * every token).  A type id >= type_vocab panics in the reference (:308-313);
 * here it returns -1. */
KO_API int ko_embed(const uint32_t *ids, const uint32_t *type_ids, const float *word,
                    const float *pos, const float *type, int64_t batch, int seq, int hidden,
                    int vocab, int max_pos, int type_vocab, int pos_offset, int scale, float *out)
{
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < batch; ++b) {
        for (int s = 0; s < seq; ++s) {
            float *o = out + (b * seq + s) * hidden;
            uint32_t id = ids[b * seq + s];
            if ((int64_t)id < vocab)
                memcpy(o, word + (int64_t)id * hidden, sizeof(float) * hidden);
            else
                memset(o, 0, sizeof(float) * hidden);
            if (scale) {
                float f = sqrtf((float)hidden);
                for (int i = 0; i < hidden; ++i) o[i] *= f;
            }
            if (pos && pos_offset + s < max_pos) {
                const float *p = pos + (int64_t)(pos_offset + s) * hidden;
                for (int i = 0; i < hidden; ++i) o[i] += p[i];
            }
            if (type && type_vocab > 0) {
                uint32_t ty = type_ids ? type_ids[b * seq + s] : 0;
                if ((int)ty >= type_vocab) {
                    bad = 1;
                    continue;
                }
                const float *tr = type + (int64_t)ty * hidden;
                for (int i = 0; i < hidden; ++i) o[i] += tr[i];
            }
        }
    }
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------- */
/* Self-attention: cpu/encoder/encoder_self_attention.rs:61-307,               */
/* utils/masks.rs:4-36, utils/linear_algebra.rs:708-740                        */
/* ------------------------------------------------------------------------- */

/* q,k,v: [batch*seq, hidden] (already projected).  ctx out: [batch*seq, hidden]
 * (heads merged, encoder_self_attention.rs:126-131 / :384-425).
 * mask: f32 [batch, seq]; a key with mask == 0 has its score OVERWRITTEN by
 * mask_value (-1e9 alloc path masks.rs:4-36; -inf no-alloc path
 * encoder_self_attention.rs:311-325).  position_bias: optional [heads,seq,seq]
 * added after scaling (:112-114 / :252-258).  scale_qk: :107-109.
 * Parallel over batch only, heads serial (linear_algebra.rs:715-718). */
KO_API void ko_attention(const float *q, const float *k, const float *v, const float *mask,
                         const float *position_bias, int64_t batch, int seq, int heads,
                         int head_dim, int scale_qk, float mask_value, float *ctx)
{
    int hidden = heads * head_dim;
    float scale = 1.0f / sqrtf((float)head_dim);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < batch; ++b) {
        float *scores = (float *)malloc(sizeof(float) * seq);
        for (int h = 0; h < heads; ++h) {
            for (int i = 0; i < seq; ++i) {
                const float *qi = q + (b * seq + i) * hidden + h * head_dim;
                for (int j = 0; j < seq; ++j) {
                    const float *kj = k + (b * seq + j) * hidden + h * head_dim;
                    float s = 0.0f;
                    for (int d = 0; d < head_dim; ++d) s += qi[d] * kj[d];
                    if (scale_qk) s *= scale;
                    if (position_bias) s += position_bias[((int64_t)h * seq + i) * seq + j];
                    if (mask && mask[b * seq + j] == 0.0f) s = mask_value;
                    scores[j] = s;
                }
                ko_softmax_row(scores, seq);
                float *o = ctx + (b * seq + i) * hidden + h * head_dim;
                for (int d = 0; d < head_dim; ++d) o[d] = 0.0f;
                for (int j = 0; j < seq; ++j) {
                    const float *vj = v + (b * seq + j) * hidden + h * head_dim;
                    float p = scores[j];
                    for (int d = 0; d < head_dim; ++d) o[d] += p * vj[d];
                }
            }
        }
        free(scores);
    }
}

/* ------------------------------------------------------------------------- */
/* Model description shared by the layer / encoder drivers                     */
/* ------------------------------------------------------------------------- */

typedef struct {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
    const float *ln1_g, *ln1_b;
    const float *w1, *b1, *w2, *b2;
    const float *ln2_g, *ln2_b;
    const float *wg; /* SwiGLU gate weight [inter, hidden] (Nomic: mlp.fc11); NULL: FC2(act(FC1(x))) */
} ko_layer;

typedef struct {
    int32_t hidden, layers, heads, inter, vocab, max_pos, type_vocab, pos_offset;
    int32_t act, prenorm, scale_embeddings, scale_qk;
    float eps;
    int32_t blocked_gemm; /* 1: use the reference's 64-row / 4x3 AVX2 blocking (timed baseline) */
    const float *word, *pos, *type, *emb_ln_g, *emb_ln_b;
    const ko_layer *L;
    /* RoPE caches [rope_len, head_dim] (cpu/rope/mod.rs:96-116), NULL when the model has position embeddings */
    const float *rope_cos, *rope_sin;
    int32_t rope_len;
} ko_model;

/* silu_scalar, activations.rs:74-82 */
static inline float ko_silu(float x)
{
    if (x <= -20.0f) return 0.0f;
    if (x >= 20.0f) return x;
    return x / (1.0f + expf(-x));
}

/* RoPE::apply_3d -> rotate_4d_in_place (cpu/rope/mod.rs:118-170, 210-245) on [batch*seq, heads*head_dim]:
 * (x0, x1) = (x[i], x[i + half]) -> (x0*cos - x1*sin, x0*sin + x1*cos) with the caches' row `s` (offset 0). */
static void ko_rope_rows(const ko_model *m, float *x, int64_t batch, int seq, int heads, int head_dim)
{
    const int half = head_dim / 2;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < batch * seq; ++t) {
        const int s = (int)(t % seq);
        const float *c = m->rope_cos + (int64_t)s * head_dim, *sn = m->rope_sin + (int64_t)s * head_dim;
        for (int h = 0; h < heads; ++h) {
            float *r = x + (t * heads + h) * head_dim;
            for (int i = 0; i < half; ++i) {
                const float x0 = r[i], x1 = r[i + half];
                r[i] = x0 * c[i] - x1 * sn[i];
                r[i + half] = x0 * sn[i] + x1 * c[i];
            }
        }
    }
}

static void ko_lin(const ko_model *m, const float *x, const float *w, const float *b, int64_t rows,
                   int k, int n, float *y)
{
    if (m->blocked_gemm)
        ko_linear_blocked(x, w, b, rows, k, n, y);
    else
        ko_linear(x, w, b, rows, k, n, y);
}

/* cpu/encoder/encoder_layer.rs:216-232 forward_postnorm / :197-214
 * forward_prenorm (and their no-alloc twins :113-179 / :63-111, which compute
 * the same values):
 *   post: h1 = LN1(x + Attn(x)); y = LN2(h1 + FFN(h1))
 *   pre : h1 = x + Attn(LN1(x)); y = h1 + FFN(LN2(h1))
 * FFN = cpu/feedforward/standard_new.rs:29-82: FC2(act(FC1(x))); with a gate weight,
 * cpu/feedforward/swiglu.rs:33-57: down(silu(gate(x)) * up(x)) (transformer_encoder.rs:163-172).
 * With RoPE caches Q and K are rotated after the projections (encoder_self_attention.rs:81-85).
 * hidden is updated in place.  tokens = batch*seq. */
KO_API void ko_encoder_layer(const ko_model *m, const ko_layer *L, float *hidden, const float *mask,
                             const float *position_bias, int64_t batch, int seq, float mask_value)
{
    int H = m->hidden, I = m->inter;
    int64_t T = batch * seq;
    float *normed = (float *)malloc(sizeof(float) * T * H);
    float *q = (float *)malloc(sizeof(float) * T * H);
    float *k = (float *)malloc(sizeof(float) * T * H);
    float *v = (float *)malloc(sizeof(float) * T * H);
    float *ctx = (float *)malloc(sizeof(float) * T * H);
    float *attn = (float *)malloc(sizeof(float) * T * H);
    float *mid = (float *)malloc(sizeof(float) * T * I);

    const float *attn_in = hidden;
    if (m->prenorm) {
        ko_layer_norm(hidden, L->ln1_g, L->ln1_b, m->eps, T, H, normed);
        attn_in = normed;
    }
    /* cpu/encoder/qkv_projection.rs:30-138: fused [3H,H] GEMM when H <= 512 is
     * numerically three independent projections. */
    ko_lin(m, attn_in, L->wq, L->bq, T, H, H, q);
    ko_lin(m, attn_in, L->wk, L->bk, T, H, H, k);
    ko_lin(m, attn_in, L->wv, L->bv, T, H, H, v);
    if (m->rope_cos) {
        ko_rope_rows(m, q, batch, seq, m->heads, H / m->heads);
        ko_rope_rows(m, k, batch, seq, m->heads, H / m->heads);
    }
    ko_attention(q, k, v, mask, position_bias, batch, seq, m->heads, H / m->heads, m->scale_qk,
                 mask_value, ctx);
    ko_lin(m, ctx, L->wo, L->bo, T, H, H, attn);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < T * H; ++i) hidden[i] += attn[i];

    const float *ffn_in = hidden;
    if (m->prenorm) {
        ko_layer_norm(hidden, L->ln2_g, L->ln2_b, m->eps, T, H, normed);
        ffn_in = normed;
    } else {
        ko_layer_norm(hidden, L->ln1_g, L->ln1_b, m->eps, T, H, normed);
        memcpy(hidden, normed, sizeof(float) * T * H);
        ffn_in = hidden;
    }
    if (L->wg) {
        float *up = (float *)malloc(sizeof(float) * T * I);
        ko_lin(m, ffn_in, L->wg, NULL, T, H, I, mid);
        ko_lin(m, ffn_in, L->w1, L->b1, T, H, I, up);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < T * I; ++i) mid[i] = ko_silu(mid[i]) * up[i];
        free(up);
    } else {
        ko_lin(m, ffn_in, L->w1, L->b1, T, H, I, mid);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < T * I; ++i) mid[i] = ko_act(mid[i], m->act);
    }
    ko_lin(m, mid, L->w2, L->b2, T, I, H, attn);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < T * H; ++i) hidden[i] += attn[i];
    if (!m->prenorm) {
        ko_layer_norm(hidden, L->ln2_g, L->ln2_b, m->eps, T, H, normed);
        memcpy(hidden, normed, sizeof(float) * T * H);
    }
    free(normed); free(q); free(k); free(v); free(ctx); free(attn); free(mid);
}

/* cpu/encoder/traits.rs:66-139 get_hidden_states_batch_from_ids and :295-313
 * forward_tokens: embed -> embed_norm (transformer_encoder.rs:303-305) ->
 * layers (:335-368); no final norm (:300-302).  mask is u32 [batch,seq]
 * converted with `as f32` (traits.rs:71).  mask_value: the caller picks -1e9
 * (alloc path, used when 1 < tokens < 1000 and always by forward_tokens) or
 * -inf (no-alloc path, tokens <= 1 or >= 1000: cpu/strategy.rs:43-44). */
KO_API int ko_encoder_forward(const ko_model *m, const uint32_t *ids, const uint32_t *mask_u32,
                              const uint32_t *type_ids, int64_t batch, int seq, float mask_value,
                              float *hidden_out)
{
    int H = m->hidden;
    int64_t T = batch * seq;
    if (T == 0) return 0;
    float *maskf = (float *)malloc(sizeof(float) * T);
    for (int64_t i = 0; i < T; ++i) maskf[i] = (float)mask_u32[i];
    float *emb = (float *)malloc(sizeof(float) * T * H);
    int rc = ko_embed(ids, type_ids, m->word, m->pos, m->type, batch, seq, H, m->vocab, m->max_pos,
                      m->type_vocab, m->pos_offset, m->scale_embeddings, emb);
    if (rc == 0) {
        if (m->emb_ln_g)
            ko_layer_norm(emb, m->emb_ln_g, m->emb_ln_b, m->eps, T, H, hidden_out);
        else
            memcpy(hidden_out, emb, sizeof(float) * T * H);
        for (int l = 0; l < m->layers; ++l)
            ko_encoder_layer(m, &m->L[l], hidden_out, maskf, NULL, batch, seq, mask_value);
    }
    free(emb);
    free(maskf);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Pooling: pooling/mod.rs:11-73, cpu/encoder/traits.rs:529-536               */
/* ------------------------------------------------------------------------- */

/* pooling/mod.rs:11-33 mean_pool: sum_s h*m / count (count==0 -> 1); rows whose
 * mask sums to 0 return token 0's hidden row. */
KO_API void ko_mean_pool(const float *hidden, const float *mask, int64_t batch, int seq, int H,
                         float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < batch; ++b) {
        float cnt = 0.0f;
        for (int s = 0; s < seq; ++s) cnt += mask[b * seq + s];
        float *o = out + b * H;
        if (cnt == 0.0f) {
            memcpy(o, hidden + b * seq * H, sizeof(float) * H);
            continue;
        }
        for (int i = 0; i < H; ++i) o[i] = 0.0f;
        for (int s = 0; s < seq; ++s) {
            float mv = mask[b * seq + s];
            const float *h = hidden + (b * seq + s) * H;
            for (int i = 0; i < H; ++i) o[i] += h[i] * mv;
        }
        for (int i = 0; i < H; ++i) o[i] = o[i] / cnt;
    }
}

/* pooling/mod.rs:36-38 cls_pool */
KO_API void ko_cls_pool(const float *hidden, int64_t batch, int seq, int H, float *out)
{
    for (int64_t b = 0; b < batch; ++b) memcpy(out + b * H, hidden + b * seq * H, sizeof(float) * H);
}

/* pooling/mod.rs:41-52 max_pool: masked positions become MASK_VALUE (-1e9),
 * fold starts from MASK_VALUE. */
KO_API void ko_max_pool(const float *hidden, const float *mask, int64_t batch, int seq, int H,
                        float *out)
{
    for (int64_t b = 0; b < batch; ++b)
        for (int i = 0; i < H; ++i) {
            float acc = -1e9f;
            for (int s = 0; s < seq; ++s) {
                float x = (mask[b * seq + s] == 0.0f) ? -1e9f : hidden[(b * seq + s) * H + i];
                acc = acc > x ? acc : x;
            }
            out[b * H + i] = acc;
        }
}

/* pooling/mod.rs:55-68 last_token_pool: last position with mask > 0, else 0. */
KO_API void ko_last_token_pool(const float *hidden, const float *mask, int64_t batch, int seq,
                               int H, float *out)
{
    for (int64_t b = 0; b < batch; ++b) {
        int last = 0;
        for (int s = seq - 1; s >= 0; --s)
            if (mask[b * seq + s] > 0.0f) {
                last = s;
                break;
            }
        memcpy(out + b * H, hidden + (b * seq + last) * H, sizeof(float) * H);
    }
}

/* cpu/encoder/traits.rs:529-536 l2_normalize_inplace: row /= ||row|| when > 0 */
KO_API void ko_l2_normalize(float *x, int64_t rows, int H)
{
    for (int64_t r = 0; r < rows; ++r) {
        float s = 0.0f;
        for (int i = 0; i < H; ++i) s += x[r * H + i] * x[r * H + i];
        float n = sqrtf(s);
        if (n > 0.0f)
            for (int i = 0; i < H; ++i) x[r * H + i] /= n;
    }
}

/* kjarni-models/src/models/sentence_encoder/model.rs:201-218 encode_batch_flat:
 * hidden states -> mean_pool -> l2_normalize (ALWAYS). */
KO_API int ko_embed_batch(const ko_model *m, const uint32_t *ids, const uint32_t *mask_u32,
                          int64_t batch, int seq, float mask_value, float *out)
{
    int H = m->hidden;
    int64_t T = batch * seq;
    if (T == 0) return 0;
    float *hidden = (float *)malloc(sizeof(float) * T * H);
    float *maskf = (float *)malloc(sizeof(float) * T);
    for (int64_t i = 0; i < T; ++i) maskf[i] = (float)mask_u32[i];
    int rc = ko_encoder_forward(m, ids, mask_u32, NULL, batch, seq, mask_value, hidden);
    if (rc == 0) {
        ko_mean_pool(hidden, maskf, batch, seq, H, out);
        ko_l2_normalize(out, batch, H);
    }
    free(hidden);
    free(maskf);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Classification head: cpu/encoder/classifier.rs:210-273                      */
/* ------------------------------------------------------------------------- */

/* CLS row -> optional dense (+act: tanh for bert.pooler / classifier.dense,
 * relu for pre_classifier) -> classifier.  w_dense == NULL skips the dense. */
KO_API void ko_cls_head(const float *hidden, int64_t batch, int seq, int H, const float *w_dense,
                        const float *b_dense, int dense_act, const float *w_cls,
                        const float *b_cls, int num_labels, float *logits)
{
    float *cls = (float *)malloc(sizeof(float) * batch * H);
    float *feat = (float *)malloc(sizeof(float) * batch * H);
    ko_cls_pool(hidden, batch, seq, H, cls);
    const float *f = cls;
    if (w_dense) {
        ko_linear(cls, w_dense, b_dense, batch, H, H, feat);
        for (int64_t i = 0; i < batch * H; ++i) feat[i] = ko_act(feat[i], dense_act);
        f = feat;
    }
    ko_linear(f, w_cls, b_cls, batch, H, num_labels, logits);
    free(cls);
    free(feat);
}

/* ------------------------------------------------------------------------- */
/* Cosine similarity + scans: kjarni-search/src/vector.rs, kjarni-rag/src/     */
/* segment.rs, kjarni/src/embedder/model.rs                                    */
/* ------------------------------------------------------------------------- */

/* kjarni-search/src/vector.rs:131-148 VectorStore::cosine_similarity */
KO_API float ko_cosine_ks(const float *a, const float *b, int n)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (int i = 0; i < n; ++i) {
        dot += a[i] * b[i];
        na += a[i] * a[i];
        nb += b[i] * b[i];
    }
    float den = sqrtf(na) * sqrtf(nb);
    if (!(den > 1e-9f)) den = 1e-9f; /* f32::max(den, 1e-9) */
    return dot / den;
}

/* kjarni-rag/src/segment.rs:355-371 cosine_similarity_with_norm */
KO_API float ko_cosine_kr(const float *q, const float *d, int n, float q_norm)
{
    float dot = 0.0f, nb = 0.0f;
    for (int i = 0; i < n; ++i) {
        dot += q[i] * d[i];
        nb += d[i] * d[i];
    }
    float dn = sqrtf(nb);
    if (dn < 1e-9f) return 0.0f;
    return dot / (q_norm * dn);
}

/* kjarni/src/embedder/model.rs:247-257 cosine_similarity (== 0 guards) */
KO_API float ko_cosine_k(const float *a, const float *b, int n)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (int i = 0; i < n; ++i) dot += a[i] * b[i];
    for (int i = 0; i < n; ++i) na += a[i] * a[i];
    for (int i = 0; i < n; ++i) nb += b[i] * b[i];
    na = sqrtf(na);
    nb = sqrtf(nb);
    if (na == 0.0f || nb == 0.0f) return 0.0f;
    return dot / (na * nb);
}

/* Scores of every document (the scan itself, single thread scalar like the
 * reference: segment.rs:320-326 / vector.rs:155-160).  mode 0 = KS, 1 = KR. */
KO_API void ko_cosine_scan(const float *query, const float *corpus, int64_t n_docs, int dim,
                           int mode, float *scores)
{
    float qn = 0.0f;
    for (int i = 0; i < dim; ++i) qn += query[i] * query[i];
    qn = sqrtf(qn);
    for (int64_t d = 0; d < n_docs; ++d)
        scores[d] = mode ? ko_cosine_kr(query, corpus + d * dim, dim, qn)
                         : ko_cosine_ks(query, corpus + d * dim, dim);
}

typedef struct {
    int64_t idx;
    float score;
} ko_hit;

/* Descending by score; ties keep ascending document index (this is what the
 * reference's stable sort_by over an index-ordered Vec yields:
 * vector.rs:162 / segment.rs:336).  NaN compares Equal in the reference
 * (partial_cmp().unwrap_or(Equal)); scans of finite vectors never make NaN. */
static int ko_hit_cmp(const void *pa, const void *pb)
{
    const ko_hit *a = (const ko_hit *)pa, *b = (const ko_hit *)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* VectorStore::search (vector.rs:150-166) / Segment::search_vectors
 * (segment.rs:307-337): full scan, top `limit` by score descending.
 * Returns the number of hits written (min(limit, n_docs); 0 when the KR query
 * norm < 1e-9, segment.rs:315-317). */
KO_API int64_t ko_search(const float *query, const float *corpus, int64_t n_docs, int dim, int mode,
                         int64_t limit, int64_t *out_idx, float *out_score)
{
    if (n_docs <= 0 || limit <= 0) return 0;
    if (mode) {
        float qn = 0.0f;
        for (int i = 0; i < dim; ++i) qn += query[i] * query[i];
        if (sqrtf(qn) < 1e-9f) return 0;
    }
    float *scores = (float *)malloc(sizeof(float) * n_docs);
    ko_hit *hits = (ko_hit *)malloc(sizeof(ko_hit) * n_docs);
    ko_cosine_scan(query, corpus, n_docs, dim, mode, scores);
    for (int64_t d = 0; d < n_docs; ++d) {
        hits[d].idx = d;
        hits[d].score = scores[d];
    }
    qsort(hits, (size_t)n_docs, sizeof(ko_hit), ko_hit_cmp);
    int64_t k = limit < n_docs ? limit : n_docs;
    for (int64_t i = 0; i < k; ++i) {
        out_idx[i] = hits[i].idx;
        out_score[i] = hits[i].score;
    }
    free(scores);
    free(hits);
    return k;
}

KO_API int ko_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

KO_API void ko_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
