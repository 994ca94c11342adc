// Kernels of the Whisper path that the BERT-family encoder does not already have: the log-mel front
// end around two GEMMs (DFT, mel filterbank), im2col for the two strided convolutions, and the
// single-sequence decoder step (row GEMV with fused epilogues, cached attention, filtered argmax).
//
//   log-mel        crates/kjarni-transformers/src/audio/mel.rs:60-262
//   conv1d         mel.rs:331-371 (as im2col + the projection GEMM)
//   decoder step   cpu/encoder_decoder/cpu_decoder.rs:457-516, decoder_cross_attn.rs:71-120,
//                  encoder_decoder/decoder_self_attn.rs:52-146
//   token choice   crates/kjarni-models/src/models/whisper/transcriber.rs:243-270
#include <atomic>

#include "device_utils.h"
#include "whisper_kernels.h"

namespace kjarni {

namespace {

constexpr float kMaskValue = -1e9f;  // utils/masks.rs MASK_VALUE

// ---- log-mel ------------------------------------------------------------------------------------

// frames[f, i] = padded[f*hop + i] * window[i]  (i < n_fft; columns up to ld are zero), where `padded`
// is the signal reflect-padded by n_fft/2 on both sides (mel.rs:139-160).  A frame that would run past
// the padded signal is all zero (the reference's loop breaks there and leaves zeros).
__global__ __launch_bounds__(256) void mel_frames_kernel(const float* __restrict__ audio, int64_t n,
                                                         const float* __restrict__ window, int n_fft, int hop,
                                                         int n_frames, int ld, float* __restrict__ frames)
{
    const int f = blockIdx.x;
    const int pad = n_fft / 2;
    const int64_t n_padded = n + 2 * (int64_t)pad;
    const bool live = (int64_t)f * hop + n_fft <= n_padded;
    for (int i = threadIdx.x; i < ld; i += 256) {
        float v = 0.0f;
        if (live && i < n_fft) {
            const int64_t j = (int64_t)f * hop + i;
            int64_t src;
            if (j < pad) {
                src = pad - j;                       // left reflection: audio[pad], ..., audio[1]
                if (src >= n) src = n - 1;
            } else if (j < pad + n) {
                src = j - pad;
            } else {
                const int64_t r = j - pad - n;       // right reflection: audio[n-2], audio[n-3], ...
                src = (n >= 2 + r) ? n - 2 - r : 0;
            }
            v = audio[src] * window[i];
        }
        frames[(int64_t)f * ld + i] = v;
    }
}

// power[f, k] = |X_k|^2 computed as the reference does: mag = sqrt(re*re + im*im); mag*mag
// (mel.rs:108-110, 256-258).  re at column k, im at column im_off + k of the DFT GEMM's output.
__global__ __launch_bounds__(256) void mel_power_kernel(const float* __restrict__ dft, int ld_dft, int im_off,
                                                        int n_bins, int ld_out, int64_t n_frames,
                                                        float* __restrict__ power)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_frames * ld_out) return;
    const int64_t f = idx / ld_out;
    const int k = (int)(idx % ld_out);
    float v = 0.0f;
    if (k < n_bins) {
        const float re = dft[f * ld_dft + k], im = dft[f * ld_dft + im_off + k];
        const float mag = sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));
        v = __fmul_rn(mag, mag);
    }
    power[idx] = v;
}

__device__ __forceinline__ uint32_t orderable_f32(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable_f32(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// log10(max(x, 1e-10)) in place over the first n_mels columns + global maximum (whisper_log_mel,
// mel.rs:124-136).  *max_bits must be zeroed before the launch.
__global__ __launch_bounds__(256) void mel_log_max_kernel(float* __restrict__ mel, int ld, int n_mels,
                                                          int64_t n_frames, uint32_t* __restrict__ max_bits)
{
    __shared__ float red[4];
    float mx = -INFINITY;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_frames * n_mels;
         idx += (int64_t)gridDim.x * 256) {
        const int64_t f = idx / n_mels;
        const int m = (int)(idx % n_mels);
        const float v = log10f(fmaxf(mel[f * ld + m], 1e-10f));
        mel[f * ld + m] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(max_bits, orderable_f32(mx));
    }
}

// out[f, m] = (max(x, max - 8) + 4) / 4, written time-major with row stride ld_out.
__global__ __launch_bounds__(256) void mel_normalize_kernel(const float* __restrict__ mel, int ld, int n_mels,
                                                            int64_t n_frames, const uint32_t* __restrict__ max_bits,
                                                            int ld_out, float* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_frames * n_mels) return;
    const int64_t f = idx / n_mels;
    const int m = (int)(idx % n_mels);
    const float mx = from_orderable_f32(*max_bits);
    out[f * ld_out + m] = (fmaxf(mel[f * ld + m], mx - 8.0f) + 4.0f) / 4.0f;
}

// ---- convolution as im2col ------------------------------------------------------------------------

// cols[t, k*C + c] = x[t*stride + k - pad, c] (zero outside, zero for columns >= 3*C): the GEMM against
// the weight re-laid as [out, k*C + c] is conv1d(kernel 3) of mel.rs:331-371 on time-major data.
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, int64_t ldx, int t_in, int channels,
                                                      int stride, int pad, int t_out, int ld_cols,
                                                      float* __restrict__ cols)
{
    const int t = blockIdx.x;
    for (int j = threadIdx.x; j < ld_cols; j += 256) {
        float v = 0.0f;
        if (j < 3 * channels) {
            const int k = j / channels, c = j - k * channels;
            const int ti = t * stride + k - pad;
            if (ti >= 0 && ti < t_in) v = x[(int64_t)ti * ldx + c];
        }
        cols[(int64_t)t * ld_cols + j] = v;
    }
    (void)t_out;
}

// x[r, :] += table[(r % period) (+ offset), :]   (position rows; rows beyond the table are left alone)
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ x, int64_t rows, int hidden, int period,
                                                       const float* __restrict__ table, int table_rows)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * hidden) return;
    const int64_t r = idx / hidden;
    const int p = (int)(r % period);
    if (p < table_rows) x[idx] += table[(int64_t)p * hidden + (idx - r * hidden)];
}

// ---- decoder step ---------------------------------------------------------------------------------

// h[s, :] = word[ids[s], :] (* sqrt(H) when scaled) + pos[offset + s, :]   (Embeddings::forward as the
// seq2seq decoder calls it, cpu_decoder.rs:432-439; ids >= vocab leave zeros as in the encoder path).
// offset_ptr (device) overrides `offset` when given: the graph-replayed step reads its position there.
__global__ __launch_bounds__(256) void decoder_embed_kernel(const uint32_t* __restrict__ ids, int hidden, int vocab,
                                                            const float* __restrict__ word, const float* __restrict__ pos,
                                                            int max_pos, int offset, const int* __restrict__ offset_ptr,
                                                            int pos_step, float scale, float* __restrict__ out)
{
    const int s = blockIdx.x;
    const uint32_t id = ids[s];
    const int p = (offset_ptr ? *offset_ptr : offset) + s * pos_step;  // pos_step 0: rows are lanes at the same position
    for (int i = threadIdx.x; i < hidden; i += 256) {
        float v = 0.0f;
        if (id < (uint32_t)vocab) v = word[(int64_t)id * hidden + i] * scale;
        if (pos && p < max_pos) v += pos[(int64_t)p * hidden + i];
        out[(int64_t)s * hidden + i] = v;
    }
}

#ifdef KJARNI_TUNING
// Measurements (tuning build): shader cycles the decode attention's register path spends per phase, summed over workgroups --
// [0] entry -> scores (the loads' round trips + the dots), [1] -> block max, [2] -> exp, weighted V, block sum, [3] -> slab
// stored, [4] workgroups counted; [5] / [6] / [7]: the one-row GEMV's entry -> dot reduced, -> stored, waves sampled.  kjarni_hip_attention_stamps reads / resets them.
__device__ unsigned long long g_att_stamp[16];   // ([15]: stamps on; [8] .. [12]: the one-row LN GEMV's entry -> row arrived, -> arguments arrived, -> weight requests issued, waves, -> every request issued)
#endif
constexpr int GEMV_MAX_ROWS = 8;

// Y[r, n] = epi(LN?(X[r, :]) . W[n, :] + bias[n]) (+ R[r, n]) for a handful of rows r: one wave per output
// column, the weight row streamed once with 16-byte loads and reused for every input row.
//  * LN: the input rows are layer-normalised on the fly (two-pass mean / variance per row, recomputed by
//    every wave: a row is a few KB and sits in L2), so LayerNorm + projection is one launch.
//  * The n_out columns may be split into up to three equal segments written to different buffers (Q | K | V);
//    segments 1 and 2 are written at row (*row_off_ptr or row_off) + r: straight into the KV cache.
template <int EPI, bool LN>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        const float* __restrict__ R, int64_t ldr, int n_out, int k, int seg,
                                                        float* __restrict__ Y0, int64_t ldy0, float* __restrict__ Y1,
                                                        float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                        const int* __restrict__ row_off_ptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_out) return;
    const int k4 = k >> 2;
    float mean[GEMV_MAX_ROWS], rstd[GEMV_MAX_ROWS];
    if (LN) {
#pragma unroll
        for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
            mean[r] = 0.0f;
            rstd[r] = 1.0f;
            if (r < rows) {
                float s = 0.0f;
                for (int i = lane; i < k4; i += 64) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
                    s += (x[0] + x[1]) + (x[2] + x[3]);
                }
                const float mu = wave_sum(s) / (float)k;
                float v = 0.0f;
                for (int i = lane; i < k4; i += 64) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v = fmaf(x[c] - mu, x[c] - mu, v);
                }
                mean[r] = mu;
                rstd[r] = 1.0f / sqrtf(wave_sum(v) / (float)k + eps);
            }
        }
    }
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + n * (int64_t)k);
    float acc[GEMV_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) acc[r] = 0.0f;
    for (int i = lane; i < k4; i += 64) {
        const f32x4 w = w4[i];
        f32x4 g = f32x4{1.f, 1.f, 1.f, 1.f}, bt = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LN) {
            g = *reinterpret_cast<const f32x4*>(gamma + i * 4);
            bt = *reinterpret_cast<const f32x4*>(beta + i * 4);
        }
#pragma unroll
        for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
            if (r < rows) {
                f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
                if (LN) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = (x[c] - mean[r]) * rstd[r] * g[c] + bt[c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r] = fmaf(x[c], w[c], acc[r]);
            }
        }
    }
    const float b = bias ? bias[n] : 0.0f;
    const int which = seg > 0 ? (int)(n / seg) : 0;
    const int64_t col = seg > 0 ? n - (int64_t)which * seg : n;
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    const int64_t ldy = which == 0 ? ldy0 : ldy12;
    const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
        if (r < rows) {
            float v = wave_sum(acc[r]) + b;
            if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
            if (EPI == EPI_BIAS_RESIDUAL) v += R[r * ldr + n];
            if (lane == 0) Y[(r0 + r) * ldy + col] = v;
        }
    }
}

// gemv_rows_kernel for ONE row whose length is 256 KCH floats, with every request of the wave issued before anything is
// waited for: the weight row (non-temporal), bias / residual, the input row, gamma / beta -- then the statistics, the
// normalisation and the dot product run on registers.  A decode step is a chain of ~5 us launches, each a chain of memory
// round trips: the general kernel pays one for the row, a second for the weights and a third for the residual.  Same
// lane-to-chunk map, LayerNorm arithmetic and summation order: results are bit-identical to gemv_rows_kernel's.
// EMBED (the first projection of a one-token decoder step): the input row is not read from X but built here, as
// decoder_embed_kernel builds it -- word[id] * scale + pos[p] (id = *emb.ids, p = *emb.pos_ptr or emb.pos; ids >= vocab and
// positions >= max_pos leave zeros) -- by every wave for itself; the wave of column 0 also stores it to x_raw_out (the residual
// stream the later projections add to).  x_norm_out (LN only): the wave of column 0 stores the NORMALISED row there (the model's
// last hidden state when the vocabulary head folds the final LayerNorm in).  One launch fewer each, the same arithmetic.
struct GemvEmbed {
    const uint32_t* ids;
    const float *word, *pos_table;
    int vocab, max_pos, pos;
    const int* pos_ptr;
    float scale;
};

// COLS output columns per wave, WAVES waves per workgroup: a wave's requests for the row, gamma and beta are the same whatever
// its column, and on a compute unit they queue behind one another in the one address pipe -- with one column per wave a wave
// of the 1536-column projection needed ~2 300 cycles just to ISSUE its eight loads (stamps: docs/history/r06.md); with COLS
// columns a wave issues 6 + 2 COLS instead of 8 COLS.  Each column's dot product keeps its own lane order and reduction.
template <int EPI, bool LN, int KCH, bool EMBED = false, int COLS = 1, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) void gemv_row_fast_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps,
                                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                                   const float* R, int n_out, int seg, float* Y0,
                                                                   float* __restrict__ Y1, float* __restrict__ Y2, int64_t ldy12,
                                                                   int row_off, const int* __restrict__ row_off_ptr,
                                                                   float* __restrict__ x_raw_out, float* __restrict__ x_norm_out,
                                                                   GemvEmbed emb)
{
    constexpr int K = 256 * KCH;
#ifdef KJARNI_TUNING
    const unsigned long long gst0 = __builtin_amdgcn_s_memtime();
#endif
    // (every argument in one batch of scalar loads: device_utils.h)
    kj_args_now(X, gamma, beta, eps, W, bias, R, n_out, seg, Y0, Y1, Y2, ldy12, row_off, row_off_ptr, x_raw_out, x_norm_out);
    if (EMBED) kj_args_now(emb.ids, emb.word, emb.pos_table, emb.vocab, emb.max_pos, emb.pos, emb.pos_ptr, emb.scale);
#ifdef KJARNI_TUNING
    unsigned long long gsta;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gsta) : "s"(x_norm_out), "s"(ldy12) : "memory");   // (arguments arrived)
#endif
    const int lane = threadIdx.x & 63;
    const int wave = WAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int64_t n0 = ((int64_t)blockIdx.x * WAVES + wave) * COLS;
    if (n0 >= n_out) return;
    f32x4 x[KCH], w[COLS][KCH], g[KCH], bt[KCH];
#if defined(KJARNI_TUNING) && defined(KJARNI_GEMV_PROBE)
    // (timing probe, results garbage: the arrays every wave shares read from a place of the wave's own instead)
    gamma = W + ((n0 * 7 + 3) % n_out) * (int64_t)K;
    beta = W + ((n0 * 13 + 5) % n_out) * (int64_t)K;
    if (KJARNI_GEMV_PROBE >= 2) X = W + ((n0 * 11 + 1) % n_out) * (int64_t)K;
#endif
    // Order of issue = order of need, and nothing that waits stands before a request: the row offset (a device counter) is a
    // scalar load waited for only at the store; the weight rows -- the one stream that comes from HBM -- go first.
    int r0s = row_off;
    if (seg > 0 && row_off_ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(r0s) : "s"(row_off_ptr));
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int64_t nc = COLS > 1 && n0 + c >= n_out ? n_out - 1 : n0 + c;   // (a ragged last wave re-reads the last row)
        const f32x4* w4 = reinterpret_cast<const f32x4*>(W + nc * (int64_t)K);
#pragma unroll
        for (int j = 0; j < KCH; ++j) w[c][j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    }
#ifdef KJARNI_TUNING
    asm volatile("" ::: "memory");
    const unsigned long long gstw = __builtin_amdgcn_s_memtime();   // (weight requests issued)
    asm volatile("" ::: "memory");
#endif
    if (EMBED) {
        const uint32_t id = emb.ids[0];
        const int p = emb.pos_ptr ? *emb.pos_ptr : emb.pos;
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (id < (uint32_t)emb.vocab) {
                v = *reinterpret_cast<const f32x4*>(emb.word + (int64_t)id * K + (lane + 64 * j) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = v[c] * emb.scale;
            }
            if (emb.pos_table && p < emb.max_pos) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(emb.pos_table + (int64_t)p * K + (lane + 64 * j) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += pv[c];
            }
            x[j] = v;
        }
        if (n0 == 0 && x_raw_out) {
#pragma unroll
            for (int j = 0; j < KCH; ++j) *reinterpret_cast<f32x4*>(x_raw_out + (lane + 64 * j) * 4) = x[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < KCH; ++j) x[j] = *reinterpret_cast<const f32x4*>(X + (lane + 64 * j) * 4);
    }
    if (LN) {
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            g[j] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * j) * 4);
            bt[j] = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * j) * 4);
        }
    }
    // (bias and residual of the wave's columns: lane c holds column n0 + c's)
    const int64_t nl = n0 + lane < n_out ? n0 + lane : n_out - 1;
    float b = 0.0f, res = 0.0f;
    if (lane < COLS) {
        if (bias) b = bias[nl];
        if (EPI == EPI_BIAS_RESIDUAL) res = R[nl];
    }
#ifdef KJARNI_TUNING
    unsigned long long gstx = 0;
    asm volatile("" ::: "memory");
    const unsigned long long gsti = __builtin_amdgcn_s_memtime();   // (every request issued)
#endif
    if (LN) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j) s += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]);
#ifdef KJARNI_TUNING
        asm volatile("" : "+v"(s));
        gstx = __builtin_amdgcn_s_memtime();   // (the input row has arrived)
#endif
        const float mu = wave_sum(s) / (float)K;
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) v = fmaf(x[j][c] - mu, x[j][c] - mu, v);
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)K + eps);
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[j][c] = (x[j][c] - mu) * rstd * g[j][c] + bt[j][c];
        if (n0 == 0 && x_norm_out) {
#pragma unroll
            for (int j = 0; j < KCH; ++j) *reinterpret_cast<f32x4*>(x_norm_out + (lane + 64 * j) * 4) = x[j];
        }
    }
    float out = 0.0f;
#pragma unroll
    for (int cc = 0; cc < COLS; ++cc) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = fmaf(x[j][c], w[cc][j][c], acc);
        const float d = wave_sum(acc);
        out = lane == cc ? d : out;
    }
    float v = out + b;
#ifdef KJARNI_TUNING
    asm volatile("" : "+v"(v));
    const unsigned long long gst1 = __builtin_amdgcn_s_memtime();
#endif
    if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
    if (EPI == EPI_BIAS_RESIDUAL) v += res;
    const int64_t n = n0 + lane;
    const int which = seg > 0 ? (int)(n >= seg) + (int)(n >= 2 * (int64_t)seg) : 0;   // (n / seg: three segments at most)
    const int64_t col = n - (int64_t)which * seg;
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0s));
    if (lane < COLS && n < n_out) Y[(which == 0 ? 0 : (int64_t)r0s * ldy12) + col] = v;
#ifdef KJARNI_TUNING
    const unsigned long long gst2 = __builtin_amdgcn_s_memtime();   // (before the flag is read: its round trip is not the store's)
    if (lane == 0 && n_out <= 4096 && (blockIdx.x & 31) == 0 && wave == 0 && g_att_stamp[15] != 0ull) {   // (a sample of the waves; not the vocabulary head; only while the stamps are being taken)
        atomicAdd(&g_att_stamp[5], gst1 - gst0);
        atomicAdd(&g_att_stamp[6], gst2 - gst1);
        atomicAdd(&g_att_stamp[7], 1ull);
        if (LN) {
            atomicAdd(&g_att_stamp[8], gstx - gst0);    // (entry -> the row arrived)
            atomicAdd(&g_att_stamp[9], gsta - gst0);    // (entry -> arguments arrived)
            atomicAdd(&g_att_stamp[10], gstw - gst0);   // (entry -> weight requests issued)
            atomicAdd(&g_att_stamp[11], 1ull);
            atomicAdd(&g_att_stamp[12], gsti - gst0);   // (entry -> every request issued)
        }
    }
#endif
}

// The same projection for 2..8 rows (several tokens of one sequence, or the lanes of a lock-step decode): the rows are
// (normalised and) staged in LDS once per workgroup -- wave w prepares rows w, w + 4 -- instead of being re-read and
// re-normalised from L2 by every wave for every row.  Lane-to-chunk assignment, LayerNorm arithmetic and reduction order
// are those of gemv_rows_kernel, so a row's result is bit-identical whichever kernel computes it.
template <int EPI, bool LN>
__global__ __launch_bounds__(256) void gemv_rows_lds_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ R, int64_t ldr, int n_out, int k, int seg,
                                                            float* __restrict__ Y0, int64_t ldy0, float* __restrict__ Y1,
                                                            float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                            const int* __restrict__ row_off_ptr)
{
    extern __shared__ __attribute__((aligned(16))) float gx[];  // [rows][k]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k4 = k >> 2;
    for (int r = wave; r < rows; r += 4) {
        float mu = 0.0f, rs = 1.0f;
        if (LN) {
            float s = 0.0f;
            for (int i = lane; i < k4; i += 64) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
                s += (x[0] + x[1]) + (x[2] + x[3]);
            }
            mu = wave_sum(s) / (float)k;
            float v = 0.0f;
            for (int i = lane; i < k4; i += 64) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) v = fmaf(x[c] - mu, x[c] - mu, v);
            }
            rs = 1.0f / sqrtf(wave_sum(v) / (float)k + eps);
        }
        for (int i = lane; i < k4; i += 64) {
            f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + i * 4);
            if (LN) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 4);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + i * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = (x[c] - mu) * rs * g[c] + bt[c];
            }
            *reinterpret_cast<f32x4*>(gx + (int64_t)r * k + i * 4) = x;
        }
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= n_out) return;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + n * (int64_t)k);
    float acc[GEMV_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) acc[r] = 0.0f;
    for (int i = lane; i < k4; i += 64) {
        const f32x4 w = w4[i];
#pragma unroll
        for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
            if (r < rows) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(gx + (int64_t)r * k + i * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r] = fmaf(x[c], w[c], acc[r]);
            }
        }
    }
    const float b = bias ? bias[n] : 0.0f;
    const int which = seg > 0 ? (int)(n / seg) : 0;
    const int64_t col = seg > 0 ? n - (int64_t)which * seg : n;
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    const int64_t ldy = which == 0 ? ldy0 : ldy12;
    const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
        if (r < rows) {
            float v = wave_sum(acc[r]) + b;
            if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
            if (EPI == EPI_BIAS_RESIDUAL) v += R[r * ldr + n];
            if (lane == 0) Y[(r0 + r) * ldy + col] = v;
        }
    }
}

// The attention output projection of a one-token step, fed by the decode attention's per-split slabs instead of a context
// row: Y[n] = merged(slabs) . W[n, :] + bias[n] + R[n].  The workgroup merges the slabs once into LDS (the arithmetic of
// decode_attention_combine_kernel, element for element) while its four weight rows are in flight, then each wave takes its
// dot product in gemv_row_fast_kernel's order -- bit-identical to combine + projection, one launch fewer.
constexpr int ATT_MERGE_MAX_SPLITS = 16;

template <int KCH>
__global__ __launch_bounds__(256) void gemv_row_att_kernel(const float* __restrict__ part, int splits, int head_dim,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* R, int n_out, float* Y)  // R == Y in place
{
    constexpr int K = 256 * KCH;
    __shared__ __attribute__((aligned(16))) float gx[K];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nc = n < n_out ? n : n_out - 1;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + nc * (int64_t)K);
    f32x4 w[KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) w[j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    const float b = bias ? bias[nc] : 0.0f;
    const float res = R[nc];
    const int stride = head_dim + 4;
    for (int q = tid; q < K / 4; q += 256) {
        const int col = q * 4, head = col / head_dim, j = col - head * head_dim;
        const float* slab = part + (int64_t)head * splits * stride;
        f32x2 hd[ATT_MERGE_MAX_SPLITS];
        f32x4 av[ATT_MERGE_MAX_SPLITS];
#pragma unroll
        for (int i = 0; i < ATT_MERGE_MAX_SPLITS; ++i) {  // every slab of the head requested at once
            const float* p = slab + (i < splits ? i : 0) * stride;
            hd[i] = *reinterpret_cast<const f32x2*>(p);
            av[i] = *reinterpret_cast<const f32x4*>(p + 4 + j);
        }
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < ATT_MERGE_MAX_SPLITS; ++i)
            if (i < splits) M = fmaxf(M, hd[i][0]);
        float L = 0.0f;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ATT_MERGE_MAX_SPLITS; ++i) {
            if (i < splits) {
                const float wgt = (hd[i][0] == -INFINITY) ? 0.0f : expf(hd[i][0] - M);
                L = fmaf(hd[i][1], wgt, L);
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fmaf(av[i][c], wgt, a[c]);
            }
        }
        const float inv = 1.0f / L;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = L > 0.0f ? a[c] * inv : a[c];
        *reinterpret_cast<f32x4*>(gx + col) = o;
    }
    __syncthreads();
    if (n >= n_out) return;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(gx + (lane + 64 * j) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = fmaf(x[c], w[j][c], acc);
    }
    const float v = wave_sum(acc) + b + res;
    if (lane == 0) Y[n] = v;
}

// gemv_rows_lds_kernel for rows of 256 KCH floats with the wave's weight row, bias and residual values requested before the
// rows are staged (they do not depend on them): the staging's three passes over X then run under the weight round trip.
// Same arithmetic and order as gemv_rows_lds_kernel: bit-identical results.
template <int EPI, bool LN, int KCH>
__global__ __launch_bounds__(256) void gemv_rows_lds_fast_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const float* R, int64_t ldr, int n_out, int seg,
                                                                 float* Y0, int64_t ldy0, float* __restrict__ Y1,
                                                                 float* __restrict__ Y2, int64_t ldy12, int row_off,
                                                                 const int* __restrict__ row_off_ptr)
{
    extern __shared__ __attribute__((aligned(16))) float gx[];  // [rows][k]
    constexpr int k = 256 * KCH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nc = n < n_out ? n : n_out - 1;  // surplus waves of the last workgroup still help staging
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + nc * (int64_t)k);
    f32x4 w[KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) w[j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    const float b = bias ? bias[nc] : 0.0f;
    float res[GEMV_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) res[r] = (EPI == EPI_BIAS_RESIDUAL && r < rows) ? R[r * ldr + nc] : 0.0f;
    const int which = seg > 0 ? (int)(nc / seg) : 0;
    const int64_t r0 = which == 0 ? 0 : (row_off_ptr ? *row_off_ptr : row_off);
    for (int r = wave; r < rows; r += 4) {
        f32x4 x[KCH];
#pragma unroll
        for (int j = 0; j < KCH; ++j) x[j] = *reinterpret_cast<const f32x4*>(X + r * ldx + (lane + 64 * j) * 4);
        if (LN) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < KCH; ++j) s += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]);
            const float mu = wave_sum(s) / (float)k;
            float v = 0.0f;
#pragma unroll
            for (int j = 0; j < KCH; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) v = fmaf(x[j][c] - mu, x[j][c] - mu, v);
            const float rs = 1.0f / sqrtf(wave_sum(v) / (float)k + eps);
#pragma unroll
            for (int j = 0; j < KCH; ++j) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * j) * 4);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * j) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[j][c] = (x[j][c] - mu) * rs * g[c] + bt[c];
            }
        }
#pragma unroll
        for (int j = 0; j < KCH; ++j) *reinterpret_cast<f32x4*>(gx + (int64_t)r * k + (lane + 64 * j) * 4) = x[j];
    }
    __syncthreads();
    if (n >= n_out) return;
    float acc[GEMV_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
#pragma unroll
        for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
            if (r < rows) {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(gx + (int64_t)r * k + (lane + 64 * j) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r] = fmaf(xr[c], w[j][c], acc[r]);
            }
        }
    }
    const int64_t col = seg > 0 ? n - (int64_t)which * seg : n;
    float* Y = which == 0 ? Y0 : (which == 1 ? Y1 : Y2);
    const int64_t ldy = which == 0 ? ldy0 : ldy12;
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
        if (r < rows) {
            float v = wave_sum(acc[r]) + b;
            if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
            if (EPI == EPI_BIAS_RESIDUAL) v += res[r];
            if (lane == 0) Y[(r0 + r) * ldy + col] = v;
        }
    }
}

// The vocabulary head for 2..8 rows: Y[r, n] = X[r, :] . W[n, :] + bias[n] over tens of thousands of columns.  A workgroup
// stages the rows in LDS once and then takes 4 CPW columns (CPW per wave, all their weight rows requested up front): with one
// column per wave the staging traffic (rows x k floats per workgroup) was twice the weight traffic.  Same per-element
// arithmetic and order as gemv_rows_lds_kernel.
template <int KCH, int CPW>
__global__ __launch_bounds__(256) void gemv_rows_head_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                             const float* __restrict__ W, const float* __restrict__ bias, int n_out,
                                                             float* __restrict__ Y, int64_t ldy)
{
    extern __shared__ __attribute__((aligned(16))) float gx[];  // [rows][k]
    constexpr int k = 256 * KCH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * CPW;
    f32x4 w[CPW][KCH];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int64_t nc = n0 + c < n_out ? n0 + c : n_out - 1;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(W + nc * (int64_t)k);
#pragma unroll
        for (int j = 0; j < KCH; ++j) w[c][j] = __builtin_nontemporal_load(w4 + lane + 64 * j);
    }
    float b[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) b[c] = bias ? bias[n0 + c < n_out ? n0 + c : n_out - 1] : 0.0f;
    for (int r = wave; r < rows; r += 4) {
#pragma unroll
        for (int j = 0; j < KCH; ++j)
            *reinterpret_cast<f32x4*>(gx + (int64_t)r * k + (lane + 64 * j) * 4) =
                *reinterpret_cast<const f32x4*>(X + r * ldx + (lane + 64 * j) * 4);
    }
    __syncthreads();
    float acc[GEMV_MAX_ROWS][CPW];
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[r][c] = 0.0f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
#pragma unroll
        for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
            if (r < rows) {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(gx + (int64_t)r * k + (lane + 64 * j) * 4);
#pragma unroll
                for (int c = 0; c < CPW; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][c] = fmaf(xr[e], w[c][j][e], acc[r][c]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < GEMV_MAX_ROWS; ++r) {
        if (r < rows) {
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const float v = wave_sum(acc[r][c]) + b[c];
                if (lane == 0 && n0 + c < n_out) Y[r * ldy + n0 + c] = v;
            }
        }
    }
}

// Cached attention for a few query rows, split over the keys ("flash decoding"): workgroup (head, split, row) reduces its
// key range to a slab (max, sum of exp, sum of exp * V); the slabs are merged by decode_attention_combine_kernel, or by the
// consumer of the context row itself (the LLM's output projection reads the slabs while its weights are in flight:
// llm_kernels.hip).  K rows are ldk floats apart with head h at columns [h*d, h*d+d); same for V.  The number of keys is
// n_keys, or *n_keys_ptr + rows when the pointer is given (graph replay: keys = cache length + new rows).
// causal >= 0 (or the pointer): query row s sees keys <= base + s (apply_causal_mask, utils/masks.rs:103-113: masked scores
// are OVERWRITTEN with -1e9, which exp() then turns into exactly 0 next to any real score).
// Slab layout per (row, head, split): [max, sum, 0, 0, acc[head_dim]] (16-byte aligned pieces).
constexpr int ATT_MAX_CHUNK = 512;
constexpr int ATT_FAST = 8;  // keys per lane group that the short-range path holds in registers (128 keys per split at d = 64)

__global__ __launch_bounds__(256) void decode_attention_partial_kernel(const float* __restrict__ q, int64_t ldq,
                                                                       const float* __restrict__ K0, int64_t ldk,
                                                                       const float* __restrict__ V0, int64_t ldv, int n_keys,
                                                                       const int* __restrict__ n_keys_ptr, int rows, int head_dim,
                                                                       float scale, int causal, int splits, int kv_group,
                                                                       int64_t k_lane_stride, int64_t v_lane_stride, int lanes,
                                                                       float* __restrict__ part)
{
    __shared__ float sc[ATT_MAX_CHUNK];
    __shared__ float red[4];
    __shared__ f32x4 accs[4 * 32];
    const int h = blockIdx.x, sp = blockIdx.y, s = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = n_keys_ptr ? *n_keys_ptr : 0;
    // lanes: the rows are independent sequences in lock step -- each attends to its own cache (row s at K + s * stride),
    // all of them hold the same number of keys (the ones before this step plus this step's own), no mask between rows.
    const int n = n_keys_ptr ? base + (lanes ? 1 : rows) : n_keys;
    const int causal_base = (causal < 0 || lanes) ? -1 : (n_keys_ptr ? base : causal);
    const float* __restrict__ K = K0 + (int64_t)blockIdx.z * k_lane_stride;
    const float* __restrict__ V = V0 + (int64_t)blockIdx.z * v_lane_stride;
    const int chunk = (n + splits - 1) / splits;
    const int t0 = sp * chunk, t1 = min(n, t0 + chunk);
    const int lpk = head_dim >> 2;          // lanes per key (16 for d = 64)
    const int groups = 256 / lpk;
    const int g = tid / lpk, l = tid - g * lpk;
    const int col = (h / kv_group) * head_dim + l * 4;  // grouped-query attention: kv_group query heads share a KV head
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + (int64_t)s * ldq + h * head_dim + l * 4);

    auto block_max = [&](float v) {
        v = wave_max(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        return m;
    };
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const float t = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
        return t;
    };

    float mx = -INFINITY, sum = 0.0f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef KJARNI_TUNING
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    unsigned long long st1 = 0, st2 = 0, st3 = 0;
    const bool fast_path = t1 - t0 <= ATT_FAST * groups;
#endif
    if (t1 - t0 <= ATT_FAST * groups) {
        // Short ranges (a decode step over a few hundred cached keys is a chain of latencies, not of bytes): every K and V
        // row of the range is requested at once and held in registers -- one memory round trip instead of one per pass.
        f32x4 kr[ATT_FAST], vr[ATT_FAST];
        const int last = t1 > t0 ? t1 - 1 : 0;
#pragma unroll
        for (int j = 0; j < ATT_FAST; ++j) kr[j] = *reinterpret_cast<const f32x4*>(K + (int64_t)min(t0 + g + j * groups, last) * ldk + col);
#pragma unroll
        for (int j = 0; j < ATT_FAST; ++j) vr[j] = *reinterpret_cast<const f32x4*>(V + (int64_t)min(t0 + g + j * groups, last) * ldv + col);
        float sv[ATT_FAST];
#pragma unroll
        for (int j = 0; j < ATT_FAST; ++j) {
            const int t = t0 + g + j * groups;
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) dot = fmaf(qv[c], kr[j][c], dot);
            if (lpk == 16) dot = row16_sum_desc(dot);  // (d = 64: a key's 16 lanes are a DPP row; the same sum as the loop below)
            else
                for (int off = lpk >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, kWave);
            float v = dot * scale;
            if (causal_base >= 0 && t > causal_base + s) v = kMaskValue;
            sv[j] = t < t1 ? v : -INFINITY;
            mx = fmaxf(mx, sv[j]);
        }
#ifdef KJARNI_TUNING
        st1 = __builtin_amdgcn_s_memtime();
#endif
        mx = block_max(mx);
#ifdef KJARNI_TUNING
        st2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int j = 0; j < ATT_FAST; ++j) {
            const float e = t0 + g + j * groups < t1 ? expf(sv[j] - mx) : 0.0f;
            if (l == 0) sum += e;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(e, vr[j][c], acc[c]);
        }
        sum = block_sum(sum);
#ifdef KJARNI_TUNING
        st3 = __builtin_amdgcn_s_memtime();
#endif
    } else {
        for (int t = t0 + g; t < t1; t += groups) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(K + (int64_t)t * ldk + col);
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) dot = fmaf(qv[c], kv[c], dot);
            if (lpk == 16) dot = row16_sum_desc(dot);  // (d = 64: a key's 16 lanes are a DPP row; the same sum as the loop below)
            else
                for (int off = lpk >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, kWave);
            float v = dot * scale;
            if (causal_base >= 0 && t > causal_base + s) v = kMaskValue;
            if (l == 0) sc[t - t0] = v;
            mx = fmaxf(mx, v);
        }
        mx = block_max(mx);
        for (int t = t0 + tid; t < t1; t += 256) {
            const float e = expf(sc[t - t0] - mx);
            sc[t - t0] = e;
            sum += e;
        }
        sum = block_sum(sum);
        for (int t = t0 + g; t < t1; t += groups) {
            const float p = sc[t - t0];
            const f32x4 vv = *reinterpret_cast<const f32x4*>(V + (int64_t)t * ldv + col);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(p, vv[c], acc[c]);
        }
    }
    // sum of exp * V over the workgroup: the lane groups of a wave by shuffles, the four waves through LDS
    if (lpk == 16) {  // (d = 64: the lane groups are the wave's four rows -- i ^ 16, then i ^ 32, without the LDS crossbar)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = sum_xor32(sum_xor16(acc[c]));
    } else {
        for (int off = lpk; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off, kWave);
        }
    }
    if (lane < lpk) accs[wave * lpk + lane] = acc;
    __syncthreads();
    float* out = part + (((int64_t)s * gridDim.x + h) * splits + sp) * (head_dim + 4);
    if (tid < lpk) *reinterpret_cast<f32x4*>(out + 4 + tid * 4) = (accs[tid] + accs[lpk + tid]) + (accs[2 * lpk + tid] + accs[3 * lpk + tid]);
    if (tid == 0) *reinterpret_cast<f32x4*>(out) = f32x4{(t1 > t0) ? mx : -INFINITY, (t1 > t0) ? sum : 0.0f, 0.0f, 0.0f};
#ifdef KJARNI_TUNING
    const unsigned long long st4 = __builtin_amdgcn_s_memtime();   // (before the flag is read)
    if (tid == 0 && fast_path && g_att_stamp[15] != 0ull) {   // (only while the stamps are being taken: the atomics cost a token 0.07 ms)
        atomicAdd(&g_att_stamp[0], st1 - st0);
        atomicAdd(&g_att_stamp[1], st2 - st1);
        atomicAdd(&g_att_stamp[2], st3 - st2);
        atomicAdd(&g_att_stamp[3], st4 - st3);
        atomicAdd(&g_att_stamp[4], 1ull);
    }
#endif
}


__global__ __launch_bounds__(128) void decode_attention_combine_kernel(const float* __restrict__ part, int heads, int splits,
                                                                       int head_dim, float* __restrict__ ctx, int64_t ldc)
{
    const int h = blockIdx.x, s = blockIdx.y, j = threadIdx.x;
    if (j >= head_dim) return;
    const int stride = head_dim + 4;
    const float* p = part + ((int64_t)s * heads + h) * splits * stride;
    if (splits <= 16) {  // every slab requested before the first is used (same arithmetic, same order: one round trip, not `splits`)
        f32x2 hd[16];
        float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* pi = p + (i < splits ? i : 0) * stride;
            hd[i] = *reinterpret_cast<const f32x2*>(pi);
            av[i] = pi[4 + j];
        }
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < splits) M = fmaxf(M, hd[i][0]);
        float L = 0.0f, a = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < splits) {
                const float w = (hd[i][0] == -INFINITY) ? 0.0f : expf(hd[i][0] - M);
                L = fmaf(hd[i][1], w, L);
                a = fmaf(av[i], w, a);
            }
        }
        ctx[(int64_t)s * ldc + h * head_dim + j] = L > 0.0f ? a * (1.0f / L) : a;
        return;
    }
    float M = -INFINITY;
    for (int i = 0; i < splits; ++i) M = fmaxf(M, p[i * stride]);
    float L = 0.0f, a = 0.0f;
    for (int i = 0; i < splits; ++i) {
        const float* pi = p + i * stride;
        const float w = (pi[0] == -INFINITY) ? 0.0f : expf(pi[0] - M);
        L = fmaf(pi[1], w, L);
        a = fmaf(pi[4 + j], w, a);
    }
    ctx[(int64_t)s * ldc + h * head_dim + j] = L > 0.0f ? a * (1.0f / L) : a;
}

// WhisperModel::pick_token (transcriber.rs:243-270): argmax over the ids that may be produced;
// Iterator::max_by returns the LAST of equal maxima.  The chosen id goes to out[0]; when `history` is given
// (graph replay) it is also appended at history[*count], *count and *pos are advanced: the next step's
// input token, position and key count then live on the device and the host only looks every few steps.
__global__ __launch_bounds__(1024) void pick_token_kernel(const float* __restrict__ logits, int vocab, int first_special,
                                                          int eos, int timestamp_begin, int allow_timestamps,
                                                          int32_t* __restrict__ out, int32_t* __restrict__ history,
                                                          int* __restrict__ count, int* __restrict__ pos, int hist_stride,
                                                          int* __restrict__ row)
{
    // One workgroup per lane (blockIdx.x): its own logits row, output slot, history and count; lane 0 advances the
    // shared position (and the interleaved cache row = position * lanes).
    const int lane_id = blockIdx.x;
    logits += (int64_t)lane_id * vocab;
    out += lane_id;
    if (history) {
        history += (int64_t)lane_id * hist_stride;
        count += lane_id;
    }
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = -1;
    for (int i = threadIdx.x; i < vocab; i += 1024) {
        const bool ok = i < first_special || i == eos || (allow_timestamps && i >= timestamp_begin);
        if (!ok) continue;
        const float v = logits[i];
        if (idx < 0 || v >= best) {  // i ascends per thread: >= keeps the last maximum
            best = v;
            idx = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(idx, off, kWave);
        if (oi >= 0 && (idx < 0 || ov > best || (ov == best && oi > idx))) {
            best = ov;
            idx = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        bv[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bi[w] >= 0 && (idx < 0 || bv[w] > best || (bv[w] == best && bi[w] > idx))) {
                best = bv[w];
                idx = bi[w];
            }
        const int tok = idx >= 0 ? idx : eos;
        *out = tok;
        if (history) {
            history[*count] = tok;
            *count += 1;
            if (lane_id == 0) {
                *pos += 1;
                if (row) *row += (int)gridDim.x;
            }
        }
    }
}

// The same choice in two small launches, for the replayed greedy step: the 1024-thread scan above takes ~22 us of a
// 430 us step.  Keys are (orderable logit << 32 | id): atomicMax keeps the largest logit and, among equals, the largest
// id -- the last maximum, as above.  Key 0 = nothing producible seen (a real key is never 0: the sign flip sets a bit).
__global__ __launch_bounds__(256) void pick_partial_kernel(const float* __restrict__ logits, int vocab, int first_special, int eos,
                                                           int timestamp_begin, int allow_timestamps,
                                                           unsigned long long* __restrict__ best)
{
    __shared__ unsigned long long red[4];
    const int lane_id = blockIdx.y;
    logits += (int64_t)lane_id * vocab;
    unsigned long long key = 0ull;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) {
        const bool ok = i < first_special || i == eos || (allow_timestamps && i >= timestamp_begin);
        if (!ok) continue;
        const float v = logits[i];
        uint32_t u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        if (v != v) u = 1u;  // NaN below every number, above "nothing seen"
        const unsigned long long k = ((unsigned long long)u << 32) | (uint32_t)i;
        key = k > key ? k : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, kWave);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = red[w] > key ? red[w] : key;
        if (key) atomicMax(best + lane_id, key);
    }
}

__global__ void pick_finalize_kernel(unsigned long long* __restrict__ best, int eos, int32_t* __restrict__ out,
                                     int32_t* __restrict__ history, int* __restrict__ count, int* __restrict__ pos, int hist_stride,
                                     int* __restrict__ row, int lanes)
{
    const int lane_id = threadIdx.x;
    if (lane_id >= lanes) return;
    const unsigned long long key = best[lane_id];
    best[lane_id] = 0ull;
    const int tok = key ? (int)(uint32_t)(key & 0xFFFFFFFFull) : eos;
    out[lane_id] = tok;
    if (history) {
        history[(int64_t)lane_id * hist_stride + count[lane_id]] = tok;
        count[lane_id] += 1;
        if (lane_id == 0) {
            *pos += 1;
            if (row) *row += lanes;
        }
    }
}

}  // namespace

#ifdef KJARNI_TUNING
hipError_t attention_stamps(unsigned long long* out16, int reset)
{
    // reset != 0: counters to zero and the stamps ON ([15] = 1); reset == 0: read them and switch the stamps OFF again
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_att_stamp), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long next[16] = {};
    if (reset) next[15] = 1ull;
    else
        for (int i = 0; i < 15; ++i) next[i] = out16[i];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_att_stamp), next, sizeof(next));
}
#endif

// ---- launchers --------------------------------------------------------------------------------------

hipError_t launch_mel_frames(const float* audio, int64_t n_samples, const float* window, int n_fft, int hop,
                             int n_frames, int ld, float* frames, hipStream_t stream)
{
    hipLaunchKernelGGL(mel_frames_kernel, dim3((unsigned)n_frames), dim3(256), 0, stream, audio, n_samples, window,
                       n_fft, hop, n_frames, ld, frames);
    return hipGetLastError();
}

hipError_t launch_mel_power(const float* dft, int ld_dft, int im_off, int n_bins, int ld_out, int64_t n_frames,
                            float* power, hipStream_t stream)
{
    const int64_t total = n_frames * ld_out;
    hipLaunchKernelGGL(mel_power_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dft, ld_dft,
                       im_off, n_bins, ld_out, n_frames, power);
    return hipGetLastError();
}

hipError_t launch_mel_log_normalize(float* mel, int ld, int n_mels, int64_t n_frames, uint32_t* max_scratch,
                                    int ld_out, float* out, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(max_scratch, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const int64_t total = n_frames * n_mels;
    int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(mel_log_max_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, stream, mel, ld,
                       n_mels, n_frames, max_scratch);
    hipLaunchKernelGGL(mel_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, mel, ld, n_mels, n_frames,
                       max_scratch, ld_out, out);
    return hipGetLastError();
}

hipError_t launch_im2col3(const float* x, int64_t ldx, int t_in, int channels, int stride, int pad, int t_out,
                          int ld_cols, float* cols, hipStream_t stream)
{
    hipLaunchKernelGGL(im2col3_kernel, dim3((unsigned)t_out), dim3(256), 0, stream, x, ldx, t_in, channels, stride, pad,
                       t_out, ld_cols, cols);
    return hipGetLastError();
}

hipError_t launch_add_rows(float* x, int64_t rows, int hidden, int period, const float* table, int table_rows,
                           hipStream_t stream)
{
    const int64_t total = rows * hidden;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, rows, hidden,
                       period, table, table_rows);
    return hipGetLastError();
}

hipError_t launch_decoder_embed(const uint32_t* ids, int n, int hidden, int vocab, const float* word, const float* pos,
                                int max_pos, int offset, const int* offset_ptr, int scale_embeddings, float* out,
                                hipStream_t stream, int lanes)
{
    const float scale = scale_embeddings ? sqrtf((float)hidden) : 1.0f;
    hipLaunchKernelGGL(decoder_embed_kernel, dim3((unsigned)n), dim3(256), 0, stream, ids, hidden, vocab, word, pos, max_pos,
                       offset, offset_ptr, lanes ? 0 : 1, scale, out);
    return hipGetLastError();
}

#ifdef KJARNI_TUNING
std::atomic<int> g_gemv_rows_variant{0};  // 1 = never stage the rows in LDS -- tuning build only
#else
constexpr int g_gemv_rows_variant = 0;
#endif
#ifdef KJARNI_TUNING
void set_gemv_rows_variant(int v) { g_gemv_rows_variant = v; }
#endif

// The one-row kernel for 512- / 2048-float rows is the only one that honours embed_* / x_raw_out / x_norm_out.
bool gemv_rows_takes_row_extras(const GemvArgs& a)
{
    const int kch = a.k / 256;
    const bool ln = a.gamma != nullptr;
    return a.rows == 1 && a.k % 256 == 0 && (kch == 2 || kch == 8) && g_gemv_rows_variant == 0 && (!ln || a.beta) &&
           (!ln || ((reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.beta)) & 15) == 0) && (a.k & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && (a.embed_ids || ((a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0));
}

hipError_t launch_gemv_rows(const GemvArgs& a, hipStream_t stream)
{
    if (a.rows <= 0 || a.n_out <= 0) return hipSuccess;
    if ((a.embed_ids || a.x_raw_out || a.x_norm_out) && !gemv_rows_takes_row_extras(a)) return hipErrorInvalidValue;
    const bool simple = !a.gamma && a.seg <= 0;
    if (a.rows > GEMV_MAX_ROWS || (a.k & 3) || (a.ldx & 3) || (reinterpret_cast<uintptr_t>(a.X) & 15) ||
        (reinterpret_cast<uintptr_t>(a.W) & 15)) {
        if (!simple) return hipErrorInvalidValue;
        return launch_gemm(a.X, a.ldx, a.W, a.bias, a.R, a.ldr, a.Y0, a.ldy0, a.rows, a.n_out, a.k, a.epi, stream);
    }
    const dim3 grid((unsigned)((a.n_out + 3) / 4));
    const size_t lds = (size_t)a.rows * a.k * sizeof(float);
    const bool staged = a.rows >= 2 && lds <= 64 * 1024 && g_gemv_rows_variant == 0;
    const bool staged_fast = staged && a.k % 256 == 0 && (a.k / 256 == 2 || a.k / 256 == 8) && (!a.gamma || a.beta) &&
                             ((reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.beta)) & 15) == 0;
    if (staged_fast && !a.gamma && a.epi == EPI_BIAS && a.seg <= 0 && a.n_out >= 8192 && a.k == 512) {  // the vocabulary head
        constexpr int CPW = 4;
        hipLaunchKernelGGL((gemv_rows_head_kernel<2, CPW>), dim3((unsigned)((a.n_out + 4 * CPW - 1) / (4 * CPW))), dim3(256), lds, stream,
                           a.X, a.ldx, a.rows, a.W, a.bias, a.n_out, a.Y0, a.ldy0);
        return hipGetLastError();
    }
#define KJ_STAGED_FAST(EPI, LN, KCH)                                                                                             \
    do {                                                                                                                         \
        auto kern = gemv_rows_lds_fast_kernel<EPI, LN, KCH>;                                                                     \
        if (lds > 48 * 1024) {                                                                                                   \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)lds);                                                                  \
            if (e != hipSuccess) return e;                                                                                       \
        }                                                                                                                        \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a.X, a.ldx, a.rows, a.gamma, a.beta, a.eps, a.W, a.bias, a.R, a.ldr, \
                           a.n_out, a.seg, a.Y0, a.ldy0, a.Y1, a.Y2, a.ldy12, a.row_off, a.row_off_ptr);                         \
    } while (0)
#define KJ_GEMV(EPI, LN)                                                                                                         \
    do {                                                                                                                         \
        if (staged_fast) {                                                                                                       \
            if (a.k == 512) KJ_STAGED_FAST(EPI, LN, 2);                                                                          \
            else KJ_STAGED_FAST(EPI, LN, 8);                                                                                     \
        } else if (staged) {                                                                                                     \
            auto kern = gemv_rows_lds_kernel<EPI, LN>;                                                                           \
            if (lds > 48 * 1024) {                                                                                               \
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
                if (e != hipSuccess) return e;                                                                                   \
            }                                                                                                                    \
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a.X, a.ldx, a.rows, a.gamma, a.beta, a.eps, a.W, a.bias, a.R, \
                               a.ldr, a.n_out, a.k, a.seg, a.Y0, a.ldy0, a.Y1, a.Y2, a.ldy12, a.row_off, a.row_off_ptr);        \
        } else {                                                                                                                 \
            hipLaunchKernelGGL((gemv_rows_kernel<EPI, LN>), grid, dim3(256), 0, stream, a.X, a.ldx, a.rows, a.gamma, a.beta,     \
                               a.eps, a.W, a.bias, a.R, a.ldr, a.n_out, a.k, a.seg, a.Y0, a.ldy0, a.Y1, a.Y2, a.ldy12,           \
                               a.row_off, a.row_off_ptr);                                                                        \
        }                                                                                                                        \
    } while (0)
    const bool ln = a.gamma != nullptr;
    // one row of 512 or 2048 floats (the Whisper-base decoder's widths): the variant with every request issued up front
    const int kch = a.k / 256;
    if (a.rows == 1 && a.k % 256 == 0 && (kch == 2 || kch == 8) && g_gemv_rows_variant == 0 && (!ln || a.beta) &&
        (!ln || ((reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.beta)) & 15) == 0)) {
        const GemvEmbed emb{a.embed_ids, a.embed_word, a.embed_pos_table, a.embed_vocab, a.embed_max_pos, a.embed_pos, a.embed_pos_ptr,
                            a.embed_scale};
        int cols = 1, waves = 4;   // (columns per wave, waves per workgroup)
#ifdef KJARNI_TUNING
        if (const char* e = getenv("KJARNI_HIP_GEMV_COLS")) cols = atoi(e);
        if (const char* e = getenv("KJARNI_HIP_GEMV_WAVES")) waves = atoi(e);
        if ((cols != 1 && cols != 2 && cols != 4) || (waves != 1 && waves != 4)) return hipErrorInvalidValue;
#endif
        const dim3 fgrid((unsigned)((a.n_out + (int64_t)cols * waves - 1) / ((int64_t)cols * waves))), fblock(64 * waves);
#define KJ_FAST4(EPI, LN, KCH, EMB, COLS, WAVES)                                                                                   \
    hipLaunchKernelGGL((gemv_row_fast_kernel<EPI, LN, KCH, EMB, COLS, WAVES>), fgrid, fblock, 0, stream, a.X, a.gamma, a.beta, a.eps, \
                       a.W, a.bias, a.R, a.n_out, a.seg, a.Y0, a.Y1, a.Y2, a.ldy12, a.row_off, a.row_off_ptr, a.x_raw_out,         \
                       a.x_norm_out, emb)
#ifdef KJARNI_TUNING
#define KJ_FAST3(EPI, LN, KCH, EMB)                                                                                                \
    do {                                                                                                                           \
        if (waves == 4) {                                                                                                          \
            if (cols == 1) KJ_FAST4(EPI, LN, KCH, EMB, 1, 4);                                                                      \
            else if (cols == 2) KJ_FAST4(EPI, LN, KCH, EMB, 2, 4);                                                                 \
            else KJ_FAST4(EPI, LN, KCH, EMB, 4, 4);                                                                                \
        } else {                                                                                                                   \
            if (cols == 1) KJ_FAST4(EPI, LN, KCH, EMB, 1, 1);                                                                      \
            else if (cols == 2) KJ_FAST4(EPI, LN, KCH, EMB, 2, 1);                                                                 \
            else KJ_FAST4(EPI, LN, KCH, EMB, 4, 1);                                                                                \
        }                                                                                                                          \
    } while (0)
#else
#define KJ_FAST3(EPI, LN, KCH, EMB) KJ_FAST4(EPI, LN, KCH, EMB, 1, 4)
#endif
        if (a.embed_ids) {  // the first projection of a one-token step builds its input row itself (LN + Q | K | V, 512-float rows)
            if (!(ln && a.epi == EPI_BIAS && kch == 2)) return hipErrorInvalidValue;
            KJ_FAST3(EPI_BIAS, true, 2, true);
            return hipGetLastError();
        }
#define KJ_FAST(EPI, LN)                                                                                                          \
    do {                                                                                                                          \
        if (kch == 2) KJ_FAST3(EPI, LN, 2, false);                                                                                \
        else KJ_FAST3(EPI, LN, 8, false);                                                                                         \
    } while (0)
        switch (a.epi) {
        case EPI_BIAS:
            if (ln) KJ_FAST(EPI_BIAS, true);
            else KJ_FAST(EPI_BIAS, false);
            break;
        case EPI_BIAS_GELU:
            if (ln) KJ_FAST(EPI_BIAS_GELU, true);
            else KJ_FAST(EPI_BIAS_GELU, false);
            break;
        case EPI_BIAS_RESIDUAL:
            if (ln) KJ_FAST(EPI_BIAS_RESIDUAL, true);
            else KJ_FAST(EPI_BIAS_RESIDUAL, false);
            break;
        default: return hipErrorInvalidValue;
        }
#undef KJ_FAST
#undef KJ_FAST3
#undef KJ_FAST4
        return hipGetLastError();
    }
    switch (a.epi) {
    case EPI_BIAS:
        if (ln) KJ_GEMV(EPI_BIAS, true);
        else KJ_GEMV(EPI_BIAS, false);
        break;
    case EPI_BIAS_GELU:
        if (ln) KJ_GEMV(EPI_BIAS_GELU, true);
        else KJ_GEMV(EPI_BIAS_GELU, false);
        break;
    case EPI_BIAS_RESIDUAL:
        if (ln) KJ_GEMV(EPI_BIAS_RESIDUAL, true);
        else KJ_GEMV(EPI_BIAS_RESIDUAL, false);
        break;
    default: return hipErrorInvalidValue;
    }
#undef KJ_GEMV
#undef KJ_STAGED_FAST
    return hipGetLastError();
}

bool gemv_row_att_supported(int k, int splits, int head_dim)
{
    return (k == 512 || k == 2048) && splits >= 1 && splits <= ATT_MERGE_MAX_SPLITS && head_dim >= 4 && (head_dim & 3) == 0 &&
           k % head_dim == 0 && g_gemv_rows_variant == 0;
}

hipError_t launch_gemv_row_att(const float* slabs, int splits, int head_dim, const float* W, const float* bias, const float* R, int n_out,
                               int k, float* Y, hipStream_t stream)
{
    if (!gemv_row_att_supported(k, splits, head_dim) || !R || (reinterpret_cast<uintptr_t>(W) & 15)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_out + 3) / 4));
    if (k == 512)
        hipLaunchKernelGGL(gemv_row_att_kernel<2>, grid, dim3(256), 0, stream, slabs, splits, head_dim, W, bias, R, n_out, Y);
    else
        hipLaunchKernelGGL(gemv_row_att_kernel<8>, grid, dim3(256), 0, stream, slabs, splits, head_dim, W, bias, R, n_out, Y);
    return hipGetLastError();
}

size_t decode_attention_scratch_floats(int rows, int heads, int head_dim, int splits)
{
    return (size_t)rows * heads * splits * (head_dim + 4);
}

hipError_t launch_decode_attention(const float* q, int64_t ldq, int rows, const float* K, int64_t ldk, const float* V,
                                   int64_t ldv, int n_keys, const int* n_keys_ptr, int max_keys, int heads, int head_dim,
                                   int causal_base, int splits, float* scratch, float* ctx, int64_t ldc, hipStream_t stream,
                                   int kv_group, int64_t k_lane_stride, int64_t v_lane_stride, int lanes)
{
    if (rows <= 0 || (n_keys <= 0 && !n_keys_ptr)) return hipSuccess;
    if (head_dim > 128 || 256 % (head_dim / 4) != 0 || (head_dim & 3) || splits < 1) return hipErrorInvalidValue;
    const int worst = n_keys_ptr ? max_keys : n_keys;
    if ((worst + splits - 1) / splits > ATT_MAX_CHUNK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_attention_partial_kernel, dim3((unsigned)heads, (unsigned)splits, (unsigned)rows), dim3(256), 0, stream,
                       q, ldq, K, ldk, V, ldv, n_keys, n_keys_ptr, rows, head_dim, 1.0f / sqrtf((float)head_dim), causal_base, splits,
                       kv_group < 1 ? 1 : kv_group, k_lane_stride, v_lane_stride, lanes, scratch);
    if (ctx)  // ctx == nullptr: the caller merges the slabs itself (layout above)
        hipLaunchKernelGGL(decode_attention_combine_kernel, dim3((unsigned)heads, (unsigned)rows), dim3(128), 0, stream, scratch, heads,
                           splits, head_dim, ctx, ldc);
    return hipGetLastError();
}

hipError_t launch_pick_token(const float* logits, int vocab, int first_special, int eos, int timestamp_begin,
                             int allow_timestamps, int32_t* out, int32_t* history, int* count, int* pos, hipStream_t stream, int lanes,
                             int hist_stride, int* row, unsigned long long* best_scratch)
{
    const int n = lanes < 1 ? 1 : lanes;
    if (best_scratch) {  // two-stage: many small workgroups + one thread per lane
        hipLaunchKernelGGL(pick_partial_kernel, dim3(48, (unsigned)n), dim3(256), 0, stream, logits, vocab, first_special, eos,
                           timestamp_begin, allow_timestamps, best_scratch);
        hipLaunchKernelGGL(pick_finalize_kernel, dim3(1), dim3(64), 0, stream, best_scratch, eos, out, history, count, pos, hist_stride, row, n);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pick_token_kernel, dim3((unsigned)n), dim3(1024), 0, stream, logits, vocab, first_special, eos, timestamp_begin,
                       allow_timestamps, out, history, count, pos, hist_stride, row);
    return hipGetLastError();
}

}  // namespace kjarni
