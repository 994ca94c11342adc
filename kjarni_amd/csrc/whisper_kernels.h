// Launchers of whisper.hip (same conventions as kernels.h: enqueue on `stream`, no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace kjarni {

// frames[f, 0..ld): windowed reflect-padded signal (mel.rs:62-101); ld >= n_fft, extra columns zero.
hipError_t launch_mel_frames(const float* audio, int64_t n_samples, const float* window, int n_fft, int hop,
                             int n_frames, int ld, float* frames, hipStream_t stream);
// power[f, k] = sqrt(re^2 + im^2)^2 from the DFT GEMM's output (re at column k, im at im_off + k).
hipError_t launch_mel_power(const float* dft, int ld_dft, int im_off, int n_bins, int ld_out, int64_t n_frames,
                            float* power, hipStream_t stream);
// whisper_log_mel (mel.rs:124-136) on mel energies [n_frames, ld] -> out [n_frames, ld_out] (time-major).
hipError_t launch_mel_log_normalize(float* mel, int ld, int n_mels, int64_t n_frames, uint32_t* max_scratch,
                                    int ld_out, float* out, hipStream_t stream);
// kernel-3 im2col on time-major data: cols [t_out, ld_cols], column k*C + c.
hipError_t launch_im2col3(const float* x, int64_t ldx, int t_in, int channels, int stride, int pad, int t_out,
                          int ld_cols, float* cols, hipStream_t stream);
// x[r, :] += table[r % period, :] for r % period < table_rows.
hipError_t launch_add_rows(float* x, int64_t rows, int hidden, int period, const float* table, int table_rows,
                           hipStream_t stream);
// offset_ptr (device int) overrides offset when given.
hipError_t launch_decoder_embed(const uint32_t* ids, int n, int hidden, int vocab, const float* word, const float* pos,
                                int max_pos, int offset, const int* offset_ptr, int scale_embeddings, float* out,
                                hipStream_t stream, int lanes = 0);  // lanes: every row sits at the same position

// Y = epi(LN?(X) W^T + b) (+ R) for up to 8 rows (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL).
//  gamma != null: X rows are layer-normalised (gamma, beta, eps) on the fly.
//  seg > 0: output columns [0,seg) -> Y0 (ld ldy0), [seg,2seg) -> Y1, [2seg,3seg) -> Y2 (ld ldy12), the latter two at
//  row (*row_off_ptr or row_off) + r.
struct GemvArgs {
    const float* X = nullptr;
    int64_t ldx = 0;
    int rows = 0;
    const float *gamma = nullptr, *beta = nullptr;
    float eps = 0.0f;
    const float *W = nullptr, *bias = nullptr, *R = nullptr;
    int64_t ldr = 0;
    int n_out = 0, k = 0, seg = 0;
    float* Y0 = nullptr;
    int64_t ldy0 = 0;
    float *Y1 = nullptr, *Y2 = nullptr;
    int64_t ldy12 = 0;
    int row_off = 0;
    const int* row_off_ptr = nullptr;
    GemmEpilogue epi = EPI_BIAS;
    // One row of 512 / 2048 floats only (gemv_rows_takes_row_extras): the kernel can build its input row itself as
    // launch_decoder_embed would (embed_ids != null: word[*embed_ids] * embed_scale + pos_table[*embed_pos_ptr | embed_pos]; X is
    // then unused; LayerNorm + EPI_BIAS, 512-float rows) and leave copies of it: x_raw_out = the row as read / built (the residual
    // stream), x_norm_out = the row after the LayerNorm (the model's last hidden state when the vocabulary head folds the final
    // norm in).
    const uint32_t* embed_ids = nullptr;
    const float *embed_word = nullptr, *embed_pos_table = nullptr;
    int embed_vocab = 0, embed_max_pos = 0, embed_pos = 0;
    const int* embed_pos_ptr = nullptr;
    float embed_scale = 1.0f;
    float *x_raw_out = nullptr, *x_norm_out = nullptr;
};
bool gemv_rows_takes_row_extras(const GemvArgs& args);
hipError_t launch_gemv_rows(const GemvArgs& args, hipStream_t stream);
#ifdef KJARNI_TUNING
hipError_t attention_stamps(unsigned long long* out16, int reset);  // measurements: decode attention's cycles per phase
void set_gemv_rows_variant(int variant);  // 0 = rows staged in LDS when there are several, 1 = always the per-wave kernel
#endif

// Attention of `rows` query rows over cached keys/values, split over `splits` key ranges + a combine pass.
// n_keys_ptr (device int) given: keys = *n_keys_ptr + rows and the causal base = *n_keys_ptr (max_keys bounds it);
// causal_base < 0 = no causal mask.  scratch: decode_attention_scratch_floats(...) floats.
size_t decode_attention_scratch_floats(int rows, int heads, int head_dim, int splits);
// One-token output projection fed by the attention's slabs (launch_decode_attention with ctx == nullptr):
// Y[n] = merged(slabs) . W[n, :] + bias[n] + R[n], k = heads * head_dim in {512, 2048}, splits <= 16.
bool gemv_row_att_supported(int k, int splits, int head_dim);
hipError_t launch_gemv_row_att(const float* slabs, int splits, int head_dim, const float* W, const float* bias, const float* R, int n_out,
                               int k, float* Y, hipStream_t stream);
hipError_t launch_decode_attention(const float* q, int64_t ldq, int rows, const float* K, int64_t ldk, const float* V,
                                   int64_t ldv, int n_keys, const int* n_keys_ptr, int max_keys, int heads, int head_dim,
                                   int causal_base, int splits, float* scratch, float* ctx, int64_t ldc, hipStream_t stream,
                                   int kv_group = 1,  // grouped-query attention: kv_group query heads per KV head
                                   // lanes: the rows are independent sequences decoded in lock step; row s reads K + s * k_lane_stride
                                   // (V alike), every row sees n_keys (or *n_keys_ptr + 1) keys and there is no causal mask.
                                   int64_t k_lane_stride = 0, int64_t v_lane_stride = 0, int lanes = 0);
// history/count/pos (device, may be null): append the token, advance the counters (graph-replayed steps).
hipError_t launch_pick_token(const float* logits, int vocab, int first_special, int eos, int timestamp_begin,
                             int allow_timestamps, int32_t* out, int32_t* history, int* count, int* pos, hipStream_t stream,
                             // lanes > 1: logits [lanes, vocab], out / count [lanes], history [lanes, hist_stride]; lane 0 advances
                             // *pos by 1 and *row (the interleaved cache row) by `lanes`
                             int lanes = 1, int hist_stride = 0, int* row = nullptr,
                             // given (>= lanes zeroed entries): the scan runs as 48 small workgroups per lane + a one-thread-per-lane finish
                             unsigned long long* best_scratch = nullptr);

}  // namespace kjarni
