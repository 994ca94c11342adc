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
hipError_t launch_decoder_embed(const uint32_t* ids, int n, int hidden, int vocab, const float* word, const float* pos,
                                int max_pos, int offset, int scale_embeddings, float* out, hipStream_t stream);
// Y = epi(X W^T + b) for up to 8 rows (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL); more rows go to launch_gemm.
hipError_t launch_gemv_rows(const float* X, int64_t ldx, int rows, const float* W, const float* bias, const float* R,
                            int64_t ldr, int n_out, int k, float* Y, int64_t ldy, GemmEpilogue epi, hipStream_t stream);
// Attention of `rows` query rows over n_keys cached keys/values; causal_base < 0 = no causal mask.
hipError_t launch_decode_attention(const float* q, int64_t ldq, int rows, const float* K, int64_t ldk, const float* V,
                                   int64_t ldv, int n_keys, int heads, int head_dim, int causal_base, float* ctx,
                                   int64_t ldc, hipStream_t stream);
hipError_t launch_pick_token(const float* logits, int vocab, int first_special, int eos, int timestamp_begin,
                             int allow_timestamps, int32_t* out, hipStream_t stream);

}  // namespace kjarni
