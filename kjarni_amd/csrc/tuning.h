// Every A/B switch of the TUNING build (make tuning -> libkjarni_ffi_tuning.so, -DKJARNI_TUNING; tools/ only) in one
// place.  The shipped library is compiled without KJARNI_TUNING: each predicate below is then a constant `false`, so
// the product dispatch in gemm.hip / attention.hip / cosine.hip reads -- and compiles -- as if the switch were absent.
//
// gemm variant (kjarni_hip_set_gemm_variant):
//    3  erf-GELU epilogue through libm-grade erff instead of the A&S 7.1.26 polynomial
//    4  residual projections without the fused LayerNorm kernel
//    5  mid-size route: one workgroup per tile instead of the persistent tile loop
//    6  no few-rows (K over the waves) kernel
//    7  no mid-size (64 x 64 tiles, K slices) route
//    9  128 x 128 tiles without an epilogue (micro-benchmark upper bound)
//   10  plain-epilogue 128 x 128 tiles without the persistent tile loop
//   11  fused LayerNorm tiles as a persistent launch (two workgroups per CU)
//   12  128 x 128 tiles: plain output stores whatever the output's size (no streaming stores for outputs >= 256 MB)
//   14  fused LayerNorm tiles read bias / gamma / beta from global memory in the epilogue (rounds 1-2) instead of from LDS
//   15  residual projection + LayerNorm of a few rows (up to gemm_few_rows_max) with a long K: the 64 x 64-tile K slices instead of the few-rows kernel's
//   31..34  persistent grid of the 64 x 64-tile kernel: 768 / 512 / 256 / 1 280 workgroups instead of 1 024
//   21..26  knock-out diagnostics of the 64 x 64-tile kernel (unsliced launches): its DIAG template parameter 1..6
//   17  f32-on-bf16 mode: the 64 x 64-tile route stays on the f32 matrix cores
//   52  128 x 128 tiles: the epilogue without its global stores (diagnostic)
//   53  128 x 128 tiles: one launch-resident workgroup per tile sequence WITH a prologue per tile (gemm_nt_f32_mfma) where the
//       continuous K-stream kernel (gemm_nt_f32_stream) would run
//   54  gemm_nt_f32_stream: wave 0 of every workgroup leaves (cycles in K-steps, cycles between them, tiles) in the first floats of
//       the output (diagnostic; the output is then not the product)
//   56  gemm_nt_f32_stream: every tile of a workgroup stored over the workgroup's first (plain stores: the output stays in the L2;
//       diagnostic -- what the stores cost as instructions, without their memory traffic)
//   55  gemm_nt_f32_stream: the output stores without the LDS transpose and the barrier behind it (diagnostic: values in the wrong places)
// 100000 + r  f32-on-bf16 mode: the split kernel takes calls from r rows (instead of 6 144: gemm.hip, split_min_rows()) in place of the 64 x 64-tile route
// 1000 + r  the few-rows kernel takes calls of up to r rows (sweeps of the few-rows / 64 x 64-tile crossover), r < 1000
//    8  mid-size calls on the 64 x 64 tiles of gemm.hip instead of gemm_flex.hip's per-call tile
// 2000 + 100 RA + CB  gemm_flex.hip: this tile (64 RA rows x 16 CB columns) instead of the launcher's choice
// 3000 + 100 RA + CB  the same tile with an LDS claim of more than half a CU (one workgroup per CU, never two)
// 5000 + 100 RA + CB  that tile for calls of any size (the per-call tile kernel against the large-batch tiles at 10^5 rows)
// 10000 d + 2000 + 100 RA + CB  the same with knock-out d (tiles 128 x 144 and 128 x 192 only): 1 no output stores, 2 no global loads in
//      the K-loop, 3 nothing but the MFMAs in the K-loop, 4 no barrier in the K-loop, 5 no staging (loads + LDS writes) in the K-loop,
//      6 / 7 / 8 staging loads with the sc1 / sc0 / nt cache-policy bit, 9 streaming (nt) output stores
// attention variant (kjarni_hip_set_attention_variant):
//    1  never the persistent pipelined kernel (seq <= 128)
//   11..16  knock-out diagnostics of the pipelined kernel (its DIAG template parameter 1..6)
//   20  two resident workgroups per CU instead of three
//   21  padded calls of seq <= 96 on the full 128-key kernel (no skipping of empty key tiles and waves), no split kernel either
//   22  small calls (<= 256 items) on the item-per-workgroup kernels instead of the split one (a key tile per wave)
// cosine variant (kjarni_hip_set_cosine_variant):
//    1  streaming passes only (no GEMM route for many queries)
//    2  one query: scores + selection as separate launches instead of the fused pass
//    3  many queries (diagnostic, results meaningless): every tile of the matrix-core scan reads the corpus' first 256 rows -- the kernel without its HBM stream
//    4  the same without the epilogue; 5 the same without the norm arithmetic of the K-loop; 6 / 7: the real corpus stream without the epilogue / the norms
//    8  many queries, large corpus: the f32 matrix-core scan selects (round 5's route) instead of the bf16 filter pass + exact rescoring
#pragma once

#ifdef KJARNI_TUNING
#include <atomic>
#endif

namespace kjarni {
namespace tune {

#ifdef KJARNI_TUNING
extern std::atomic<int> g_gemm, g_attention, g_cosine;  // defined in hip_api.cpp
inline int gemm() { return g_gemm.load(std::memory_order_relaxed); }
inline int attention() { return g_attention.load(std::memory_order_relaxed); }
inline int cosine() { return g_cosine.load(std::memory_order_relaxed); }
#else
constexpr int gemm() { return 0; }
constexpr int attention() { return 0; }
constexpr int cosine() { return 0; }
#endif

inline bool gelu_through_libm() { return gemm() == 3; }
inline bool no_fused_layernorm() { return gemm() == 4; }
inline bool mid_one_workgroup_per_tile() { return gemm() == 5; }
inline bool no_few_rows_route() { return gemm() == 6; }
inline bool no_mid_route() { return gemm() == 7; }
inline bool tiles_without_epilogue() { return gemm() == 9; }
inline bool layernorm_params_from_global() { return gemm() == 14; }
inline bool no_persistent_tile_loop() { return gemm() == 10; }
inline bool persistent_layernorm_tiles() { return gemm() == 11; }
inline bool no_few_rows_k_slices() { return gemm() == 15; }
inline int mid_grid_override() { return gemm() >= 31 && gemm() <= 34 ? (gemm() == 31 ? 768 : gemm() == 32 ? 512 : gemm() == 33 ? 256 : 1280) : 0; }
inline bool mid_split_off() { return gemm() == 17; }
inline int mid_knockout() { return gemm() >= 21 && gemm() <= 26 ? gemm() - 20 : 0; }
inline int split_min_rows_override() { return gemm() >= 100000 ? gemm() - 100000 : 0; }  // (measurements: 100000 + rows)
inline int few_rows_max_override() { return gemm() >= 1000 && gemm() < 2000 ? gemm() - 1000 : 0; }  // (measurements: 1000 + rows)
inline int flex_config_override() { return gemm() < 100000 && gemm() % 10000 >= 2000 && gemm() % 10000 < 6000 && gemm() % 10000 / 1000 != 4 ? gemm() % 1000 : 0; }  // (2000 + 100 RA + CB: gemm_flex.hip's tile)
inline bool flex_one_workgroup_per_cu() { return gemm() < 100000 && gemm() % 10000 >= 3000 && gemm() % 10000 < 4000; }  // (3000 + 100 RA + CB: claim more than half of the LDS)
inline bool flex_any_rows() { return gemm() < 100000 && gemm() % 10000 >= 5000 && gemm() % 10000 < 6000; }  // (5000 + 100 RA + CB: that tile at ANY row count)
inline int flex_knockout() { return gemm() >= 12000 && gemm() < 100000 && flex_config_override() ? gemm() / 10000 : 0; }
inline bool no_flex_route() { return gemm() == 8; }
inline bool no_streaming_output_stores() { return gemm() == 12; }
inline bool tiles_without_stores() { return gemm() == 52; }
inline bool no_k_stream_tiles() { return gemm() == 53; }
inline bool tiles_with_cycle_counts() { return gemm() == 54; }
inline bool tiles_without_transpose() { return gemm() == 55; }
inline bool tiles_stored_in_place() { return gemm() == 56; }

inline bool no_pipelined_attention() { return attention() == 1; }
inline int attention_knockout() { return attention() >= 11 && attention() <= 16 ? attention() - 10 : 0; }
inline bool attention_two_workgroups_per_cu() { return attention() == 20; }
inline bool no_short_attention() { return attention() == 21; }
inline bool no_split_attention() { return attention() == 22 || attention() == 21; }

inline bool scan_streaming_only() { return cosine() == 1; }
inline bool scan_two_launches() { return cosine() == 2; }
inline bool scan_f32_select() { return cosine() == 8; }
inline int scan_diag() { return cosine() >= 3 && cosine() <= 7 ? cosine() - 2 : 0; }  // 1 same tile, 2 + no epilogue, 3 + no norms, 4 / 5: the real stream without epilogue / norms

}  // namespace tune
}  // namespace kjarni
