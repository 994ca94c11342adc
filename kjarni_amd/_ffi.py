"""ctypes binding of libkjarni_ffi.so (include/kjarni.h + include/kjarni_hip.h).

Mirrors the reference's Python binding (crates/kjarni-ffi/bindings/python/
kjarni/_ffi.py): same struct mirrors, same check_error contract.  The library
is the product; if it is missing this module raises -- there is no Python or
CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KJARNI_FFI_LIB", os.path.join(_HERE, "lib", "libkjarni_ffi.so"))  # override: sanitizer builds


class KjarniError:
    OK = 0
    NULL_POINTER = 1
    INVALID_UTF8 = 2
    MODEL_NOT_FOUND = 3
    LOAD_FAILED = 4
    INFERENCE_FAILED = 5
    GPU_UNAVAILABLE = 6
    INVALID_CONFIG = 7
    CANCELLED = 8
    TIMEOUT = 9
    STREAM_ENDED = 10
    UNKNOWN = 255


class KjarniDevice:
    CPU = 0
    GPU = 1


class KjarniException(Exception):
    def __init__(self, code: int, message: str):
        self.code = code
        self.message = message
        super().__init__(f"{error_name(code)}: {message}")


class KjarniFloatArray(Structure):
    _fields_ = [("data", POINTER(c_float)), ("len", c_size_t)]

    def to_numpy(self) -> np.ndarray:
        if not self.data or self.len == 0:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(self.data, shape=(self.len,)).copy()

    def free(self):
        lib().kjarni_float_array_free(byref(self))


class KjarniFloat2DArray(Structure):
    _fields_ = [("data", POINTER(c_float)), ("rows", c_size_t), ("cols", c_size_t)]

    def to_numpy(self) -> np.ndarray:
        if not self.data or self.rows == 0 or self.cols == 0:
            return np.zeros((0, 0), np.float32)
        return np.ctypeslib.as_array(self.data, shape=(self.rows, self.cols)).copy()

    def free(self):
        lib().kjarni_float_2d_array_free(byref(self))


class KjarniStringArray(Structure):
    _fields_ = [("strings", POINTER(c_char_p)), ("len", c_size_t)]

    def to_list(self):
        return [self.strings[i].decode("utf-8") for i in range(self.len)]

    def free(self):
        lib().kjarni_string_array_free(byref(self))


class KjarniTokenBatch(Structure):
    _fields_ = [("ids", POINTER(C.c_uint32)), ("attention_mask", POINTER(C.c_uint32)),
                ("type_ids", POINTER(C.c_uint32)), ("batch", c_size_t), ("seq", c_size_t)]


class KjarniHipKernelStat(Structure):
    _fields_ = [("kind", c_char_p), ("symbol", c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


class KjarniEmbedderConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p),
                ("model_path", c_char_p), ("normalize", c_int32), ("quiet", c_int32)]


class KjarniClassResult(Structure):
    _fields_ = [("label", c_char_p), ("score", c_float)]


class KjarniClassResults(Structure):
    _fields_ = [("results", POINTER(KjarniClassResult)), ("len", c_size_t)]

    def to_list(self):
        return [(self.results[i].label.decode("utf-8"), float(self.results[i].score))
                for i in range(self.len)]

    def free(self):
        lib().kjarni_class_results_free(byref(self))


class KjarniClassifierConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p),
                ("model_path", c_char_p), ("labels", POINTER(c_char_p)), ("num_labels", c_size_t),
                ("multi_label", c_int32), ("quiet", c_int32)]


class KjarniRerankResult(Structure):
    _fields_ = [("index", c_size_t), ("score", c_float)]


class KjarniRerankResults(Structure):
    _fields_ = [("results", POINTER(KjarniRerankResult)), ("len", c_size_t)]

    def to_list(self):
        return [(int(self.results[i].index), float(self.results[i].score)) for i in range(self.len)]

    def free(self):
        lib().kjarni_rerank_results_free(byref(self))


class KjarniSearchResult(Structure):
    _fields_ = [("score", c_float), ("document_id", c_size_t), ("text", c_char_p), ("metadata_json", c_char_p)]


class KjarniSearchResults(Structure):
    _fields_ = [("results", POINTER(KjarniSearchResult)), ("len", c_size_t)]

    def to_list(self):
        import json
        out = []
        for i in range(self.len):
            r = self.results[i]
            out.append(dict(score=float(r.score), document_id=int(r.document_id),
                            text=r.text.decode("utf-8") if r.text else "",
                            metadata=json.loads(r.metadata_json.decode("utf-8")) if r.metadata_json else {}))
        return out

    def free(self):
        lib().kjarni_search_results_free(byref(self))


class KjarniSearchOptions(Structure):
    _fields_ = [("mode", c_int32), ("top_k", c_size_t), ("use_reranker", c_int32), ("threshold", c_float),
                ("source_pattern", c_char_p), ("filter_key", c_char_p), ("filter_value", c_char_p)]


class KjarniSearcherConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p), ("rerank_model", c_char_p),
                ("default_mode", c_int32), ("default_top_k", c_size_t), ("quiet", c_int32)]


class KjarniProgress(Structure):
    _fields_ = [("stage", c_int32), ("current", c_size_t), ("total", c_size_t), ("message", c_char_p)]


KjarniProgressCallbackFn = C.CFUNCTYPE(None, KjarniProgress, c_void_p)


class KjarniIndexStats(Structure):
    _fields_ = [("documents_indexed", c_size_t), ("chunks_created", c_size_t), ("dimension", c_size_t),
                ("size_bytes", C.c_uint64), ("files_processed", c_size_t), ("files_skipped", c_size_t),
                ("elapsed_ms", C.c_uint64)]


class KjarniIndexInfo(Structure):
    # char* kept as void* so the pointers survive to kjarni_index_info_free
    _fields_ = [("path", c_void_p), ("document_count", c_size_t), ("segment_count", c_size_t),
                ("dimension", c_size_t), ("size_bytes", C.c_uint64), ("embedding_model", c_void_p)]


class KjarniIndexerConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p), ("chunk_size", c_size_t),
                ("chunk_overlap", c_size_t), ("batch_size", c_size_t), ("extensions", c_char_p),
                ("exclude_patterns", c_char_p), ("recursive", c_int32), ("include_hidden", c_int32),
                ("max_file_size", c_size_t), ("quiet", c_int32)]


class KjarniToken(Structure):
    _fields_ = [("text", c_char_p), ("token_id", C.c_uint32), ("is_special", C.c_bool)]


KjarniTokenCallbackFn = C.CFUNCTYPE(C.c_bool, KjarniToken, c_void_p)


class KjarniTranscriberConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p), ("model_path", c_char_p),
                ("language", c_char_p), ("task", c_int32), ("timestamps", c_int32), ("max_tokens_per_chunk", c_size_t),
                ("quiet", c_int32)]


class KjarniTranscriptionSegment(Structure):
    _fields_ = [("start", c_float), ("end", c_float), ("text", c_char_p)]


class KjarniTranscription(Structure):
    _fields_ = [("text", c_char_p), ("segments", POINTER(KjarniTranscriptionSegment)), ("num_segments", c_size_t),
                ("language", c_char_p), ("duration_secs", c_float)]


class KjarniTranscriptionProgress(Structure):
    _fields_ = [("stage", c_int32), ("current", c_size_t), ("total", c_size_t), ("message", c_char_p)]


KjarniTranscriptionProgressFn = C.CFUNCTYPE(None, KjarniTranscriptionProgress, c_void_p)


class KjarniChatConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p), ("model_path", c_char_p),
                ("system_prompt", c_char_p), ("mode", c_int32), ("quiet", c_int32)]


class KjarniGenerationConfig(Structure):
    _fields_ = [("temperature", c_float), ("top_k", c_int32), ("top_p", c_float), ("min_p", c_float),
                ("repetition_penalty", c_float), ("max_new_tokens", c_int32), ("do_sample", c_int32)]


class KjarniResolvedGeneration(Structure):
    _fields_ = [("strategy", c_int32), ("temperature", c_float), ("top_k", c_int64), ("top_p", c_float), ("min_p", c_float),
                ("repetition_penalty", c_float), ("no_repeat_ngram_size", c_size_t), ("max_new_tokens", c_int64),
                ("max_length", c_size_t), ("add_bos_token", c_int32)]


KjarniStreamCallbackFn = C.CFUNCTYPE(C.c_bool, c_char_p, c_void_p)


class KjarniRerankerConfig(Structure):
    _fields_ = [("device", c_int32), ("cache_dir", c_char_p), ("model_name", c_char_p),
                ("model_path", c_char_p), ("quiet", c_int32)]


_u32p = POINTER(C.c_uint32)
_f32p = POINTER(c_float)
_i64p = POINTER(c_int64)

# name -> (restype, argtypes).  Everything include/*.h declares is listed here;
# tests/test_abi.py checks the two stay in sync.
SIGNATURES = {
    # kjarni.h
    "kjarni_error_name": (c_char_p, [c_int32]),
    "kjarni_error_code_to_string": (c_char_p, [c_int32]),
    "kjarni_last_error_message": (c_char_p, []),
    "kjarni_clear_error": (None, []),
    "kjarni_init": (c_int32, []),
    "kjarni_shutdown": (None, []),
    "kjarni_version": (c_char_p, []),
    "kjarni_float_array_free": (None, [POINTER(KjarniFloatArray)]),
    "kjarni_float_2d_array_free": (None, [POINTER(KjarniFloat2DArray)]),
    "kjarni_string_free": (None, [c_void_p]),
    "kjarni_string_array_free": (None, [POINTER(KjarniStringArray)]),
    "kjarni_cosine_similarity": (c_float, [_f32p, _f32p, c_size_t]),
    "kjarni_embedder_config_default": (KjarniEmbedderConfig, []),
    "kjarni_embedder_new": (c_int32, [POINTER(KjarniEmbedderConfig), POINTER(c_void_p)]),
    "kjarni_embedder_free": (None, [c_void_p]),
    "kjarni_embedder_encode": (c_int32, [c_void_p, c_char_p, POINTER(KjarniFloatArray)]),
    "kjarni_embedder_encode_batch": (c_int32, [c_void_p, POINTER(c_char_p), c_size_t,
                                               POINTER(KjarniFloat2DArray)]),
    "kjarni_embedder_similarity": (c_int32, [c_void_p, c_char_p, c_char_p, POINTER(c_float)]),
    "kjarni_embedder_dim": (c_size_t, [c_void_p]),
    "kjarni_class_results_free": (None, [POINTER(KjarniClassResults)]),
    "kjarni_classifier_config_default": (KjarniClassifierConfig, []),
    "kjarni_classifier_new": (c_int32, [POINTER(KjarniClassifierConfig), POINTER(c_void_p)]),
    "kjarni_classifier_free": (None, [c_void_p]),
    "kjarni_classifier_classify": (c_int32, [c_void_p, c_char_p, POINTER(KjarniClassResults)]),
    "kjarni_classifier_labels": (c_int32, [c_void_p, POINTER(KjarniStringArray)]),
    "kjarni_classifier_num_labels": (c_size_t, [c_void_p]),
    "kjarni_rerank_results_free": (None, [POINTER(KjarniRerankResults)]),
    "kjarni_reranker_config_default": (KjarniRerankerConfig, []),
    "kjarni_reranker_new": (c_int32, [POINTER(KjarniRerankerConfig), POINTER(c_void_p)]),
    "kjarni_reranker_free": (None, [c_void_p]),
    "kjarni_reranker_score": (c_int32, [c_void_p, c_char_p, c_char_p, POINTER(c_float)]),
    "kjarni_reranker_rerank": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t,
                                         POINTER(KjarniRerankResults)]),
    "kjarni_reranker_rerank_top_k": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t, c_size_t,
                                               POINTER(KjarniRerankResults)]),
    "kjarni_search_results_free": (None, [POINTER(KjarniSearchResults)]),
    "kjarni_search_options_default": (KjarniSearchOptions, []),
    "kjarni_searcher_config_default": (KjarniSearcherConfig, []),
    "kjarni_searcher_new": (c_int32, [POINTER(KjarniSearcherConfig), POINTER(c_void_p)]),
    "kjarni_searcher_free": (None, [c_void_p]),
    "kjarni_searcher_search": (c_int32, [c_void_p, c_char_p, c_char_p, POINTER(KjarniSearchResults)]),
    "kjarni_searcher_search_with_options": (c_int32, [c_void_p, c_char_p, c_char_p, POINTER(KjarniSearchOptions),
                                                      POINTER(KjarniSearchResults)]),
    "kjarni_search_keywords": (c_int32, [c_char_p, c_char_p, c_size_t, POINTER(KjarniSearchResults)]),
    "kjarni_searcher_has_reranker": (C.c_bool, [c_void_p]),
    "kjarni_searcher_default_mode": (c_int32, [c_void_p]),
    "kjarni_searcher_default_top_k": (c_size_t, [c_void_p]),
    "kjarni_searcher_model_name": (c_size_t, [c_void_p, c_char_p, c_size_t]),
    "kjarni_searcher_reranker_model": (c_size_t, [c_void_p, c_char_p, c_size_t]),
    "kjarni_cancel_token_new": (c_void_p, []),
    "kjarni_cancel_token_cancel": (None, [c_void_p]),
    "kjarni_cancel_token_is_cancelled": (C.c_bool, [c_void_p]),
    "kjarni_cancel_token_reset": (None, [c_void_p]),
    "kjarni_cancel_token_free": (None, [c_void_p]),
    "kjarni_index_info_free": (None, [KjarniIndexInfo]),
    "kjarni_indexer_config_default": (KjarniIndexerConfig, []),
    "kjarni_indexer_new": (c_int32, [POINTER(KjarniIndexerConfig), POINTER(c_void_p)]),
    "kjarni_indexer_free": (None, [c_void_p]),
    "kjarni_indexer_create": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t, c_int32,
                                        POINTER(KjarniIndexStats)]),
    "kjarni_indexer_create_with_callback": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t, c_int32,
                                                      KjarniProgressCallbackFn, c_void_p, c_void_p,
                                                      POINTER(KjarniIndexStats)]),
    "kjarni_indexer_add": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t, POINTER(c_size_t)]),
    "kjarni_indexer_add_with_callback": (c_int32, [c_void_p, c_char_p, POINTER(c_char_p), c_size_t,
                                                   KjarniProgressCallbackFn, c_void_p, c_void_p,
                                                   POINTER(c_size_t)]),
    "kjarni_index_info": (c_int32, [c_char_p, POINTER(KjarniIndexInfo)]),
    "kjarni_index_delete": (c_int32, [c_char_p]),
    "kjarni_indexer_model_name": (c_size_t, [c_void_p, c_char_p, c_size_t]),
    "kjarni_indexer_dimension": (c_size_t, [c_void_p]),
    "kjarni_indexer_chunk_size": (c_size_t, [c_void_p]),
    "kjarni_transcriber_config_default": (KjarniTranscriberConfig, []),
    "kjarni_transcriber_new": (c_int32, [POINTER(KjarniTranscriberConfig), POINTER(c_void_p)]),
    "kjarni_transcriber_free": (None, [c_void_p]),
    "kjarni_transcription_free": (None, [POINTER(KjarniTranscription)]),
    "kjarni_transcriber_transcribe_audio": (c_int32, [c_void_p, _f32p, c_size_t, C.c_uint32, POINTER(KjarniTranscription)]),
    "kjarni_transcriber_transcribe_file": (c_int32, [c_void_p, c_char_p, POINTER(KjarniTranscription)]),
    "kjarni_transcriber_transcribe_audio_with_callbacks": (c_int32, [
        c_void_p, _f32p, c_size_t, C.c_uint32, KjarniTranscriptionProgressFn, c_void_p, KjarniTokenCallbackFn, c_void_p,
        c_void_p, POINTER(KjarniTranscription)]),
    "kjarni_transcriber_transcribe_file_with_callbacks": (c_int32, [
        c_void_p, c_char_p, KjarniTranscriptionProgressFn, c_void_p, KjarniTokenCallbackFn, c_void_p, c_void_p,
        POINTER(KjarniTranscription)]),
    "kjarni_transcriber_model_name": (c_size_t, [c_void_p, c_char_p, c_size_t]),
    "kjarni_float_array_free_by_value": (None, [KjarniFloatArray]),
    "kjarni_float_2d_array_free_by_value": (None, [KjarniFloat2DArray]),
    "kjarni_string_array_free_by_value": (None, [KjarniStringArray]),
    "kjarni_class_results_free_by_value": (None, [KjarniClassResults]),
    "kjarni_rerank_results_free_by_value": (None, [KjarniRerankResults]),
    "kjarni_search_results_free_by_value": (None, [KjarniSearchResults]),
    # kjarni_hip.h
    "kjarni_hip_whisper_load": (c_int32, [c_char_p, c_int32, POINTER(c_void_p)]),
    "kjarni_hip_whisper_free": (None, [c_void_p]),
    "kjarni_hip_whisper_dims": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "kjarni_hip_whisper_log_mel": (c_int32, [c_void_p, _f32p, c_size_t, _f32p]),
    "kjarni_hip_whisper_encode_mel": (c_int32, [c_void_p, _f32p, c_int32, _f32p]),
    "kjarni_hip_whisper_encode_audio": (c_int32, [c_void_p, _f32p, c_size_t, _f32p]),
    "kjarni_hip_whisper_decode_begin": (c_int32, [c_void_p]),
    "kjarni_hip_whisper_decode_forward": (c_int32, [c_void_p, _u32p, c_int32, _f32p, _f32p]),
    "kjarni_hip_whisper_greedy": (c_int32, [c_void_p, _u32p, c_int32, c_int32, c_size_t, _u32p, c_size_t, POINTER(c_size_t)]),
    "kjarni_hip_whisper_decode_text": (c_int32, [c_void_p, _u32p, c_size_t, c_int32, POINTER(c_void_p)]),
    "kjarni_audio_load_wav": (c_int32, [c_char_p, POINTER(KjarniFloatArray), POINTER(C.c_uint32)]),
    "kjarni_bytelevel_decode": (c_int32, [c_char_p, _u32p, c_size_t, c_int32, POINTER(c_void_p)]),
    "kjarni_chat_config_default": (KjarniChatConfig, []),
    "kjarni_generation_config_default": (KjarniGenerationConfig, []),
    "kjarni_chat_new": (c_int32, [POINTER(KjarniChatConfig), POINTER(c_void_p)]),
    "kjarni_chat_free": (None, [c_void_p]),
    "kjarni_chat_send": (c_int32, [c_void_p, c_char_p, POINTER(KjarniGenerationConfig), POINTER(c_void_p)]),
    "kjarni_chat_stream": (c_int32, [c_void_p, c_char_p, POINTER(KjarniGenerationConfig), KjarniStreamCallbackFn, c_void_p, c_void_p]),
    "kjarni_chat_send_with_history": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_char_p), c_size_t, c_char_p,
                                                POINTER(KjarniGenerationConfig), POINTER(c_void_p)]),
    "kjarni_chat_conversation_new": (c_int32, [c_void_p, POINTER(c_void_p)]),
    "kjarni_chat_conversation_free": (None, [c_void_p]),
    "kjarni_chat_conversation_send": (c_int32, [c_void_p, c_char_p, POINTER(KjarniGenerationConfig), POINTER(c_void_p)]),
    "kjarni_chat_conversation_stream": (c_int32, [c_void_p, c_char_p, POINTER(KjarniGenerationConfig), KjarniStreamCallbackFn,
                                                  c_void_p, c_void_p]),
    "kjarni_chat_conversation_len": (c_size_t, [c_void_p]),
    "kjarni_chat_conversation_clear": (None, [c_void_p, c_int32]),
    "kjarni_chat_model_name": (c_size_t, [c_void_p, c_char_p, c_size_t]),
    "kjarni_chat_context_size": (c_size_t, [c_void_p]),
    "kjarni_bpe_tokenizer_load": (c_int32, [c_char_p, POINTER(c_void_p)]),
    "kjarni_bpe_tokenizer_free": (None, [c_void_p]),
    "kjarni_bpe_tokenizer_encode": (c_int32, [c_void_p, c_char_p, c_size_t, _u32p, c_size_t, POINTER(c_size_t)]),
    "kjarni_bpe_tokenizer_decode": (c_int32, [c_void_p, _u32p, c_size_t, c_int32, POINTER(c_void_p)]),
    "kjarni_bpe_tokenizer_pre_tokenize": (c_int32, [c_void_p, c_char_p, POINTER(KjarniStringArray)]),
    "kjarni_chat_template_apply": (c_int32, [c_int32, POINTER(c_int32), POINTER(c_char_p), c_size_t, POINTER(c_void_p)]),
    "kjarni_sampling_distribution": (c_int32, [_f32p, c_size_t, c_float, c_int64, c_float, c_float, _f32p]),
    "kjarni_sampling_distribution_candidates": (c_int32, [_f32p, c_size_t, c_float, c_float, c_int64, c_float, c_float, _f32p,
                                                        POINTER(c_int32), POINTER(c_size_t)]),
    "kjarni_sample_from_probs": (C.c_uint32, [_f32p, c_size_t, c_float]),
    "kjarni_logits_process": (c_int32, [_f32p, c_size_t, _u32p, c_size_t, c_float, c_size_t]),
    "kjarni_generation_resolve": (c_int32, [c_char_p, c_size_t, c_char_p, c_int32, POINTER(KjarniGenerationConfig),
                                            POINTER(KjarniResolvedGeneration)]),
    "kjarni_hip_chat_resolve": (c_int32, [c_void_p, POINTER(KjarniGenerationConfig), POINTER(KjarniResolvedGeneration)]),
    "kjarni_hip_chat_format_prompt": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_char_p), c_size_t, c_char_p, POINTER(c_void_p)]),
    "kjarni_hip_chat_encode": (c_int32, [c_void_p, c_char_p, POINTER(KjarniGenerationConfig), _u32p, c_size_t, POINTER(c_size_t)]),
    "kjarni_hip_chat_seed": (None, [c_void_p, C.c_uint64]),
    "kjarni_hip_chat_set_device_sampling": (None, [c_void_p, c_int32]),
    "kjarni_hip_chat_sampling_counters": (None, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "kjarni_hip_decoder_load": (c_int32, [c_char_p, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "kjarni_hip_decoder_free": (None, [c_void_p]),
    "kjarni_hip_decoder_dims": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                          POINTER(c_int32), POINTER(C.c_uint64)]),
    "kjarni_hip_decoder_reset": (c_int32, [c_void_p]),
    "kjarni_hip_decoder_tile_gemm_calls": (c_uint64, [c_void_p]),
    "kjarni_hip_decoder_set_device_sampling": (None, [c_void_p, c_int32]),
    "kjarni_hip_decoder_forward": (c_int32, [c_void_p, _u32p, c_int32, _f32p, _f32p]),
    "kjarni_hip_decoder_generate": (c_int32, [c_void_p, _u32p, c_size_t, c_size_t, c_float, c_int32, KjarniTokenCallbackFn, c_void_p,
                                              _u32p, c_size_t, POINTER(c_size_t)]),
    "kjarni_text_split": (c_int32, [c_char_p, c_size_t, c_size_t, c_char_p, POINTER(KjarniStringArray)]),
    "kjarni_collect_files": (c_int32, [POINTER(KjarniIndexerConfig), POINTER(c_char_p), c_size_t,
                                       POINTER(KjarniStringArray)]),
    "kjarni_index_write": (c_int32, [c_char_p, c_size_t, c_size_t, c_char_p, POINTER(c_char_p), POINTER(c_char_p),
                                     _f32p, c_size_t, c_int32]),
    "kjarni_bm25_tokenize": (c_int32, [c_char_p, POINTER(KjarniStringArray)]),
    "kjarni_glob_match": (c_int32, [c_char_p, c_char_p]),
    "kjarni_rrf_fuse": (c_int32, [POINTER(c_size_t), c_size_t, POINTER(c_size_t), c_size_t, c_size_t,
                                  POINTER(c_size_t), _f32p, POINTER(c_size_t)]),
    "kjarni_hip_index_search": (c_int32, [c_char_p, c_char_p, _f32p, c_size_t, POINTER(KjarniSearchOptions),
                                          POINTER(KjarniSearchResults)]),
    "kjarni_hip_device_count": (c_int32, []),
    "kjarni_hip_encoder_load": (c_int32, [c_char_p, c_int32, POINTER(c_void_p)]),
    "kjarni_hip_encoder_free": (None, [c_void_p]),
    "kjarni_hip_encoder_hidden_size": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_num_layers": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_max_seq_len": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_vocab_size": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_num_labels": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_device": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_set_chunk_tokens": (c_int32, [c_void_p, c_int64]),
    "kjarni_hip_encoder_set_packing": (c_int32, [c_void_p, c_int32]),
    "kjarni_hip_encoder_set_combining": (c_int32, [c_void_p, c_int32]),
    "kjarni_hip_encoder_set_two_lanes": (c_int32, [c_void_p, c_int32]),
    "kjarni_hip_set_f32_on_bf16": (c_int32, [c_int32]),
    "kjarni_hip_get_f32_on_bf16": (c_int32, []),
    "kjarni_hip_selftest_reductions": (c_int32, [c_int32, C.c_uint32, C.c_uint32, POINTER(C.c_uint32)]),
    "kjarni_hip_clock_probe": (c_int32, [c_void_p, C.c_uint32, c_void_p]),
    "kjarni_hip_clock_trace": (c_int32, [c_void_p, C.c_uint32, C.c_uint32, c_void_p]),
    "kjarni_hip_measurement_stream": (c_void_p, []),
    "kjarni_hip_measurement_stream_release": (None, []),
    "kjarni_hip_encoder_hidden_states": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                   c_int32, c_void_p, c_void_p]),
    "kjarni_hip_encoder_embed": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                           c_int32, c_int32, c_void_p, c_void_p]),
    "kjarni_hip_encoder_logits": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                            c_void_p, c_void_p]),
    "kjarni_hip_encoder_hidden_states_host": (c_int32, [c_void_p, _u32p, _u32p, _u32p, c_int64, c_int32,
                                                        c_int32, _f32p]),
    "kjarni_hip_encoder_embed_host": (c_int32, [c_void_p, _u32p, _u32p, _u32p, c_int64, c_int32, c_int32,
                                                c_int32, c_int32, _f32p]),
    "kjarni_hip_encoder_logits_host": (c_int32, [c_void_p, _u32p, _u32p, _u32p, c_int64, c_int32, c_int32,
                                                 _f32p]),
    "kjarni_hip_op_linear": (c_int32, [c_int32, _f32p, _f32p, _f32p, _f32p, c_int64, c_int32, c_int32, c_int32,
                                       _f32p, c_int32, _f32p]),
    "kjarni_hip_op_linear_bf16_weights": (c_int32, [c_int32, _f32p, c_void_p, _f32p, _f32p, c_int64, c_int32, c_int32, c_int32,
                                                    _f32p, c_int32, _f32p]),
    "kjarni_hip_op_attention": (c_int32, [c_int32, _f32p, _u32p, c_int64, c_int32, c_int32, c_int32, c_float,
                                          _f32p, c_int32, _f32p]),
    "kjarni_hip_op_pool": (c_int32, [c_int32, _f32p, _u32p, c_int64, c_int32, c_int32, c_int32, c_int32, _f32p]),
    "kjarni_hip_op_attention_biased": (c_int32, [c_int32, _f32p, _u32p, _f32p, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32,
                                                 c_float, _f32p]),
    "kjarni_hip_op_layer_norm": (c_int32, [c_int32, _f32p, _f32p, _f32p, c_float, c_int64, c_int32, _f32p,
                                           c_int32, _f32p]),
    "kjarni_hip_op_linear_layer_norm": (c_int32, [c_int32, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, c_float, c_int64,
                                                  c_int32, c_int32, _f32p, c_int32, _f32p]),
    "kjarni_hip_group_load": (c_int32, [c_char_p, POINTER(c_int32), c_size_t, POINTER(c_void_p)]),
    "kjarni_hip_group_free": (None, [c_void_p]),
    "kjarni_hip_group_size": (c_size_t, [c_void_p]),
    "kjarni_hip_group_device": (c_int32, [c_void_p, c_size_t]),
    "kjarni_hip_group_hidden_size": (c_int32, [c_void_p]),
    "kjarni_hip_group_num_labels": (c_int32, [c_void_p]),
    "kjarni_hip_group_shard": (c_int32, [c_void_p, c_int64, c_size_t, POINTER(c_int64), POINTER(c_int64)]),
    "kjarni_hip_group_gather_plan": (c_int32, [c_int64, c_size_t, c_int64, c_void_p, c_size_t, POINTER(c_size_t)]),
    "kjarni_hip_group_embed_host": (c_int32, [c_void_p, _u32p, _u32p, _u32p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                              _f32p]),
    "kjarni_hip_group_logits_host": (c_int32, [c_void_p, _u32p, _u32p, _u32p, c_int64, c_int32, c_int32, _f32p]),
    "kjarni_hip_group_embed_allgather": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                   c_int64, c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "kjarni_hip_group_logits_allgather": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                    c_int64, c_int32, c_int32, POINTER(c_void_p)]),
    "kjarni_hip_group_transport": (c_char_p, [c_void_p]),
    "kjarni_hip_encoder_profile_begin": (c_int32, [c_void_p]),
    "kjarni_hip_encoder_profile_begin_kinds": (c_int32, [c_void_p, C.c_uint32]),
    "kjarni_hip_encoder_profile_end": (c_int32, [c_void_p, POINTER(KjarniHipKernelStat), c_size_t,
                                                 POINTER(c_size_t)]),
    "kjarni_hip_cosine_scores": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_int32,
                                           c_void_p, c_void_p]),
    "kjarni_hip_cosine_topk_workspace_bytes": (c_size_t, [c_int32, c_int64, c_int32]),
    "kjarni_hip_cosine_topk": (c_int32, [c_int32, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "kjarni_hip_search_breakdown": (None, [POINTER(C.c_double), c_size_t]),
    "kjarni_hip_set_keyword_parallel_min_docs": (None, [c_size_t]),
    "kjarni_hip_cosine_search_workspace_bytes": (c_size_t, [c_int32, c_int64, c_int32, c_int32]),
    "kjarni_hip_cosine_search": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "kjarni_hip_cosine_search_host": (c_int32, [c_int32, _f32p, c_int32, _f32p, c_int64, c_int32, c_int32,
                                                c_int32, _i64p, _f32p, _i64p]),
    "kjarni_tokenizer_load": (c_int32, [c_char_p, c_size_t, POINTER(c_void_p)]),
    "kjarni_tokenizer_free": (None, [c_void_p]),
    "kjarni_tokenizer_encode_batch": (c_int32, [c_void_p, POINTER(c_char_p), POINTER(c_char_p), c_size_t,
                                                POINTER(KjarniTokenBatch)]),
    "kjarni_token_batch_free": (None, [POINTER(KjarniTokenBatch)]),
    "kjarni_hip_malloc": (c_int32, [c_int32, c_size_t, POINTER(c_void_p)]),
    "kjarni_hip_free": (c_int32, [c_int32, c_void_p]),
    "kjarni_hip_memcpy_h2d": (c_int32, [c_int32, c_void_p, c_void_p, c_size_t]),
    "kjarni_hip_memcpy_d2h": (c_int32, [c_int32, c_void_p, c_void_p, c_size_t]),
    "kjarni_hip_synchronize": (c_int32, [c_int32]),
}

_lib = None


def lib():
    """Load the native library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C kjarni_amd/csrc` "
                "(or __graft_entry__.build()).  There is no fallback implementation.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(L, name):
                continue  # reported by tests/test_abi.py; partial builds stay importable
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def error_name(code: int) -> str:
    return lib().kjarni_error_name(int(code)).decode()


def last_error() -> str:
    msg = lib().kjarni_last_error_message()
    return msg.decode("utf-8", "replace") if msg else ""


def check_error(code: int):
    if code != KjarniError.OK:
        raise KjarniException(int(code), last_error())
