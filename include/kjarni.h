/*
 * kjarni.h -- the kjarni-ffi C ABI, served by the MI355X encoder core.
 *
 * Drop-in surface for the reference's `libkjarni_ffi.so`
 * (crates/kjarni-ffi/Cargo.toml:7-8).  Every declaration below names the Rust
 * `extern "C"` item it replaces; struct layouts are the reference's
 * `#[repr(C)]` layouts byte for byte.  Authoritative source is the Rust code,
 * not the stale cbindgen header (crates/kjarni-ffi/include/kjarni.h): in
 * particular every `*_free` takes a POINTER to the struct, as the Rust source
 * (kjarni-ffi/src/lib.rs:131-174, reranker.rs:51-57, classifier.rs:55-67) and
 * the Go binding (bindings/go/embedder.go:227-234) do.
 *
 * Behavioural differences from the reference, by design:
 *   - inference always runs on an AMD GPU through hand-written HIP kernels;
 *     `device` (Cpu/Gpu) is accepted for layout compatibility and both values
 *     select the GPU.  Without a usable HIP device `*_new` returns
 *     KJARNI_ERROR_GPU_UNAVAILABLE -- there is no CPU fallback.
 *   - models are never downloaded (no network): a registry name must already
 *     be present under <cache>/<org>_<repo>/, otherwise
 *     KJARNI_ERROR_MODEL_NOT_FOUND.  `model_path` is honoured by all three
 *     components (the reference's Embedder ignores it, kjarni/src/embedder/
 *     model.rs:51-116).
 */
#ifndef KJARNI_H
#define KJARNI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- errors: kjarni-ffi/src/error.rs:7-100 ---------------------------------- */
typedef enum KjarniErrorCode {
    KJARNI_OK = 0,
    KJARNI_ERROR_NULL_POINTER = 1,
    KJARNI_ERROR_INVALID_UTF8 = 2,
    KJARNI_ERROR_MODEL_NOT_FOUND = 3,
    KJARNI_ERROR_LOAD_FAILED = 4,
    KJARNI_ERROR_INFERENCE_FAILED = 5,
    KJARNI_ERROR_GPU_UNAVAILABLE = 6,
    KJARNI_ERROR_INVALID_CONFIG = 7,
    KJARNI_ERROR_CANCELLED = 8,
    KJARNI_ERROR_TIMEOUT = 9,
    KJARNI_ERROR_STREAM_ENDED = 10,
    KJARNI_ERROR_UNKNOWN = 255,
} KjarniErrorCode;

/* error.rs:63-85: static "KJARNI_OK" / "KJARNI_ERROR_*" names. */
const char* kjarni_error_name(KjarniErrorCode err);
const char* kjarni_error_code_to_string(KjarniErrorCode err); /* error.rs:56-61 */
/* error.rs:87-100: thread-local; valid until the next failing call or clear on
 * this thread; NULL when none. */
const char* kjarni_last_error_message(void);
void kjarni_clear_error(void);

/* ---- runtime: kjarni-ffi/src/lib.rs:35-57 ----------------------------------- */
KjarniErrorCode kjarni_init(void);   /* optional, idempotent */
void kjarni_shutdown(void);          /* no-op, as in the reference */
const char* kjarni_version(void);    /* static "0.1.0" */

/* ---- arrays: kjarni-ffi/src/lib.rs:59-174 -----------------------------------
 * The library allocates, the caller frees exactly once with the matching free.
 * Empty = {NULL, 0[, 0]}. */
typedef struct KjarniFloatArray {
    float* data;
    size_t len;
} KjarniFloatArray;

typedef struct KjarniFloat2DArray {
    float* data; /* row-major */
    size_t rows;
    size_t cols;
} KjarniFloat2DArray;

typedef struct KjarniStringArray {
    char** strings;
    size_t len;
} KjarniStringArray;

void kjarni_float_array_free(const KjarniFloatArray* arr);       /* lib.rs:131-138 */
void kjarni_float_2d_array_free(const KjarniFloat2DArray* arr);  /* lib.rs:141-149 */
void kjarni_string_free(char* s);                                /* lib.rs:152-159 */
void kjarni_string_array_free(const KjarniStringArray* arr);     /* lib.rs:162-174 */

/* lib.rs:176-188 -> kjarni/src/embedder/model.rs:247-257; 0.0 on NULL / len 0. */
float kjarni_cosine_similarity(const float* a, const float* b, size_t len);

/* ---- device: kjarni-ffi/src/embedder.rs:13-18 ------------------------------- */
typedef enum KjarniDevice {
    KJARNI_DEVICE_CPU = 0,
    KJARNI_DEVICE_GPU = 1,
} KjarniDevice;

/* ---- Embedder: kjarni-ffi/src/embedder.rs:20-275 ---------------------------- */
typedef struct KjarniEmbedderConfig {
    KjarniDevice device;
    const char* cache_dir;  /* NULL = default cache */
    const char* model_name; /* NULL = "minilm-l6-v2" */
    const char* model_path; /* NULL = registry */
    int32_t normalize;      /* honoured by encode / similarity, not by encode_batch */
    int32_t quiet;
} KjarniEmbedderConfig;

typedef struct KjarniEmbedder KjarniEmbedder;

KjarniEmbedderConfig kjarni_embedder_config_default(void);                          /* :39-48 */
KjarniErrorCode kjarni_embedder_new(const KjarniEmbedderConfig* config, KjarniEmbedder** out); /* :56-127 */
void kjarni_embedder_free(KjarniEmbedder* embedder);                                /* :130-135 */
KjarniErrorCode kjarni_embedder_encode(KjarniEmbedder* embedder, const char* text,
                                       KjarniFloatArray* out);                      /* :138-171 */
KjarniErrorCode kjarni_embedder_encode_batch(KjarniEmbedder* embedder, const char* const* texts,
                                             size_t num_texts, KjarniFloat2DArray* out); /* :174-224 */
KjarniErrorCode kjarni_embedder_similarity(KjarniEmbedder* embedder, const char* text1,
                                           const char* text2, float* out);          /* :227-262 */
size_t kjarni_embedder_dim(const KjarniEmbedder* embedder);                         /* :265-275 */

/* ---- Classifier: kjarni-ffi/src/classifier.rs:10-297 ------------------------ */
typedef struct KjarniClassResult {
    char* label;
    float score;
} KjarniClassResult;

typedef struct KjarniClassResults {
    KjarniClassResult* results;
    size_t len;
} KjarniClassResults;

typedef struct KjarniClassifierConfig {
    KjarniDevice device;
    const char* cache_dir;
    const char* model_name; /* NULL = "sentiment" */
    const char* model_path;
    const char* const* labels; /* NULL = model labels */
    size_t num_labels;
    int32_t multi_label;
    int32_t quiet;
} KjarniClassifierConfig;

typedef struct KjarniClassifier KjarniClassifier;

void kjarni_class_results_free(const KjarniClassResults* results);                  /* :55-67 */
KjarniClassifierConfig kjarni_classifier_config_default(void);                      /* :91-103 */
KjarniErrorCode kjarni_classifier_new(const KjarniClassifierConfig* config, KjarniClassifier** out);
void kjarni_classifier_free(KjarniClassifier* classifier);
KjarniErrorCode kjarni_classifier_classify(KjarniClassifier* classifier, const char* text,
                                           KjarniClassResults* out);               /* :226-256 */
KjarniErrorCode kjarni_classifier_labels(const KjarniClassifier* classifier, KjarniStringArray* out);
size_t kjarni_classifier_num_labels(const KjarniClassifier* classifier);

/* ---- Reranker: kjarni-ffi/src/reranker.rs:10-322 ---------------------------- */
typedef struct KjarniRerankResult {
    size_t index; /* position in the input array */
    float score;
} KjarniRerankResult;

typedef struct KjarniRerankResults {
    KjarniRerankResult* results;
    size_t len;
} KjarniRerankResults;

typedef struct KjarniRerankerConfig {
    KjarniDevice device;
    const char* cache_dir;
    const char* model_name; /* NULL = "minilm-l6-v2-cross-encoder" */
    const char* model_path;
    int32_t quiet;
} KjarniRerankerConfig;

typedef struct KjarniReranker KjarniReranker;

void kjarni_rerank_results_free(const KjarniRerankResults* results);                /* :51-57 */
KjarniRerankerConfig kjarni_reranker_config_default(void);                          /* :70-79 */
KjarniErrorCode kjarni_reranker_new(const KjarniRerankerConfig* config, KjarniReranker** out); /* :88-163 */
void kjarni_reranker_free(KjarniReranker* reranker);                                /* :166-171 */
KjarniErrorCode kjarni_reranker_score(KjarniReranker* reranker, const char* query,
                                      const char* document, float* out);            /* :174-212 */
KjarniErrorCode kjarni_reranker_rerank(KjarniReranker* reranker, const char* query,
                                       const char* const* documents, size_t num_docs,
                                       KjarniRerankResults* out);                   /* :215-269 */
KjarniErrorCode kjarni_reranker_rerank_top_k(KjarniReranker* reranker, const char* query,
                                             const char* const* documents, size_t num_docs,
                                             size_t top_k, KjarniRerankResults* out); /* :272-322 */

/* ---- Searcher: kjarni-ffi/src/searcher.rs:12-491 ------------------------------
 * The index is the reference's segmented on-disk layout (crates/kjarni-rag/src/segment.rs,
 * index_reader.rs); the semantic scan runs on the GPU, BM25 / rank fusion / filters on the host. */
typedef enum KjarniSearchMode {
    KJARNI_SEARCH_KEYWORD = 0,
    KJARNI_SEARCH_SEMANTIC = 1,
    KJARNI_SEARCH_HYBRID = 2,
} KjarniSearchMode;

typedef struct KjarniSearchResult {
    float score;
    size_t document_id;
    char* text;
    char* metadata_json;
} KjarniSearchResult;

typedef struct KjarniSearchResults {
    KjarniSearchResult* results;
    size_t len;
} KjarniSearchResults;

/* Sentinels mean "use the searcher's default": mode -1, top_k 0, use_reranker -1, threshold 0.0. */
typedef struct KjarniSearchOptions {
    int32_t mode;
    size_t top_k;
    int32_t use_reranker;
    float threshold;
    const char* source_pattern; /* glob on the "source" metadata value */
    const char* filter_key;     /* metadata key that must equal filter_value */
    const char* filter_value;
} KjarniSearchOptions;

typedef struct KjarniSearcherConfig {
    KjarniDevice device;
    const char* cache_dir;
    const char* model_name;   /* NULL = "minilm-l6-v2" */
    const char* rerank_model; /* NULL = no reranker */
    KjarniSearchMode default_mode; /* default Hybrid */
    size_t default_top_k;          /* default 10 */
    int32_t quiet;
} KjarniSearcherConfig;

typedef struct KjarniSearcher KjarniSearcher;

void kjarni_search_results_free(const KjarniSearchResults* results);                 /* :99-117 */
KjarniSearchOptions kjarni_search_options_default(void);                             /* :129-140 */
KjarniSearcherConfig kjarni_searcher_config_default(void);                           /* :154-165 */
KjarniErrorCode kjarni_searcher_new(const KjarniSearcherConfig* config, KjarniSearcher** out); /* :172-241 */
void kjarni_searcher_free(KjarniSearcher* searcher);                                 /* :243-250 */
KjarniErrorCode kjarni_searcher_search(KjarniSearcher* searcher, const char* index_path, const char* query,
                                       KjarniSearchResults* out);                    /* :253-262 */
KjarniErrorCode kjarni_searcher_search_with_options(KjarniSearcher* searcher, const char* index_path,
                                                    const char* query, const KjarniSearchOptions* options,
                                                    KjarniSearchResults* out);       /* :265-360 */
/* BM25 only; needs neither a model nor a GPU. */
KjarniErrorCode kjarni_search_keywords(const char* index_path, const char* query, size_t top_k,
                                       KjarniSearchResults* out);                    /* :363-395 */
bool kjarni_searcher_has_reranker(const KjarniSearcher* searcher);                   /* :398-404 */
KjarniSearchMode kjarni_searcher_default_mode(const KjarniSearcher* searcher);       /* :407-415 */
size_t kjarni_searcher_default_top_k(const KjarniSearcher* searcher);                /* :418-424 */
/* Copy the name into buf (NUL-terminated, truncated to buf_len); return the size needed incl. NUL. */
size_t kjarni_searcher_model_name(const KjarniSearcher* searcher, char* buf, size_t buf_len);     /* :427-457 */
size_t kjarni_searcher_reranker_model(const KjarniSearcher* searcher, char* buf, size_t buf_len); /* :460-491 */

/* ---- progress + cancellation: kjarni-ffi/src/callback.rs:7-101 ---------------- */
typedef enum KjarniProgressStage {
    KJARNI_PROGRESS_SCANNING = 0,
    KJARNI_PROGRESS_LOADING = 1,
    KJARNI_PROGRESS_EMBEDDING = 2,
    KJARNI_PROGRESS_WRITING = 3,
    KJARNI_PROGRESS_COMMITTING = 4,
    KJARNI_PROGRESS_SEARCHING = 5,
    KJARNI_PROGRESS_RERANKING = 6,
} KjarniProgressStage;

typedef struct KjarniProgress {
    KjarniProgressStage stage;
    size_t current;
    size_t total;        /* 0 = unknown */
    const char* message; /* may be NULL; valid for the duration of the callback */
} KjarniProgress;

/* The struct is passed BY VALUE (callback.rs:32-33). */
typedef void (*KjarniProgressCallbackFn)(KjarniProgress progress, void* user_data);

typedef struct KjarniCancelToken KjarniCancelToken;
KjarniCancelToken* kjarni_cancel_token_new(void);                          /* :53-58 */
void kjarni_cancel_token_cancel(KjarniCancelToken* token);                 /* :60-67; may be called from any thread */
bool kjarni_cancel_token_is_cancelled(const KjarniCancelToken* token);     /* :69-76; false on NULL */
void kjarni_cancel_token_reset(KjarniCancelToken* token);                  /* :78-85 */
void kjarni_cancel_token_free(KjarniCancelToken* token);                   /* :87-94 */

/* ---- Indexer: kjarni-ffi/src/indexer.rs:14-675 ---------------------------------
 * Files and directories -> chunks (crates/kjarni-rag/src/{loader,splitter}.rs) -> embeddings on the
 * GPU -> the reference's segmented on-disk index (crates/kjarni-rag/src/{index_writer,segment}.rs). */
typedef struct KjarniIndexStats {
    size_t documents_indexed;
    size_t chunks_created;
    size_t dimension;
    uint64_t size_bytes;
    size_t files_processed;
    size_t files_skipped;
    uint64_t elapsed_ms;
} KjarniIndexStats;

typedef struct KjarniIndexInfo {
    char* path;            /* owned; free with kjarni_index_info_free */
    size_t document_count;
    size_t segment_count;
    size_t dimension;
    uint64_t size_bytes;
    char* embedding_model; /* owned, may be NULL */
} KjarniIndexInfo;

typedef struct KjarniIndexerConfig {
    KjarniDevice device;
    const char* cache_dir;
    const char* model_name;       /* NULL = "minilm-l6-v2" */
    size_t chunk_size;            /* characters; default 512 */
    size_t chunk_overlap;         /* characters; default 50 */
    size_t batch_size;            /* chunks per embed/progress step; default 32 */
    const char* extensions;       /* comma separated; NULL = the default text extensions */
    const char* exclude_patterns; /* comma separated globs */
    int32_t recursive;            /* default 1 */
    int32_t include_hidden;       /* default 0 */
    size_t max_file_size;         /* bytes; default 10 MiB; 0 = keep the default */
    int32_t quiet;
} KjarniIndexerConfig;

typedef struct KjarniIndexer KjarniIndexer;

/* By VALUE, as the Rust source has it (indexer.rs:85-93). */
void kjarni_index_info_free(KjarniIndexInfo info);
KjarniIndexerConfig kjarni_indexer_config_default(void);                              /* :126-143 */
KjarniErrorCode kjarni_indexer_new(const KjarniIndexerConfig* config, KjarniIndexer** out); /* :153-243 */
void kjarni_indexer_free(KjarniIndexer* indexer);                                     /* :246-251 */
/* Errors: index exists without force / no inputs / dimension mismatch -> INVALID_CONFIG;
 * missing input path or index -> MODEL_NOT_FOUND; cancelled -> CANCELLED; else INFERENCE_FAILED. */
KjarniErrorCode kjarni_indexer_create(KjarniIndexer* indexer, const char* index_path, const char* const* inputs,
                                      size_t num_inputs, int32_t force, KjarniIndexStats* out);      /* :301-344 */
KjarniErrorCode kjarni_indexer_create_with_callback(KjarniIndexer* indexer, const char* index_path,
                                                    const char* const* inputs, size_t num_inputs, int32_t force,
                                                    KjarniProgressCallbackFn progress_callback, void* user_data,
                                                    const KjarniCancelToken* cancel_token,
                                                    KjarniIndexStats* out);                          /* :347-433 */
KjarniErrorCode kjarni_indexer_add(KjarniIndexer* indexer, const char* index_path, const char* const* inputs,
                                   size_t num_inputs, size_t* documents_added);                      /* :437-483 */
KjarniErrorCode kjarni_indexer_add_with_callback(KjarniIndexer* indexer, const char* index_path,
                                                 const char* const* inputs, size_t num_inputs,
                                                 KjarniProgressCallbackFn progress_callback, void* user_data,
                                                 const KjarniCancelToken* cancel_token,
                                                 size_t* documents_added);                           /* :486-578 */
/* Neither needs a model or a GPU. */
KjarniErrorCode kjarni_index_info(const char* index_path, KjarniIndexInfo* out);      /* :582-604 */
KjarniErrorCode kjarni_index_delete(const char* index_path);                          /* :607-625 */
size_t kjarni_indexer_model_name(const KjarniIndexer* indexer, char* buf, size_t buf_len); /* :629-657 */
size_t kjarni_indexer_dimension(const KjarniIndexer* indexer);                        /* :660-666 */
size_t kjarni_indexer_chunk_size(const KjarniIndexer* indexer);                       /* :669-675 */

/* ---- Chat (decoder-only LLMs): kjarni-ffi/src/chat.rs:13-758 -----------------------------------
 * model_name is a registry name ("llama3.2-1b-instruct", "qwen2.5-0.5b-instruct", ...): it selects the
 * architecture, the chat template (Llama 3 / ChatML / Mistral [INST]) and the directory <cache_dir>/<org>_<repo>.  The reference
 * never reads model_path (chat.rs:190-250); here a non-NULL model_path replaces that directory (config.json,
 * tokenizer.json, model.safetensors, optional generation_config.json), everything else still follows model_name.
 * Both device values run on the MI355X; KJARNI_HIP_DEVICE picks the ordinal, KJARNI_HIP_CHAT_CONTEXT the KV-cache
 * capacity in tokens (default 32768, the model's max_position_embeddings is what kjarni_chat_context_size reports).
 * Errors of kjarni_chat_new (chat.rs:126-139): unknown name / files not on disk -> MODEL_NOT_FOUND; encoder, seq2seq,
 * speech or template-less model -> INVALID_CONFIG; no GPU -> GPU_UNAVAILABLE; anything else -> LOAD_FAILED
 * (including Phi-3, which the reference cannot load either). */
typedef struct KjarniChatConfig {      /* chat.rs:13-30 */
    KjarniDevice device;
    const char* cache_dir;     /* NULL = default cache */
    const char* model_name;    /* required */
    const char* model_path;    /* NULL = <cache_dir>/<org>_<repo> */
    const char* system_prompt; /* NULL = the template's default system prompt */
    int32_t mode;              /* 0 default (temperature 0.7, 512 new tokens), 1 creative (0.9, 1024), 2 reasoning (0.3, 2048) */
    int32_t quiet;
} KjarniChatConfig;

typedef struct KjarniGenerationConfig { /* chat.rs:32-49; negative = keep the resolved default */
    float temperature;
    int32_t top_k;
    float top_p;
    float min_p;
    float repetition_penalty;
    int32_t max_new_tokens;
    int32_t do_sample;         /* -1 default, 0 greedy, 1 sample */
} KjarniGenerationConfig;

/* Receives each generated token's text; return false to stop. */
typedef bool (*KjarniStreamCallbackFn)(const char* text, void* user_data);

typedef struct KjarniChat KjarniChat;
typedef struct KjarniChatConversation KjarniChatConversation;

KjarniChatConfig kjarni_chat_config_default(void);                                   /* chat.rs:56-67 */
KjarniGenerationConfig kjarni_generation_config_default(void);                       /* :70-81 */
KjarniErrorCode kjarni_chat_new(const KjarniChatConfig* config, KjarniChat** out);   /* :174-250 */
void kjarni_chat_free(KjarniChat* chat);                                             /* :253-258 */
/* One-shot message with the configured (or the template's default) system prompt; the reply is trimmed and a
 * trailing template stop sequence is stripped (crates/kjarni/src/chat/model.rs:283-303).  Free with kjarni_string_free. */
KjarniErrorCode kjarni_chat_send(KjarniChat* chat, const char* message, const KjarniGenerationConfig* gen_config,
                                 char** out);                                         /* :271-322 */
/* The same, one callback per generated token (single-token decode, special tokens included); the cancel token is
 * looked at before each callback. */
KjarniErrorCode kjarni_chat_stream(KjarniChat* chat, const char* message, const KjarniGenerationConfig* gen_config,
                                   KjarniStreamCallbackFn callback, void* user_data,
                                   const KjarniCancelToken* cancel_token);            /* :338-401 */
/* roles[i]: 0 system (restarts the history), 1 user, 2 assistant; anything else -> INVALID_CONFIG. */
KjarniErrorCode kjarni_chat_send_with_history(KjarniChat* chat, const int32_t* roles, const char* const* contents,
                                              size_t history_len, const char* message,
                                              const KjarniGenerationConfig* gen_config, char** out); /* :403-484 */
/* Stateful conversation: the handle owns the history and must not outlive its chat. */
KjarniErrorCode kjarni_chat_conversation_new(KjarniChat* chat, KjarniChatConversation** out);        /* :487-511 */
void kjarni_chat_conversation_free(KjarniChatConversation* convo);                                   /* :514-519 */
KjarniErrorCode kjarni_chat_conversation_send(KjarniChatConversation* convo, const char* message,
                                              const KjarniGenerationConfig* gen_config, char** out); /* :521-583 */
KjarniErrorCode kjarni_chat_conversation_stream(KjarniChatConversation* convo, const char* message,
                                                const KjarniGenerationConfig* gen_config,
                                                KjarniStreamCallbackFn callback, void* user_data,
                                                const KjarniCancelToken* cancel_token);              /* :589-693 */
size_t kjarni_chat_conversation_len(const KjarniChatConversation* convo);                            /* :699-707 */
void kjarni_chat_conversation_clear(KjarniChatConversation* convo, int32_t keep_system);             /* :709-717 */
/* Without a buffer: the name's byte length; with one: bytes copied, NUL excluded (chat.rs:723-745). */
size_t kjarni_chat_model_name(const KjarniChat* chat, char* buf, size_t buf_len);
size_t kjarni_chat_context_size(const KjarniChat* chat);                                             /* :747-758 */

/* ---- streamed tokens: kjarni-ffi/src/callback.rs:36-47 -------------------------------- */
typedef struct KjarniToken {
    const char* text; /* valid for the duration of the callback */
    uint32_t token_id;
    bool is_special;
} KjarniToken;

/* Return false to stop the stream.  The struct is passed BY VALUE. */
typedef bool (*KjarniTokenCallbackFn)(KjarniToken token, void* user_data);

/* ---- Transcriber (Whisper) ---------------------------------------------------------------
 * NOT part of the reference's kjarni-ffi crate: the reference exposes transcription only through its
 * Rust API (crates/kjarni/src/transcriber/{builder,model,types,validation}.rs).  This group mirrors that
 * API in the style of the groups above so the C#/Go/Python front-ends can reach it the same way.
 * Pipeline per 30-second chunk (crates/kjarni-models/src/models/whisper/transcriber.rs:85-460):
 * log-mel -> conv stem -> encoder -> greedy decode, all on the GPU; WAV decoding, resampling, chunking,
 * timestamp parsing and stitching on the host. */
typedef enum KjarniTranscribeTask {
    KJARNI_TASK_TRANSCRIBE = 0,
    KJARNI_TASK_TRANSLATE = 1,
} KjarniTranscribeTask;

typedef struct KjarniTranscriberConfig {
    KjarniDevice device;
    const char* cache_dir;
    const char* model_name; /* NULL = "whisper-small"; builder.rs:145-159 names plus tiny/base/medium */
    const char* model_path; /* local directory with config.json, tokenizer.json, model.safetensors */
    const char* language;   /* NULL = auto (stubbed as English, transcriber.rs:277-279) */
    KjarniTranscribeTask task;
    int32_t timestamps;     /* parse timestamp tokens into timed segments */
    size_t max_tokens_per_chunk; /* default 448; 1..4096 (validation.rs:52-64) */
    int32_t quiet;
} KjarniTranscriberConfig;

typedef struct KjarniTranscriptionSegment {
    float start; /* seconds */
    float end;
    char* text;
} KjarniTranscriptionSegment;

typedef struct KjarniTranscription {       /* types.rs:34-55 */
    char* text;
    KjarniTranscriptionSegment* segments;
    size_t num_segments;
    char* language;
    float duration_secs;
} KjarniTranscription;

typedef enum KjarniTranscriptionStage {    /* types.rs:78-88 */
    KJARNI_TRANSCRIPTION_LOADING_AUDIO = 0,
    KJARNI_TRANSCRIPTION_ENCODING = 1,
    KJARNI_TRANSCRIPTION_DECODING = 2,
    KJARNI_TRANSCRIPTION_STITCHING = 3,
} KjarniTranscriptionStage;

typedef struct KjarniTranscriptionProgress {
    KjarniTranscriptionStage stage;
    size_t current;
    size_t total;        /* 0 = unknown */
    const char* message; /* may be NULL */
} KjarniTranscriptionProgress;

typedef void (*KjarniTranscriptionProgressFn)(KjarniTranscriptionProgress progress, void* user_data);

typedef struct KjarniTranscriber KjarniTranscriber;

KjarniTranscriberConfig kjarni_transcriber_config_default(void);
/* Errors: unknown model / bad language / bad max_tokens -> INVALID_CONFIG (builder.rs:118-127); files not on
 * disk -> MODEL_NOT_FOUND; no GPU -> GPU_UNAVAILABLE; anything else -> LOAD_FAILED. */
KjarniErrorCode kjarni_transcriber_new(const KjarniTranscriberConfig* config, KjarniTranscriber** out);
void kjarni_transcriber_free(KjarniTranscriber* transcriber);
void kjarni_transcription_free(const KjarniTranscription* transcription);
/* Transcriber::transcribe_audio (model.rs:91-106): mono f32 samples, resampled linearly to 16 kHz when needed. */
KjarniErrorCode kjarni_transcriber_transcribe_audio(KjarniTranscriber* transcriber, const float* samples, size_t num_samples,
                                                    uint32_t sample_rate, KjarniTranscription* out);
/* Transcriber::transcribe_file (model.rs:72-88): WAV (PCM 8/16/24/32, float 32); missing / not a file ->
 * MODEL_NOT_FOUND, unsupported extension -> INVALID_CONFIG, unreadable audio -> INFERENCE_FAILED. */
KjarniErrorCode kjarni_transcriber_transcribe_file(KjarniTranscriber* transcriber, const char* path, KjarniTranscription* out);
/* The same with the progress callback (model.rs:301-318), the per-token stream of stream_audio (model.rs:196-262;
 * on_token returning false ends the stream, the text so far is still returned) and a cancel token -> CANCELLED. */
KjarniErrorCode kjarni_transcriber_transcribe_audio_with_callbacks(
    KjarniTranscriber* transcriber, const float* samples, size_t num_samples, uint32_t sample_rate,
    KjarniTranscriptionProgressFn progress, void* progress_user_data, KjarniTokenCallbackFn on_token, void* token_user_data,
    const KjarniCancelToken* cancel_token, KjarniTranscription* out);
KjarniErrorCode kjarni_transcriber_transcribe_file_with_callbacks(
    KjarniTranscriber* transcriber, const char* path, KjarniTranscriptionProgressFn progress, void* progress_user_data,
    KjarniTokenCallbackFn on_token, void* token_user_data, const KjarniCancelToken* cancel_token, KjarniTranscription* out);
size_t kjarni_transcriber_model_name(const KjarniTranscriber* transcriber, char* buf, size_t buf_len);

/* ---- by-value twins of the frees ------------------------------------------------------
 * The reference's committed cbindgen header and its C# / Python bindings pass these structs BY VALUE
 * (crates/kjarni-ffi/include/kjarni.h:440-455, 534, 581, 734; bindings/csharp/Kjarni/Native.cs:376-394),
 * the Rust source by pointer.  The canonical names above follow the Rust source; a binding written
 * against the header binds these instead and keeps its declarations. */
void kjarni_float_array_free_by_value(KjarniFloatArray arr);
void kjarni_float_2d_array_free_by_value(KjarniFloat2DArray arr);
void kjarni_string_array_free_by_value(KjarniStringArray arr);
void kjarni_class_results_free_by_value(KjarniClassResults results);
void kjarni_rerank_results_free_by_value(KjarniRerankResults results);
void kjarni_search_results_free_by_value(KjarniSearchResults results);

#ifdef __cplusplus
}
#endif

#endif /* KJARNI_H */
