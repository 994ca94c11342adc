"""C-ABI checks that need no GPU: the library loads, exports every symbol the
headers declare, mirrors the reference's #[repr(C)] layouts and its
error/NULL/UTF-8 behaviour (crates/kjarni-ffi/src/lib.rs:190-330, error.rs:102-200)."""
import ctypes as C
import os
import re
import threading

import numpy as np
import pytest

import kjarni_amd
from kjarni_amd import _ffi
from kjarni_amd._ffi import KjarniError as E
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = kjarni_amd.lib()


def _declared():
    names = set()
    for h in ("kjarni.h", "kjarni_hip.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(kjarni_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared()
    assert len(declared) > 120
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    unbound = sorted(declared - set(_ffi.SIGNATURES))
    assert not unbound, f"declared but missing from kjarni_amd._ffi.SIGNATURES: {unbound}"
    extra = sorted(set(_ffi.SIGNATURES) - declared)
    assert not extra, f"bound in _ffi.py but not declared in a header: {extra}"


def test_struct_layouts_match_reference_repr_c():
    # embedder.rs:20-35 (40 bytes, pointers 8-aligned; Go mirror ffi.go:34-42)
    assert C.sizeof(_ffi.KjarniEmbedderConfig) == 40
    assert _ffi.KjarniEmbedderConfig.cache_dir.offset == 8
    assert _ffi.KjarniEmbedderConfig.normalize.offset == 32 and _ffi.KjarniEmbedderConfig.quiet.offset == 36
    assert C.sizeof(_ffi.KjarniRerankerConfig) == 40          # reranker.rs:60-67
    assert C.sizeof(_ffi.KjarniClassifierConfig) == 56        # classifier.rs:69-88
    assert C.sizeof(_ffi.KjarniRerankResult) == 16            # reranker.rs:11-15
    assert C.sizeof(_ffi.KjarniClassResult) == 16             # classifier.rs:11-15
    assert C.sizeof(_ffi.KjarniFloatArray) == 16 and C.sizeof(_ffi.KjarniFloat2DArray) == 24  # lib.rs:59-84


def test_defaults():
    c = L.kjarni_embedder_config_default()   # embedder.rs:39-48
    assert (c.device, c.cache_dir, c.model_name, c.model_path, c.normalize, c.quiet) == (0, None, None, None, 1, 0)
    r = L.kjarni_reranker_config_default()   # reranker.rs:70-79
    assert (r.device, r.cache_dir, r.model_name, r.model_path, r.quiet) == (0, None, None, None, 0)
    k = L.kjarni_classifier_config_default()  # classifier.rs:91-103
    assert (k.device, k.num_labels, k.multi_label, k.quiet) == (0, 0, 0, 0) and not k.labels


def test_runtime_and_error_names():
    assert L.kjarni_init() == E.OK and L.kjarni_init() == E.OK   # idempotent (lib.rs:35-45)
    L.kjarni_shutdown()
    assert L.kjarni_version() == b"0.1.0"
    names = {0: "KJARNI_OK", 1: "KJARNI_ERROR_NULL_POINTER", 2: "KJARNI_ERROR_INVALID_UTF8",
             3: "KJARNI_ERROR_MODEL_NOT_FOUND", 4: "KJARNI_ERROR_LOAD_FAILED", 5: "KJARNI_ERROR_INFERENCE_FAILED",
             6: "KJARNI_ERROR_GPU_UNAVAILABLE", 7: "KJARNI_ERROR_INVALID_CONFIG", 8: "KJARNI_ERROR_CANCELLED",
             9: "KJARNI_ERROR_TIMEOUT", 10: "KJARNI_ERROR_STREAM_ENDED", 255: "KJARNI_ERROR_UNKNOWN"}
    for code, name in names.items():   # error.rs:63-85
        assert L.kjarni_error_name(code) == name.encode()
        assert L.kjarni_error_code_to_string(code) == name.encode()


def test_last_error_is_thread_local_and_clearable():
    L.kjarni_clear_error()
    assert L.kjarni_last_error_message() is None          # error.rs:87-95: NULL when none
    cfg = L.kjarni_embedder_config_default()
    cfg.model_name = b"definitely-not-a-model"
    h = C.c_void_p()
    assert L.kjarni_embedder_new(C.byref(cfg), C.byref(h)) == E.MODEL_NOT_FOUND
    msg = L.kjarni_last_error_message()
    assert msg and b"Unknown model 'definitely-not-a-model'" in msg
    seen = []
    t = threading.Thread(target=lambda: seen.append(L.kjarni_last_error_message()))
    t.start()
    t.join()
    assert seen == [None]                                  # another thread sees no error
    L.kjarni_clear_error()
    assert L.kjarni_last_error_message() is None


def test_null_pointer_handling():
    h = C.c_void_p()
    assert L.kjarni_embedder_new(None, None) == E.NULL_POINTER
    assert L.kjarni_reranker_new(None, None) == E.NULL_POINTER
    assert L.kjarni_classifier_new(None, None) == E.NULL_POINTER
    fa = _ffi.KjarniFloatArray()
    assert L.kjarni_embedder_encode(None, b"x", C.byref(fa)) == E.NULL_POINTER
    f2 = _ffi.KjarniFloat2DArray()
    assert L.kjarni_embedder_encode_batch(None, None, 0, C.byref(f2)) == E.NULL_POINTER
    assert L.kjarni_embedder_dim(None) == 0               # embedder.rs:265-275
    assert L.kjarni_classifier_num_labels(None) == 0
    # frees tolerate NULL and empty structs (lib.rs:131-174)
    L.kjarni_float_array_free(None)
    L.kjarni_float_2d_array_free(None)
    L.kjarni_string_array_free(None)
    L.kjarni_string_free(None)
    L.kjarni_rerank_results_free(None)
    L.kjarni_class_results_free(None)
    L.kjarni_float_array_free(C.byref(_ffi.KjarniFloatArray()))
    L.kjarni_embedder_free(None)
    L.kjarni_hip_encoder_free(None)


def test_invalid_utf8_in_config():
    cfg = L.kjarni_embedder_config_default()
    cfg.model_name = b"\xff\xfe"
    h = C.c_void_p()
    assert L.kjarni_embedder_new(C.byref(cfg), C.byref(h)) == E.INVALID_UTF8


def test_registry_resolution_messages():
    def err(name):
        cfg = L.kjarni_reranker_config_default()
        cfg.model_name = name.encode()
        h = C.c_void_p()
        rc = L.kjarni_reranker_new(C.byref(cfg), C.byref(h))
        return rc, (L.kjarni_last_error_message() or b"").decode()
    # registry.rs:725-739: substring matches are suggested
    rc, msg = err("minilm")
    assert rc == E.MODEL_NOT_FOUND
    assert msg == "Unknown model 'minilm'. Did you mean: minilm-l6-v2, minilm-l6-v2-cross-encoder?"
    # the classifier's default name "sentiment" is itself unknown in the reference (classifier.rs:133)
    h = C.c_void_p()
    assert L.kjarni_classifier_new(None, C.byref(h)) == E.MODEL_NOT_FOUND
    assert b"Did you mean: distilbert-sentiment, roberta-sentiment, bert-sentiment-multilingual?" in \
        L.kjarni_last_error_message()
    # registry.rs:741-751: Levenshtein suggestions
    rc, msg = err("minilm-l6-v3")
    assert rc == E.MODEL_NOT_FOUND and "Did you mean: minilm-l6-v2" in msg
    # known name, case-insensitive + HF alias, but nothing on disk -> "not downloaded" (ModelNotFound)
    for name in ("MiniLM-L6-v2-Cross-Encoder", "cross-encoder/ms-marco-MiniLM-L-6-v2"):
        cfg = L.kjarni_reranker_config_default()
        cfg.model_name = name.encode()
        cfg.cache_dir = b"/nonexistent-cache"
        assert L.kjarni_reranker_new(C.byref(cfg), C.byref(h)) == E.MODEL_NOT_FOUND
        assert b"/nonexistent-cache/cross-encoder_ms-marco-MiniLM-L-6-v2" in L.kjarni_last_error_message()
    # a decoder model is not valid for reranking -> LoadFailed (IncompatibleModel maps to `_`)
    rc, msg = err("gpt2")
    assert rc == E.LOAD_FAILED


def test_cosine_similarity_helper():
    a = np.array([1, 2, 3], np.float32)
    f = lambda x, y: L.kjarni_cosine_similarity(x.ctypes.data_as(_ffi._f32p), y.ctypes.data_as(_ffi._f32p), len(x))
    assert abs(f(a, a) - 1.0) < 1e-6
    assert abs(f(a, -a) + 1.0) < 1e-6
    assert f(np.zeros(3, np.float32), a) == 0.0            # embedder/model.rs:253-255
    assert L.kjarni_cosine_similarity(None, None, 3) == 0.0  # lib.rs:181-183
    assert L.kjarni_cosine_similarity(a.ctypes.data_as(_ffi._f32p), a.ctypes.data_as(_ffi._f32p), 0) == 0.0


def test_encode_batch_zero_texts_is_ok_and_empty():
    # embedder.rs:185-188 returns Ok + empty before touching the handle's model; needs a non-NULL handle
    # and texts pointer, so use dummies (never dereferenced for n == 0).
    f2 = _ffi.KjarniFloat2DArray()
    dummy = (C.c_char_p * 1)(b"x")
    assert L.kjarni_embedder_encode_batch(C.c_void_p(1), dummy, 0, C.byref(f2)) == E.OK
    assert not f2.data and f2.rows == 0 and f2.cols == 0
    rr = _ffi.KjarniRerankResults()
    assert L.kjarni_reranker_rerank(C.c_void_p(1), b"q", dummy, 0, C.byref(rr)) == E.OK
    assert not rr.results and rr.len == 0


@pytest.mark.skipif(kjarni_amd.device_count() > 0, reason="only meaningful on a host without a GPU")
def test_no_gpu_means_gpu_unavailable_not_a_cpu_fallback(tmp_path):
    d = str(tmp_path / "m")
    synth.minilm_embedder(d, seed=0, num_hidden_layers=1)
    synth.add_tokenizer(d)
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.HipEncoder(d, 0)
    assert ei.value.code == E.GPU_UNAVAILABLE
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.Embedder(model_path=d)
    assert ei.value.code == E.GPU_UNAVAILABLE
    q = np.ones(4, np.float32)
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.cosine_search(q, np.ones((3, 4), np.float32), 2)
    assert ei.value.code == E.GPU_UNAVAILABLE


def test_bad_model_files(tmp_path):
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.HipEncoder(str(tmp_path / "nothing"), 0)
    assert ei.value.code == E.MODEL_NOT_FOUND
    with pytest.raises(kjarni_amd.KjarniException) as ei:
        kjarni_amd.Reranker(model_path=str(tmp_path / "nothing"))
    assert ei.value.code == E.LOAD_FAILED
