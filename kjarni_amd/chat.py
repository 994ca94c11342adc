"""Chat over the kjarni_chat_* C ABI (mirror of the reference's Go / C# bindings:
crates/kjarni-ffi/bindings/go/chat.go, bindings/csharp/Kjarni/Chat.cs) plus the stage-wise hooks of kjarni_hip.h
(BPE tokenizer, templates, sampling, generation-config resolution) the parity tests use."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import KjarniDevice, check_error, lib

MODES = {"default": 0, "creative": 1, "reasoning": 2}
ROLES = {"system": 0, "user": 1, "assistant": 2}
TEMPLATES = {"llama3": 0, "chatml": 1, "mistral": 2}
STRATEGIES = ["greedy", "sample", "beam_search"]


class GenerationConfig(NamedTuple):
    """Negative / None = keep the resolved default (KjarniGenerationConfig)."""
    temperature: Optional[float] = None
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    min_p: Optional[float] = None
    repetition_penalty: Optional[float] = None
    max_new_tokens: Optional[int] = None
    do_sample: Optional[bool] = None

    @staticmethod
    def greedy(max_new_tokens: Optional[int] = None) -> "GenerationConfig":
        return GenerationConfig(do_sample=False, max_new_tokens=max_new_tokens)


class ResolvedGeneration(NamedTuple):
    strategy: str
    temperature: float
    top_k: Optional[int]
    top_p: Optional[float]
    min_p: Optional[float]
    repetition_penalty: float
    no_repeat_ngram_size: int
    max_new_tokens: Optional[int]
    max_length: int
    add_bos_token: bool


def _gen(cfg: Optional[GenerationConfig]):
    if cfg is None:
        return None
    c = lib().kjarni_generation_config_default()
    if cfg.temperature is not None:
        c.temperature = cfg.temperature
    if cfg.top_k is not None:
        c.top_k = cfg.top_k
    if cfg.top_p is not None:
        c.top_p = cfg.top_p
    if cfg.min_p is not None:
        c.min_p = cfg.min_p
    if cfg.repetition_penalty is not None:
        c.repetition_penalty = cfg.repetition_penalty
    if cfg.max_new_tokens is not None:
        c.max_new_tokens = cfg.max_new_tokens
    if cfg.do_sample is not None:
        c.do_sample = int(cfg.do_sample)
    return C.byref(c)


def _resolved(r: _ffi.KjarniResolvedGeneration) -> ResolvedGeneration:
    return ResolvedGeneration(STRATEGIES[r.strategy], float(r.temperature), None if r.top_k < 0 else int(r.top_k),
                              None if r.top_p < 0 else float(r.top_p), None if r.min_p < 0 else float(r.min_p),
                              float(r.repetition_penalty), int(r.no_repeat_ngram_size),
                              None if r.max_new_tokens < 0 else int(r.max_new_tokens), int(r.max_length), bool(r.add_bos_token))


def _take_string(p: C.c_void_p) -> str:
    s = C.string_at(p).decode("utf-8") if p.value else ""
    if p.value:
        lib().kjarni_string_free(p)
    return s


def _stream_cb(on_token: Callable[[str], bool]):
    def cb(text, _user):
        r = on_token((text or b"").decode("utf-8", errors="replace"))
        return True if r is None else bool(r)
    return _ffi.KjarniStreamCallbackFn(cb)


def _messages(history: Sequence[Tuple[str, str]]):
    n = len(history)
    roles = (C.c_int32 * max(n, 1))(*[ROLES[r] if isinstance(r, str) else int(r) for r, _ in history])
    keep = [c.encode("utf-8") for _, c in history]
    contents = (C.c_char_p * max(n, 1))(*keep)
    return roles, contents, n, keep


class Chat:
    def __init__(self, model: str, model_path: Optional[str] = None, cache_dir: Optional[str] = None,
                 system_prompt: Optional[str] = None, mode: str = "default", device: str = "cpu", quiet: bool = True):
        cfg = lib().kjarni_chat_config_default()
        cfg.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        self._keep = [s.encode("utf-8") if s is not None else None for s in (cache_dir, model, model_path, system_prompt)]
        cfg.cache_dir, cfg.model_name, cfg.model_path, cfg.system_prompt = self._keep
        cfg.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        cfg.quiet = int(quiet)
        self._handle = C.c_void_p()
        check_error(lib().kjarni_chat_new(C.byref(cfg), C.byref(self._handle)))

    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_chat_free(self._handle)
            self._handle = C.c_void_p()

    __del__ = close

    @property
    def model_name(self) -> str:
        need = lib().kjarni_chat_model_name(self._handle, None, 0)
        buf = C.create_string_buffer(need + 1)
        lib().kjarni_chat_model_name(self._handle, buf, need + 1)
        return buf.value.decode("utf-8")

    @property
    def context_size(self) -> int:
        return int(lib().kjarni_chat_context_size(self._handle))

    def send(self, message: str, config: Optional[GenerationConfig] = None) -> str:
        out = C.c_void_p()
        check_error(lib().kjarni_chat_send(self._handle, message.encode("utf-8"), _gen(config), C.byref(out)))
        return _take_string(out)

    def stream(self, message: str, on_token: Callable[[str], bool], config: Optional[GenerationConfig] = None, cancel=None):
        cb = _stream_cb(on_token)
        check_error(lib().kjarni_chat_stream(self._handle, message.encode("utf-8"), _gen(config), cb, None,
                                             cancel._handle if cancel is not None else None))

    def send_with_history(self, history: Sequence[Tuple[str, str]], message: str, config: Optional[GenerationConfig] = None) -> str:
        roles, contents, n, _keep = _messages(history)
        out = C.c_void_p()
        check_error(lib().kjarni_chat_send_with_history(self._handle, roles, contents, n, message.encode("utf-8"), _gen(config),
                                                        C.byref(out)))
        return _take_string(out)

    def conversation(self) -> "ChatConversation":
        return ChatConversation(self)

    # ---- kjarni_hip.h hooks on a live handle ----
    def resolve(self, config: Optional[GenerationConfig] = None) -> ResolvedGeneration:
        r = _ffi.KjarniResolvedGeneration()
        check_error(lib().kjarni_hip_chat_resolve(self._handle, _gen(config), C.byref(r)))
        return _resolved(r)

    def format_prompt(self, history: Optional[Sequence[Tuple[str, str]]], message: Optional[str]) -> str:
        out = C.c_void_p()
        msg = message.encode("utf-8") if message is not None else None
        if history is None:
            check_error(lib().kjarni_hip_chat_format_prompt(self._handle, None, None, 0, msg, C.byref(out)))
        else:
            roles, contents, n, _keep = _messages(history)
            check_error(lib().kjarni_hip_chat_format_prompt(self._handle, roles, contents, n, msg, C.byref(out)))
        return _take_string(out)

    def encode(self, prompt: str, config: Optional[GenerationConfig] = None) -> List[int]:
        n = C.c_size_t()
        check_error(lib().kjarni_hip_chat_encode(self._handle, prompt.encode("utf-8"), _gen(config), None, 0, C.byref(n)))
        ids = np.zeros(max(n.value, 1), np.uint32)
        check_error(lib().kjarni_hip_chat_encode(self._handle, prompt.encode("utf-8"), _gen(config),
                                                 ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size, C.byref(n)))
        return ids[: n.value].tolist()

    def seed(self, seed: int):
        lib().kjarni_hip_chat_seed(self._handle, seed)

    def set_device_sampling(self, on: bool):
        """Sampling / logits processors with the O(vocab) work on the device (default) or on a host copy of the logits."""
        lib().kjarni_hip_chat_set_device_sampling(self._handle, 1 if on else 0)

    def sampling_counters(self):
        """(tokens decided from the device's candidates, tokens that needed the full logits)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().kjarni_hip_chat_sampling_counters(self._handle, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)


class ChatConversation:
    def __init__(self, chat: Chat):
        self._chat = chat  # the parent must outlive the conversation
        self._handle = C.c_void_p()
        check_error(lib().kjarni_chat_conversation_new(chat._handle, C.byref(self._handle)))

    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_chat_conversation_free(self._handle)
            self._handle = C.c_void_p()

    __del__ = close

    def send(self, message: str, config: Optional[GenerationConfig] = None) -> str:
        out = C.c_void_p()
        check_error(lib().kjarni_chat_conversation_send(self._handle, message.encode("utf-8"), _gen(config), C.byref(out)))
        return _take_string(out)

    def stream(self, message: str, on_token: Callable[[str], bool], config: Optional[GenerationConfig] = None, cancel=None):
        cb = _stream_cb(on_token)
        check_error(lib().kjarni_chat_conversation_stream(self._handle, message.encode("utf-8"), _gen(config), cb, None,
                                                          cancel._handle if cancel is not None else None))

    def __len__(self) -> int:
        return int(lib().kjarni_chat_conversation_len(self._handle))

    def clear(self, keep_system: bool = True):
        lib().kjarni_chat_conversation_clear(self._handle, int(keep_system))


# ---- host-side stages --------------------------------------------------------------------------------

class BpeTokenizer:
    def __init__(self, tokenizer_json_path: str):
        self._handle = C.c_void_p()
        check_error(lib().kjarni_bpe_tokenizer_load(tokenizer_json_path.encode("utf-8"), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_bpe_tokenizer_free(self._handle)
            self._handle = C.c_void_p()

    def encode(self, text: str, max_length: int = 0) -> List[int]:
        raw = text.encode("utf-8")
        n = C.c_size_t()
        cap = len(raw) + 16
        ids = np.zeros(cap, np.uint32)
        check_error(lib().kjarni_bpe_tokenizer_encode(self._handle, raw, max_length, ids.ctypes.data_as(C.POINTER(C.c_uint32)), cap,
                                                      C.byref(n)))
        return ids[: n.value].tolist()

    def decode(self, ids: Sequence[int], skip_special: bool = False) -> str:
        a = np.ascontiguousarray(ids, np.uint32)
        out = C.c_void_p()
        check_error(lib().kjarni_bpe_tokenizer_decode(self._handle, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, int(skip_special),
                                                      C.byref(out)))
        return _take_string(out)

    def pre_tokenize(self, text: str) -> List[str]:
        arr = _ffi.KjarniStringArray()
        check_error(lib().kjarni_bpe_tokenizer_pre_tokenize(self._handle, text.encode("utf-8"), C.byref(arr)))
        out = [C.string_at(arr.strings[i]).decode("utf-8") for i in range(arr.len)]
        lib().kjarni_string_array_free(C.byref(arr))
        return out


def chat_template_apply(template: str, conversation: Sequence[Tuple[str, str]]) -> str:
    roles, contents, n, _keep = _messages(conversation)
    out = C.c_void_p()
    check_error(lib().kjarni_chat_template_apply(TEMPLATES[template], roles, contents, n, C.byref(out)))
    return _take_string(out)


def sampling_distribution(logits, temperature: float = 1.0, top_k: Optional[int] = None, top_p: Optional[float] = None,
                          min_p: Optional[float] = None) -> np.ndarray:
    lg = np.ascontiguousarray(logits, np.float32)
    out = np.zeros_like(lg)
    f = C.POINTER(C.c_float)
    check_error(lib().kjarni_sampling_distribution(lg.ctypes.data_as(f), lg.size, temperature, -1 if top_k is None else top_k,
                                                   -1.0 if top_p is None else top_p, -1.0 if min_p is None else min_p,
                                                   out.ctypes.data_as(f)))
    return out


def sampling_distribution_candidates(logits, tau: float, temperature: float = 1.0, top_k: Optional[int] = None,
                                     top_p: Optional[float] = None, min_p: Optional[float] = None):
    """The distribution decided from the candidates within `tau` of the maximum (what the device hands to the decode loop).
    Returns (probs or None when the candidates do not decide it, number of candidates)."""
    lg = np.ascontiguousarray(logits, np.float32)
    out = np.zeros_like(lg)
    f = C.POINTER(C.c_float)
    decided, n = C.c_int32(0), C.c_size_t(0)
    check_error(lib().kjarni_sampling_distribution_candidates(lg.ctypes.data_as(f), lg.size, tau, temperature,
                                                              -1 if top_k is None else top_k, -1.0 if top_p is None else top_p,
                                                              -1.0 if min_p is None else min_p, out.ctypes.data_as(f),
                                                              C.byref(decided), C.byref(n)))
    return (out if decided.value else None), int(n.value)


def sample_from_probs(probs, uniform: float) -> int:
    p = np.ascontiguousarray(probs, np.float32)
    return int(lib().kjarni_sample_from_probs(p.ctypes.data_as(C.POINTER(C.c_float)), p.size, uniform))


def logits_process(logits, tokens: Sequence[int], repetition_penalty: float = 1.0, no_repeat_ngram: int = 0) -> np.ndarray:
    lg = np.array(logits, np.float32)
    t = np.ascontiguousarray(tokens, np.uint32)
    check_error(lib().kjarni_logits_process(lg.ctypes.data_as(C.POINTER(C.c_float)), lg.size, t.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            t.size, repetition_penalty, no_repeat_ngram))
    return lg


def generation_resolve(model_type: str, max_position_embeddings: int, generation_config_json: Optional[str] = None,
                       mode: Optional[str] = "default", config: Optional[GenerationConfig] = None) -> ResolvedGeneration:
    r = _ffi.KjarniResolvedGeneration()
    m = -1 if mode is None else MODES[mode]
    hf = generation_config_json.encode("utf-8") if generation_config_json is not None else None
    check_error(lib().kjarni_generation_resolve(model_type.encode("utf-8"), max_position_embeddings, hf, m, _gen(config), C.byref(r)))
    return _resolved(r)
