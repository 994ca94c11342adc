"""Transcriber (mirror of the Rust API in crates/kjarni/src/transcriber) and the stage-wise Whisper hooks."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, NamedTuple, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import KjarniDevice, check_error, lib

STAGES = ["loading_audio", "encoding", "decoding", "stitching"]


class Segment(NamedTuple):
    start: float
    end: float
    text: str


class Transcription(NamedTuple):
    text: str
    segments: List[Segment]
    language: str
    duration_secs: float


def _take(t: _ffi.KjarniTranscription) -> Transcription:
    segs = [Segment(float(t.segments[i].start), float(t.segments[i].end), (t.segments[i].text or b"").decode("utf-8"))
            for i in range(t.num_segments)]
    out = Transcription((t.text or b"").decode("utf-8"), segs, (t.language or b"").decode("utf-8"), float(t.duration_secs))
    lib().kjarni_transcription_free(C.byref(t))
    return out


class Transcriber:
    def __init__(self, model: Optional[str] = None, model_path: Optional[str] = None, cache_dir: Optional[str] = None,
                 language: Optional[str] = None, translate: bool = False, timestamps: bool = False, max_tokens: int = 448,
                 device: str = "cpu", quiet: bool = True):
        cfg = lib().kjarni_transcriber_config_default()
        cfg.device = KjarniDevice.GPU if device == "gpu" else KjarniDevice.CPU
        self._keep = [s.encode("utf-8") if s is not None else None for s in (cache_dir, model, model_path, language)]
        cfg.cache_dir, cfg.model_name, cfg.model_path, cfg.language = self._keep
        cfg.task = 1 if translate else 0
        cfg.timestamps = int(timestamps)
        cfg.max_tokens_per_chunk = max_tokens
        cfg.quiet = int(quiet)
        self._handle = C.c_void_p()
        check_error(lib().kjarni_transcriber_new(C.byref(cfg), C.byref(self._handle)))

    def __del__(self):
        if getattr(self, "_handle", None) and self._handle.value:
            lib().kjarni_transcriber_free(self._handle)
            self._handle = C.c_void_p()

    @property
    def model_name(self) -> str:
        need = lib().kjarni_transcriber_model_name(self._handle, None, 0)
        buf = C.create_string_buffer(max(need, 1))
        lib().kjarni_transcriber_model_name(self._handle, buf, need)
        return buf.value.decode("utf-8")

    @staticmethod
    def _callbacks(on_progress, on_token):
        def prog(p, _u):
            on_progress(STAGES[p.stage], int(p.current), int(p.total), p.message.decode("utf-8") if p.message else None)

        def tok(t, _u):
            r = on_token(int(t.token_id), (t.text or b"").decode("utf-8", "replace"), bool(t.is_special))
            return True if r is None else bool(r)
        return (_ffi.KjarniTranscriptionProgressFn(prog) if on_progress else _ffi.KjarniTranscriptionProgressFn(),
                _ffi.KjarniTokenCallbackFn(tok) if on_token else _ffi.KjarniTokenCallbackFn())

    def transcribe_audio(self, samples, sample_rate: int = 16000, on_progress=None, on_token=None, cancel_token=None) -> Transcription:
        s = np.ascontiguousarray(samples, np.float32)
        out = _ffi.KjarniTranscription()
        p = s.ctypes.data_as(C.POINTER(C.c_float))
        if on_progress is None and on_token is None and cancel_token is None:
            check_error(lib().kjarni_transcriber_transcribe_audio(self._handle, p, s.size, sample_rate, C.byref(out)))
        else:
            pc, tc = self._callbacks(on_progress, on_token)
            check_error(lib().kjarni_transcriber_transcribe_audio_with_callbacks(
                self._handle, p, s.size, sample_rate, pc, None, tc, None, cancel_token._handle if cancel_token else None, C.byref(out)))
        return _take(out)

    def transcribe_file(self, path: str, on_progress=None, on_token=None, cancel_token=None) -> Transcription:
        out = _ffi.KjarniTranscription()
        if on_progress is None and on_token is None and cancel_token is None:
            check_error(lib().kjarni_transcriber_transcribe_file(self._handle, path.encode("utf-8"), C.byref(out)))
        else:
            pc, tc = self._callbacks(on_progress, on_token)
            check_error(lib().kjarni_transcriber_transcribe_file_with_callbacks(
                self._handle, path.encode("utf-8"), pc, None, tc, None, cancel_token._handle if cancel_token else None, C.byref(out)))
        return _take(out)


class HipWhisper:
    """Stage-wise access to the Whisper path on the GPU (kjarni_hip_whisper_*)."""

    def __init__(self, model_dir: str, device: int = 0):
        self._h = C.c_void_p()
        check_error(lib().kjarni_hip_whisper_load(model_dir.encode("utf-8"), device, C.byref(self._h)))
        d, m, v, f = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check_error(lib().kjarni_hip_whisper_dims(self._h, C.byref(d), C.byref(m), C.byref(v), C.byref(f)))
        self.d_model, self.n_mels, self.vocab = d.value, m.value, v.value

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().kjarni_hip_whisper_free(self._h)
            self._h = C.c_void_p()

    @staticmethod
    def _f(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def log_mel(self, samples) -> np.ndarray:
        s = np.ascontiguousarray(samples, np.float32)
        out = np.empty((self.n_mels, 3000), np.float32)
        check_error(lib().kjarni_hip_whisper_log_mel(self._h, self._f(s), s.size, self._f(out)))
        return out

    def encode_mel(self, mel, fetch: bool = True) -> Optional[np.ndarray]:
        mel = np.ascontiguousarray(mel, np.float32)
        frames = mel.shape[1]
        out = np.empty(((frames + 2 - 3) // 2 + 1, self.d_model), np.float32) if fetch else None
        check_error(lib().kjarni_hip_whisper_encode_mel(self._h, self._f(mel), frames, self._f(out) if fetch else None))
        return out

    def encode_audio(self, samples, fetch: bool = True) -> Optional[np.ndarray]:
        s = np.ascontiguousarray(samples, np.float32)
        out = np.empty((1500, self.d_model), np.float32) if fetch else None
        check_error(lib().kjarni_hip_whisper_encode_audio(self._h, self._f(s), s.size, self._f(out) if fetch else None))
        return out

    def decode_begin(self):
        check_error(lib().kjarni_hip_whisper_decode_begin(self._h))

    def decode_forward(self, ids: Sequence[int]):
        a = np.ascontiguousarray(ids, np.uint32)
        hidden = np.empty((a.size, self.d_model), np.float32)
        logits = np.empty(self.vocab, np.float32)
        check_error(lib().kjarni_hip_whisper_decode_forward(self._h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size,
                                                            self._f(hidden), self._f(logits)))
        return hidden, logits

    def greedy(self, prompt: Sequence[int], timestamps: bool = False, max_tokens: int = 448) -> List[int]:
        p = np.ascontiguousarray(prompt, np.uint32)
        out = np.empty(max_tokens + 2, np.uint32)
        n = C.c_size_t(0)
        check_error(lib().kjarni_hip_whisper_greedy(self._h, p.ctypes.data_as(C.POINTER(C.c_uint32)), p.size, int(timestamps),
                                                    max_tokens, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size, C.byref(n)))
        return out[:n.value].tolist()

    def decode_text(self, ids: Sequence[int], skip_special: bool = True) -> str:
        a = np.ascontiguousarray(ids, np.uint32)
        p = C.c_void_p()
        check_error(lib().kjarni_hip_whisper_decode_text(self._h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, int(skip_special), C.byref(p)))
        s = C.string_at(p).decode("utf-8")
        lib().kjarni_string_free(p)
        return s


def bytelevel_decode(tokenizer_json: str, ids: Sequence[int], skip_special: bool = True) -> str:
    a = np.ascontiguousarray(ids, np.uint32)
    p = C.c_void_p()
    check_error(lib().kjarni_bytelevel_decode(tokenizer_json.encode("utf-8"), a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size,
                                              int(skip_special), C.byref(p)))
    s = C.string_at(p).decode("utf-8")
    lib().kjarni_string_free(p)
    return s


def load_wav(path: str):
    arr = _ffi.KjarniFloatArray()
    rate = C.c_uint32(0)
    check_error(lib().kjarni_audio_load_wav(path.encode("utf-8"), C.byref(arr), C.byref(rate)))
    out = np.ctypeslib.as_array(arr.data, shape=(arr.len,)).copy() if arr.len else np.zeros(0, np.float32)
    lib().kjarni_float_array_free(C.byref(arr))
    return out, rate.value
