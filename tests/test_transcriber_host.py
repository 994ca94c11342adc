"""Host-side pieces of the transcription path, no GPU: ByteLevel decode against the `tokenizers` core the
reference links (Tokenizer::decode, called at crates/kjarni-models/src/models/whisper/transcriber.rs:176, 317),
WAV loading against the oracle (audio/loader.rs:125-300), config validation (transcriber/validation.rs:37-98,
builder.rs:118-159) and the C layouts."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from kjarni_amd import _ffi
from kjarni_amd.transcriber import bytelevel_decode, load_wav
from oracle import whisper_oracle as W
from tests import synth

F32 = np.float32


@pytest.fixture(scope="module")
def tok_json(tmp_path_factory):
    return synth.whisper_tokenizer_json(str(tmp_path_factory.mktemp("tok") / "tokenizer.json"))


def test_struct_layouts_and_defaults():
    assert C.sizeof(_ffi.KjarniToken) == 16                      # callback.rs:36-41
    assert C.sizeof(_ffi.KjarniTranscriberConfig) == 64
    assert C.sizeof(_ffi.KjarniTranscriptionSegment) == 16
    assert C.sizeof(_ffi.KjarniTranscription) == 40
    assert C.sizeof(_ffi.KjarniTranscriptionProgress) == 32
    c = _ffi.lib().kjarni_transcriber_config_default()           # builder.rs:26-39
    assert (c.device, c.task, c.timestamps, c.max_tokens_per_chunk, c.quiet) == (0, 0, 0, 448, 0)
    assert c.model_name is None and c.model_path is None and c.language is None and c.cache_dir is None


def test_bytelevel_decode_matches_tokenizers(tok_json):
    from tokenizers import Tokenizer
    hf = Tokenizer.from_file(tok_json)
    rng = np.random.default_rng(3)
    cases = [[], [256], list(range(256, 276)), [262, 263], [263], [267, 268], [268, 267], [274, 275], [275],
             [50257, 300, 50259, 301, 50364, 50365], list(range(1, 256)), [60000, 300]]
    for _ in range(60):
        n = int(rng.integers(1, 40))
        ids = rng.integers(0, 50257, n).tolist()
        if rng.random() < 0.5:
            ids[int(rng.integers(0, n))] = int(rng.integers(50257, 51865))
        cases.append(ids)
    for ids in cases:
        for skip in (True, False):
            known = [i for i in ids if i < 51865]
            assert bytelevel_decode(tok_json, ids, skip) == hf.decode(known, skip_special_tokens=skip), (ids, skip)
    assert bytelevel_decode(tok_json, [50259], False) == "<|en|>" and bytelevel_decode(tok_json, [50259], True) == ""
    assert bytelevel_decode(tok_json, [50364 + 75], False) == "<|1.50|>"


def _wav(path, fmt_tag, channels, rate, bits, payload, extensible=False):
    block = channels * bits // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, channels, rate, rate * block, block, bits, 22, bits, 0, fmt_tag,
                          b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
        b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


@pytest.mark.parametrize("bits,tag,channels,rate,ext", [(16, 1, 1, 16000, False), (16, 1, 2, 44100, False), (8, 1, 1, 8000, False),
                                                        (24, 1, 2, 22050, False), (32, 1, 1, 48000, False), (32, 3, 2, 16000, False),
                                                        (16, 1, 3, 32000, True)])
def test_wav_loader_matches_oracle(tmp_path, bits, tag, channels, rate, ext):
    rng = np.random.default_rng(bits + channels)
    n = 997 * channels
    if tag == 3:
        payload = rng.uniform(-1, 1, n).astype("<f4").tobytes()
    elif bits == 8:
        payload = rng.integers(0, 256, n).astype(np.uint8).tobytes()
    elif bits == 16:
        payload = rng.integers(-32768, 32768, n).astype("<i2").tobytes()
    elif bits == 24:
        v = rng.integers(-(1 << 23), 1 << 23, n)
        payload = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    else:
        payload = rng.integers(-(1 << 31), 1 << 31, n).astype("<i4").tobytes()
    path = _wav(str(tmp_path / "a.wav"), tag, channels, rate, bits, payload, ext)
    got, got_rate = load_wav(path)
    exp, exp_rate = W.read_wav(open(path, "rb").read())
    assert got_rate == exp_rate == rate
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert np.abs(got).max() <= 1.0 + 1e-6


def test_wav_errors(tmp_path):
    L = _ffi.lib()
    arr = _ffi.KjarniFloatArray()
    assert L.kjarni_audio_load_wav(str(tmp_path / "missing.wav").encode(), C.byref(arr), None) == _ffi.KjarniError.INFERENCE_FAILED
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not a wav file at all")
    assert L.kjarni_audio_load_wav(str(bad).encode(), C.byref(arr), None) == _ffi.KjarniError.INFERENCE_FAILED
    assert b"WAV" in L.kjarni_last_error_message()
    assert L.kjarni_audio_load_wav(None, C.byref(arr), None) == _ffi.KjarniError.NULL_POINTER


def test_transcriber_new_validation(tmp_path):
    """builder.rs:118-127 + validation.rs:37-67: these fail before anything touches the GPU."""
    L = _ffi.lib()

    def new(**kw):
        cfg = L.kjarni_transcriber_config_default()
        keep = []
        for k, v in kw.items():
            if isinstance(v, str):
                v = v.encode()
                keep.append(v)
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = L.kjarni_transcriber_new(C.byref(cfg), C.byref(h))
        return rc, (L.kjarni_last_error_message() or b"").decode()

    rc, msg = new(model_name="whisper-enormous")
    assert rc == _ffi.KjarniError.INVALID_CONFIG and "Unknown model: 'whisper-enormous'" in msg
    rc, msg = new(language="")
    assert rc == _ffi.KjarniError.INVALID_CONFIG and "Language code cannot be empty" in msg
    rc, msg = new(language="a-very-long-language")
    assert rc == _ffi.KjarniError.INVALID_CONFIG and "too long" in msg
    rc, msg = new(max_tokens_per_chunk=0)
    assert rc == _ffi.KjarniError.INVALID_CONFIG and "must be > 0" in msg
    rc, msg = new(max_tokens_per_chunk=10_000)
    assert rc == _ffi.KjarniError.INVALID_CONFIG and "too large: 10000 (max 4096)" in msg
    rc, msg = new(model_name="Whisper-Small", cache_dir=str(tmp_path))     # known name, nothing on disk, no downloads
    assert rc == _ffi.KjarniError.MODEL_NOT_FOUND and "openai_whisper-small" in msg
    assert L.kjarni_transcriber_new(None, None) == _ffi.KjarniError.NULL_POINTER
    L.kjarni_transcriber_free(None)
    L.kjarni_transcription_free(None)
    assert L.kjarni_transcriber_model_name(None, None, 0) == 0
