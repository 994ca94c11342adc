"""Pins oracle/whisper_oracle.py with the reference's own model-free golden tests (CPU only).

  conv front end        crates/kjarni-transformers/src/audio/mel.rs:2078-2119   (fixture: golden/whisper_conv_frontend.json)
  cross attention       cpu/encoder_decoder/decoder_cross_attn.rs:194-437
  cross decoder layer   cpu/encoder_decoder/decoder_cross_attn_layer.rs:384-700
  Whisper decoder       cpu/encoder_decoder/cpu_decoder.rs:871-934, weights :1003-1132
  Whisper encoder       cpu/encoder_decoder/cpu_encoder.rs:913-972, weights :704-770
  chunking / stitching  crates/kjarni-models/src/models/whisper/transcriber.rs:460-535
  resampling            crates/kjarni/src/transcriber/model.rs:333-356
Tolerances are the reference's (1e-4, 1e-3 for the two Whisper scenarios)."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from oracle import whisper_oracle as W

F32 = np.float32
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seq(start, n, step=0.01):
    """start, start+step, ... as the literals in the reference tests (two/three decimals)."""
    return np.asarray([round(start + i * step, 6) for i in range(n)], F32)


# ---------------------------------------------------------------- conv front end
def test_conv_frontend_golden():
    g = json.load(open(os.path.join(GOLD, "whisper_conv_frontend.json")))
    a = lambda k: np.asarray(g[k]["data"], F32).reshape(g[k]["shape"])  # noqa: E731
    t = {"model.conv1.weight": a("conv1_weight"), "model.conv1.bias": a("conv1_bias"),
         "model.conv2.weight": a("conv2_weight"), "model.conv2.bias": a("conv2_bias"),
         "model.embed_positions.weight": a("embed_positions")}
    out = W.conv_frontend(a("mel_input"), t, prefix="model", max_positions=1500)
    assert out.shape == (1, 5, 8)
    assert np.abs(out - a("frontend_output")).max() < 1e-4


def test_sinusoidal_fallback_and_shapes():
    e = W.sinusoidal_embeddings(6, 4)
    assert e[0].tolist() == [0.0, 1.0, 0.0, 1.0]
    np.testing.assert_allclose(e[1], [np.sin(1.0), np.cos(1.0), np.sin(0.01), np.cos(0.01)], atol=1e-6)
    w = W.hann_window(400)                                          # mel.rs:2135-2139
    assert len(w) == 400 and abs(w[0]) < 1e-6


# ---------------------------------------------------------------- cross attention (decoder_cross_attn.rs:126-437)
def _cross_attn_params(q0=0.01):
    return dict(q_w=seq(q0, 16).reshape(4, 4), q_b=np.full(4, 0.01, F32),
                k_w=seq(q0 + 0.16, 16).reshape(4, 4), k_b=np.full(4, 0.01, F32),
                v_w=seq(q0 + 0.32, 16).reshape(4, 4), v_b=np.full(4, 0.01, F32),
                o_w=seq(q0 + 0.48, 16).reshape(4, 4), o_b=np.full(4, 0.01, F32))


DEC2 = seq(0.0, 8, 0.1).reshape(1, 2, 4)
ENC3 = seq(1.0, 12, 0.1).reshape(1, 3, 4)


def test_cross_attention_goldens():
    p = _cross_attn_params()
    k_t, v = W.precompute_cross_kv(ENC3, p, 2)
    out = W.cross_attention(DEC2, k_t, v, p, 2)
    g1 = [5.161280, 5.568286, 5.975293, 6.382300, 5.237971, 5.650974, 6.063977, 6.476981]
    assert np.abs(out.reshape(-1) - F32(g1)).max() < 1e-4
    out = W.cross_attention(DEC2, k_t, v, p, 2, mask=F32([[1, 1, 0]]))
    g2 = [4.482478, 4.835866, 5.189254, 5.542642, 4.511338, 4.866982, 5.222627, 5.578271]
    assert np.abs(out.reshape(-1) - F32(g2)).max() < 1e-4
    # precomputed K/V (test 3): K golden is [b,h,s,d], the cache holds [b,h,d,s]
    gk = F32([0.866, 1.05, 1.162, 1.41, 1.458, 1.77, 1.234, 1.418, 1.658, 1.906, 2.082, 2.394]).reshape(1, 2, 3, 2)
    gv = F32([1.602, 1.786, 2.154, 2.402, 2.706, 3.018, 1.97, 2.154, 2.65, 2.898, 3.33, 3.642]).reshape(1, 2, 3, 2)
    assert np.abs(v - gv).max() < 1e-4 and np.abs(k_t - gk.transpose(0, 1, 3, 2)).max() < 1e-4
    # batched with masks (test 4)
    dec = seq(0.0, 16, 0.1).reshape(2, 2, 4)
    enc = seq(1.0, 24, 0.1).reshape(2, 3, 4)
    k_t, v = W.precompute_cross_kv(enc, p, 2)
    out = W.cross_attention(dec, k_t, v, p, 2, mask=F32([[1, 1, 1], [1, 1, 0]]))
    g4 = [5.161280, 5.568287, 5.975293, 6.382300, 5.237972, 5.650974, 6.063977, 6.476981,
          8.476384, 9.145302, 9.814219, 10.483137, 8.504461, 9.175573, 9.846687, 10.517800]
    assert np.abs(out.reshape(-1) - F32(g4)).max() < 1e-4
    # single-token decode (test 5)
    k_t, v = W.precompute_cross_kv(seq(0.0, 12, 0.1).reshape(1, 3, 4), p, 2)
    out = W.cross_attention(F32([0.5, 0.6, 0.7, 0.8]).reshape(1, 1, 4), k_t, v, p, 2)
    assert np.abs(out.reshape(-1) - F32([1.976542, 2.131828, 2.287114, 2.442401])).max() < 1e-4


# ---------------------------------------------------------------- cross decoder layer (decoder_cross_attn_layer.rs:252-700)
def _layer_params():
    ln = dict(g=np.ones(4, F32), b=np.full(4, 0.01, F32))
    return {"self": _cross_attn_params(0.01), "self_ln": ln, "cross": _cross_attn_params(0.65), "cross_ln": ln,
            "ffn": dict(fc1_w=seq(1.29, 32).reshape(8, 4), fc1_b=np.full(8, 0.01, F32),
                        fc2_w=seq(1.61, 32).reshape(4, 8), fc2_b=np.full(4, 0.01, F32)), "ffn_ln": ln}


ENC_L = seq(0.5, 12, 0.1).reshape(1, 3, 4)


def test_cross_decoder_layer_goldens():
    p = _layer_params()
    post = F32([-1.331635, -0.437212, 0.457212, 1.351635, -1.331635, -0.437212, 0.457212, 1.351634])
    out, (nk, nv) = W.cross_decoder_layer(DEC2, ENC_L, p, 2, False, 1e-5)
    assert np.abs(out.reshape(-1) - post).max() < 1e-4 and nk.ndim == 3 and nv.ndim == 3
    pre = F32([22.014660, 22.899734, 23.784811, 24.669884, 22.414780, 23.299860, 24.184940, 25.070023])
    out_pre, _ = W.cross_decoder_layer(DEC2, ENC_L, p, 2, True, 1e-5)
    assert np.abs(out_pre.reshape(-1) - pre).max() < 1e-4
    assert np.abs(out_pre - out).max() > 1.0                                 # prenorm vs postnorm differ
    out, _ = W.cross_decoder_layer(DEC2, ENC_L, p, 2, False, 1e-5, cross_mask=F32([[1, 1, 0]]))
    g3 = F32([-1.331635, -0.437211, 0.457212, 1.351634, -1.331635, -0.437211, 0.457211, 1.351635])
    assert np.abs(out.reshape(-1) - g3).max() < 1e-4
    out, _ = W.cross_decoder_layer(DEC2, ENC_L, p, 2, False, 1e-5, cross_kv=W.precompute_cross_kv(ENC_L, p["cross"], 2))
    assert np.abs(out.reshape(-1) - post).max() < 1e-4
    out, _ = W.cross_decoder_layer(seq(0.0, 16, 0.1).reshape(2, 2, 4), seq(0.5, 24, 0.1).reshape(2, 3, 4), p, 2, False,
                                   1e-5, cross_mask=F32([[1, 1, 1], [1, 0, 0]]))
    g5 = F32([-1.331635, -0.437212, 0.457212, 1.351634, -1.331635, -0.437211, 0.457211, 1.351635,
              -1.331634, -0.437212, 0.457211, 1.351635, -1.331635, -0.437212, 0.457212, 1.351635])
    assert np.abs(out.reshape(-1) - g5).max() < 1e-4
    out, _ = W.cross_decoder_layer(F32([0.1, 0.2, 0.3, 0.4]).reshape(1, 1, 4), seq(0.0, 12, 0.1).reshape(1, 3, 4), p, 2,
                                   False, 1e-5)
    assert np.abs(out.reshape(-1) - F32([-1.331635, -0.437212, 0.457212, 1.351635])).max() < 1e-4


# ---------------------------------------------------------------- Whisper decoder scenario (cpu_decoder.rs:871-934)
def _count_weights(names_shapes):
    """gen_weights_helper (cpu_decoder.rs:1003-1017): consecutive i * 0.001, counting from 1."""
    out, count = {}, 1
    for name, shape in names_shapes:
        n = int(np.prod(shape))
        out[name] = (np.arange(count, count + n).astype(F32) * F32(0.001)).reshape(shape)
        count += n
    return out


def test_decoder_whisper_golden():
    w = _count_weights([("token_emb", (10, 4)), ("sa_q", (4, 4)), ("sa_k", (4, 4)), ("sa_v", (4, 4)), ("sa_o", (4, 4)),
                        ("ca_q", (4, 4)), ("ca_k", (4, 4)), ("ca_v", (4, 4)), ("ca_o", (4, 4)),
                        ("fc1", (8, 4)), ("fc2", (4, 8))])
    ln = dict(g=np.ones(4, F32), b=np.full(4, 0.01, F32))
    p = {"self": dict(q_w=w["sa_q"], k_w=w["sa_k"], v_w=w["sa_v"], o_w=w["sa_o"]), "self_ln": ln,
         "cross": dict(q_w=w["ca_q"], k_w=w["ca_k"], v_w=w["ca_v"], o_w=w["ca_o"]), "cross_ln": ln,
         "ffn": dict(fc1_w=w["fc1"], fc1_b=None, fc2_w=w["fc2"], fc2_b=None), "ffn_ln": ln}
    enc = F32([0.336690, 0.128809, 0.234462, 0.230333, -1.122856, -0.186328, 2.208201, -0.637997,
               0.461657, 0.267351, 0.534905, 0.809357]).reshape(1, 3, 4)
    # Seq2SeqCPUDecoder::forward (cpu_decoder.rs:216-263): embed -> sinusoidal -> embed LN -> layer -> final LN
    h = O.embed(np.asarray([[1, 2]], np.uint32), None, w["token_emb"], None, None)
    h = (h + W.sinusoidal_embeddings(1024, 4)[None, :2]).astype(F32)
    h = O.layer_norm(h, ln["g"], ln["b"], 1e-5)
    h, _ = W.cross_decoder_layer(h, enc, p, 2, True, 1e-5)
    h = O.layer_norm(h, ln["g"], ln["b"], 1e-5)
    golden = F32([-0.995535, 1.004425, -0.984425, 1.015535, 0.646058, -0.145752, -1.544646, 1.084340])
    assert np.abs(h.reshape(-1) - golden).max() < 1e-3


# ---------------------------------------------------------------- Whisper encoder scenario (cpu_encoder.rs:913-972)
def test_encoder_whisper_prenorm_sinusoidal_golden():
    H = 4
    t = {"model.encoder.layer_norm.weight": np.ones(H, F32), "model.encoder.layer_norm.bias": np.full(H, 0.01, F32)}
    pre = "model.encoder.layers.0"
    for name, start in (("q_proj", 0.041), ("k_proj", 0.057), ("v_proj", 0.073), ("out_proj", 0.089)):
        t[f"{pre}.self_attn.{name}.weight"] = seq(start, 16, 0.001).reshape(4, 4)
    for n in ("self_attn_layer_norm", "final_layer_norm"):
        t[f"{pre}.{n}.weight"], t[f"{pre}.{n}.bias"] = np.ones(H, F32), np.full(H, 0.01, F32)
    # fc1 = 0.105 + i * 0.001, fc2 starts at 0.105 + 0.032 (f32 arithmetic as in the Rust closure)
    t[f"{pre}.fc1.weight"] = (F32(0.105) + np.arange(32).astype(F32) * F32(0.001)).reshape(8, 4)
    t[f"{pre}.fc2.weight"] = (F32(F32(0.105) + F32(0.032)) + np.arange(32).astype(F32) * F32(0.001)).reshape(4, 8)
    t[f"{pre}.fc1.bias"], t[f"{pre}.fc2.bias"] = np.zeros(8, F32), np.zeros(4, F32)
    cfg = dict(d_model=4, encoder_layers=1, decoder_layers=0, encoder_attention_heads=2, max_source_positions=1500)
    m = W.WhisperOracle(t, cfg)
    x = seq(0.1, 12, 0.1).reshape(1, 3, 4)
    x = (x + W.sinusoidal_embeddings(1024, 4)[None, :3]).astype(F32)
    out = m.encoder_forward(x)
    golden = F32([-1.153330, 0.814141, -0.794141, 1.173330, 0.247473, -0.264928, -1.361826, 1.419281,
                  0.621133, -1.347122, -0.484890, 1.250878])
    assert np.abs(out.reshape(-1) - golden).max() < 1e-3


# ---------------------------------------------------------------- host-side pieces (transcriber.rs:460-535)
def test_chunk_audio_reference_cases():
    c = W.chunk_audio(np.ones(100_000, F32))
    assert len(c) == 1 and len(c[0]) == W.CHUNK_SAMPLES and c[0][0] == 1.0 and c[0][100_000] == 0.0
    assert len(W.chunk_audio(np.ones(W.CHUNK_SAMPLES, F32))) == 1
    c = W.chunk_audio(np.ones(W.CHUNK_SAMPLES + 100, F32))
    assert len(c) == 2 and all(len(x) == W.CHUNK_SAMPLES for x in c) and c[1][99] == 1.0 and c[1][100] == 0.0
    assert W.chunk_audio(np.zeros(0, F32)) == []


def test_stitch_and_boundaries_reference_cases():
    text, segs = W.stitch([dict(text="Hello ", segments=[dict(start=0.0, end=30.0, text="Hello ")]),
                           dict(text="world.", segments=[dict(start=30.0, end=45.0, text="world.")])])
    assert text == "Hello world." and len(segs) == 1 and segs[0]["text"] == "Hello world."
    assert abs(segs[0]["start"]) < 0.01 and abs(segs[0]["end"] - 45.0) < 0.01
    assert W.is_chunk_boundary(30.0) and W.is_chunk_boundary(60.0) and W.is_chunk_boundary(29.99)
    assert not W.is_chunk_boundary(15.0) and not W.is_chunk_boundary(29.5)


def test_timestamp_segments_and_pick_token():
    dec = lambda ids: "".join(f"<{i}>" for i in ids)  # noqa: E731
    T = W.TIMESTAMP_BEGIN
    segs = W.parse_timestamp_segments([T, 5, 6, T + 100, T + 100, 7, T + 250, 9], dec, 30.0)
    assert [(round(s["start"], 2), round(s["end"], 2), s["text"]) for s in segs] == \
        [(30.0, 32.0, "<5><6>"), (32.0, 35.0, "<7>"), (35.0, 65.0, "<9>")]
    r = W.finalize_chunk([5, 50257 + 3, 6, W.EOT_TOKEN], dec, False, 60.0)
    assert r["text"] == "<5><6>" and r["segments"] == [dict(start=60.0, end=90.0, text="<5><6>")]
    logits = np.zeros(51865, F32)
    logits[[7, 50300, 50400]] = [1.0, 5.0, 3.0]
    assert W.WhisperOracle.pick_token(logits, False, 50257) == 7            # specials and timestamps suppressed
    assert W.WhisperOracle.pick_token(logits, True, 50257) == 50400         # timestamps allowed
    logits[50257] = 9.0
    assert W.WhisperOracle.pick_token(logits, False, 50257) == 50257        # EOS always allowed
    tie = np.zeros(51865, F32)
    assert W.WhisperOracle.pick_token(tie, False, 50257) == 50257           # max_by keeps the last maximum


def test_resample_linear_and_wav():
    x = np.arange(8, dtype=F32)
    assert np.array_equal(W.resample_linear(x, 16000, 16000), x)
    up = W.resample_linear(x, 8000, 16000)
    assert len(up) == 16 and np.allclose(up[:15], np.arange(15) / 2.0) and up[15] == 7.0
    down = W.resample_linear(x, 16000, 8000)
    assert np.array_equal(down, x[::2])
    # 16-bit stereo WAV at 8 kHz -> mono, resampled to 16 kHz
    pcm = np.stack([np.arange(100, dtype=np.int16) * 100, np.arange(100, dtype=np.int16) * 300], 1)
    data = pcm.astype("<i2").tobytes()
    wav = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 8000, 32000, 4, 16) + \
        b"data" + struct.pack("<I", len(data)) + data
    s, rate = W.read_wav(wav)
    assert rate == 8000 and len(s) == 200
    assert np.allclose(s[::2], np.arange(100) * 200 / 32768.0, atol=1e-6)


# ---------------------------------------------------------------- log-mel: independent float64 cross-check
def test_log_mel_against_float64():
    """The reference holds no value test for compute_mel_spectrogram (only shapes, mel.rs:2121-2133); the
    restatement is checked against the same definition evaluated in float64."""
    rng = np.random.default_rng(0)
    t = np.arange(48_000) / 16000.0
    audio = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1800 * t) + 0.05 * rng.standard_normal(len(t))).astype(F32)
    mel = W.log_mel(audio)
    assert mel.shape == (80, 3000) and mel.dtype == F32
    x = np.pad(audio.astype(np.float64), 200, mode="reflect")
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(400) / 400))
    n_use = 1 + (len(x) - 400) // 160
    frames = np.stack([x[i * 160:i * 160 + 400] * win for i in range(n_use)])
    spec = np.zeros((201, 3000))
    spec[:, :n_use] = (np.abs(np.fft.rfft(frames, axis=1)) ** 2).T
    ref = np.log10(np.maximum(W.mel_filterbank().astype(np.float64) @ spec, 1e-10))
    ref = (np.maximum(ref, ref.max() - 8.0) + 4.0) / 4.0
    assert np.abs(mel - ref).max() < 2e-3                      # the reference's f32 twiddle angles cost ~1e-3 here
    assert np.abs(mel[:, n_use:] - mel[0, -1]).max() == 0      # frames past the signal: floor value
    fb = W.mel_filterbank()
    assert fb.shape == (80, 201) and (fb >= 0).all() and fb[:, 0].sum() == 0
