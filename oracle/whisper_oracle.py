"""CPU restatement of the reference's Whisper path -- TEST INFRASTRUCTURE ONLY.

Nothing here is imported by the product (kjarni_amd/); only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of the benchmarks may use it.  Every function cites the reference lines it follows.

  log-mel front end       crates/kjarni-transformers/src/audio/mel.rs:60-262
  conv front end          mel.rs:265-391
  encoder (pre-norm)      cpu/encoder_decoder/cpu_encoder.rs:193-262, cpu/encoder/encoder_layer.rs:195-212
  decoder                 cpu/encoder_decoder/cpu_decoder.rs:216-516, decoder_cross_attn_layer.rs:123-151,
                          decoder_cross_attn.rs:48-120, encoder_decoder/decoder_self_attn.rs:52-146
  greedy transcription    crates/kjarni-models/src/models/whisper/transcriber.rs:85-460
  resampling, chunk loop  crates/kjarni/src/transcriber/model.rs:91-176, 333-356
  WAV decoding            crates/kjarni-transformers/src/audio/loader.rs:125-300 (hound 3.5 for the container)

Parity status: PINNED by the reference's model-free goldens (tests/test_whisper_oracle.py): conv front
end (mel.rs:2078-2119), cross attention (decoder_cross_attn.rs:194-437), cross decoder layer
(decoder_cross_attn_layer.rs:384-856), Whisper decoder (cpu_decoder.rs:871-934), Whisper encoder
scenario (cpu_encoder.rs:913-972), chunking / stitching (transcriber.rs:460-535).  The log-mel stage has
only shape tests in the reference (mel.rs:2121-2140): its values are pinned against an independent
float64 computation, not against reference outputs.
"""
from __future__ import annotations

import math
import struct
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import oracle as O

F32 = np.float32
MASK_VALUE = F32(-1e9)  # utils/masks.rs MASK_VALUE

SOT_TOKEN, EOT_TOKEN = 50258, 50257
TRANSCRIBE_TOKEN, TRANSLATE_TOKEN, NO_TIMESTAMPS_TOKEN = 50359, 50360, 50363
TIMESTAMP_BEGIN, FIRST_SPECIAL_TOKEN = 50364, 50257
TIMESTAMP_RESOLUTION = F32(0.02)
CHUNK_SECS = F32(30.0)
SAMPLE_RATE = 16000
CHUNK_SAMPLES = 480_000


# ----------------------------------------------------------------------------- log-mel (mel.rs:60-262)
def pad_reflect(audio: np.ndarray, pad: int) -> np.ndarray:
    """mel.rs:139-160 (numpy 'reflect', with the reference's clamps for very short inputs)."""
    n = len(audio)
    left = [audio[i if i < n else n - 1] for i in range(pad, 0, -1)]
    right = [audio[n - 2 - i if n >= 2 + i else 0] for i in range(pad)]
    return np.concatenate([np.asarray(left, F32), audio.astype(F32), np.asarray(right, F32)])


def mel_filterbank(sample_rate=16000, n_fft=400, n_mels=80, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """mel.rs:163-233: Slaney scale + Slaney normalisation, all in f32."""
    n_bins = n_fft // 2 + 1
    sr = F32(sample_rate)
    f_sp = F32(200.0) / F32(3.0)
    min_log_hz = F32(1000.0)
    min_log_mel = min_log_hz / f_sp
    logstep = F32(math.log(float(F32(6.4)))) / F32(27.0)  # 6.4_f32.ln()

    def ln(x):
        return F32(np.log(F32(x)))

    def hz_to_mel(hz):
        hz = F32(hz)
        return hz / f_sp if hz < min_log_hz else min_log_mel + ln(hz / min_log_hz) / logstep

    def mel_to_hz(mel):
        mel = F32(mel)
        return mel * f_sp if mel < min_log_mel else min_log_hz * F32(np.exp(F32(logstep * (mel - min_log_mel))))

    mel_min, mel_max = hz_to_mel(fmin), hz_to_mel(fmax)
    pts = [F32(mel_min + (mel_max - mel_min) * F32(i) / F32(n_mels + 1)) for i in range(n_mels + 2)]
    mel_f = np.asarray([mel_to_hz(m) for m in pts], F32)
    fdiff = (mel_f[1:] - mel_f[:-1]).astype(F32)
    fft_freqs = np.asarray([sr * F32(i) / F32(n_fft) for i in range(n_bins)], F32)
    w = np.zeros((n_mels, n_bins), F32)
    for i in range(n_mels):
        lower = (fft_freqs - mel_f[i]) / fdiff[i]
        upper = (mel_f[i + 2] - fft_freqs) / fdiff[i + 1]
        w[i] = np.maximum(F32(0.0), np.minimum(lower, upper))
        w[i] *= F32(2.0) / (mel_f[i + 2] - mel_f[i])
    return w


def hann_window(size: int) -> np.ndarray:
    """mel.rs:236-240 (periodic Hann, f32)."""
    i = np.arange(size, dtype=F32)
    return (F32(0.5) * (F32(1.0) - np.cos((F32(2.0) * F32(np.pi) * i / F32(size)).astype(F32)))).astype(F32)


def dft_tables(n: int) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin of mel.rs:251-254: angle = -2.0 * PI * (k*i) as f32 / n as f32, evaluated in f32."""
    k = np.arange(n // 2 + 1, dtype=np.int64)[:, None]
    i = np.arange(n, dtype=np.int64)[None, :]
    angle = ((F32(-2.0) * F32(np.pi)) * (k * i).astype(F32) / F32(n)).astype(F32)
    return np.cos(angle).astype(F32), np.sin(angle).astype(F32)


def log_mel(audio: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """compute_mel_spectrogram with MelConfig::whisper() (mel.rs:44-121): centred STFT (reflect pad 200),
    exactly 3000 frames (frames past the signal stay zero), power = |X|^2 via sqrt then square,
    mel filterbank, log10(max(x, 1e-10)), clamp to max - 8, (x + 4) / 4.  -> [n_mels, 3000]."""
    n_fft, hop, n_frames = 400, 160, 3000
    x = pad_reflect(np.asarray(audio, F32), n_fft // 2)
    win = hann_window(n_fft)
    cos_t, sin_t = dft_tables(n_fft)
    usable = 0
    while usable < n_frames and usable * hop + n_fft <= len(x):
        usable += 1
    idx = (np.arange(usable)[:, None] * hop + np.arange(n_fft)[None, :])
    frames = (x[idx] * win[None, :]).astype(F32)                    # [frames, 400]
    re = frames @ cos_t.T
    im = frames @ sin_t.T
    mag = np.sqrt((re * re + im * im).astype(F32)).astype(F32)
    spec = np.zeros((n_fft // 2 + 1, n_frames), F32)
    spec[:, :usable] = (mag * mag).T
    mel = mel_filterbank(n_mels=n_mels) @ spec
    log_spec = np.log10(np.maximum(mel, F32(1e-10))).astype(F32)
    mx = log_spec.max()
    return ((np.maximum(log_spec, mx - F32(8.0)) + F32(4.0)) / F32(4.0)).astype(F32)


# ----------------------------------------------------------------------------- conv front end (mel.rs:265-391)
def gelu_tanh(x: np.ndarray) -> np.ndarray:
    """mel.rs:374-376 (the conv front end uses the tanh form)."""
    x = x.astype(F32)
    return (x * F32(0.5) * (F32(1.0) + np.tanh((x * F32(0.7978845608) * (F32(1.0) + F32(0.044715) * x * x)).astype(F32)))).astype(F32)


def conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, padding: int) -> np.ndarray:
    """mel.rs:331-371.  x [B, Cin, T], w [Cout, Cin, K] -> [B, Cout, Tout]."""
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    Tout = (T + 2 * padding - K) // stride + 1
    xp = np.zeros((B, Cin, T + 2 * padding), F32)
    xp[:, :, padding:padding + T] = x
    cols = np.stack([xp[:, :, k:k + stride * Tout:stride] for k in range(K)], axis=2)   # [B, Cin, K, Tout]
    out = np.einsum("bckt,ock->bot", cols, w.astype(F32), optimize=True).astype(F32)
    return (out + b.astype(F32)[None, :, None]).astype(F32)


def sinusoidal_embeddings(max_len: int, dim: int) -> np.ndarray:
    """mel.rs:379-391 / cpu_decoder.rs:330-342."""
    e = np.zeros((max_len, dim), F32)
    pos = np.arange(max_len, dtype=F32)[:, None]
    i = np.arange(dim // 2, dtype=F32)[None, :]
    angle = (pos / np.power(F32(10000.0), (F32(2.0) * i / F32(dim)).astype(F32)).astype(F32)).astype(F32)
    e[:, 0::2][:, :dim // 2] = np.sin(angle)
    e[:, 1::2][:, :dim // 2] = np.cos(angle)
    return e


def conv_frontend(mel: np.ndarray, t: Dict[str, np.ndarray], prefix: str = "model.encoder",
                  max_positions: int = 1500) -> np.ndarray:
    """AudioConvFrontend::forward (mel.rs:303-328).  mel [B, n_mels, T] -> [B, T/2, hidden]."""
    x = gelu_tanh(conv1d(mel.astype(F32), t[f"{prefix}.conv1.weight"], t[f"{prefix}.conv1.bias"], 1, 1))
    x = gelu_tanh(conv1d(x, t[f"{prefix}.conv2.weight"], t[f"{prefix}.conv2.bias"], 2, 1))
    x = np.ascontiguousarray(x.transpose(0, 2, 1))
    pos = t.get(f"{prefix}.embed_positions.weight")
    if pos is None:
        pos = sinusoidal_embeddings(max_positions, x.shape[2])
    n = min(x.shape[1], pos.shape[0])
    x[:, :n, :] += pos[:n].astype(F32)
    return x


# ----------------------------------------------------------------------------- attention pieces
def _split_heads(x: np.ndarray, heads: int) -> np.ndarray:
    B, S, H = x.shape
    return x.reshape(B, S, heads, H // heads).transpose(0, 2, 1, 3)


def _softmax_rows(s: np.ndarray) -> np.ndarray:
    """activations.rs:223-242 per row: exp(x - max), * (1/sum) when sum > 0."""
    m = s.max(axis=-1, keepdims=True)
    e = np.exp((s - m).astype(F32)).astype(F32)
    z = e.sum(axis=-1, keepdims=True, dtype=F32)
    inv = np.where(z > 0, F32(1.0) / np.where(z > 0, z, F32(1.0)), F32(1.0)).astype(F32)
    return (e * inv).astype(F32)


def cross_attention(hidden, k_t, v, p: Dict[str, np.ndarray], heads: int, mask=None) -> np.ndarray:
    """DecoderCrossAttention::forward (decoder_cross_attn.rs:71-120).  k_t [B,h,d,Sk], v [B,h,Sk,d]."""
    B, S, H = hidden.shape
    q = _split_heads(O.linear(hidden, p["q_w"], p.get("q_b")), heads)
    scores = np.matmul(q, k_t).astype(F32) * F32(1.0 / math.sqrt(H // heads))
    if mask is not None and mask.shape[1] == scores.shape[3]:
        scores = np.where(mask[:, None, None, :] == 0, MASK_VALUE, scores).astype(F32)
    ctx = np.matmul(_softmax_rows(scores), v).astype(F32)
    ctx = np.ascontiguousarray(ctx.transpose(0, 2, 1, 3)).reshape(B, S, H)
    return O.linear(ctx, p["o_w"], p.get("o_b"))


def precompute_cross_kv(enc: np.ndarray, p: Dict[str, np.ndarray], heads: int):
    """decoder_cross_attn.rs:48-69."""
    k = _split_heads(O.linear(enc, p["k_w"], p.get("k_b")), heads)
    v = _split_heads(O.linear(enc, p["v_w"], p.get("v_b")), heads)
    return np.ascontiguousarray(k.transpose(0, 1, 3, 2)), np.ascontiguousarray(v)


def self_attention(hidden, p: Dict[str, np.ndarray], heads: int, past_kv=None, mask=None):
    """DecoderSelfAttention::forward (decoder_self_attn.rs:52-146): returns (out, new_k, new_v) with
    new_k / new_v the [B, S, H] projections of THIS call's tokens."""
    B, S, H = hidden.shape
    q = _split_heads(O.linear(hidden, p["q_w"], p.get("q_b")), heads)
    k_new = O.linear(hidden, p["k_w"], p.get("k_b"))
    v_new = O.linear(hidden, p["v_w"], p.get("v_b"))
    cache_len = 0
    if past_kv is not None:
        cache_len = past_kv[0].shape[1]
        full_k = np.concatenate([past_kv[0], k_new], axis=1)
        full_v = np.concatenate([past_kv[1], v_new], axis=1)
    else:
        full_k, full_v = k_new, v_new
    kh, vh = _split_heads(full_k, heads), _split_heads(full_v, heads)
    scores = np.matmul(q, kh.transpose(0, 1, 3, 2)).astype(F32) * F32(1.0 / math.sqrt(H // heads))
    total = scores.shape[3]
    # apply_attention_mask (utils/linear_algebra.rs:921-942): a [rows, total] mask is read as
    # [batch, keys]; it only applies when rows broadcasts against the batch
    if mask is not None and mask.shape[1] == total and mask.shape[0] in (1, B):
        scores = np.where(mask[:, None, None, :] == 0, MASK_VALUE, scores).astype(F32)
    if S > 1:  # apply_causal_mask (utils/masks.rs:103-113)
        qpos = cache_len + np.arange(S)[:, None]
        scores = np.where(np.arange(total)[None, :] > qpos, MASK_VALUE, scores).astype(F32)
    ctx = np.matmul(_softmax_rows(scores), vh).astype(F32)
    ctx = np.ascontiguousarray(ctx.transpose(0, 2, 1, 3)).reshape(B, S, H)
    return O.linear(ctx, p["o_w"], p.get("o_b")), k_new, v_new


def _gelu_erf(x: np.ndarray) -> np.ndarray:
    flat = np.ascontiguousarray(x, F32).reshape(-1)
    from scipy.special import erf
    return (F32(0.5) * flat * (F32(1.0) + erf((flat * F32(0.7071067811865475)).astype(F32)).astype(F32))).astype(F32).reshape(x.shape)


def ffn(x, p: Dict[str, np.ndarray]) -> np.ndarray:
    return O.linear(_gelu_erf(O.linear(x, p["fc1_w"], p["fc1_b"])), p["fc2_w"], p["fc2_b"])


def create_causal_mask(q_len: int, total_len: int) -> np.ndarray:
    """utils/masks.rs:121-142."""
    past = total_len - q_len
    return (np.arange(total_len)[None, :] <= (past + np.arange(q_len))[:, None]).astype(F32)


# ----------------------------------------------------------------------------- decoder layer
def cross_decoder_layer(hidden, enc, p: Dict[str, Dict[str, np.ndarray]], heads: int, pre_norm: bool, eps: float,
                        self_mask=None, cross_mask=None, past_kv=None, cross_kv=None):
    """CrossDecoderLayer::forward (decoder_cross_attn_layer.rs:55-151)."""
    ln = lambda x, q: O.layer_norm(x, p[q]["g"], p[q]["b"], eps)  # noqa: E731
    if cross_kv is None:
        cross_kv = precompute_cross_kv(enc, p["cross"], heads)
    if pre_norm:
        a, nk, nv = self_attention(ln(hidden, "self_ln"), p["self"], heads, past_kv, self_mask)
        h = (hidden + a).astype(F32)
        h = (h + cross_attention(ln(h, "cross_ln"), cross_kv[0], cross_kv[1], p["cross"], heads, cross_mask)).astype(F32)
        h = (h + ffn(ln(h, "ffn_ln"), p["ffn"])).astype(F32)
    else:
        a, nk, nv = self_attention(hidden, p["self"], heads, past_kv, self_mask)
        h = ln((hidden + a).astype(F32), "self_ln")
        h = ln((h + cross_attention(h, cross_kv[0], cross_kv[1], p["cross"], heads, cross_mask)).astype(F32), "cross_ln")
        h = ln((h + ffn(h, p["ffn"])).astype(F32), "ffn_ln")
    return h, (nk, nv)


# ----------------------------------------------------------------------------- the model
class WhisperOracle:
    """WhisperModel restated over a {hf_tensor_name: ndarray} dict + config dict
    (crates/kjarni-models/src/models/whisper/{config,model,transcriber}.rs)."""

    def __init__(self, tensors: Dict[str, np.ndarray], config: dict):
        self.t = {k: np.ascontiguousarray(v, F32) for k, v in tensors.items()}
        self.c = config
        self.H = config["d_model"]
        self.heads = config["encoder_attention_heads"]     # metadata(): decoder layers use the same count
        self.eps = 1e-5
        self.enc_layers = [self._enc_layer(i) for i in range(config["encoder_layers"])]
        self.dec_layers = [self._dec_layer(i) for i in range(config["decoder_layers"])]

    def _attn(self, pre: str) -> Dict[str, np.ndarray]:
        t = self.t
        d = dict(q_w=t[f"{pre}.q_proj.weight"], k_w=t[f"{pre}.k_proj.weight"], v_w=t[f"{pre}.v_proj.weight"],
                 o_w=t[f"{pre}.out_proj.weight"])
        for s, n in (("q_b", "q_proj"), ("k_b", "k_proj"), ("v_b", "v_proj"), ("o_b", "out_proj")):
            if f"{pre}.{n}.bias" in t:                     # with_optional_bias: Whisper's k_proj has none
                d[s] = t[f"{pre}.{n}.bias"]
        return d

    def _ln(self, pre: str):
        return dict(g=self.t[f"{pre}.weight"], b=self.t[f"{pre}.bias"])

    def _ffn(self, pre: str):
        t = self.t
        return dict(fc1_w=t[f"{pre}.fc1.weight"], fc1_b=t[f"{pre}.fc1.bias"], fc2_w=t[f"{pre}.fc2.weight"],
                    fc2_b=t[f"{pre}.fc2.bias"])

    def _enc_layer(self, i: int):
        pre = f"model.encoder.layers.{i}"
        return {"self": self._attn(f"{pre}.self_attn"), "self_ln": self._ln(f"{pre}.self_attn_layer_norm"),
                "ffn": self._ffn(pre), "ffn_ln": self._ln(f"{pre}.final_layer_norm")}

    def _dec_layer(self, i: int):
        pre = f"model.decoder.layers.{i}"
        return {"self": self._attn(f"{pre}.self_attn"), "self_ln": self._ln(f"{pre}.self_attn_layer_norm"),
                "cross": self._attn(f"{pre}.encoder_attn"), "cross_ln": self._ln(f"{pre}.encoder_attn_layer_norm"),
                "ffn": self._ffn(pre), "ffn_ln": self._ln(f"{pre}.final_layer_norm")}

    # -- encoder ------------------------------------------------------------------------------
    def encoder_forward(self, hidden: np.ndarray) -> np.ndarray:
        """Seq2SeqCPUEncoder::forward on hidden input (cpu_encoder.rs:193-262), pre-norm layers
        (encoder_layer.rs:195-212), mask of ones, final LayerNorm."""
        h = hidden.astype(F32)
        for p in self.enc_layers:
            n = O.layer_norm(h, p["self_ln"]["g"], p["self_ln"]["b"], self.eps)
            sp = p["self"]
            q, k, v = O.linear(n, sp["q_w"], sp.get("q_b")), O.linear(n, sp["k_w"], sp.get("k_b")), O.linear(n, sp["v_w"], sp.get("v_b"))
            ctx = O.attention(q, k, v, np.ones(h.shape[:2], F32), self.heads)
            h = (h + O.linear(ctx, sp["o_w"], sp.get("o_b"))).astype(F32)
            n = O.layer_norm(h, p["ffn_ln"]["g"], p["ffn_ln"]["b"], self.eps)
            h = (h + ffn(n, p["ffn"])).astype(F32)
        return O.layer_norm(h, self.t["model.encoder.layer_norm.weight"], self.t["model.encoder.layer_norm.bias"], self.eps)

    def encode_mel(self, mel: np.ndarray) -> np.ndarray:
        """WhisperModel::encode_mel (transcriber.rs:122-141).  mel [n_mels, T] -> [1, T/2, H]."""
        return self.encoder_forward(conv_frontend(mel[None], self.t, max_positions=self.c["max_source_positions"]))

    # -- decoder ------------------------------------------------------------------------------
    def decoder_embed(self, ids: np.ndarray, offset: int) -> np.ndarray:
        """Embeddings::forward(ids, None, offset, scale_embedding) with the learned decoder positions."""
        scale = bool(self.c.get("scale_embedding", False))
        return O.embed(ids, None, self.t["model.decoder.embed_tokens.weight"],
                       self.t["model.decoder.embed_positions.weight"], None, pos_offset=offset, scale=scale)

    def precompute_cross_kv(self, enc: np.ndarray):
        return [precompute_cross_kv(enc, p["cross"], self.heads) for p in self.dec_layers]

    def decoder_forward(self, ids: np.ndarray, enc: np.ndarray, cache: List[Optional[Tuple[np.ndarray, np.ndarray]]],
                        cross_kv) -> np.ndarray:
        """CpuCrossDecoder::forward + cache update as decode_chunk drives it (cpu_decoder.rs:399-516,
        transcriber.rs:168-183): returns the final-normed hidden states; `cache` is extended in place."""
        offset = 0 if cache[0] is None else cache[0][0].shape[1]
        h = self.decoder_embed(ids, offset)
        S = h.shape[1]
        self_mask = create_causal_mask(S, offset + S) * np.ones((1, S), F32)  # combine_masks: broadcast quirk kept
        for i, p in enumerate(self.dec_layers):
            h, (nk, nv) = cross_decoder_layer(h, enc, p, self.heads, True, self.eps, self_mask,
                                              np.ones((1, enc.shape[1]), F32), cache[i], cross_kv[i])
            cache[i] = (nk, nv) if cache[i] is None else (np.concatenate([cache[i][0], nk], 1),
                                                          np.concatenate([cache[i][1], nv], 1))
        return O.layer_norm(h, self.t["model.decoder.layer_norm.weight"], self.t["model.decoder.layer_norm.bias"], self.eps)

    def logits(self, hidden: np.ndarray) -> np.ndarray:
        """lm_head = the shared token embedding, no bias (config.rs:83-87)."""
        w = self.t.get("proj_out.weight", self.t["model.decoder.embed_tokens.weight"])
        return O.linear(hidden, w, None)

    # -- greedy transcription -------------------------------------------------------------------
    @staticmethod
    def pick_token(logits: np.ndarray, timestamps: bool, eos: int) -> int:
        """transcriber.rs:243-270: argmax over allowed ids; Iterator::max_by keeps the LAST maximum."""
        ids = np.arange(logits.shape[0])
        ok = (ids < FIRST_SPECIAL_TOKEN) | (ids == eos)
        if timestamps:
            ok |= ids >= TIMESTAMP_BEGIN
        cand = ids[ok]
        vals = logits[ok]
        best = vals.max()
        return int(cand[np.nonzero(vals == best)[0][-1]])

    def prompt_tokens(self, language_token: int, translate: bool, timestamps: bool) -> List[int]:
        """transcriber.rs:273-293."""
        toks = [SOT_TOKEN, language_token, TRANSLATE_TOKEN if translate else TRANSCRIBE_TOKEN]
        if not timestamps:
            toks.append(NO_TIMESTAMPS_TOKEN)
        return toks

    def decode_chunk_ids(self, enc: np.ndarray, language_token: int = 50259, translate=False, timestamps=False,
                         max_tokens: int = 448, eos: Optional[int] = None) -> List[int]:
        """decode_chunk (transcriber.rs:144-240) up to the generated id list."""
        eos = self.c.get("eos_token_id", EOT_TOKEN) if eos is None else eos
        prompt = self.prompt_tokens(language_token, translate, timestamps)
        cross = self.precompute_cross_kv(enc)
        cache: List = [None] * len(self.dec_layers)
        h = self.decoder_forward(np.asarray([prompt], np.uint32), enc, cache, cross)
        nxt = self.pick_token(self.logits(h[:, -1:, :])[0, 0], timestamps, eos)
        out = [nxt]
        for _ in range(max_tokens):
            if nxt == eos:
                break
            h = self.decoder_forward(np.asarray([[nxt]], np.uint32), enc, cache, cross)
            nxt = self.pick_token(self.logits(h)[0, 0], timestamps, eos)
            out.append(nxt)
        return out


# ----------------------------------------------------------------------------- host-side pieces
def chunk_audio(samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> List[np.ndarray]:
    """transcriber.rs:87-119."""
    size = int(F32(30.0) * F32(sample_rate))
    if len(samples) == 0:
        return []
    out = []
    for off in range(0, len(samples), size) if len(samples) > size else [0]:
        c = np.zeros(size, F32)
        part = samples[off:off + size]
        c[:len(part)] = part
        out.append(c)
    return out


def resample_linear(samples: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    """kjarni/src/transcriber/model.rs:333-356 (f64 positions, f32 interpolation)."""
    if from_rate == to_rate or len(samples) == 0:
        return samples.astype(F32)
    ratio = to_rate / from_rate
    out_len = int(math.ceil(len(samples) * ratio))
    src = np.arange(out_len, dtype=np.float64) / ratio
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, len(samples) - 1)
    frac = (src - lo).astype(F32)
    s = samples.astype(F32)
    return (s[lo] + (s[hi] - s[lo]) * frac).astype(F32)


def parse_timestamp_segments(ids: Sequence[int], decode: Callable[[List[int]], str], offset: float):
    """transcriber.rs:340-409."""
    segs, start, cur = [], None, []
    for i in ids:
        if i >= TIMESTAMP_BEGIN:
            t = float(F32(i - TIMESTAMP_BEGIN) * TIMESTAMP_RESOLUTION + F32(offset))
            if start is None:
                start = t
            else:
                text = decode([x for x in cur if x < FIRST_SPECIAL_TOKEN])
                if text.strip():
                    segs.append(dict(start=start, end=t, text=text))
                start, cur = t, []
        elif i < FIRST_SPECIAL_TOKEN:
            cur.append(i)
    if start is not None and cur:
        text = decode([x for x in cur if x < FIRST_SPECIAL_TOKEN])
        if text.strip():
            segs.append(dict(start=start, end=float(F32(start) + CHUNK_SECS), text=text))
    return segs


def finalize_chunk(ids: Sequence[int], decode: Callable[[List[int]], str], timestamps: bool, offset: float):
    """transcriber.rs:301-338."""
    if timestamps:
        segs = parse_timestamp_segments(ids, decode, offset)
        return dict(segments=segs, text="".join(s["text"] for s in segs))
    text = decode([i for i in ids if i < FIRST_SPECIAL_TOKEN])
    return dict(segments=[dict(start=float(F32(offset)), end=float(F32(offset) + CHUNK_SECS), text=text)], text=text)


def is_chunk_boundary(t: float) -> bool:
    """transcriber.rs:452-455 (f32 remainder)."""
    rem = float(np.fmod(F32(t), CHUNK_SECS))
    return rem < 0.02 or float(CHUNK_SECS - F32(rem)) < 0.02


def stitch(chunks: Sequence[dict]):
    """transcriber.rs:412-449."""
    text = "".join(c["text"] for c in chunks)
    merged: List[dict] = []
    for c in chunks:
        for s in c["segments"]:
            if merged and abs(float(F32(merged[-1]["end"]) - F32(s["start"]))) < 0.02 and is_chunk_boundary(merged[-1]["end"]):
                merged[-1]["end"] = s["end"]
                merged[-1]["text"] += s["text"]
            else:
                merged.append(dict(s))
    return text, merged


def read_wav(data: bytes, mono: bool = True, target_rate: int = SAMPLE_RATE):
    """loader.rs:125-208 over a RIFF/WAVE image (PCM 8/16/24/32-bit, IEEE float 32): samples scaled by
    2^(bits-1), channels averaged, linear resampling to `target_rate`.  Returns (samples, original_rate)."""
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body, 0)
            if fmt[0] == 0xFFFE and len(body) >= 26:       # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                fmt = (struct.unpack_from("<H", body, 24)[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    tag, channels, rate, _, _, bits = fmt
    if tag == 3:
        x = np.frombuffer(pcm, "<f4").astype(F32)
    elif bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.int16) - 128).astype(F32) / F32(128.0)
    elif bits == 16:
        x = np.frombuffer(pcm, "<i2").astype(F32) / F32(32768.0)
    elif bits == 24:
        b = np.frombuffer(pcm, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = v.astype(F32) / F32(1 << 23)
    else:
        x = np.frombuffer(pcm, "<i4").astype(F32) / F32(2147483648.0)
    if mono and channels > 1:
        x = x[:len(x) // channels * channels].reshape(-1, channels)
        acc = np.zeros(x.shape[0], F32)
        for c in range(channels):
            acc = (acc + x[:, c]).astype(F32)
        x = (acc / F32(channels)).astype(F32)
    return resample_linear(x, rate, target_rate), rate
