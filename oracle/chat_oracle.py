"""CPU restatement of the reference's chat path around the decoder forward -- TEST INFRASTRUCTURE ONLY.

Nothing under kjarni_amd/ imports this module; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

  templates             crates/kjarni-transformers/src/chat/llama3.rs:38-96, chatml.rs:15-48, mistral.rs:16-80
  Conversation/History  crates/kjarni-transformers/src/chat/templates.rs:51-131, crates/kjarni/src/chat/types.rs:197-238
  Chat                  crates/kjarni/src/chat/model.rs:30-352
  config resolution     crates/kjarni/src/generation/resolution.rs:8-85
  model defaults        crates/kjarni-models/src/models/llama/model.rs:373-396, qwen/model.rs:261-282,
                        crates/kjarni-transformers/src/common/mod.rs:26-35, 297-349
  sampling              crates/kjarni-transformers/src/common/sampling.rs:81-184, activations.rs:223-242
  generation loop       crates/kjarni-transformers/src/decoder/generator.rs:141-163, 228-381

Pinned by the reference's own unit tests for these files, restated in tests/test_chat_oracle.py (template strings of
mistral.rs:98-147 and chatml.rs:56-60, the resolution cases of resolution.rs:112-197, the filter cases of
sampling.rs:311-368, HFGenerationDefaults of common/mod.rs:374-420).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, replace
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

SYSTEM, USER, ASSISTANT = "system", "user", "assistant"
Message = Tuple[str, str]  # (role, content)


# ---- templates -----------------------------------------------------------------------------------------

def apply_llama3(conv: Sequence[Message]) -> str:
    """Llama3ChatTemplate::for_generation().apply (llama3.rs:57-75)."""
    p = "<|begin_of_text|>"
    for role, content in conv:
        p += f"<|start_header_id|>{role}<|end_header_id|>\n\n{content}<|eot_id|>"
    return p + "<|start_header_id|>assistant<|end_header_id|>\n\n"


def apply_chatml(conv: Sequence[Message]) -> str:
    """ChatMLTemplate::new().apply (chatml.rs:15-34)."""
    p = ""
    for role, content in conv:
        p += f"<|im_start|>{role}\n{content}<|im_end|>\n"
    return p + "<|im_start|>assistant\n"


def apply_mistral(conv: Sequence[Message]) -> str:
    """MistralChatTemplate::new().apply (mistral.rs:16-75): the system prompt rides on the first user turn."""
    if not conv:
        return ""
    p = "<s>"
    msgs = list(conv)
    system = None
    if msgs[0][0] == SYSTEM:
        system = msgs.pop(0)[1]
    first_user = True
    for role, content in msgs:
        if role == USER:
            p += "[INST] "
            if first_user:
                if system is not None:
                    p += system + "\n\n"
                first_user = False
            p += content + " [/INST]"
        elif role == ASSISTANT:
            p += " " + content + "</s>"
    return p


TEMPLATES = {"llama3": apply_llama3, "chatml": apply_chatml, "mistral": apply_mistral}
STOP_SEQUENCES = {"llama3": ["<|eot_id|>", "<|end_of_text|>"], "chatml": ["<|im_end|>", "<|endoftext|>"], "mistral": ["</s>"]}
DEFAULT_SYSTEM = {"llama3": "You are a helpful, harmless, and honest assistant.", "chatml": "You are a helpful assistant.", "mistral": None}


# ---- generation config ---------------------------------------------------------------------------------

@dataclass
class GenerationConfig:
    max_new_tokens: Optional[int]
    max_length: int
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    add_bos_token: bool = True
    strategy: str = "greedy"  # greedy | sample | beam_search
    temperature: float = 0.7   # SamplingParams (common/mod.rs:26-35)
    top_k: Optional[int] = 50
    top_p: Optional[float] = 0.9
    min_p: Optional[float] = 0.1
    num_beams: int = 4
    length_penalty: float = 1.0


@dataclass
class Overrides:
    temperature: Optional[float] = None
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    min_p: Optional[float] = None
    repetition_penalty: Optional[float] = None
    no_repeat_ngram_size: Optional[int] = None
    max_new_tokens: Optional[int] = None
    do_sample: Optional[bool] = None
    num_beams: Optional[int] = None
    length_penalty: Optional[float] = None


def _first(*vals):
    for v in vals:
        if v is not None:
            return v
    return None


def resolve_generation_config(defaults: GenerationConfig, user: Overrides, runtime: Overrides) -> GenerationConfig:
    """resolution.rs:8-85."""
    c = replace(defaults)
    beams = _first(runtime.num_beams, user.num_beams)
    do_sample = _first(runtime.do_sample, user.do_sample)
    if beams is not None and beams > 1:
        if c.strategy != "beam_search":
            c.num_beams, c.length_penalty = 4, 1.0
        c.strategy = "beam_search"
    elif do_sample is False:
        c.strategy = "greedy"
    elif do_sample is True:
        if c.strategy != "sample":
            c.temperature, c.top_k, c.top_p, c.min_p = 0.7, 50, 0.9, 0.1
        c.strategy = "sample"
    v = _first(runtime.max_new_tokens, user.max_new_tokens)
    if v is not None:
        c.max_new_tokens = v
    v = _first(runtime.repetition_penalty, user.repetition_penalty)
    if v is not None:
        c.repetition_penalty = v
    v = _first(runtime.no_repeat_ngram_size, user.no_repeat_ngram_size)
    if v is not None:
        c.no_repeat_ngram_size = v
    if c.strategy == "sample":
        for name in ("temperature", "top_k", "top_p", "min_p"):
            v = _first(getattr(runtime, name), getattr(user, name))
            if v is not None:
                setattr(c, name, v)
    elif c.strategy == "beam_search":
        if beams is not None:
            c.num_beams = beams
        v = _first(runtime.length_penalty, user.length_penalty)
        if v is not None:
            c.length_penalty = v
    return c


def hf_generation_defaults(text: str, max_seq_len: int) -> Optional[GenerationConfig]:
    """HFGenerationDefaults::from_json + into_generation_config (common/mod.rs:297-349); None when serde would fail."""
    try:
        j = json.loads(text)
    except ValueError:
        return None
    if not isinstance(j, dict):
        return None

    def field(name, kinds, nullable):
        if name not in j:
            return None
        v = j[name]
        if v is None:
            if nullable:
                return None
            raise TypeError(name)
        if isinstance(v, bool) and bool not in kinds:
            raise TypeError(name)
        if not isinstance(v, kinds):
            raise TypeError(name)
        return v

    try:
        do_sample = field("do_sample", (bool,), False) or False
        temperature = field("temperature", (int, float), False)
        top_p = field("top_p", (int, float), True)
        top_k = field("top_k", (int,), True)
        max_new = field("max_new_tokens", (int,), True)
        max_len = field("max_length", (int,), True)
        rep = field("repetition_penalty", (int, float), True)
        field("decoder_start_token_id", (int,), True)
    except TypeError:
        return None
    c = GenerationConfig(max_new_tokens=max_new, max_length=max_len if max_len is not None else max_seq_len,
                         repetition_penalty=float(rep) if rep is not None else 1.0, no_repeat_ngram_size=0, add_bos_token=True)
    if do_sample:
        c.strategy = "sample"
        c.temperature = float(temperature) if temperature is not None else 1.0
        c.top_k, c.top_p, c.min_p = top_k, (float(top_p) if top_p is not None else None), None
    else:
        c.strategy = "greedy"
    return c


def model_default_generation_config(model_type: str, max_pos: int, generation_config_json: Optional[str]) -> GenerationConfig:
    if generation_config_json is not None and model_type != "mistral":  # mistral/model.rs:236 never looks at the file
        c = hf_generation_defaults(generation_config_json, max_pos)
        if c is not None:
            return c
    if model_type == "mistral":  # mistral/model.rs:236-252
        return GenerationConfig(max_new_tokens=512, max_length=max_pos, repetition_penalty=1.15, add_bos_token=True, strategy="sample",
                                temperature=0.7, top_k=40, top_p=0.9, min_p=0.05)
    if model_type == "qwen2":  # qwen/model.rs:267-281
        return GenerationConfig(max_new_tokens=512, max_length=max_pos, repetition_penalty=1.1, add_bos_token=False, strategy="sample",
                                temperature=0.7, top_k=40, top_p=0.8, min_p=0.05)
    return GenerationConfig(max_new_tokens=256, max_length=max_pos, repetition_penalty=1.0, add_bos_token=True, strategy="sample",
                            temperature=0.6, top_k=None, top_p=0.9, min_p=0.05)  # llama/model.rs:381-395


MODE_TEMPERATURE = {"default": 0.7, "creative": 0.9, "reasoning": 0.3}  # chat/types.rs:129-144
MODE_MAX_TOKENS = {"default": 512, "creative": 1024, "reasoning": 2048}


def chat_generation_config(model_type: str, max_pos: int, generation_config_json: Optional[str], mode: Optional[str],
                           runtime: Overrides = Overrides()) -> GenerationConfig:
    """Chat::from_builder + Generator::generate_with_config: defaults -> mode overrides (as user) -> runtime."""
    defaults = model_default_generation_config(model_type, max_pos, generation_config_json)
    user = Overrides()
    if mode is not None:
        user = Overrides(temperature=MODE_TEMPERATURE[mode], max_new_tokens=MODE_MAX_TOKENS[mode])
    built = resolve_generation_config(defaults, user, Overrides())
    return resolve_generation_config(built, user, runtime)


# ---- sampling ------------------------------------------------------------------------------------------

NEG_INF = np.float32(-np.inf)


def softmax_inplace(x: np.ndarray) -> np.ndarray:
    """activations.rs:223-242: f32, running sum in index order, multiply by the reciprocal."""
    x = x.astype(np.float32)
    mx = np.float32(-np.inf)
    for v in x:
        mx = max(mx, v)
    e = np.exp((x - mx).astype(np.float32)).astype(np.float32)
    s = np.float32(0.0)
    for v in e:
        s = np.float32(s + v)
    if s > 0:
        e = (e * np.float32(np.float32(1.0) / s)).astype(np.float32)
    return e


def _sorted_desc(logits: np.ndarray) -> np.ndarray:
    # sort_by(|a, b| logits[b].partial_cmp(logits[a])): stable, descending by value
    return np.argsort(-logits.astype(np.float64), kind="stable")


def top_k_filtering(logits: np.ndarray, k: int) -> np.ndarray:
    out = logits.astype(np.float32).copy()
    if k >= out.size:
        return out  # the reference would index past the end; a k >= vocab is a no-op here
    out[_sorted_desc(out)[k:]] = NEG_INF
    return out


def top_p_filtering(logits: np.ndarray, p: float) -> np.ndarray:
    out = logits.astype(np.float32).copy()
    order = _sorted_desc(out)
    probs = softmax_inplace(out)
    cumulative = np.float32(0.0)
    for i, idx in enumerate(order):
        cumulative = np.float32(cumulative + probs[idx])
        if cumulative > np.float32(p):
            out[order[i + 1:]] = NEG_INF
            break
    return out


def min_p_filtering(logits: np.ndarray, min_p: float) -> np.ndarray:
    out = logits.astype(np.float32).copy()
    probs = softmax_inplace(out)
    cutoff = np.float32(max(np.float32(0.0), probs.max()) * np.float32(min_p))
    out[probs < cutoff] = NEG_INF
    return out


def sampling_distribution(logits: np.ndarray, temperature: float, top_k: Optional[int], top_p: Optional[float],
                          min_p: Optional[float]) -> np.ndarray:
    """sample_token up to the draw (sampling.rs:89-108)."""
    lg = logits.astype(np.float32).copy()
    if top_k is not None:
        lg = top_k_filtering(lg, top_k)
    if top_p is not None:
        lg = top_p_filtering(lg, top_p)
    if min_p is not None:
        lg = min_p_filtering(lg, min_p)
    temp = np.float32(1.0) if temperature < 1e-5 else np.float32(temperature)
    return softmax_inplace((lg / temp).astype(np.float32))


def sample_from_probs(probs: np.ndarray, uniform: float) -> int:
    cumulative = np.float32(0.0)
    u = np.float32(uniform)
    for i, p in enumerate(probs.astype(np.float32)):
        cumulative = np.float32(cumulative + p)
        if cumulative >= u:
            return i
    return len(probs) - 1


def sample_token(logits: np.ndarray, config: GenerationConfig, uniform: Optional[float]) -> int:
    if config.strategy == "greedy":
        best = 0
        for i in range(1, len(logits)):  # max_by keeps the last of equal maxima
            if logits[i] >= logits[best]:
                best = i
        return best
    if config.strategy == "sample":
        return sample_from_probs(sampling_distribution(logits, config.temperature, config.top_k, config.top_p, config.min_p), uniform)
    raise ValueError("Beam search is not supported in this generator.")


# ---- Chat ----------------------------------------------------------------------------------------------

RUST_WHITESPACE = "\t\n\x0b\x0c\r \x85\xa0\u1680" + "".join(chr(c) for c in range(0x2000, 0x200B)) + "\u2028\u2029\u202f\u205f\u3000"


def rust_trim(s: str) -> str:
    return s.strip(RUST_WHITESPACE)


class ChatOracle:
    """Chat + Generator + run_generation_loop over an LlmOracle-like model and a tokenizer with encode/decode/token_to_id."""

    def __init__(self, model, tokenizer, template: str, model_type: str, max_pos: int, bos_id: Optional[int], eos_ids: Sequence[int],
                 generation_config_json: Optional[str] = None, system_prompt: Optional[str] = None, mode: str = "default"):
        self.model, self.tokenizer, self.template = model, tokenizer, template
        self.model_type, self.max_pos, self.bos_id = model_type, max_pos, bos_id
        self.generation_config_json, self.system_prompt, self.mode = generation_config_json, system_prompt, mode
        self.stop_ids = set()
        if eos_ids:
            self.stop_ids.add(eos_ids[0])  # models/base.rs:261-271
        eot = tokenizer.token_to_id("<|eot_id|>")
        if eot is not None:
            self.stop_ids.add(eot)

    def create_conversation(self) -> List[Message]:
        if self.system_prompt is not None:
            return [(SYSTEM, self.system_prompt)]
        d = DEFAULT_SYSTEM[self.template]
        return [(SYSTEM, d)] if d is not None else []

    def history_to_conversation(self, history: Sequence[Message]) -> List[Message]:
        conv: List[Message] = []
        has_system = False
        for role, content in history:
            if role == SYSTEM:
                conv = [(SYSTEM, content)]
                has_system = True
            else:
                conv.append((role, content))
        if not has_system and self.system_prompt is not None:
            return [(SYSTEM, self.system_prompt)] + [m for m in conv if m[0] != SYSTEM]
        return conv

    def format_prompt(self, conv: Sequence[Message]) -> str:
        return TEMPLATES[self.template](conv)

    def resolve(self, runtime: Overrides = Overrides()) -> GenerationConfig:
        return chat_generation_config(self.model_type, self.max_pos, self.generation_config_json, self.mode, runtime)

    def encode(self, prompt: str, config: GenerationConfig) -> List[int]:
        ids = list(self.tokenizer.encode(prompt))[: self.max_pos]
        if config.add_bos_token and self.bos_id is not None and (not ids or ids[0] != self.bos_id):
            ids.insert(0, self.bos_id)
        return ids

    def stream(self, prompt: str, runtime: Overrides = Overrides(), uniforms: Optional[Callable[[], float]] = None,
               context_limit: Optional[int] = None) -> List[Tuple[int, str]]:
        """run_generation_loop: (id, text) of every generated token."""
        from oracle.llm_oracle import apply_no_repeat_ngram, apply_repetition_penalty  # noqa: WPS433
        config = self.resolve(runtime)
        tokens = self.encode(prompt, config)
        if not tokens:
            raise ValueError("cannot generate from empty prompt")
        limit = self.max_pos if context_limit is None else min(self.max_pos, context_limit)
        max_len = len(tokens) + config.max_new_tokens if config.max_new_tokens is not None else config.max_length
        max_new = config.max_new_tokens if config.max_new_tokens is not None else max_len - len(tokens)
        cache = self.model.new_cache()
        logits = self.model.logits(self.model.forward(tokens, cache)[0, -1])
        all_tokens = list(tokens)
        out: List[Tuple[int, str]] = []
        for _ in range(max_new):
            if len(all_tokens) >= limit or len(all_tokens) >= max_len:
                break
            lg = np.array(logits, np.float32)
            if config.repetition_penalty != 1.0:
                apply_repetition_penalty(lg, all_tokens, config.repetition_penalty)  # in place
            if config.no_repeat_ngram_size > 0:
                apply_no_repeat_ngram(lg, all_tokens, config.no_repeat_ngram_size)
            nxt = sample_token(lg, config, uniforms() if config.strategy == "sample" else None)
            if nxt in self.stop_ids:
                break
            all_tokens.append(nxt)
            out.append((nxt, self.tokenizer.decode([nxt], skip_special_tokens=False)))
            if len(all_tokens) >= limit or len(all_tokens) >= max_len:
                break
            logits = self.model.logits(self.model.forward([nxt], cache)[0, -1])
        return out

    def generate(self, prompt: str, runtime: Overrides = Overrides(), uniforms=None, context_limit=None) -> str:
        """Chat::generate (model.rs:283-303)."""
        cleaned = rust_trim("".join(t for _, t in self.stream(prompt, runtime, uniforms, context_limit)))
        for stop in STOP_SEQUENCES[self.template]:
            if cleaned.endswith(stop):
                cleaned = rust_trim(cleaned[: -len(stop)])
        return cleaned
