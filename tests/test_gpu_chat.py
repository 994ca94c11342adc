"""kjarni_chat_* end to end on the GPU against oracle/chat_oracle.py driving oracle/llm_oracle.py and
oracle/bpe_oracle.py: prompts, token ids, per-token texts, cleaned replies, histories, stop / length / context limits,
callbacks and cancellation, sampling support and seeding, error codes."""
import json
import os
import shutil

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
LLAMA_SPECIAL = {"bos": 700, "eot": 704, "end_of_text": 701, "eom": 705}


def _make(tmp, name, base, tokenizer, hf=None, **over):
    d = os.path.join(str(tmp), name)
    cfg, tensors = synth.llm_model(d, base, seed=11, **over)
    shutil.copy(os.path.join(GOLDEN, f"bpe_{tokenizer}_tokenizer.json"), os.path.join(d, "tokenizer.json"))
    if hf is not None:
        with open(os.path.join(d, "generation_config.json"), "w") as f:
            f.write(hf)
    return d, cfg, tensors


def _oracle(d, cfg, tensors, template, hf=None, system_prompt=None, mode="default"):
    from oracle.bpe_oracle import BpeOracle
    from oracle.chat_oracle import ChatOracle
    from oracle.llm_oracle import LlmOracle
    eos = cfg["eos_token_id"]
    return ChatOracle(LlmOracle(tensors, cfg), BpeOracle(os.path.join(d, "tokenizer.json")), template, cfg["model_type"],
                      cfg["max_position_embeddings"], cfg.get("bos_token_id"), eos if isinstance(eos, list) else [eos], hf, system_prompt, mode)


@pytest.fixture(scope="module")
def llama(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("chat_llama")
    d, cfg, tensors = _make(tmp, "llama", synth.LLAMA_TEST, "llama3", vocab_size=720, bos_token_id=LLAMA_SPECIAL["bos"],
                            eos_token_id=[LLAMA_SPECIAL["end_of_text"], LLAMA_SPECIAL["eom"], LLAMA_SPECIAL["eot"]])
    return d, cfg, tensors


@pytest.fixture(scope="module")
def llama_chat(llama):
    from kjarni_amd.chat import Chat
    return Chat("llama3.2-1b-instruct", model_path=llama[0])


def _greedy(n=None):
    from kjarni_amd.chat import GenerationConfig
    return GenerationConfig(do_sample=False, max_new_tokens=n)


def _ov(**kw):
    from oracle.chat_oracle import Overrides
    return Overrides(**kw)


def test_accessors_and_resolved_defaults(llama, llama_chat):
    d, cfg, tensors = llama
    assert llama_chat.model_name == "llama3.2-1b-instruct"
    assert llama_chat.context_size == cfg["max_position_embeddings"]
    want = _oracle(d, cfg, tensors, "llama3").resolve()
    got = llama_chat.resolve()
    assert (got.strategy, got.top_k, got.max_new_tokens, got.add_bos_token) == (want.strategy, want.top_k, want.max_new_tokens, True)
    assert abs(got.temperature - 0.7) < 1e-6 and abs(got.top_p - 0.9) < 1e-6 and abs(got.min_p - 0.05) < 1e-6


def test_prompt_and_ids_match_oracle(llama, llama_chat):
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    conv = o.create_conversation() + [("user", "Hello, how are you?")]
    prompt = llama_chat.format_prompt(None, "Hello, how are you?")
    assert prompt == o.format_prompt(conv)
    assert "You are a helpful, harmless, and honest assistant." in prompt
    ids = llama_chat.encode(prompt)
    assert ids == o.encode(prompt, o.resolve())
    assert ids[0] == LLAMA_SPECIAL["bos"] and ids[1] != LLAMA_SPECIAL["bos"]  # the template's BOS is not doubled
    history = [("user", "one"), ("assistant", "two"), ("system", "late system"), ("user", "three")]
    assert llama_chat.format_prompt(history, "four") == o.format_prompt(o.history_to_conversation(history) + [("user", "four")])


def test_greedy_send_matches_oracle(llama, llama_chat):
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    for message, n in [("Hello!", 24), ("What is 2+2?\nAnswer briefly.", 40), ("þæö 日本語 😀", 12)]:
        prompt = o.format_prompt(o.create_conversation() + [("user", message)])
        want = o.generate(prompt, _ov(do_sample=False, max_new_tokens=n))
        assert llama_chat.send(message, _greedy(n)) == want


def test_stream_pieces_callbacks_and_cancel(llama, llama_chat):
    from kjarni_amd.indexer import CancelToken
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    prompt = o.format_prompt(o.create_conversation() + [("user", "Tell me a story.")])
    want = [t for _, t in o.stream(prompt, _ov(do_sample=False, max_new_tokens=30))]
    got = []
    llama_chat.stream("Tell me a story.", lambda t: got.append(t) or True, _greedy(30))
    assert got == want and len(got) == 30
    got = []
    llama_chat.stream("Tell me a story.", lambda t: (got.append(t), len(got) < 5)[1], _greedy(30))
    assert got == want[:5]  # false from the callback ends the stream after that token
    token = CancelToken()
    got = []

    def on_token(t):
        got.append(t)
        if len(got) == 3:
            token.cancel()
        return True

    llama_chat.stream("Tell me a story.", on_token, _greedy(30), cancel=token)
    assert got == want[:3]  # cancellation is seen before the next token is handed over


def test_stop_token_and_length_and_context_limits(tmp_path, llama):
    from kjarni_amd.chat import Chat
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    prompt = o.format_prompt(o.create_conversation() + [("user", "Count.")])
    from kjarni_amd.chat import GenerationConfig
    # a repetition penalty keeps the random-weight model from repeating one token, so a stop token can be picked
    ov = _ov(do_sample=False, max_new_tokens=20, repetition_penalty=1.6)
    g = GenerationConfig(do_sample=False, max_new_tokens=20, repetition_penalty=1.6)
    ids = [i for i, _ in o.stream(prompt, ov)]
    k = next(k for k in range(3, len(ids)) if ids[k] not in ids[:k])
    stop = ids[k]
    # a model whose FIRST eos id is that token stops there; its other eos ids do not stop it (models/base.rs:261-271)
    d2, cfg2, tensors2 = _make(tmp_path, "stop", synth.LLAMA_TEST, "llama3", vocab_size=720, bos_token_id=700, eos_token_id=[stop, ids[1]])
    chat = Chat("llama3.2-1b-instruct", model_path=d2)
    o2 = _oracle(d2, cfg2, tensors2, "llama3")
    got = []
    chat.stream("Count.", lambda t: got.append(t) or True, g)
    want = [t for _, t in o2.stream(prompt, ov)]
    assert got == want and len(got) == k
    # default budget (512 new tokens) on a 256-token context: generation ends when the context is full
    got = []
    chat2 = Chat("llama3.2-1b-instruct", model_path=d)
    chat2.stream("Count.", lambda t: got.append(t) or True, _greedy())
    n_prompt = len(o.encode(prompt, o.resolve()))
    assert len(got) == cfg["max_position_embeddings"] - n_prompt
    assert got == [t for _, t in o.stream(prompt, _ov(do_sample=False))]
    assert chat2.send("Count.", _greedy(0)) == ""


def test_history_and_conversation(llama, llama_chat):
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    history = [("user", "My name is Xylophone7492."), ("assistant", "Nice to meet you.")]
    prompt = o.format_prompt(o.history_to_conversation(history) + [("user", "What is my name?")])
    assert llama_chat.send_with_history(history, "What is my name?", _greedy(16)) == o.generate(prompt, _ov(do_sample=False, max_new_tokens=16))

    convo = llama_chat.conversation()
    assert len(convo) == 0  # no configured system prompt: the history starts empty (chat.rs:497-501)
    first = convo.send("My name is Xylophone7492.", _greedy(10))
    h = [("user", "My name is Xylophone7492.")]
    assert first == o.generate(o.format_prompt(o.history_to_conversation(h)), _ov(do_sample=False, max_new_tokens=10))
    assert len(convo) == 2
    h.append(("assistant", first))
    h.append(("user", "What is my name?"))
    second = convo.send("What is my name?", _greedy(10))
    assert second == o.generate(o.format_prompt(o.history_to_conversation(h)), _ov(do_sample=False, max_new_tokens=10))
    assert len(convo) == 4
    pieces = []
    convo.stream("And again?", lambda t: pieces.append(t) or True, _greedy(6))
    assert len(pieces) == 6 and len(convo) == 6  # the streamed reply is stored as streamed
    convo.clear(keep_system=True)
    assert len(convo) == 0
    with pytest.raises(Exception, match="Invalid role"):
        llama_chat.send_with_history([(7, "x")], "y", _greedy(1))


def test_system_prompt_conversation_and_modes(llama):
    from kjarni_amd.chat import Chat
    d, cfg, tensors = llama
    chat = Chat("llama3.2-1b-instruct", model_path=d, system_prompt="You are a pirate.", mode="creative")
    o = _oracle(d, cfg, tensors, "llama3", system_prompt="You are a pirate.", mode="creative")
    r = chat.resolve()
    assert r.max_new_tokens == 1024 and abs(r.temperature - 0.9) < 1e-6
    assert chat.format_prompt(None, "Ahoy") == o.format_prompt([("system", "You are a pirate."), ("user", "Ahoy")])
    convo = chat.conversation()
    assert len(convo) == 1
    convo.send("Ahoy", _greedy(4))
    assert len(convo) == 3
    convo.clear(keep_system=True)
    assert len(convo) == 1
    convo.clear(keep_system=False)
    assert len(convo) == 0
    # a history without a system entry gets the configured one in front (model.rs:168-182)
    h = [("user", "a"), ("assistant", "b")]
    assert chat.format_prompt(h, "c") == o.format_prompt([("system", "You are a pirate.")] + h + [("user", "c")])


def test_sampling_stays_in_the_oracle_support_and_is_seedable(llama, llama_chat):
    from kjarni_amd.chat import GenerationConfig
    from oracle import chat_oracle as co
    d, cfg, tensors = llama
    o = _oracle(d, cfg, tensors, "llama3")
    prompt = o.format_prompt(o.create_conversation() + [("user", "Sample something.")])
    cfgs = [GenerationConfig(max_new_tokens=24), GenerationConfig(max_new_tokens=24, temperature=1.5, top_k=5, top_p=0.95, min_p=0.0,
                                                                  repetition_penalty=1.3)]
    for g in cfgs:
        resolved = o.resolve(_ov(temperature=g.temperature, top_k=g.top_k, top_p=g.top_p, min_p=g.min_p,
                                 repetition_penalty=g.repetition_penalty, max_new_tokens=g.max_new_tokens))
        runs = []
        for seed in (1, 1, 2):
            llama_chat.seed(seed)
            pieces = []
            llama_chat.stream("Sample something.", lambda t: pieces.append(t) or True, g)
            runs.append(pieces)
        assert runs[0] == runs[1]  # same seed, same draw sequence
        # every sampled token has non-zero probability under the oracle's distribution for the same prefix
        from oracle.llm_oracle import apply_repetition_penalty
        tok = o.tokenizer
        tokens = o.encode(prompt, resolved)
        cache = o.model.new_cache()
        logits = o.model.logits(o.model.forward(tokens, cache)[0, -1])
        for piece in runs[2]:
            lg = np.array(logits, np.float32)
            if resolved.repetition_penalty != 1.0:
                apply_repetition_penalty(lg, tokens, resolved.repetition_penalty)
            probs = co.sampling_distribution(lg, resolved.temperature, resolved.top_k, resolved.top_p, resolved.min_p)
            cands = [i for i in np.flatnonzero(probs > 0) if tok.decode([int(i)]) == piece]
            assert cands, piece
            if len(cands) != 1:
                break  # two surviving tokens with the same text: the prefix is no longer identifiable from the stream
            nxt = int(cands[0])
            tokens.append(nxt)
            logits = o.model.logits(o.model.forward([nxt], cache)[0, -1])


def test_qwen_chatml_with_generation_config(tmp_path):
    from kjarni_amd.chat import Chat
    hf = '{"do_sample": true, "temperature": 0.7, "top_p": 0.8, "top_k": 20, "repetition_penalty": 1.05, "bos_token_id": 700}'
    d, cfg, tensors = _make(tmp_path, "qwen", synth.QWEN_TEST, "qwen2", hf=hf, vocab_size=720, bos_token_id=700, eos_token_id=702)
    chat = Chat("qwen2.5-0.5b-instruct", model_path=d)
    o = _oracle(d, cfg, tensors, "chatml", hf=hf)
    assert chat.model_name == "qwen2.5-0.5b-instruct"
    r = chat.resolve()
    assert (r.strategy, r.top_k, r.add_bos_token, r.max_new_tokens) == ("sample", 20, True, 512)
    assert abs(r.repetition_penalty - 1.05) < 1e-6 and r.min_p is None
    prompt = chat.format_prompt(None, "Hi")
    assert prompt == "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\nHi<|im_end|>\n<|im_start|>assistant\n"
    ids = chat.encode(prompt)
    assert ids == o.encode(prompt, o.resolve()) and ids[0] == 700  # generation_config.json turns add_bos_token on
    from kjarni_amd.chat import GenerationConfig
    g = GenerationConfig(do_sample=False, max_new_tokens=20)
    assert chat.send("Hi", g) == o.generate(prompt, _ov(do_sample=False, max_new_tokens=20))  # repetition penalty 1.05 on the host path
    # without generation_config.json: the Qwen fallback (no BOS, repetition penalty 1.1)
    d2, cfg2, tensors2 = _make(tmp_path, "qwen_nohf", synth.QWEN_TEST, "qwen2", vocab_size=720, bos_token_id=700, eos_token_id=702)
    chat2 = Chat("qwen2.5-0.5b-instruct", model_path=d2)
    o2 = _oracle(d2, cfg2, tensors2, "chatml")
    assert chat2.encode(prompt)[0] != 700
    assert chat2.send("Hi", g) == o2.generate(prompt, _ov(do_sample=False, max_new_tokens=20))


def test_error_codes(tmp_path, llama):
    from kjarni_amd import _ffi
    from kjarni_amd.chat import Chat

    def code(fn):
        with pytest.raises(_ffi.KjarniException) as e:
            fn()
        return e.value.code, str(e.value)

    c, msg = code(lambda: Chat("not-a-real-model"))
    assert c == _ffi.KjarniError.MODEL_NOT_FOUND and "not-a-real-model" in msg
    c, msg = code(lambda: Chat("minilm-l6-v2"))
    assert c == _ffi.KjarniError.INVALID_CONFIG and "is an encoder and cannot generate text" in msg
    c, msg = code(lambda: Chat("flan-t5-base"))
    assert c == _ffi.KjarniError.INVALID_CONFIG and "seq2seq" in msg
    c, msg = code(lambda: Chat("whisper-small"))
    assert c == _ffi.KjarniError.INVALID_CONFIG and "speech-to-text" in msg
    c, msg = code(lambda: Chat("llama3.2-1b-instruct", cache_dir=str(tmp_path / "empty")))
    assert c == _ffi.KjarniError.MODEL_NOT_FOUND and "not downloaded" in msg
    c, msg = code(lambda: Chat("gpt2", model_path=llama[0]))
    assert c == _ffi.KjarniError.INVALID_CONFIG and "does not have a chat template" in msg
    # a tokenizer.json this library cannot reproduce is a load error, never a silent approximation
    d = str(tmp_path / "bad_tok")
    shutil.copytree(llama[0], d)
    with open(os.path.join(d, "tokenizer.json")) as f:
        j = json.load(f)
    j["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = r"\w+"
    with open(os.path.join(d, "tokenizer.json"), "w") as f:
        json.dump(j, f)
    c, msg = code(lambda: Chat("llama3.2-1b-instruct", model_path=d))
    assert c == _ffi.KjarniError.LOAD_FAILED and "unsupported pre-tokenizer regex" in msg


class _HfTokenizer:
    """The `tokenizers` package itself as the tokenizer of the oracle chat (SentencePiece-style BPE has no Python
    restatement in oracle/; the product tokenizer is pinned to the same package in tests/test_bpe_tokenizer.py)."""

    def __init__(self, path):
        import tokenizers
        self.t = tokenizers.Tokenizer.from_file(path)

    def encode(self, text):
        return self.t.encode(text, add_special_tokens=False).ids

    def decode(self, ids, skip_special_tokens=False):
        return self.t.decode(list(ids), skip_special_tokens=skip_special_tokens)

    def token_to_id(self, token):
        return self.t.token_to_id(token)


def test_mistral_chat(tmp_path):
    pytest.importorskip("tokenizers")
    from kjarni_amd.chat import Chat, GenerationConfig
    from oracle.chat_oracle import ChatOracle
    from oracle.llm_oracle import LlmOracle
    hf = '{"do_sample": false, "max_new_tokens": 5}'  # present on disk, never read for Mistral (mistral/model.rs:236-252)
    base = dict(synth.LLAMA_TEST, model_type="mistral", rope_theta=10000.0, tie_word_embeddings=False)
    base.pop("rope_scaling")
    d, cfg, tensors = _make(tmp_path, "mistral", base, "llama3", hf=hf, vocab_size=768, bos_token_id=1, eos_token_id=2)
    shutil.copy(os.path.join(GOLDEN, "spbpe_legacy_tokenizer.json"), os.path.join(d, "tokenizer.json"))
    tensors["lm_head.weight"] = tensors.get("lm_head.weight", tensors["model.embed_tokens.weight"])
    chat = Chat("mistral-7b", model_path=d)
    tok = _HfTokenizer(os.path.join(d, "tokenizer.json"))
    o = ChatOracle(LlmOracle(tensors, cfg), tok, "mistral", "mistral", cfg["max_position_embeddings"], 1, [2], hf)
    assert chat.model_name == "mistral-7b"
    r = chat.resolve()
    assert (r.strategy, r.top_k, r.max_new_tokens, r.add_bos_token) == ("sample", 40, 512, True)
    assert abs(r.repetition_penalty - 1.15) < 1e-6 and abs(r.min_p - 0.05) < 1e-6 and abs(r.temperature - 0.7) < 1e-6
    prompt = chat.format_prompt(None, "Hello there")
    assert prompt == "<s>[INST] Hello there [/INST]" == o.format_prompt(o.create_conversation() + [("user", "Hello there")])
    ids = chat.encode(prompt)
    assert ids == o.encode(prompt, o.resolve()) and ids[0] == 1 and ids[1] != 1
    history = [("system", "Assistant is friendly."), ("user", "Hello!"), ("assistant", "Hi there!")]
    assert chat.format_prompt(history, "How are you?") == \
        "<s>[INST] Assistant is friendly.\n\nHello! [/INST] Hi there!</s>[INST] How are you? [/INST]"
    g = GenerationConfig(do_sample=False, max_new_tokens=24)
    ov = _ov(do_sample=False, max_new_tokens=24)
    assert chat.send("Hello there", g) == o.generate(prompt, ov)  # repetition penalty 1.15 applied on the way
    pieces = []
    chat.stream("Hello there", lambda t: pieces.append(t) or True, g)
    assert pieces == [t for _, t in o.stream(prompt, ov)]


# ---- sampling with the O(vocab) work on the device (llm_kernels.hip: processors, candidate cut) against the same loop on
# ---- a host copy of the logits (the checker): same seed, same tokens
@pytest.fixture(scope="module")
def peaked_chat(tmp_path_factory):
    """The llama fixture with its final norm scaled 8x: a peaked next-token distribution (as a trained model's), so that
    the candidates within reach of top-p / min-p are a small part of the vocabulary."""
    from safetensors.numpy import save_file

    from kjarni_amd.chat import Chat
    tmp = tmp_path_factory.mktemp("chat_peaked")
    d, cfg, tensors = _make(tmp, "llama", synth.LLAMA_TEST, "llama3", vocab_size=720, bos_token_id=LLAMA_SPECIAL["bos"],
                            eos_token_id=[LLAMA_SPECIAL["end_of_text"], LLAMA_SPECIAL["eom"], LLAMA_SPECIAL["eot"]])
    tensors = dict(tensors)
    tensors["model.norm.weight"] = (tensors["model.norm.weight"] * np.float32(8.0)).astype(np.float32)
    save_file({k: np.ascontiguousarray(v) for k, v in tensors.items()}, os.path.join(d, "model.safetensors"))
    return Chat("llama3.2-1b-instruct", model_path=d)


@pytest.mark.parametrize("which", ["peaked", "flat"])
def test_device_sampling_equals_host_sampling(llama_chat, peaked_chat, which):
    from kjarni_amd.chat import GenerationConfig
    chat = peaked_chat if which == "peaked" else llama_chat
    cfgs = [GenerationConfig(max_new_tokens=40),                                                    # top-p 0.9 + min-p 0.05, T 0.6
            GenerationConfig(max_new_tokens=40, temperature=0.7, top_k=40, top_p=0.9, min_p=0.05),
            GenerationConfig(max_new_tokens=40, temperature=1.5, top_k=5, top_p=0.95, min_p=0.0, repetition_penalty=1.3),
            GenerationConfig(max_new_tokens=40, temperature=1.0, top_p=0.8, repetition_penalty=1.15),
            GenerationConfig(max_new_tokens=40, temperature=0.9, min_p=0.1),
            GenerationConfig(max_new_tokens=40, temperature=1.2),                                   # temperature only: needs the logits
            GenerationConfig(max_new_tokens=30, do_sample=False, repetition_penalty=1.4)]
    before = chat.sampling_counters()
    for g in cfgs:
        for seed in (3, 4):
            runs = []
            for on in (True, False):
                chat.set_device_sampling(on)
                chat.seed(seed)
                pieces = []
                chat.stream("Tell me about sampling.", lambda t: pieces.append(t) or True, g)
                runs.append(pieces)
            chat.set_device_sampling(True)
            assert runs[0] == runs[1], (which, g, seed)
            assert len(runs[0]) > 0
    cand, full = (a - b for a, b in zip(chat.sampling_counters(), before))
    assert cand + full > 0
    if which == "peaked":
        assert cand > 3 * full      # the candidates decide nearly every token; temperature-only always needs the logits
