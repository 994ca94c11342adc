"""oracle/chat_oracle.py pinned by the reference's own unit tests for the chat path (expected values restated from the
cited test functions), then the host side of the C ABI (templates, config resolution, sampling, logits processors)
compared with the oracle.  No GPU needed."""
import numpy as np
import pytest

from kjarni_amd import chat as kc
from oracle import chat_oracle as co
from oracle.chat_oracle import ASSISTANT, SYSTEM, USER, GenerationConfig, Overrides


# ---- the reference's template tests ---------------------------------------------------------------------

def test_mistral_template_reference_cases():  # chat/mistral.rs:90-147
    assert co.apply_mistral([]) == ""
    assert co.apply_mistral([(USER, "Hello there")]) == "<s>[INST] Hello there [/INST]"
    assert co.apply_mistral([(SYSTEM, "You are a helpful assistant."), (USER, "What is 2 + 2?")]) == \
        "<s>[INST] You are a helpful assistant.\n\nWhat is 2 + 2? [/INST]"
    assert co.apply_mistral([(USER, "Tell me a joke."), (ASSISTANT, "Why did the chicken cross the road?")]) == \
        "<s>[INST] Tell me a joke. [/INST] Why did the chicken cross the road?</s>"
    conv = [(SYSTEM, "Assistant is friendly."), (USER, "Hello!"), (ASSISTANT, "Hi there!"), (USER, "How are you?"),
            (ASSISTANT, "I'm good, thank you!")]
    assert co.apply_mistral(conv) == \
        "<s>[INST] Assistant is friendly.\n\nHello! [/INST] Hi there!</s>[INST] How are you? [/INST] I'm good, thank you!</s>"
    assert co.STOP_SEQUENCES["mistral"] == ["</s>"]


def test_chatml_template_reference_cases():  # chat/chatml.rs:54-100
    assert co.apply_chatml([]) == "<|im_start|>assistant\n"
    p = co.apply_chatml([(SYSTEM, "You are a helpful assistant."), (USER, "Hello!")])
    assert p.startswith("<|im_start|>system\n") and "You are a helpful assistant." in p and "<|im_start|>user\n" in p
    assert p.endswith("<|im_start|>assistant\n")
    assert co.STOP_SEQUENCES["chatml"] == ["<|im_end|>", "<|endoftext|>"]


def test_llama3_template_reference_cases():  # chat/llama3.rs:257-310, 365-388, 436-444
    p = co.apply_llama3([(USER, "Hello!")])
    assert p.startswith("<|begin_of_text|>") and "<|start_header_id|>user<|end_header_id|>" in p and "<|eot_id|>" in p
    assert p.endswith("<|start_header_id|>assistant<|end_header_id|>\n\n")
    p = co.apply_llama3([(USER, "What is 2+2?"), (ASSISTANT, "2+2 equals 4."), (USER, "And 3+3?")])
    assert p.count("<|start_header_id|>user<|end_header_id|>") == 2 and p.count("<|start_header_id|>assistant<|end_header_id|>") == 2
    p = co.apply_llama3([(USER, "Hi"), (ASSISTANT, "Hello!"), (USER, "How are you?"), (ASSISTANT, "I am good.")])
    assert p.count("<|start_header_id|>assistant<|end_header_id|>") == 3
    assert "<|eot_id|>" in co.STOP_SEQUENCES["llama3"] and "<|end_of_text|>" in co.STOP_SEQUENCES["llama3"]
    assert co.apply_llama3([]).startswith("<|begin_of_text|>")


# ---- the reference's generation-config tests ------------------------------------------------------------

def _default_config():  # resolution.rs:93-110
    return GenerationConfig(max_new_tokens=256, max_length=2048, strategy="sample", temperature=0.7, top_k=50, top_p=0.9, min_p=None)


def test_resolution_reference_cases():  # resolution.rs:112-197
    d = _default_config()
    assert co.resolve_generation_config(d, Overrides(), Overrides()).max_new_tokens == 256
    r = co.resolve_generation_config(d, Overrides(temperature=0.5, max_new_tokens=512), Overrides())
    assert r.max_new_tokens == 512 and r.strategy == "sample" and r.temperature == 0.5
    r = co.resolve_generation_config(d, Overrides(temperature=0.5), Overrides(temperature=0.9))
    assert r.temperature == 0.9
    assert co.resolve_generation_config(d, Overrides(do_sample=False), Overrides()).strategy == "greedy"
    r = co.resolve_generation_config(d, Overrides(num_beams=4), Overrides())
    assert r.strategy == "beam_search" and r.num_beams == 4


def test_hf_generation_defaults_reference_cases():  # common/mod.rs:374-420
    c = co.hf_generation_defaults('{"do_sample": false, "max_length": 2048}', 4096)
    assert c.strategy == "greedy" and c.max_length == 2048
    c = co.hf_generation_defaults('{"do_sample": true, "temperature": 0.8, "top_p": 0.95, "top_k": 40, "repetition_penalty": 1.1}', 1024)
    assert c.strategy == "sample" and c.temperature == 0.8 and c.top_p == 0.95 and c.top_k == 40
    assert c.repetition_penalty == 1.1 and c.max_length == 1024 and c.min_p is None
    c = co.hf_generation_defaults("{}", 512)
    assert c.strategy == "greedy" and c.max_length == 512
    assert co.hf_generation_defaults("{ invalid_json }", 512) is None


def test_chat_mode_defaults():  # chat/tests.rs:9-18
    assert co.MODE_TEMPERATURE == {"default": 0.7, "creative": 0.9, "reasoning": 0.3}
    assert co.MODE_MAX_TOKENS == {"default": 512, "creative": 1024, "reasoning": 2048}


# ---- the reference's sampling tests ---------------------------------------------------------------------

def test_sampling_filters_reference_cases():  # sampling.rs:257-368
    assert np.allclose(co.softmax_inplace(np.array([1.0, 1.0, 1.0, 1.0], np.float32)), 0.25, atol=1e-6)
    p = co.softmax_inplace(np.array([1000.0, 1001.0, 1002.0], np.float32))
    assert abs(p.sum() - 1.0) < 1e-6 and np.isfinite(p).all()
    f = co.top_k_filtering(np.array([1.0, 5.0, 3.0, 4.0, 2.0], np.float32), 3)
    assert np.isfinite(f[[1, 3, 2]]).all() and f[0] == -np.inf and f[4] == -np.inf
    assert np.isfinite(co.top_k_filtering(np.array([1.0, 2.0, 3.0], np.float32), 3)).all()
    f = co.top_k_filtering(np.array([1.0, 5.0, 3.0], np.float32), 1)
    assert np.isfinite(f[1]) and f[0] == -np.inf and f[2] == -np.inf
    assert np.isfinite(co.top_p_filtering(np.array([0.0, 1.0, 2.0, 3.0], np.float32), 0.9)[3])
    assert np.isfinite(co.top_p_filtering(np.array([1.0, 2.0, 3.0, 4.0], np.float32), 1.0)).all()
    assert np.isfinite(co.top_p_filtering(np.array([1.0, 2.0, 10.0], np.float32), 0.01)[2])


def test_logits_processors_reference_cases():  # sampling.rs:371-460
    from oracle.llm_oracle import apply_repetition_penalty
    def pen(values, tokens, penalty):
        a = np.array(values, np.float32)
        apply_repetition_penalty(a, tokens, penalty)  # in place
        return a.tolist()

    assert pen([2.0, 4.0, 6.0], [1], 2.0) == [2.0, 2.0, 6.0]
    assert pen([-2.0, -4.0, 1.0], [0, 1], 2.0) == [-4.0, -8.0, 1.0]
    assert pen([-1.0, 0.0, 2.0], [0, 2], 2.0) == [-2.0, 0.0, 1.0]
    assert kc.logits_process([2.0, 4.0, 6.0], [1], 2.0).tolist() == [2.0, 2.0, 6.0]
    assert kc.logits_process([-2.0, -4.0, 1.0], [0, 1], 2.0).tolist() == [-4.0, -8.0, 1.0]
    assert kc.logits_process([1.0, 2.0, 3.0], [100], 2.0).tolist() == [1.0, 2.0, 3.0]
    assert kc.logits_process([1.0, 2.0, 3.0], [0, 1], 1.0).tolist() == [1.0, 2.0, 3.0]


# ---- the C ABI host side against the oracle --------------------------------------------------------------

CONVERSATIONS = [
    [],
    [(USER, "Hello!")],
    [(SYSTEM, "You are a pirate."), (USER, "Hello!")],
    [(USER, "What is 2+2?"), (ASSISTANT, "2+2 equals 4."), (USER, "And 3+3?")],
    [(SYSTEM, "Assistant is friendly."), (USER, "Hello!"), (ASSISTANT, "Hi there!"), (USER, "How are you?"), (ASSISTANT, "I'm good.")],
    [(ASSISTANT, "I speak first"), (USER, "multi\nline\n\ncontent with <|eot_id|> inside and unicode: þæö 日本")],
    [(USER, "one"), (USER, "two"), (SYSTEM, "late system")],
]


@pytest.mark.parametrize("template", ["llama3", "chatml", "mistral"])
def test_templates_match_oracle(template):
    for conv in CONVERSATIONS:
        assert kc.chat_template_apply(template, conv) == co.TEMPLATES[template](conv), (template, conv)


def _as_tuple(c: GenerationConfig):
    sample = c.strategy == "sample"
    return (c.strategy, round(c.temperature, 6) if sample else None, c.top_k if sample else None,
            None if not sample or c.top_p is None else round(c.top_p, 6), None if not sample or c.min_p is None else round(c.min_p, 6),
            round(c.repetition_penalty, 6), c.no_repeat_ngram_size, c.max_new_tokens, c.max_length, c.add_bos_token)


def _resolved_tuple(r: kc.ResolvedGeneration):
    sample = r.strategy == "sample"
    return (r.strategy, round(r.temperature, 6) if sample else None, r.top_k if sample else None,
            None if not sample or r.top_p is None else round(r.top_p, 6), None if not sample or r.min_p is None else round(r.min_p, 6),
            round(r.repetition_penalty, 6), r.no_repeat_ngram_size, r.max_new_tokens, r.max_length, r.add_bos_token)


HF_FILES = [None, "{}", '{"do_sample": true, "temperature": 0.6, "top_p": 0.9, "bos_token_id": 128000, "eos_token_id": [128001, 128008]}',
            '{"do_sample": false, "max_length": 2048}', '{"do_sample": true, "top_k": 20, "repetition_penalty": 1.05, "max_new_tokens": 77}',
            "{ invalid_json }", '{"do_sample": "yes"}', '{"temperature": null}', '{"top_p": null, "do_sample": true}', "[1, 2]"]
RUNTIMES = [None, kc.GenerationConfig(), kc.GenerationConfig(do_sample=False), kc.GenerationConfig(do_sample=True),
            kc.GenerationConfig(temperature=0.25, top_k=7, top_p=0.5, min_p=0.2, repetition_penalty=1.3, max_new_tokens=9),
            kc.GenerationConfig(temperature=0.0, do_sample=False, max_new_tokens=0), kc.GenerationConfig(top_k=0, do_sample=True)]


def _to_overrides(g):
    if g is None:
        return Overrides()
    return Overrides(temperature=g.temperature, top_k=g.top_k, top_p=g.top_p, min_p=g.min_p, repetition_penalty=g.repetition_penalty,
                     max_new_tokens=g.max_new_tokens, do_sample=g.do_sample)


@pytest.mark.parametrize("model_type", ["llama", "qwen2", "mistral"])
def test_generation_resolution_matches_oracle(model_type):
    for hf in HF_FILES:
        for mode in ["default", "creative", "reasoning", None]:
            for rt in RUNTIMES:
                want = co.chat_generation_config(model_type, 4096, hf, mode, _to_overrides(rt))
                got = kc.generation_resolve(model_type, 4096, hf, mode, rt)
                assert _resolved_tuple(got) == _as_tuple(want), (model_type, hf, mode, rt)


def test_generation_defaults_are_the_reference_ones():
    # llama fallback (llama/model.rs:381-395) under ChatMode::Default: the mode's temperature and token budget win
    r = kc.generation_resolve("llama", 131072, None, "default")
    assert (r.strategy, r.top_k, r.max_new_tokens, r.add_bos_token) == ("sample", None, 512, True)
    assert abs(r.temperature - 0.7) < 1e-6 and abs(r.top_p - 0.9) < 1e-6 and abs(r.min_p - 0.05) < 1e-6
    r = kc.generation_resolve("qwen2", 32768, None, "reasoning")
    assert (r.strategy, r.top_k, r.max_new_tokens, r.add_bos_token) == ("sample", 40, 2048, False)
    assert abs(r.temperature - 0.3) < 1e-6 and abs(r.repetition_penalty - 1.1) < 1e-6
    # a greedy generation_config.json: the mode's temperature has nothing to apply to
    r = kc.generation_resolve("llama", 4096, '{"do_sample": false}', "creative")
    assert r.strategy == "greedy" and r.max_new_tokens == 1024
    # Mistral never reads generation_config.json (mistral/model.rs:236-252)
    r = kc.generation_resolve("mistral", 32768, '{"do_sample": false}', "default")
    assert (r.strategy, r.top_k, r.max_new_tokens) == ("sample", 40, 512) and abs(r.repetition_penalty - 1.15) < 1e-6
    # forcing sampling on a greedy default picks SamplingParams::default()
    r = kc.generation_resolve("llama", 4096, '{"do_sample": false}', None, kc.GenerationConfig(do_sample=True))
    assert (r.strategy, r.top_k) == ("sample", 50) and abs(r.min_p - 0.1) < 1e-6 and abs(r.temperature - 0.7) < 1e-6


def _logit_cases():
    rng = np.random.default_rng(5)
    yield rng.normal(0, 3, 1000).astype(np.float32)
    yield rng.normal(0, 0.01, 257).astype(np.float32)          # nearly flat: top-p runs deep into the tail
    x = rng.normal(0, 2, 4096).astype(np.float32)
    x[17] = x[900] = x[901] = x.max() + 1.0                    # ties at the top: stable order decides
    yield x
    yield np.round(rng.normal(0, 2, 512)).astype(np.float32)  # many exact ties
    x = rng.normal(0, 5, 50000).astype(np.float32)
    x[rng.integers(0, 50000, 100)] = -np.inf
    yield x
    yield np.array([3.0], np.float32)
    yield np.array([0.0, 0.0, 0.0, 0.0], np.float32)


PARAMS = [dict(temperature=1.0), dict(temperature=0.7, top_k=50), dict(temperature=0.6, top_p=0.9, min_p=0.05),
          dict(temperature=0.7, top_k=40, top_p=0.8, min_p=0.05), dict(temperature=0.0, top_k=1), dict(temperature=2.0, top_p=0.999),
          dict(temperature=0.3, min_p=0.5), dict(temperature=1.0, top_k=10**6), dict(temperature=0.9, top_p=1.0), dict(temperature=1.0, top_p=0.0)]


def test_sampling_distribution_matches_oracle():
    for lg in _logit_cases():
        for p in PARAMS:
            want = co.sampling_distribution(lg, p["temperature"], p.get("top_k"), p.get("top_p"), p.get("min_p"))
            got = kc.sampling_distribution(lg, **p)
            diff = np.flatnonzero((want > 0) != (got > 0))
            if lg.size < 4096:
                assert diff.size == 0, (lg.size, p, diff[:5])  # small vocabularies: the reference's own summation order
            else:
                # large vocabularies use a vectorised exp (< 2 ulp from libm): the support may differ only where the
                # reference's own decision hangs on the last bit of a running sum, i.e. on negligible mass
                assert np.maximum(want, got)[diff].sum() < 1e-5, (lg.size, p, diff[:5])
            # a 10^5-term f32 running sum (the reference's) carries ~1e-5 relative rounding of its own: when no filter shrinks
            # the set first, the normaliser here is a more accurate 8-lane sum
            assert np.allclose(got, want, rtol=1e-5 if lg.size < 4096 else 1e-4, atol=1e-7), (lg.size, p)


def test_sample_from_probs_matches_oracle():
    rng = np.random.default_rng(11)
    for lg in _logit_cases():
        probs = co.sampling_distribution(lg, 0.8, 50, 0.9, None)
        for u in [0.0, 1e-9, 0.1, 0.5, 0.9, 0.999999, 1.0] + rng.random(20).tolist():
            assert kc.sample_from_probs(probs, u) == co.sample_from_probs(probs, u), (lg.size, u)
    assert kc.sample_from_probs(np.array([0.0, 0.0, 1.0], np.float32), 0.0) == 0   # `cumulative >= uniform` holds at index 0
    assert kc.sample_from_probs(np.array([0.25, 0.25, 0.25], np.float32), 0.99) == 2  # never reached: the last index


def test_logits_processors_match_oracle():
    from oracle.llm_oracle import apply_no_repeat_ngram, apply_repetition_penalty
    rng = np.random.default_rng(3)
    for _ in range(20):
        lg = rng.normal(0, 2, 64).astype(np.float32)
        toks = rng.integers(0, 8, rng.integers(0, 30)).tolist()
        for pen, n in [(1.0, 0), (1.3, 0), (1.0, 2), (1.1, 3), (0.8, 1), (2.0, 5)]:
            want = lg.copy()
            if pen != 1.0:
                apply_repetition_penalty(want, toks, pen)
            if n > 0:
                apply_no_repeat_ngram(want, toks, n)
            got = kc.logits_process(lg, toks, pen, n)
            assert np.array_equal(got, np.asarray(want, np.float32)), (toks, pen, n)


# ---- the candidate form of the sampler (what the decode loop uses: the device cuts the vocabulary down to the tokens within
# ---- tau of the maximum and sums the exponentials; kjarni_amd/csrc/sampling.cpp, sampling_distribution_candidates)
def _logit_sets():
    rng = np.random.default_rng(7)
    yield "peaked", (rng.standard_normal(5000) * 2.5).astype(np.float32)
    yield "flat", (rng.standard_normal(5000) * 0.05).astype(np.float32)
    big = (rng.standard_normal(128256) * 1.5).astype(np.float32)
    big[[17, 4000, 99999]] += np.float32(9.0)
    yield "vocab-128k", big
    ties = np.round(rng.standard_normal(3000) * 2).astype(np.float32)      # many exact ties: the stable order matters
    yield "ties", ties
    lone = np.full(2000, -30.0, np.float32)
    lone[123] = 5.0
    yield "one-token", lone


@pytest.mark.parametrize("params", [dict(top_k=40, top_p=0.9, min_p=0.05, temperature=0.7), dict(top_p=0.9, min_p=0.05, temperature=0.6),
                                    dict(top_k=50, top_p=0.9, min_p=0.1, temperature=0.7), dict(top_k=5), dict(top_p=0.5),
                                    dict(min_p=0.2, temperature=1.3), dict(top_p=0.999, temperature=0.3), dict(top_k=1),
                                    dict(temperature=0.8)])
def test_candidate_form_equals_the_full_sampler_or_declines(params):
    """Whenever the candidates decide the distribution it is the full sampler's, bit for bit; when they cannot (a filter
    reaches past them, temperature only, a crossing within the rounding of the sum) the call says so."""
    from kjarni_amd import chat as K
    decided_some = False
    for name, logits in _logit_sets():
        full = K.sampling_distribution(logits, **params)
        for tau in (0.5, 3.0, 8.0, 20.0, 60.0):
            got, n = K.sampling_distribution_candidates(logits, tau, **params)
            assert 1 <= n <= logits.size
            if got is None:
                continue
            decided_some = True
            assert np.array_equal(got, full), (name, tau, params)
    only_temperature = not any(k in params for k in ("top_k", "top_p", "min_p"))
    assert decided_some != only_temperature       # temperature only needs the whole vocabulary: always declined


def test_candidate_form_declines_what_it_cannot_decide():
    from kjarni_amd import chat as K
    rng = np.random.default_rng(3)
    logits = (rng.standard_normal(4000) * 0.1).astype(np.float32)     # nearly flat: top-p 0.9 needs ~90 % of the vocabulary
    got, n = K.sampling_distribution_candidates(logits, 0.05, top_p=0.9)
    assert got is None and n < 4000
    got, _ = K.sampling_distribution_candidates(logits, 0.05, top_k=3000)
    assert got is None                                                # fewer candidates than k
    got, _ = K.sampling_distribution_candidates(logits, 1.0, min_p=0.01)
    assert got is None                                                # ln(1 / 0.01) = 4.6 > tau: survivors may lie outside
    got, n = K.sampling_distribution_candidates(logits, 50.0, top_p=0.9)
    assert got is None and n == 4000                                  # nothing was cut: the plain path is the same work
