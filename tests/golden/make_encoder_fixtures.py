#!/usr/bin/env python3
"""Second opinion for the oracle at MODEL scale (last two rows).

The reference-held encoder goldens are one layer at hidden 4 (cpu/encoder/encoder_layer.rs:244-307): nothing the
reference ships pins the model-level wiring -- tensor-name map, fused Q|K|V order, 12 heads of 32, six post-norm layers at
hidden 384, mean-pool + L2, bert.pooler + classifier.  Hugging Face `transformers.BertModel` /
`BertForSequenceClassification` compute the identical graph (post-norm, erf-GELU, eps inside the sqrt), so this script --
BUILD CONTAINER ONLY, it needs torch + transformers -- loads the seeded random weights of tests/synth.py into them,
evaluates in float64, and writes

    tests/golden/encoder_fixtures.npz

for TWO weight families of tests/synth.py -- "init" (N(0, 0.02): uniform softmax, GELU in its linear part) and "trained"
(keys prefixed `trained_`: peaked softmax, LayerNorm gain outliers, FC1 pre-activations beyond +-6; the stand-in for the
real checkpoints of sentence_encoder/tests.rs:411-1184 / cross_encoder/tests.rs:38-100, which cannot be fetched here) --
with (ids, mask[, type ids]) -> embeddings [B, 384] (mean pool + L2, SentenceEncoder::encode_batch_flat semantics),
raw last hidden states for the small cases, and -> logits [B, 1] for (query, doc) pairs, B in {1, 3, 64}, S in {8, 128},
ragged masks.  Only the .npz travels; tests/test_oracle_fixtures.py holds oracle/ to it at 1e-5 on the CPU and
tests/test_gpu_encoder.py holds the HIP path to it at 1e-4 on the GPU box.

    python tests/golden/make_encoder_fixtures.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import synth  # noqa: E402

CASES = [(1, 8), (3, 8), (64, 8), (1, 128), (3, 128), (64, 128)]
EMBED_SEED, CROSS_SEED = 0, 1


def weights_digest(tensors) -> str:
    """sha256 over the tensors in name order: the test regenerates the weights from the seed and checks this first, so a
    change in numpy's generator shows up as a digest mismatch, not as a parity failure."""
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(np.ascontiguousarray(tensors[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def embed_inputs(B, S):
    ids, mask = synth.synthetic_ids(B, S, seed=1000 * B + S, ragged=(B > 1))
    return ids, mask


def pair_inputs(B, S):
    ids, mask, types = synth.synthetic_pairs(B, S, seed=2000 * B + S, qlen=min(16, max(1, S // 3)))
    rng = np.random.default_rng(B * 31 + S)
    for i in range(B):               # every second pair shorter than the padded length
        if i % 2 == 1 and S > 8:
            n = int(rng.integers(S // 2, S))
            ids[i, n - 1] = 102
            ids[i, n:] = 0
            mask[i, n:] = 0
            types[i, n:] = 0
    return ids, mask, types


def main():
    import torch
    from transformers import BertConfig, BertForSequenceClassification, BertModel

    torch.set_grad_enabled(False)
    out = {}

    def hf_config(cfg, **kw):
        return BertConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                          num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                          intermediate_size=cfg["intermediate_size"], hidden_act="gelu",
                          max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"],
                          layer_norm_eps=cfg["layer_norm_eps"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          attn_implementation="eager", **kw)

    def load(model, tensors):
        sd = {k: torch.from_numpy(v.astype(np.float64)) for k, v in tensors.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        missing = [k for k in missing if "position_ids" not in k]
        assert not missing and not unexpected, (missing, unexpected)
        return model.double().eval()

    for family, pre in (("init", ""), ("trained", "trained_")):
        make = synth.trained_bert_tensors if family == "trained" else synth.bert_tensors
        # ---- Embedder: BertModel without the pooler; mean pool over real tokens + L2 in float64
        cfg = dict(synth.MINILM, model_type="bert", hidden_act="gelu", layer_norm_eps=1e-12)
        tensors = make(cfg, EMBED_SEED)
        out[pre + "embed_weights_sha256"] = np.array(weights_digest(tensors))
        model = load(BertModel(hf_config(cfg), add_pooling_layer=False), tensors)
        worst32 = 0.0
        for B, S in CASES:
            ids, mask = embed_inputs(B, S)
            ti, tm = torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(mask.astype(np.int64))
            h = model(input_ids=ti, attention_mask=tm).last_hidden_state          # [B, S, H] float64
            m = tm.double().unsqueeze(-1)
            pooled = (h * m).sum(1) / m.sum(1).clamp(min=1.0)
            emb = pooled / pooled.norm(dim=1, keepdim=True)
            tag = f"{pre}embed_{B}x{S}"
            out[tag + "_ids"], out[tag + "_mask"] = ids, mask
            out[tag + "_embeddings"] = emb.numpy().astype(np.float32)
            if B * S <= 64 or (family == "trained" and B * S <= 128):
                out[tag + "_hidden"] = h.numpy().astype(np.float32)
            if (B, S) == (3, 128):
                h3x128 = h.numpy()
            h32 = model.float()(input_ids=ti, attention_mask=tm).last_hidden_state
            worst32 = max(worst32, float((h32.double() - h).abs().mul(m).max()))
            model.double()
        out[pre + "embed_hf_f32_vs_f64_hidden_max_abs"] = np.array(worst32)
        # which numerical regime these weights put a batch in (tests assert it: a fixture that drifted back to uniform
        # softmax would otherwise still pass)
        ids, mask = embed_inputs(3, 128)
        st = synth.regime_stats(tensors, cfg, ids, mask)
        hidden64 = st.pop("last_hidden")
        assert float(np.abs(hidden64 - h3x128)[mask.astype(bool)].max()) < 1e-9, "numpy float64 forward != HF float64 forward"
        for k, v in st.items():
            out[f"{pre}regime_{k}"] = np.array(v)
        print(family, {k: round(v, 3) for k, v in st.items()})

        # ---- Reranker: BertForSequenceClassification (bert.pooler.dense + tanh -> classifier), one label
        ccfg = dict(cfg)
        ctens = make(ccfg, CROSS_SEED, prefix="bert.", head="cross")
        out[pre + "cross_weights_sha256"] = np.array(weights_digest(ctens))
        cmodel = load(BertForSequenceClassification(hf_config(ccfg, num_labels=1)), ctens)
        worst_logit = 0.0
        for B, S in CASES:
            ids, mask, types = pair_inputs(B, S)
            logits = cmodel(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask.astype(np.int64)),
                            token_type_ids=torch.from_numpy(types.astype(np.int64))).logits
            tag = f"{pre}pairs_{B}x{S}"
            worst_logit = max(worst_logit, float(logits.abs().max()))
            out[tag + "_ids"], out[tag + "_mask"], out[tag + "_types"] = ids, mask, types
            out[tag + "_logits"] = logits.numpy().astype(np.float32)

        out[pre + "pairs_logit_absmax"] = np.array(worst_logit)

    path = os.path.join(HERE, "encoder_fixtures.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(out)} arrays")


if __name__ == "__main__":
    main()
