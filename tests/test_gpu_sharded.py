"""Sharded checkpoints (`model.safetensors.index.json` + shards, the layout of the registry's 3B-8B models and of bge-m3's
mirror copies): weights/safetensors_loader.rs:46-129.  A sharded directory must load to exactly the model the single file
gives, through every loader (encoder, decoder, Whisper), and a broken index must fail at load."""
import json
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def test_sharded_encoder_is_the_same_model(tmp_path):
    import kjarni_amd
    a, b = str(tmp_path / "one"), str(tmp_path / "many")
    cfg, _ = synth.minilm_embedder(a, seed=3, num_hidden_layers=2)
    synth.minilm_embedder(b, seed=3, num_hidden_layers=2)
    wm = synth.shard_model_dir(b, 3)
    assert not os.path.exists(os.path.join(b, "model.safetensors")) and len(set(wm.values())) == 3
    ids, mask = synth.synthetic_ids(4, 32, seed=1, ragged=True)
    one, many = kjarni_amd.HipEncoder(a), kjarni_amd.HipEncoder(b)
    assert np.array_equal(one.hidden_states(ids, mask), many.hidden_states(ids, mask))
    # string level: the directory checks accept the index in place of the single file
    synth.add_tokenizer(b)
    emb = kjarni_amd.Embedder(model_path=b)
    assert emb.encode_batch(["hello world"]).shape == (1, cfg["hidden_size"])


def test_sharded_decoder_and_bf16(tmp_path):
    import kjarni_amd
    a, b = str(tmp_path / "one"), str(tmp_path / "many")
    cfg, _ = synth.llm_model(a, synth.LLAMA_TEST, seed=5, store_bf16=True)
    synth.llm_model(b, synth.LLAMA_TEST, seed=5, store_bf16=True)
    synth.shard_model_dir(b, 4)
    one, many = kjarni_amd.HipDecoder(a), kjarni_amd.HipDecoder(b)
    ids = np.random.default_rng(0).integers(4, cfg["vocab_size"], 40).tolist()
    h1, l1 = one.forward(ids)
    h2, l2 = many.forward(ids)
    assert np.array_equal(l1, l2) and np.array_equal(h1, h2)


def test_broken_index_fails_at_load(tmp_path):
    import kjarni_amd
    d = str(tmp_path / "m")
    synth.minilm_embedder(d, seed=3, num_hidden_layers=1)
    wm = synth.shard_model_dir(d, 2)
    index = os.path.join(d, "model.safetensors.index.json")
    good = json.load(open(index))
    # a tensor mapped to the shard that does not hold it
    bad = json.loads(json.dumps(good))
    name = next(iter(wm))
    bad["weight_map"][name] = next(f for f in set(wm.values()) if f != wm[name])
    json.dump(bad, open(index, "w"))
    with pytest.raises(kjarni_amd.KjarniException, match="is not in the shard"):
        kjarni_amd.HipEncoder(d)
    # a shard that is not there
    bad = json.loads(json.dumps(good))
    bad["weight_map"][name] = "model-00009-of-00009.safetensors"
    json.dump(bad, open(index, "w"))
    with pytest.raises(kjarni_amd.KjarniException, match="cannot open"):
        kjarni_amd.HipEncoder(d)
    # shard names may not leave the directory
    bad["weight_map"][name] = "../elsewhere.safetensors"
    json.dump(bad, open(index, "w"))
    with pytest.raises(kjarni_amd.KjarniException, match="shard name"):
        kjarni_amd.HipEncoder(d)
    # no weight_map
    json.dump({"metadata": {}}, open(index, "w"))
    with pytest.raises(kjarni_amd.KjarniException, match="missing 'weight_map'"):
        kjarni_amd.HipEncoder(d)
    # a tensor the index leaves out is not part of the model (the loader resolves names through the map)
    pruned = json.loads(json.dumps(good))
    del pruned["weight_map"]["embeddings.LayerNorm.weight"]
    json.dump(pruned, open(index, "w"))
    with pytest.raises(kjarni_amd.KjarniException, match="embeddings.LayerNorm.weight"):
        kjarni_amd.HipEncoder(d)
