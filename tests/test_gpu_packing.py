"""Ragged batches on packed rows.

embed / logits run the encoder over the KEPT tokens of a call only (kjarni_hip.h: kjarni_hip_encoder_set_packing): a padded
token is observable through neither output -- pooling skips it (pooling/mod.rs:11-33), the head reads token 0
(cpu/encoder/classifier.rs:219), and as a key its score is overwritten so it adds exactly 0 to every softmax row
(utils/masks.rs:4-36, encoder_self_attention.rs:311-325).  Held here to the oracle (which computes every padded row, as
the reference does) and to the padded layout of the same library, across the cases that decide the layout: holes in the
mask, token 0 masked, all-masked rows, mask values other than 0 / 1, every pooling mode, the head, sequences past one
attention chunk, other head widths, RoPE, chunk boundaries, device and host pointers."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def minilm(tmp_path_factory):
    import kjarni_amd
    d = str(tmp_path_factory.mktemp("pk_minilm"))
    cfg, t = synth.minilm_embedder(d, seed=11, num_hidden_layers=2)
    enc = kjarni_amd.HipEncoder(d, 0)
    yield enc, O.OracleModel(t, cfg, blocked_gemm=True)
    enc.close()


@pytest.fixture(scope="module")
def cross(tmp_path_factory):
    import kjarni_amd
    d = str(tmp_path_factory.mktemp("pk_cross"))
    cfg, t = synth.minilm_cross_encoder(d, seed=12, num_hidden_layers=2)
    enc = kjarni_amd.HipEncoder(d, 0)
    yield enc, O.OracleModel(t, cfg, blocked_gemm=True)
    enc.close()


def both_layouts(enc, fn):
    """fn() with packing on (the default) and off."""
    packed = fn()
    enc.set_packing(False)
    try:
        padded = fn()
    finally:
        enc.set_packing(True)
    return packed, padded


@pytest.mark.parametrize("B,S", [(2, 8), (5, 37), (40, 64), (64, 128), (130, 128), (700, 100), (3, 200), (9, 512), (300, 16)])
def test_ragged_embed_equals_oracle_and_padded_layout(minilm, B, S):
    enc, orc = minilm
    ids, mask = synth.synthetic_ids(B, S, seed=B * 7 + S, ragged=True)
    packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask))
    ref = orc.embed_batch(ids, mask)
    assert float(np.abs(packed - ref).max()) < TOL and float(np.abs(padded - ref).max()) < TOL
    assert float(np.abs(packed - padded).max()) < 1e-5


@pytest.mark.parametrize("pooling", ["mean", "cls", "max", "last_token"])
@pytest.mark.parametrize("normalize", [False, True])
def test_every_pooling_mode_on_packed_rows(minilm, pooling, normalize):
    enc, orc = minilm
    ids, mask = synth.synthetic_ids(33, 48, seed=5, ragged=True)
    h = orc.forward(ids, mask, None, O.strategy_mask_value(ids.size))
    mf = mask.astype(np.float32)
    ref = {"mean": O.mean_pool, "max": O.max_pool, "last_token": O.last_token_pool}.get(pooling, lambda a, m: O.cls_pool(a))(h, mf)
    if normalize:
        ref = O.l2_normalize(ref)
    packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask, pooling=pooling, normalize=normalize))
    assert float(np.abs(packed - ref).max()) < TOL and float(np.abs(packed - padded).max()) < 1e-5


def test_holes_in_the_mask(minilm):
    """A mask is any 0 / 1 pattern, not only right padding: kept tokens keep their own positions (position embeddings)
    and their order."""
    enc, orc = minilm
    rng = np.random.default_rng(3)
    ids, mask = synth.synthetic_ids(50, 96, seed=3, ragged=True)
    holes = rng.random(mask.shape) < 0.25
    holes[:, 0] = False
    mask[holes] = 0
    for pooling in ("mean", "last_token", "max"):
        h = orc.forward(ids, mask, None, O.strategy_mask_value(ids.size))
        ref = {"mean": O.mean_pool, "max": O.max_pool, "last_token": O.last_token_pool}[pooling](h, mask.astype(np.float32))
        packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask, pooling=pooling, normalize=False))
        assert float(np.abs(packed - ref).max()) < TOL, pooling
        assert float(np.abs(packed - padded).max()) < 1e-5, pooling


def test_calls_that_must_keep_the_padded_layout(minilm):
    """Token 0 masked in one sentence, an all-masked sentence, a mask value of 2: the results are the reference's (the
    oracle's), which only the padded rows can give."""
    import kjarni_amd
    enc, orc = minilm
    ids, mask = synth.synthetic_ids(6, 24, seed=8, ragged=True)
    m1 = mask.copy()
    m1[2, 0] = 0                       # CLS pooling reads a masked token's hidden state
    for pooling, fn in (("cls", lambda a, m: O.cls_pool(a)), ("mean", O.mean_pool)):
        ref = fn(orc.forward(ids, m1, None, O.MASK_ALLOC), m1.astype(np.float32))
        got = enc.embed(ids, m1, pooling=pooling, normalize=False, fill=kjarni_amd.MASK_NEG_1E9)
        assert float(np.abs(got - ref).max()) < TOL
    m2 = mask.copy()
    m2[4, :] = 0                       # all-masked: mean pool returns token 0's row (pooling/mod.rs:25-27)
    ref = orc.embed_batch(ids, m2, O.MASK_ALLOC)
    got = enc.embed(ids, m2, fill=kjarni_amd.MASK_NEG_1E9)
    assert float(np.abs(got - ref).max()) < TOL
    m3 = mask.copy()
    m3[1, 3] = 2                       # `as f32`: the mean pool weighs that token twice (traits.rs:71)
    ref = O.mean_pool(orc.forward(ids, m3, None, O.MASK_ALLOC), m3.astype(np.float32))
    got = enc.embed(ids, m3, normalize=False, fill=kjarni_amd.MASK_NEG_1E9)
    assert float(np.abs(got - ref).max()) < TOL


@pytest.mark.parametrize("B,S", [(3, 16), (70, 64), (300, 128), (1200, 128)])
def test_ragged_rerank_logits(cross, B, S):
    enc, orc = cross
    ids, mask, types = synth.synthetic_pairs(B, S, seed=B + S, qlen=min(16, S // 3))
    rng = np.random.default_rng(B)
    for i in range(B):
        if i % 4:
            n = int(rng.integers(S // 2, S + 1))
            ids[i, n - 1] = 102
            ids[i, n:] = 0
            mask[i, n:] = 0
            types[i, n:] = 0
    packed, padded = both_layouts(enc, lambda: enc.logits(ids, mask, types))
    ref = orc.rerank_scores(ids, mask, types)
    assert float(np.abs(packed[:, 0] - ref).max()) < TOL and float(np.abs(packed - padded).max()) < 1e-5


def test_packed_chunks_cover_the_call(minilm):
    """Chunks are runs of whole sentences within the token budget: any budget gives the same vectors, also one smaller
    than two sentences."""
    enc, _ = minilm
    ids, mask = synth.synthetic_ids(200, 64, seed=21, ragged=True)
    base = enc.embed(ids, mask)
    try:
        for budget in (64, 100, 1000, 5000):
            enc.set_chunk_tokens(budget)
            assert float(np.abs(enc.embed(ids, mask) - base).max()) < 1e-5, budget
    finally:
        enc.set_chunk_tokens(262144)


def test_device_pointer_calls_pack_too(minilm):
    """kjarni_hip_encoder_embed on device pointers, packing mode 2 (opt-in: the call synchronises its stream): the lengths
    come from a device-side scan of the mask.  In the default mode the same call takes the padded layout."""
    import torch
    enc, orc = minilm
    ids, mask = synth.synthetic_ids(150, 128, seed=31, ragged=True)
    dev = torch.device("cuda", 0)
    ti = torch.from_numpy(ids.view(np.int32)).to(dev)
    tm = torch.from_numpy(mask.view(np.int32)).to(dev)
    out = torch.empty((150, 384), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    enc.embed_dev(ti.data_ptr(), tm.data_ptr(), 150, 128, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    padded = out.cpu().numpy()
    assert float(np.abs(padded - orc.embed_batch(ids, mask)).max()) < TOL
    enc.set_packing(2)
    try:
        out.zero_()
        enc.embed_dev(ti.data_ptr(), tm.data_ptr(), 150, 128, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    finally:
        enc.set_packing(1)
    assert float(np.abs(got - orc.embed_batch(ids, mask)).max()) < TOL
    assert float(np.abs(got - padded).max()) < 1e-6
    assert float(np.abs(got - enc.embed(ids, mask)).max()) < 1e-6   # host-pointer call, same packed layout


def test_other_head_widths_and_rope(tmp_path):
    """d = 64 heads (DistilBERT-base shape: the pipelined kernel at two workgroups per CU up to 128 tokens -- 300 sentences x
    2 heads is more items than its 512 workgroups -- the general kernel above), a toy head width (the any-width kernel),
    and Nomic's RoPE (positions of packed rows come from their padded index)."""
    import kjarni_amd
    cases = [("distil", lambda d: synth.distilbert_sentiment(d, dim=128, n_layers=2, n_heads=2, hidden_dim=256)),
             ("toy", lambda d: synth.minilm_embedder(d, seed=4, hidden_size=48, num_hidden_layers=2, num_attention_heads=4,
                                                     intermediate_size=96)),
             ("nomic", lambda d: synth.nomic_embedder(d))]
    for name, make in cases:
        d = str(tmp_path / name)
        cfg, t = make(d)
        enc = kjarni_amd.HipEncoder(d, 0)
        orc = O.OracleModel(t, cfg)
        vocab = cfg.get("vocab_size", 30522)
        for B, S in ((7, 40), (5, 150), (300, 100)):
            ids, mask = synth.synthetic_ids(B, S, vocab=vocab, seed=B + S, ragged=True)
            packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask))
            ref = orc.embed_batch(ids, mask)
            assert float(np.abs(packed - ref).max()) < TOL, (name, B, S)
            assert float(np.abs(packed - padded).max()) < 1e-5, (name, B, S)
        enc.close()


def test_random_masks_packed_equals_padded(minilm, cross):
    """Seeded sweep: random batch sizes, padded lengths and mask patterns (right padding, holes, full rows, a one-token
    sentence), embed with a random pooling mode and rerank logits: the packed layout against the padded one."""
    rng = np.random.default_rng(2024)
    enc, _ = minilm
    cenc, _ = cross
    for case in range(30):
        B = int(rng.choice([1, 2, 3, 9, 33, 64, 65, 130, 700]))
        S = int(rng.choice([2, 7, 16, 33, 64, 100, 128, 129, 200, 384]))
        if B * S > 60000:
            B = max(1, 60000 // S)
        ids, mask = synth.synthetic_ids(B, S, seed=case, ragged=True)
        kind = case % 3
        if kind == 1:                                  # holes
            holes = rng.random(mask.shape) < 0.3
            holes[:, 0] = False
            mask[holes] = 0
        elif kind == 2:                                # a mix of full rows and one-token rows
            mask[::3] = 1
            ids[::3][ids[::3] == 0] = 1999
            if B > 1:
                mask[1, 1:] = 0
        pooling = ["mean", "cls", "max", "last_token"][case % 4]
        packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask, pooling=pooling, normalize=bool(case & 1)))
        assert np.isfinite(packed).all()
        assert float(np.abs(packed - padded).max()) < 1e-5, (case, B, S, pooling)
        types = (rng.random(mask.shape) < 0.5).astype(np.uint32)
        packed, padded = both_layouts(cenc, lambda: cenc.logits(ids, mask, types))
        assert float(np.abs(packed - padded).max()) < 1e-5, (case, B, S)


def test_out_of_vocabulary_ids_on_packed_rows(minilm):
    """ids >= vocab leave zero rows before the embedding LayerNorm (embeddings/mod.rs:232-236), also when the row is found
    through tok_src."""
    enc, orc = minilm
    ids, mask = synth.synthetic_ids(12, 40, seed=77, ragged=True)
    ids[3, 2] = 40000
    ids[7, 1] = 2 ** 31
    packed, padded = both_layouts(enc, lambda: enc.embed(ids, mask))
    ref = orc.embed_batch(ids, mask)
    assert float(np.abs(packed - ref).max()) < TOL and float(np.abs(packed - padded).max()) < 1e-5
