"""Several devices behind one handle (kjarni_hip_group_*, KJARNI_HIP_DEVICES for the string-level handles) and
concurrent calls on one handle.  The GPU box has one device, so the fan-out is exercised with that device listed
twice (two replicas, two host threads, two streams) and the RCCL path with a one-rank communicator; results must be
oracle-equal and equal to the single-replica result to rounding (1e-6): the projections pick their tile route by the
number of token rows in a call (<= 256, 257 .. 8 192, more), so a row block of a call and the whole call
are bit-equal only when both land in the same range."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_fanout_matches_single_device_and_oracle(tmp_path):
    import kjarni_amd
    from oracle import oracle as O
    cfg, t = synth.minilm_embedder(str(tmp_path / "e"), seed=0, num_hidden_layers=2)
    one = kjarni_amd.HipEncoder(str(tmp_path / "e"))
    grp = kjarni_amd.HipEncoderGroup(str(tmp_path / "e"), [0, 0])
    assert grp.size == 2 and grp.devices == [0, 0] and grp.hidden_size == cfg["hidden_size"]
    assert [grp.shard(9, i) for i in range(2)] == [(0, 5), (5, 4)]
    orc = O.OracleModel(t, cfg, blocked_gemm=True)
    # below 16 rows only one replica works; 37 = uneven blocks; 200 x 64 padded = 12 800 token rows whole, 6 400 per block:
    # the whole call and its blocks straddle the 8 192-row boundary between the tile routes
    for n, ragged in ((1, True), (15, True), (16, True), (37, True), (260, True), (200, False)):
        ids, mask = synth.synthetic_ids(n, 64, seed=n, ragged=ragged)
        got = grp.embed(ids, mask)
        assert float(np.abs(got - one.embed(ids, mask)).max()) <= 1e-6, n
        assert float(np.abs(got - orc.embed_batch(ids, mask)).max()) < 1e-4, n
    ids, mask = synth.synthetic_ids(1, 64, seed=1, ragged=True)   # one replica, same rows, same route: bit-equal
    assert np.array_equal(grp.embed(ids, mask), one.embed(ids, mask))
    cfg, t = synth.minilm_cross_encoder(str(tmp_path / "c"), seed=1, num_hidden_layers=2)
    one = kjarni_amd.HipEncoder(str(tmp_path / "c"))
    grp = kjarni_amd.HipEncoderGroup(str(tmp_path / "c"), [0, 0])
    ids, mask, types = synth.synthetic_pairs(41, 64, seed=2)
    got = grp.logits(ids, mask, types)
    assert float(np.abs(got - one.logits(ids, mask, types)).max()) <= 1e-6
    assert float(np.abs(got[:, 0] - O.OracleModel(t, cfg).rerank_scores(ids, mask, types)).max()) < 1e-4


@pytest.mark.parametrize("devices,transport", [([0, 0], "memcpy"), ([0], "rccl")])
@pytest.mark.parametrize("n", [32, 37])
def test_device_resident_allgather(tmp_path, devices, transport, n):
    """Every replica encodes its row block into its own full-size buffer; after the collective every buffer holds all
    rows.  [0, 0]: peer-copy transport (one device listed twice); [0]: ncclAllGather on a one-rank RCCL communicator."""
    import torch

    import kjarni_amd
    cfg, _ = synth.minilm_embedder(str(tmp_path / "e"), seed=0, num_hidden_layers=2)
    one = kjarni_amd.HipEncoder(str(tmp_path / "e"))
    grp = kjarni_amd.HipEncoderGroup(str(tmp_path / "e"), devices)
    assert grp.transport == transport
    ids, mask = synth.synthetic_ids(n, 32, seed=n, ragged=True)
    want = one.embed(ids, mask, fill=kjarni_amd.MASK_NEG_INF)
    blocks = [grp.shard(n, i) for i in range(grp.size)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()  # noqa: E731
    ids_d = [dev(ids[s:s + c]) for s, c in blocks]
    mask_d = [dev(mask[s:s + c]) for s, c in blocks]
    outs = [torch.full((n, grp.hidden_size), float("nan"), device="cuda") for _ in blocks]
    torch.cuda.synchronize()
    grp.embed_allgather([t.data_ptr() for t in ids_d], [t.data_ptr() for t in mask_d], n, 32,
                        [t.data_ptr() for t in outs], fill=kjarni_amd.MASK_NEG_INF)
    for o in outs:
        got = o.cpu().numpy()
        assert np.array_equal(got, outs[0].cpu().numpy())   # every buffer holds the same rows after the collective
        if len(devices) == 1:
            assert np.array_equal(got, want)                 # one replica = the same call
        else:
            assert float(np.abs(got - want).max()) <= 1e-6


def test_string_level_handles_fan_out_with_kjarni_hip_devices(tmp_path):
    """kjarni_embedder_encode_batch / kjarni_reranker_rerank in a process started with KJARNI_HIP_DEVICES=0,0 return what
    the same process gets with KJARNI_HIP_DEVICES=0 (the handles read the variable when they load)."""
    d = str(tmp_path / "e")
    synth.minilm_embedder(d, seed=0, num_hidden_layers=2)
    synth.add_tokenizer(d)
    c = str(tmp_path / "c")
    synth.minilm_cross_encoder(c, seed=1, num_hidden_layers=2)
    synth.add_tokenizer(c)
    code = (
        "import sys, numpy as np, kjarni_amd\\n"
        "texts = [('hello world ' * (1 + i % 7)).strip() for i in range(45)]\\n"
        "e = kjarni_amd.Embedder(model_path=sys.argv[1]); r = kjarni_amd.Reranker(model_path=sys.argv[2])\\n"
        "emb = e.encode_batch(texts)\\n"
        "rr = r.rerank('hello', texts)\\n"
        "np.save(sys.argv[3], emb); np.save(sys.argv[4], np.array([[x.index, x.score] for x in rr]))\\n")
    outs = {}
    for devs in ("0", "0,0"):
        env = dict(os.environ, KJARNI_HIP_DEVICES=devs, PYTHONPATH=ROOT)
        a, b = str(tmp_path / f"emb_{len(devs)}.npy"), str(tmp_path / f"rr_{len(devs)}.npy")
        p = subprocess.run([sys.executable, "-c", code.replace("\\n", "\n"), d, c, a, b], env=env, capture_output=True,
                           text=True, timeout=300, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-1500:]
        outs[devs] = (np.load(a), np.load(b))
    assert outs["0"][0].shape == (45, 384)
    assert float(np.abs(outs["0"][0] - outs["0,0"][0]).max()) <= 1e-6
    # rerank: the same (index -> score) pairs to rounding; equal texts tie, and their order may follow the rounding
    by_index = lambda a: a[np.argsort(a[:, 0], kind="stable")]  # noqa: E731
    a, b = by_index(outs["0"][1]), by_index(outs["0,0"][1])
    assert np.array_equal(a[:, 0], b[:, 0]) and float(np.abs(a[:, 1] - b[:, 1]).max()) <= 1e-6
    for r in (outs["0"][1], outs["0,0"][1]):
        assert (r[:-1, 1] >= r[1:, 1]).all()


def test_bad_device_list_is_invalid_config(tmp_path):
    import kjarni_amd
    synth.minilm_embedder(str(tmp_path / "e"), seed=0, num_hidden_layers=1)
    with pytest.raises(kjarni_amd.KjarniException):
        kjarni_amd.HipEncoderGroup(str(tmp_path / "e"), [0, 99])


@pytest.mark.parametrize("combining", [False, True])
def test_two_threads_on_one_handle_are_oracle_equal(tmp_path, monkeypatch, combining):
    """The reference serialises nothing on a handle (kjarni-ffi/src/lib.rs:25-32).  Sixteen host threads hammer ONE
    token-level handle and ONE string-level handle with different inputs -- batches and single sentences.  By default every call
    runs its own forward on a leased workspace: results equal the oracle at 1e-4 and the single-threaded result BIT FOR BIT (a
    shared workspace would mix the threads' activations).  With combining opted in (KJARNI_HIP_COMBINE=1, read when a handle
    loads its model; kjarni_hip.h: kjarni_hip_encoder_set_combining) small calls share forwards while larger ones run on
    workspaces of their own: a combined call takes the packed layout, hence 1e-6 and not bit equality."""
    monkeypatch.setenv("KJARNI_HIP_COMBINE", "1" if combining else "0")
    import kjarni_amd
    from oracle import oracle as O
    d = str(tmp_path / "e")
    cfg, t = synth.minilm_embedder(d, seed=0, num_hidden_layers=2)
    synth.add_tokenizer(d)
    enc = kjarni_amd.HipEncoder(d)
    oracle = O.OracleModel(t, cfg)
    n_threads = 16
    # threads 0-3: ragged batches of 5 .. 14 sentences (5 and 8 rows are small calls, the others are not); 4-15: one sentence
    inputs = [synth.synthetic_ids(5 + 3 * i, 32 + 16 * i, seed=10 + i, ragged=True) for i in range(4)]
    inputs += [synth.synthetic_ids(1, 12 + 3 * i, seed=40 + i) for i in range(n_threads - 4)]
    refs = [oracle.embed_batch(i, m) for i, m in inputs]
    solo = [enc.embed(i, m) for i, m in inputs]
    emb = kjarni_amd.Embedder(model_path=d)
    texts = [[f"sentence number {j} of thread {i} " * (1 + j % 3) for j in range(6 + i)] for i in range(4)]
    texts += [[f"a lone sentence from thread {i} " * (1 + i % 4)] for i in range(4, n_threads)]
    text_refs = [emb.encode_batch(tx) for tx in texts]
    errors = []

    def work(i):
        try:
            for _ in range(25):
                got = enc.embed(*inputs[i])
                if float(np.abs(got - refs[i]).max()) >= 1e-4:
                    errors.append(f"token-level thread {i} vs oracle: {float(np.abs(got - refs[i]).max())}")
                    return
                if (float(np.abs(got - solo[i]).max()) > 1e-6) if combining else (not np.array_equal(got, solo[i])):
                    errors.append(f"token-level thread {i} vs solo: {float(np.abs(got - solo[i]).max())}")
                    return
                got = emb.encode_batch(texts[i])
                if got.shape != text_refs[i].shape or ((float(np.abs(got - text_refs[i]).max()) > 1e-6) if combining
                                                       else (not np.array_equal(got, text_refs[i]))):
                    errors.append(f"string-level thread {i}")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors


@pytest.mark.parametrize("explicit", [False, True])
def test_without_combining_threads_get_the_solo_bits(tmp_path, monkeypatch, explicit):
    """The default (and kjarni_hip_encoder_set_combining(0) after an opt-in): every call runs alone, so concurrent callers get
    exactly the single-threaded bits -- the reference's behaviour (each call its own deterministic result)."""
    import kjarni_amd
    d = str(tmp_path / "e")
    synth.minilm_embedder(d, seed=0, num_hidden_layers=2)
    if explicit:
        monkeypatch.setenv("KJARNI_HIP_COMBINE", "1")
    else:
        monkeypatch.delenv("KJARNI_HIP_COMBINE", raising=False)
    enc = kjarni_amd.HipEncoder(d)
    if explicit:
        enc.set_combining(False)
    inputs = [synth.synthetic_ids(1 + i % 3, 16 + 5 * i, seed=70 + i, ragged=True) for i in range(8)]
    solo = [enc.embed(i, m) for i, m in inputs]
    errors = []

    def work(i):
        for _ in range(30):
            if not np.array_equal(enc.embed(*inputs[i]), solo[i]):
                errors.append(i)
                return

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors


def test_failed_workspace_allocation_leaves_the_handle_usable(tmp_path):
    """An oversized batch must fail with an error code -- and the next, small call must work (the workspace capacity is
    reset before anything is freed, so no kernel ever runs on null buffers)."""
    import kjarni_amd
    from kjarni_amd import _ffi
    cfg, _ = synth.minilm_embedder(str(tmp_path / "e"), seed=0, num_hidden_layers=1)
    enc = kjarni_amd.HipEncoder(str(tmp_path / "e"))
    ids, mask = synth.synthetic_ids(4, 16, seed=0)
    before = enc.embed(ids, mask)
    enc.set_chunk_tokens(1 << 42)
    L = kjarni_amd.lib()
    dummy = C.c_void_p()
    assert L.kjarni_hip_malloc(0, 4096, C.byref(dummy)) == 0
    # 2^31 sentences x 16 tokens: the activation buffers alone would need 10^14 bytes; nothing is dereferenced
    rc = L.kjarni_hip_encoder_embed(enc._h, dummy, dummy, None, 1 << 31, 16, 0, 1, 0, dummy, None)
    assert rc == _ffi.KjarniError.INFERENCE_FAILED and b"hipMalloc" in L.kjarni_last_error_message()
    enc.set_chunk_tokens(131072)
    assert np.array_equal(enc.embed(ids, mask), before)
    L.kjarni_hip_free(0, dummy)
