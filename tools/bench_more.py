#!/usr/bin/env python3
"""The other measurements SURVEY.md section 8(d) asks for, next to bench.py's headline line.

    python tools/bench_more.py [rerank] [scan] [ragged] [host] [strings] [indexer] [whisper] [llm]   (default: all)

One JSON line per measurement (1 GPU; the N > 1 driver is bench.py):
  rerank   BASELINE.json configs[2] on one GPU: 100 000 synthetic (query, doc) pairs, S = 128, fp32,
           ids/mask/types resident in HBM -> 100 000 logits.  pairs/s + fraction of the fp32 MFMA peak.
  scan     cosine scan + top-10 over unit-norm Gaussian corpora [N, 384], N in {1e5, 1e6, 1e7}, 1 and 64
           queries, corpus resident in HBM.  docs/s and the scan kernel's algorithmic GB/s vs 8 TB/s.
  ragged   the headline embed workload with lengths ~ U{16..128} (right-padded): masking cost.
  host     the headline embed workload through the host-pointer entry point: H2D of ids/mask and D2H of
           the embeddings inside the timed region (the PCIe-inclusive rate; never the headline value).
  strings  kjarni_embedder_encode_batch on generated ASCII sentences (tokenisation on the host included).
  indexer  kjarni_indexer_create over a generated directory tree: chunks/s end to end.
  whisper  BASELINE.json configs[3]: Whisper-base shaped model (random init), 30 s of synthetic audio: log-mel,
           conv stem + encoder, greedy decode of 448 tokens; per-stage ms, x real time, and the CPU restatement
           (oracle) timed on a bounded sample of the same work.
  llm      BASELINE.json configs[4]: Llama-3.2-1B-shaped decoder (random init), weights bf16 in HBM, f32 KV cache:
           prefill of a 128-token prompt and greedy decode of 256 tokens; tokens/s and the weight stream vs HBM peak.
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
S = 128


def flops_per_row(H=384, L=6, I=1536, seq=S, head=0):
    per_layer = 2 * seq * H * 3 * H + 2 * 2 * seq * seq * H + 2 * seq * H * H + 2 * 2 * seq * H * I
    return L * per_layer + head


def usable_cores():
    """Threads for a CPU leg: physical cores, capped by the affinity mask and the cgroup CPU quota (as bench.py's)."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def timed(fn, sync, steps, warmup):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    which = set(sys.argv[1:]) or {"rerank", "scan", "ragged", "host", "sweep", "strings", "latency", "families", "indexer", "search", "whisper", "llm", "chat"}  # "llm8b" only on request (writes 16 GB)
    import numpy as np
    import torch

    import kjarni_amd
    from kjarni_amd import _ffi
    from tests import synth

    assert torch.cuda.is_available() and kjarni_amd.device_count() >= 1, "needs an AMD GPU"
    dev = torch.device("cuda", 0)
    L = _ffi.lib()
    stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    sync = torch.cuda.synchronize

    def emit(d):
        print(json.dumps(d), flush=True)

    tmp = tempfile.mkdtemp(prefix="kjarni_bench_more_")
    emb_dir = os.path.join(tmp, "cache", "sentence-transformers_all-MiniLM-L6-v2")
    cfg, _ = synth.minilm_embedder(emb_dir, seed=0)
    synth.add_tokenizer(emb_dir)

    if "rerank" in which:
        d = os.path.join(tmp, "ce")
        cfg_r, _ = synth.minilm_cross_encoder(d, seed=1)
        enc = kjarni_amd.HipEncoder(d, 0)
        N = 100_000
        ids, mask, types = synth.synthetic_pairs(N, S, seed=1)
        t_ids, t_mask, t_types = (torch.from_numpy(a.view(np.int32)).to(dev) for a in (ids, mask, types))
        out = torch.empty((N, enc.num_labels), dtype=torch.float32, device=dev)
        dt = timed(lambda: enc.logits_dev(t_ids.data_ptr(), t_mask.data_ptr(), t_types.data_ptr(), N, S, out.data_ptr(),
                                          stream=stream()), sync, steps=3, warmup=1)
        fl = flops_per_row(head=2 * 384 * 384 + 2 * 384)
        emit({"metric": "pairs/sec minilm-l6-v2-cross-encoder rerank (seq=128)", "value": round(N / dt, 1), "unit": "pairs/s",
              "n_gpus": 1, "ms_per_step": round(dt * 1e3, 2), "dtype": "f32", "data": "synthetic",
              "config": {"workload": "BASELINE.json configs[2] on one GPU: 100 000 synthetic query-doc pairs, seq_len=128"},
              "e2e_tflops": round(N / dt * fl / 1e12, 2),
              "e2e_frac_fp32_mfma_peak": round(N / dt * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)})
        del enc, t_ids, t_mask, t_types, out

    if "families" in which:
        # The other registry embedders at their real shapes (random weights): nomic-embed-text-v1.5 (RoPE + SwiGLU, 768 x 12
        # layers, inner 3072), all-mpnet-base-v2 (BERT-base shape) and bge-m3 (XLM-R large: 1024 x 24 layers, inner 4096,
        # 250 002-row embedding table, up to 8 192 tokens).  Same token-level entry point as the headline.
        for name, make, over, gated in (
                ("nomic-embed-text", synth.nomic_embedder, dict(n_embd=768, n_layer=12, n_head=12, n_inner=3072, n_positions=8192), True),
                ("mpnet-base-v2", synth.mpnet_embedder, dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                                             intermediate_size=3072), False),
                ("bge-m3", synth.xlmr_embedder, dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                                     vocab_size=250002, max_position_embeddings=8194), False)):
            d = os.path.join(tmp, name)
            cfg_f, _ = make(d, **over)
            enc = kjarni_amd.HipEncoder(d, 0)
            H = cfg_f.get("hidden_size", cfg_f.get("n_embd"))
            Lf = cfg_f.get("num_hidden_layers", cfg_f.get("n_layer"))
            If = cfg_f.get("intermediate_size", cfg_f.get("n_inner"))
            shapes = ((16384, 128), (1024, 2048)) if gated else ((8192, 128), (256, 8192)) if name == "bge-m3" else ((16384, 128), (4096, 512))
            for N, seq in shapes:
                ids, mask = synth.synthetic_ids(N, seq, vocab=cfg_f["vocab_size"], seed=0)
                t_ids, t_mask = (torch.from_numpy(a.view(np.int32)).to(dev) for a in (ids, mask))
                out = torch.empty((N, H), dtype=torch.float32, device=dev)
                run = lambda: enc.embed_dev(t_ids.data_ptr(), t_mask.data_ptr(), N, seq, out.data_ptr(), stream=stream())  # noqa: E731
                dt = timed(run, sync, steps=2, warmup=1)
                enc.profile_begin()
                run()
                stats = [k for k in enc.profile_end() if k["launches"]]
                ffn = (3 if gated else 2) * 2 * seq * H * If
                fl = Lf * (2 * seq * H * 3 * H + 4 * seq * seq * H + 2 * seq * H * H + ffn)
                emit({"metric": f"sentences/sec {name} batch encode (seq={seq})", "value": round(N / dt, 1), "unit": "sentences/s",
                      "n_gpus": 1, "ms_per_step": round(dt * 1e3, 2), "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"{N} sentences x {seq} tokens, ids/mask in HBM, mean pool + L2"},
                      "tokens_per_s": round(N * seq / dt, 0), "e2e_tflops": round(N / dt * fl / 1e12, 2),
                      "e2e_frac_fp32_mfma_peak": round(N / dt * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                      "kernels": [{"kind": k["kind"], "launches": k["launches"], "ms": round(k["total_ms"], 2),
                                   "tflops": round(k["flops"] / k["total_ms"] / 1e9, 1) if k["flops"] else None,
                                   "gbs": round(k["bytes"] / k["total_ms"] / 1e6, 0)} for k in stats]})
                del t_ids, t_mask, out
            del enc

    if "ragged" in which or "host" in which:
        enc = kjarni_amd.HipEncoder(emb_dir, 0)
        N = 65536
        if "ragged" in which:
            ids, mask = synth.synthetic_ids(N, S, seed=0, ragged=True)
            t_ids, t_mask = (torch.from_numpy(a.view(np.int32)).to(dev) for a in (ids, mask))
            out = torch.empty((N, 384), dtype=torch.float32, device=dev)
            # device-pointer calls keep the padded layout unless the caller opts in (mode 2 reads the lengths back and
            # synchronises the stream once per call: include/kjarni_hip.h); both are reported
            dt_padded = timed(lambda: enc.embed_dev(t_ids.data_ptr(), t_mask.data_ptr(), N, S, out.data_ptr(), stream=stream()),
                              sync, steps=2, warmup=1)
            enc.set_packing(2)
            dt = timed(lambda: enc.embed_dev(t_ids.data_ptr(), t_mask.data_ptr(), N, S, out.data_ptr(), stream=stream()),
                       sync, steps=3, warmup=1)
            enc.set_packing(1)
            emit({"metric": "sentences/sec minilm-l6-v2 batch encode (seq=128), ragged lengths U{16..128} right-padded",
                  "value_padded_layout": round(N / dt_padded, 1),
                  "value": round(N / dt, 1), "unit": "sentences/s", "n_gpus": 1, "ms_per_step": round(dt * 1e3, 2),
                  "dtype": "f32", "data": "synthetic", "config": {"workload": "65 536 sentences padded to 128, mean length 72"},
                  "real_tokens_per_s": round(float(mask.sum()) / dt, 0)})
            del t_ids, t_mask, out
        if "host" in which:
            ids, mask = synth.synthetic_ids(N, S, seed=0)
            dt = timed(lambda: enc.embed(ids, mask), lambda: None, steps=2, warmup=1)
            emit({"metric": "sentences/sec minilm-l6-v2 batch encode (seq=128), host pointers (H2D ids/mask + D2H embeddings timed)",
                  "value": round(N / dt, 1), "unit": "sentences/s", "n_gpus": 1, "ms_per_step": round(dt * 1e3, 2),
                  "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "65 536 sentences x 128 through kjarni_hip_encoder_embed_host (pageable host memory)"}})
        del enc

    if "sweep" in which:
        # Calls of the sizes real callers make (the reference's default batch is 32, kjarni-ffi/src/embedder.rs): host
        # pointers in and out, one call at a time, sentences/s against the call size and the padded length.
        enc = kjarni_amd.HipEncoder(emb_dir, 0)
        rows = []
        for seq in (32, 128):
            for b in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096):
                ids, mask = synth.synthetic_ids(b, seq, seed=1)
                reps = max(3, min(200, 4096 // b))
                for _ in range(50 if not rows else 1):  # (the first size also absorbs the clock ramp from idle)
                    enc.embed(ids, mask)
                t0 = time.perf_counter()
                for _ in range(reps):
                    enc.embed(ids, mask)
                dt = (time.perf_counter() - t0) / reps
                rows.append({"batch": b, "seq": seq, "ms_per_call": round(dt * 1e3, 4), "sentences_per_s": round(b / dt, 1),
                             "frac_fp32_mfma_peak": round(b * flops_per_row(seq=seq) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)})
        emit({"metric": "sentences/sec minilm-l6-v2 embed against the call size (host pointers, one call at a time)",
              "unit": "sentences/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic", "value": rows[-1]["sentences_per_s"],
              "config": {"workload": "kjarni_hip_encoder_embed_host, batch 1 .. 4096 x seq 32 / 128"}, "calls": rows})
        del enc

    if "scan" in which:
        dim, k = 384, 10
        for n in (100_000, 1_000_000, 10_000_000):
            g = torch.Generator(device=dev).manual_seed(2)
            corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
            corpus /= torch.linalg.vector_norm(corpus, dim=1, keepdim=True)
            for nq in (1, 64):
                q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
                scores = torch.empty((nq, n), dtype=torch.float32, device=dev)
                ws = torch.empty(L.kjarni_hip_cosine_topk_workspace_bytes(nq, n, k), dtype=torch.uint8, device=dev)
                idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
                sc = torch.empty((nq, k), dtype=torch.float32, device=dev)

                def scan():
                    _ffi.check_error(L.kjarni_hip_cosine_scores(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, 1,
                                                                scores.data_ptr(), stream()))

                def topk():
                    _ffi.check_error(L.kjarni_hip_cosine_topk(0, scores.data_ptr(), nq, n, k, ws.data_ptr(), idx.data_ptr(),
                                                              sc.data_ptr(), stream()))
                steps = 20 if n <= 1_000_000 else 5
                t_scan = timed(scan, sync, steps, 2)
                t_topk = timed(topk, sync, steps, 2)
                t_all = timed(lambda: (scan(), topk()), sync, steps, 2)
                two_idx, two_sc = idx.clone(), sc.clone()
                # the one-call form (kjarni_hip_cosine_search): for one query a single fused pass, no score array
                ws2 = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, n, dim, k), dtype=torch.uint8, device=dev)

                def search():
                    _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, 1, k, ws2.data_ptr(),
                                                                idx.data_ptr(), sc.data_ptr(), stream()))
                t_search = timed(search, sync, steps, 2)
                fused_equal = bool(torch.equal(idx, two_idx)) and bool(torch.equal(sc, two_sc))
                del ws2
                ref_idx = torch.topk(q @ corpus.T if n <= 1_000_000 else scores, k, dim=1).indices
                same = bool(torch.equal(torch.sort(ref_idx, dim=1).values, torch.sort(idx, dim=1).values))
                alg = n * dim * 4
                if nq < 20:   # one streaming pass over the corpus per group of 4 queries: HBM-bound
                    passes = (nq + 3) // 4
                    roof = {"kernel": "cosine_scores_stream_kernel", "bound": "hbm",
                            "achieved": round(alg * passes / t_scan / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(alg * passes / t_scan / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_launch": alg, "launches_per_scan": passes}
                else:         # the fused matrix-core scan: dots, document norms and cosines in one launch per 64 queries
                    fl = 2.0 * nq * n * dim
                    roof = {"kernel": "cosine_scan_mfma_kernel", "bound": "mfma",
                            "achieved": round(fl / t_scan / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fl / t_scan / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                            "algorithmic_flops": fl, "corpus_gbs": round(alg * ((nq + 63) // 64) / t_scan / 1e9, 1),
                            "note": "32 flop per corpus byte at 64 queries: 4.9 TB/s of HBM at the f32 MFMA peak"}
                emit({"metric": "doc-queries/sec cosine scan + top-10 (dim 384)", "value": round(n * nq / t_all, 0),
                      "unit": "doc-queries/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"corpus [{n}, 384] unit-norm Gaussian rows resident in HBM, {nq} quer{'y' if nq == 1 else 'ies'}, k=10"},
                      "topk_set_equals_torch_topk": same, "ms_scan": round(t_scan * 1e3, 4), "ms_topk": round(t_topk * 1e3, 4),
                      "ms_total": round(t_all * 1e3, 4), "ms_search_one_call": round(t_search * 1e3, 4),
                      "search_gbs": round(alg * ((nq + 3) // 4 if nq < 20 else (nq + 63) // 64) / t_search / 1e9, 1),
                      "one_call_equals_two_calls_bit_for_bit": fused_equal, "roofline": roof})
                del scores, ws, idx, sc, q
            del corpus
            torch.cuda.empty_cache()

    if "strings" in which:
        rng = np.random.default_rng(0)
        vocab = json.load(open(os.path.join(emb_dir, "tokenizer.json")))["model"]["vocab"]
        words = [w for w in vocab if w.isalpha() and len(w) > 2][:4000] or ["alpha", "beta", "gamma"]
        n = 65536
        texts = [" ".join(rng.choice(words, 100)) for _ in range(n)]
        emb = kjarni_amd.Embedder("minilm-l6-v2", cache_dir=os.path.join(tmp, "cache"))
        emb.encode_batch(texts[:1024])
        t0 = time.perf_counter()
        out = emb.encode_batch(texts)
        dt = time.perf_counter() - t0
        tok = kjarni_amd.Tokenizer(os.path.join(emb_dir, "tokenizer.json"), 512)
        t0 = time.perf_counter()
        ids, mask, _ = tok.encode_batch(texts)
        dt_tok = time.perf_counter() - t0
        emit({"metric": "sentences/sec kjarni_embedder_encode_batch (strings in, host tokenisation included)",
              "value": round(n / dt, 1), "unit": "sentences/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"{n} generated 100-word ASCII sentences, padded length {ids.shape[1]}"},
              "tokenise_only_sentences_per_s": round(n / dt_tok, 1), "host_threads": os.cpu_count(), "rows": int(out.shape[0])})
        # real text is ragged: the same handle on sentences of 8 .. 100 words (BatchLongest pads the call to the longest,
        # pipeline/encoder/loader.rs:98-115; the layers run over the kept tokens only)
        texts = [" ".join(rng.choice(words, int(rng.integers(8, 101)))) for _ in range(n)]
        emb.encode_batch(texts[:1024])
        t0 = time.perf_counter()
        out = emb.encode_batch(texts)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        ids, mask, _ = tok.encode_batch(texts)
        dt_tok = time.perf_counter() - t0
        emit({"metric": "sentences/sec kjarni_embedder_encode_batch (ragged strings in, host tokenisation included)",
              "value": round(n / dt, 1), "unit": "sentences/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"{n} generated ASCII sentences of 8 .. 100 words, padded length {ids.shape[1]}, "
                                     f"kept tokens {float(mask.sum()) / mask.size:.3f} of the padded"},
              "tokenise_only_sentences_per_s": round(n / dt_tok, 1), "host_threads": os.cpu_count(), "rows": int(out.shape[0])})
        del emb

    if "latency" in which:
        # BASELINE.json configs[0] is a single-sentence classify; the reference serves it on the calling thread
        # (kjarni-ffi/src/lib.rs:25-32).  Per-call latency of the string-level entry points for ONE ~16-token sentence
        # (tokenise + H2D + forward + D2H + result shaping), single caller and four concurrent callers on one handle.
        import threading
        cls_dir = os.path.join(tmp, "cache", "distilbert_distilbert-base-uncased-finetuned-sst-2-english")
        synth.distilbert_sentiment(cls_dir, seed=2, n_layers=6, dim=768, n_heads=12, hidden_dim=3072)
        synth.add_tokenizer(cls_dir)
        sentence = "the quick brown fox jumps over the lazy dog near the old river bank today"
        emb = kjarni_amd.Embedder("minilm-l6-v2", cache_dir=os.path.join(tmp, "cache"))
        clf = kjarni_amd.Classifier(model_path=cls_dir)
        tok = kjarni_amd.Tokenizer(os.path.join(emb_dir, "tokenizer.json"), 512)
        n_tok = int(tok.encode_batch([sentence])[0].shape[1])

        def lat(fn, n=400, warm=30):
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            return {"p50_ms": round(ts[len(ts) // 2], 4), "p99_ms": round(ts[int(len(ts) * 0.99)], 4), "min_ms": round(ts[0], 4)}

        def concurrent(fn, threads=4, n=200):
            res = [None] * threads

            def run(i):
                res[i] = lat(fn, n=n, warm=10)
            th = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            return {"threads": threads, "calls_per_s": round(threads * (n + 10) / wall, 1),
                    "p50_ms": round(float(np.median([r["p50_ms"] for r in res])), 4),
                    "p99_ms": round(max(r["p99_ms"] for r in res), 4)}

        emit({"metric": "latency of one sentence through the string-level C ABI", "unit": "ms", "tokens": n_tok, "n_gpus": 1,
              "dtype": "f32", "data": "synthetic",
              "kjarni_embedder_encode (minilm-l6-v2 shape)": lat(lambda: emb.encode(sentence)),
              "kjarni_classifier_classify (distilbert-sst2 shape, 6 x 768)": lat(lambda: clf.classify(sentence)),
              "kjarni_embedder_encode, 4 threads on one handle": concurrent(lambda: emb.encode(sentence)),
              "kjarni_embedder_encode, 16 threads on one handle": concurrent(lambda: emb.encode(sentence), threads=16),
              "kjarni_classifier_classify, 4 threads on one handle": concurrent(lambda: clf.classify(sentence)),
              "kjarni_classifier_classify, 16 threads on one handle": concurrent(lambda: clf.classify(sentence), threads=16),
              "combining": os.environ.get("KJARNI_HIP_COMBINE", "0") not in ("", "0"),
              "note": "with KJARNI_HIP_COMBINE=1 small calls that arrive while another is on the device are combined into one "
                      "packed forward (kjarni_hip_encoder_set_combining: opt-in, default off)"})
        del emb, clf

    if "indexer" in which:
        rng = np.random.default_rng(0)
        words = ["alpha", "beta", "gamma", "delta", "kernel", "vector", "index", "search", "wave", "matrix", "iceland", "river"]
        docs = os.path.join(tmp, "docs")
        os.makedirs(docs)
        for i in range(400):
            paras = [" ".join(rng.choice(words, int(rng.integers(20, 80)))) + "." for _ in range(40)]
            with open(os.path.join(docs, f"doc{i:04d}.txt"), "w") as f:
                f.write("\n\n".join(paras))
        ix = kjarni_amd.Indexer(cache_dir=os.path.join(tmp, "cache"), quiet=True)
        ix.create(os.path.join(tmp, "warm"), [os.path.join(docs, "doc0000.txt")])
        t0 = time.perf_counter()
        st = ix.create(os.path.join(tmp, "index"), [docs])
        dt = time.perf_counter() - t0
        emit({"metric": "chunks/sec kjarni_indexer_create (files -> chunks -> embeddings -> segments on disk)",
              "value": round(st.documents_indexed / dt, 1), "unit": "chunks/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"400 generated text files, chunk_size 512 / overlap 50 / batch_size 32 (defaults): "
                                     f"{st.documents_indexed} chunks, index {st.size_bytes} bytes"},
              "elapsed_ms": st.elapsed_ms})

    if "search" in which:
        # End-to-end retrieval over an on-disk index in the reference's segmented layout: 200 000 documents in 20 segments of
        # 10 000 (the default max_docs_per_segment), dim 384; every query re-opens the index as Searcher::search does, the
        # segments' vectors stay resident in HBM between queries.
        from kjarni_amd.indexer import index_write
        from kjarni_amd.searcher import index_search, search_keywords
        rng = np.random.default_rng(3)
        n_docs, dim = 200_000, 384
        words = ["alpha", "beta", "gamma", "delta", "kernel", "vector", "index", "search", "wave", "matrix", "iceland", "river", "glacier",
                 "basalt", "harbour", "northern", "lights", "wool", "fjord", "geyser"]
        ipath = os.path.join(tmp, "big_index")
        t0 = time.perf_counter()
        for start in range(0, n_docs, 50_000):
            texts = [" ".join(rng.choice(words, 12)) + f" doc{start + i}" for i in range(50_000)]
            emb = rng.standard_normal((50_000, dim), dtype=np.float32)
            index_write(ipath, dim, texts, emb, [{"source": f"f{(start + i) % 97}.txt"} for i in range(50_000)], append=start > 0)
        t_build = time.perf_counter() - t0
        queries = rng.standard_normal((64, dim), dtype=np.float32)
        index_search(ipath, None, queries[0], mode="semantic", top_k=10)      # first query uploads the segments
        runs = {}
        for label, fn in (("semantic", lambda q: index_search(ipath, None, q, mode="semantic", top_k=10)),
                          ("keyword", lambda q: search_keywords(ipath, "glacier fjord basalt", 10)),
                          ("hybrid", lambda q: index_search(ipath, "glacier fjord basalt", q, mode="hybrid", top_k=10)),
                          ("semantic_filtered", lambda q: index_search(ipath, None, q, mode="semantic", top_k=10, source_pattern="f1*.txt"))):
            fn(queries[1])
            from kjarni_amd.searcher import search_breakdown
            t0 = time.perf_counter()
            acc = {}
            for q in queries[:32]:
                r = fn(q)
                for k_, v_ in search_breakdown().items():
                    acc[k_] = acc.get(k_, 0.0) + v_ / 32
            dt = (time.perf_counter() - t0) / 32
            # per query, microseconds: re-opening the index | the scan over the device image (of it the device round trip) | the
            # rest on the host (BM25, fusion, reading the hits' documents and metadata back)
            runs[label] = {"ms_per_query": round(dt * 1e3, 3), "queries_per_s": round(1.0 / dt, 1), "results": len(r),
                           "breakdown_us": {k_: round(v_, 1) for k_, v_ in acc.items()}}
        # the whole Searcher: query string -> tokenise -> embed on the GPU -> retrieval (-> cross-encoder over 5x candidates)
        ce_dir = os.path.join(tmp, "cache", "cross-encoder_ms-marco-MiniLM-L-6-v2")
        synth.minilm_cross_encoder(ce_dir, seed=1)
        synth.add_tokenizer(ce_dir)
        for label, kw in (("searcher_semantic", dict(mode="semantic")), ("searcher_hybrid", dict(mode="hybrid")),
                          ("searcher_hybrid_rerank", dict(mode="hybrid", rerank=True))):
            sr = kjarni_amd.Searcher("minilm-l6-v2", "minilm-l6-v2-cross-encoder" if "rerank" in label else None, device="gpu",
                                     cache_dir=os.path.join(tmp, "cache"))
            qs = [" ".join(rng.choice(words, 6)) for _ in range(33)]
            sr.search(ipath, qs[0], top_k=10, **kw)
            t0 = time.perf_counter()
            for q in qs[1:]:
                r = sr.search(ipath, q, top_k=10, **kw)
            dt = (time.perf_counter() - t0) / 32
            runs[label] = {"ms_per_query": round(dt * 1e3, 3), "queries_per_s": round(1.0 / dt, 1), "results": len(r)}
            del sr
        emit({"metric": "ms per query, retrieval over an on-disk index (200 000 docs x 384, 20 segments), top-10", "unit": "ms",
              "value": runs["semantic"]["ms_per_query"], "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "kjarni_hip_index_search / kjarni_search_keywords on an index written by kjarni_index_write; "
                                     "documents and metadata are read back from docs.bin / metadata.jsonl for the 10 hits"},
              "index_build_seconds": round(t_build, 2), "runs": runs})

    if "whisper" in which:
        d = os.path.join(tmp, "whisper-base")
        cfg_w, t_w = synth.whisper_model(d, seed=0, base=True)
        wm = kjarni_amd.HipWhisper(d)
        audio = synth.synthetic_audio(30.0, seed=1)
        n_tok = 448
        prompt = [50258, 50259, 50359, 50363]
        wm.encode_audio(audio, fetch=False)
        wm.greedy(prompt, False, 8)                                   # warm-up
        t_mel = timed(lambda: wm.log_mel(audio), lambda: None, 5, 1)  # includes the D2H of the mel
        t_enc = timed(lambda: wm.encode_audio(audio, fetch=False), lambda: None, 5, 1)   # mel + stem + encoder, synchronised
        t0 = time.perf_counter()
        ids = wm.greedy(prompt, False, n_tok)
        t_dec = time.perf_counter() - t0
        total = t_enc + t_dec
        H, L_, I, S_, V = 512, 6, 2048, 1500, 51865
        enc_flops = L_ * (2 * S_ * H * 3 * H + 4 * S_ * S_ * H + 2 * S_ * H * H + 4 * S_ * H * I) + 2 * 3000 * 512 * 240 + 2 * 1500 * 512 * 1536
        # decoder step: weights streamed once per token (HBM-bound): 6 layers x (8 HxH + 2 HxI) + lm head, + cross K/V reads
        dec_bytes = 4 * (L_ * (8 * H * H + 2 * H * I) + V * H + L_ * 2 * S_ * H)
        res = {"metric": "x real time, Whisper-base shaped transcribe of 30 s audio (log-mel + encoder + greedy decode)",
               "value": round(30.0 / total, 1), "unit": "x real time", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[3]: whisper-base shape (d=512, 6+6 layers, vocab 51865), random init, "
                                      f"30 s synthetic audio, {len(ids)} generated tokens (EOS is never the argmax with random weights)"},
               "ms_log_mel_incl_d2h": round(t_mel * 1e3, 3), "ms_mel_stem_encoder": round(t_enc * 1e3, 3),
               "ms_decode": round(t_dec * 1e3, 2), "ms_per_token": round(t_dec * 1e3 / len(ids), 4),
               "tokens_per_s": round(len(ids) / t_dec, 1),
               "encoder_tflops": round(enc_flops / t_enc / 1e12, 2),
               "decode_weight_stream_gbs": round(dec_bytes * len(ids) / t_dec / 1e9, 1),
               "decode_frac_hbm_peak": round(dec_bytes * len(ids) / t_dec / 1e9 / PEAK_HBM_GBS, 4)}
        # CPU restatement (oracle): one encoder pass + a few decoder steps, extrapolated to the same token count
        from oracle import whisper_oracle as WO
        cores = usable_cores()
        from oracle import oracle as O
        O.lib().ko_set_num_threads(int(cores))
        orc = WO.WhisperOracle(t_w, cfg_w)
        t0 = time.perf_counter()
        mel = WO.log_mel(audio)
        c_mel = time.perf_counter() - t0
        t0 = time.perf_counter()
        enc = orc.encode_mel(mel)
        c_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.decode_chunk_ids(enc, max_tokens=15)
        c_dec16 = time.perf_counter() - t0
        c_total = c_mel + c_enc + c_dec16 / 16 * len(ids)
        res["cpu_baseline"] = {"value": round(30.0 / c_total, 3), "unit": "x real time", "cores": int(cores), "kind": "port",
                               "sample": f"oracle: log-mel {c_mel:.2f} s (vectorised DFT, not the reference's scalar O(n^2) loop), "
                                         f"encoder {c_enc:.2f} s, 16 decoder steps {c_dec16:.2f} s extrapolated to {len(ids)} tokens"}
        emit(res)
        # A long recording through the string-level Transcriber: 16 chunks of 30 s (8 minutes), 448 tokens per chunk,
        # chunk by chunk vs eight chunks per decoder launch (the default without a per-token callback).
        tr = kjarni_amd.Transcriber(model_path=d, max_tokens=n_tok)
        long_audio = np.concatenate([synth.synthetic_audio(30.0, seed=20 + i) for i in range(16)])
        tr.transcribe_audio(long_audio[: 16000 * 45], 16000)          # warm-up (graphs for 2 lanes do not matter below)
        rows = {}
        for label, lanes in (("sequential", "1"), ("lock_step_8", "8")):
            os.environ["KJARNI_HIP_WHISPER_LANES"] = lanes
            tr.transcribe_audio(long_audio, 16000)
            t0 = time.perf_counter()
            out = tr.transcribe_audio(long_audio, 16000)
            dt = time.perf_counter() - t0
            rows[label] = {"seconds": round(dt, 3), "x_real_time": round(480.0 / dt, 1), "chars": len(out.text)}
        os.environ.pop("KJARNI_HIP_WHISPER_LANES", None)
        emit({"metric": "x real time, Whisper-base shaped transcribe of 8 min audio (16 chunks x 448 tokens) through kjarni_transcriber_*",
              "value": rows["lock_step_8"]["x_real_time"], "unit": "x real time", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "whisper-base shape, random init, 480 s synthetic audio, 449 generated tokens per chunk"},
              "runs": rows, "same_text": rows["sequential"]["chars"] == rows["lock_step_8"]["chars"]})

    if "llm" in which:
        d = os.path.join(tmp, "llama-1b")
        cfg_l, t_l = synth.llm_model(d, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=4096, eos_token_id=[])
        dec = kjarni_amd.HipDecoder(d, max_context=2048)
        prompt = np.random.default_rng(0).integers(1000, 100000, 128).tolist()
        n_new = 256
        dec.generate(prompt, 8)                                               # warm-up (graph capture)
        t0 = time.perf_counter()
        dec.reset()
        dec.forward(prompt, fetch=False)
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = dec.generate(prompt, n_new)
        t_total = time.perf_counter() - t0
        t_dec = t_total - t_prefill
        kv_dim = cfg_l["num_key_value_heads"] * 64
        kv_bytes = 2 * cfg_l["num_hidden_layers"] * kv_dim * 4 * (128 + n_new / 2)   # average cache read per step
        per_tok = dec.weight_bytes + kv_bytes
        res = {"metric": "tokens/sec greedy decode, Llama-3.2-1B shape, bf16 weights, batch 1", "value": round(len(out) / t_dec, 1),
               "unit": "tokens/s", "n_gpus": 1, "dtype": "bf16 weights, f32 activations/accumulate/KV", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[4]: Llama-3.2-1B geometry (2048 hidden, 16 layers, 32/8 heads, vocab 128256), "
                                      f"random init, 128-token prompt, {len(out)} generated tokens"},
               "ms_prefill_128": round(t_prefill * 1e3, 2), "prefill_tokens_per_s": round(128 / t_prefill, 1),
               "ms_per_token": round(t_dec * 1e3 / len(out), 4), "weight_bytes": dec.weight_bytes,
               "roofline": {"kernel": "llm_gemv_stream_kernel (weight stream) + decode_attention_partial", "bound": "hbm",
                            "achieved": round(per_tok * len(out) / t_dec / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(per_tok * len(out) / t_dec / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_token": int(per_tok)}}
        if not os.environ.get("KJARNI_BENCH_NO_CPU"):                         # kernel A/B runs skip the CPU leg
            from oracle import llm_oracle as LO
            from oracle import oracle as O
            cores = usable_cores()
            O.lib().ko_set_num_threads(int(cores))
            orc = LO.LlmOracle(t_l, cfg_l)
            cache = orc.new_cache()
            t0 = time.perf_counter()
            h = orc.forward(prompt[:16], cache)
            c_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            for i in range(4):
                h = orc.forward([int(out[i])], cache)
                orc.logits(h[0, -1])
            c_dec = (time.perf_counter() - t0) / 4
            res["cpu_baseline"] = {"value": round(1.0 / c_dec, 2), "unit": "tokens/s", "cores": int(cores), "kind": "port",
                                   "sample": f"oracle: 16-token prefill {c_pre:.2f} s, 4 decode steps at {c_dec * 1e3:.0f} ms each (f32 weights)"}
        emit(res)

    if "llm8b" in which:
        # The same decode path on the Llama-3.1-8B geometry (16 GB of bf16 weights): the per-kernel floor that bounds the 1B
        # shape is amortised over 6.5x more bytes per launch, so this shows what the weight-streaming kernels reach.
        d = os.path.join(tmp, "llama-8b")
        cfg_l = synth.llm_model_streamed(d, synth.LLAMA_8B, seed=0, max_position_embeddings=4096, eos_token_id=[])
        t0 = time.perf_counter()
        dec = kjarni_amd.HipDecoder(d, max_context=2048)
        t_load = time.perf_counter() - t0
        prompt = np.random.default_rng(0).integers(1000, 100000, 512).tolist()
        n_new = 128
        dec.generate(prompt[:32], 8)
        dec.reset()
        dec.forward(prompt, fetch=False)
        t0 = time.perf_counter()
        dec.reset()
        dec.forward(prompt, fetch=False)
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = dec.generate(prompt, n_new)
        t_dec = time.perf_counter() - t0 - t_prefill
        kv_bytes = 2 * cfg_l["num_hidden_layers"] * cfg_l["num_key_value_heads"] * 128 * 4 * (512 + n_new / 2)
        per_tok = dec.weight_bytes - 128256 * 4096 * 2 + kv_bytes   # the embedding table is not streamed (one row is read)
        flops_prefill = 2.0 * 512 * (dec.weight_bytes / 2 - 2 * 128256 * 4096)
        emit({"metric": "tokens/sec greedy decode, Llama-3.1-8B shape, bf16 weights, batch 1", "value": round(len(out) / t_dec, 1),
              "unit": "tokens/s", "n_gpus": 1, "dtype": "bf16 weights, f32 activations/accumulate/KV", "data": "synthetic",
              "config": {"workload": "Llama-3.1-8B geometry (4096 hidden, 32 layers, 32/8 heads of 128, inter 14336, vocab 128256, untied head), "
                                     f"random init, 512-token prompt, {len(out)} generated tokens"},
              "ms_per_token": round(t_dec * 1e3 / len(out), 4), "ms_prefill_512": round(t_prefill * 1e3, 2),
              "prefill_tokens_per_s": round(512 / t_prefill, 1), "prefill_tflops": round(flops_prefill / t_prefill / 1e12, 1),
              "weight_bytes": dec.weight_bytes, "load_seconds": round(t_load, 1),
              "roofline": {"kernel": "llm_gemv_splitk_kernel (weight stream)", "bound": "hbm", "achieved": round(per_tok * len(out) / t_dec / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(per_tok * len(out) / t_dec / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                           "algorithmic_bytes_per_token": int(per_tok)}})

    if "chat" in which:
        # The string-level path of the C ABI (kjarni_chat_send / kjarni_chat_stream) on the same Llama-1B shape: template,
        # BPE encode, prefill, decode with the sampler, per-token text.  The tokenizer is the small Llama-3-style fixture
        # (the model's 128 256-row head is kept, so the sampler sees a full-size vocabulary).
        import shutil
        from kjarni_amd.chat import Chat, GenerationConfig
        d = os.path.join(tmp, "llama-1b-chat")
        synth.llm_model(d, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=4096, bos_token_id=700, eos_token_id=[701])
        shutil.copy(os.path.join(ROOT, "tests", "golden", "bpe_llama3_tokenizer.json"), os.path.join(d, "tokenizer.json"))
        chat = Chat("llama3.2-1b-instruct", model_path=d)
        message = "Summarise the history of Iceland in three paragraphs, please. " * 4
        n_prompt = len(chat.encode(chat.format_prompt(None, message)))
        n_new = 256
        rows = {}
        for label, g in [("greedy", GenerationConfig(do_sample=False, max_new_tokens=n_new)),
                         ("sample_default", GenerationConfig(max_new_tokens=n_new)),
                         ("sample_top_k40_rep1.1", GenerationConfig(max_new_tokens=n_new, top_k=40, repetition_penalty=1.1))]:
            chat.seed(1)
            chat.send(message, GenerationConfig(do_sample=g.do_sample, max_new_tokens=8))   # warm-up
            pieces = []
            first = []
            t0 = time.perf_counter()

            def on_token(text, pieces=pieces, first=first, t0=t0):
                if not pieces:
                    first.append(time.perf_counter() - t0)
                pieces.append(text)
                return True

            chat.stream(message, on_token, g)
            dt = time.perf_counter() - t0
            rows[label] = {"tokens": len(pieces), "ms_to_first_token": round(first[0] * 1e3, 2) if first else None,
                           "tokens_per_s_after_first": round((len(pieces) - 1) / (dt - first[0]), 1) if len(pieces) > 1 else None,
                           "ms_total": round(dt * 1e3, 1)}
        # Random weights give a nearly flat next-token distribution (top-p 0.9 keeps thousands of tokens: the sampler's worst
        # case).  A trained model's is peaked; the same checkpoint with its final norm scaled 8x stands in for that.
        from safetensors.torch import load_file, save_file
        d2 = os.path.join(tmp, "llama-1b-chat-peaked")
        os.makedirs(d2)
        for name in ("config.json", "tokenizer.json"):
            shutil.copy(os.path.join(d, name), os.path.join(d2, name))
        tensors = load_file(os.path.join(d, "model.safetensors"))
        tensors["model.norm.weight"] = tensors["model.norm.weight"] * 8.0
        save_file(tensors, os.path.join(d2, "model.safetensors"))
        del tensors
        chat2 = Chat("llama3.2-1b-instruct", model_path=d2)
        chat2.seed(1)
        chat2.send(message, GenerationConfig(max_new_tokens=8))
        pieces, first = [], []
        t0 = time.perf_counter()

        def on_token2(text):
            if not pieces:
                first.append(time.perf_counter() - t0)
            pieces.append(text)
            return True

        chat2.stream(message, on_token2, GenerationConfig(max_new_tokens=n_new))
        dt = time.perf_counter() - t0
        rows["sample_default_peaked_logits"] = {"tokens": len(pieces), "ms_to_first_token": round(first[0] * 1e3, 2),
                                                "tokens_per_s_after_first": round((len(pieces) - 1) / (dt - first[0]), 1),
                                                "ms_total": round(dt * 1e3, 1)}
        del chat2
        t0 = time.perf_counter()
        for _ in range(20):
            chat.encode(chat.format_prompt(None, message))
        t_enc = (time.perf_counter() - t0) / 20
        emit({"metric": "chat tokens/sec through kjarni_chat_stream, Llama-3.2-1B shape, bf16 weights", "unit": "tokens/s",
              "value": rows["greedy"]["tokens_per_s_after_first"], "n_gpus": 1, "data": "synthetic",
              "config": {"workload": f"Llama-3.2-1B geometry, random init, {n_prompt}-token templated prompt, {n_new} new tokens; "
                                     "greedy = device-resident graph loop, sample = logits to the host + sampler per token"},
              "prompt_tokens": n_prompt, "ms_template_plus_bpe_encode": round(t_enc * 1e3, 3), "runs": rows})


if __name__ == "__main__":
    main()
