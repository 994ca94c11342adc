#!/usr/bin/env python3
"""Headline benchmark: sentences/s, minilm-l6-v2 batch encode (seq 128, fp32).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N > 1 through torch.distributed.run).
A "step" is one pass of the hot path over the whole synthetic workload of this
rank: BASELINE.json configs[1] = 65 536 sentences x 128 tokens through
ids -> embeddings+LN -> 6 encoder layers -> mean-pool -> L2 (the token-level
boundary get_hidden_states_batch_from_ids + encode_batch_flat), inputs already
resident in HBM.  With N > 1 every rank encodes its own 65 536-sentence shard
(weak scaling) and the step ends with the RCCL all-gather of the [65 536, 384]
output slabs, so each rank holds all N*65 536 vectors.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel, timed live with HIP events on its launch
                  stream over the timed region (libkjarni_ffi's profiler)
  cpu_baseline -- the CPU restatement of the reference path (oracle/, with the
                  reference's GEMM blocking) timed on this host, N = 1 only.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SENTENCES_PER_GPU = 65536
SEQ = 128
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def flops_per_sentence(H=384, L=6, I=1536, S=128):
    """GEMMs only, 2*M*N*K."""
    per_layer = 2 * S * H * 3 * H + 2 * 2 * S * S * H + 2 * S * H * H + 2 * 2 * S * H * I
    return L * per_layer


def cpu_baseline(cfg, tensors, budget_s=15.0):
    """CPU restatement of the reference path (oracle, blocked AVX2 GEMM as in
    cpu/ops/matmul.rs:571-686), B = 32 sentences per call (the Indexer default,
    kjarni-ffi/src/indexer.rs:132), one thread per physical core."""
    import numpy as np
    from oracle import oracle as O
    from tests import synth
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        cores = os.cpu_count() or 1
    O.lib().ko_set_num_threads(int(cores))
    model = O.OracleModel(tensors, cfg, blocked_gemm=True)
    B = 32
    ids, mask = synth.synthetic_ids(B * 64, SEQ, seed=0)
    model.embed_batch(ids[:B], mask[:B])  # warm-up (page in weights, spin up threads)
    done, t0 = 0, time.perf_counter()
    while done < ids.shape[0]:
        model.embed_batch(ids[done:done + B], mask[done:done + B])
        done += B
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 2), "unit": "sentences/s", "cores": int(cores), "kind": "port",
            "sample": f"{done} sentences x {SEQ} tokens in calls of {B} ({dt:.1f} s), "
                      "oracle/kjarni_oracle.c with the reference's 64-row / 4x3 AVX2 GEMM blocking"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sentences", type=int, default=SENTENCES_PER_GPU, help="sentences per GPU per step")
    ap.add_argument("--chunk-tokens", type=int, default=0, help="override the encoder's chunk size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import numpy as np
    import torch  # imported before libkjarni_ffi.so so both share torch's HIP runtime
    import torch.distributed as dist

    import kjarni_amd
    from tests import synth

    if not torch.cuda.is_available() or kjarni_amd.device_count() < 1:
        print("bench.py needs an AMD GPU (there is no CPU fallback for the product path)", file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    N, S = args.sentences, SEQ
    with tempfile.TemporaryDirectory(prefix=f"kjarni_bench_r{rank}_") as tmp:
        cfg, tensors = synth.minilm_embedder(tmp, seed=0)  # random-init MiniLM-L6-v2 shaped weights
        enc = kjarni_amd.HipEncoder(tmp, local_rank)
    if args.chunk_tokens:
        enc.set_chunk_tokens(args.chunk_tokens)
    H = enc.hidden_size

    ids_np, mask_np = synth.synthetic_ids(N, S, seed=rank)
    ids = torch.from_numpy(ids_np.view(np.int32)).to(dev)
    mask = torch.from_numpy(mask_np.view(np.int32)).to(dev)
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * N, H), dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        enc.embed_dev(ids.data_ptr(), mask.data_ptr(), N, S, out.data_ptr(), stream=stream)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # The matrix-core GEMMs hold ~85 % of the time, so the dominant kernel is one of them: only
    # their launches are bracketed by HIP events inside the timed region (events on every one of
    # the 44 launches per chunk cost ~2.5 % throughput).  The full per-kernel table comes from one
    # extra, untimed step below.
    GEMM_KINDS = ("gemm_qkv", "gemm_out_proj", "gemm_fc1", "gemm_fc2")
    if not args.no_profile:
        enc.profile_begin(GEMM_KINDS)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    stats = enc.profile_end() if not args.no_profile else []
    all_stats = []
    if not args.no_profile and rank == 0 and world == 1:
        enc.profile_begin()
        step()
        all_stats = enc.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: outputs are unit vectors
    norms = torch.linalg.vector_norm(out[:1024], dim=1)
    assert torch.allclose(norms, torch.ones_like(norms), atol=1e-4), "embeddings are not L2-normalised"

    if rank == 0:
        total = world * N * args.steps
        value = total / elapsed
        fps = flops_per_sentence(H, enc.num_layers, cfg["intermediate_size"], S)
        result = {
            "metric": "sentences/sec minilm-l6-v2 batch encode (seq=128)",
            "value": round(value, 1),
            "unit": "sentences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "minilm-l6-v2 Embedder: 65 536 synthetic sentences per GPU, seq_len=128, fp32 "
                                   "(BASELINE.json configs[1]); random-init weights of that architecture",
                       "sentences_per_gpu": N, "seq_len": S, "sharding": f"rows x{world}" +
                       (" + RCCL all-gather of [N,384] outputs" if world > 1 else "")},
            "e2e_tflops": round(value * fps / 1e12, 2),
            "e2e_frac_fp32_mfma_peak": round(value * fps / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
        }
        if stats:
            by_symbol = {}
            for s in stats:
                if s["launches"] == 0:
                    continue
                b = by_symbol.setdefault(s["symbol"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                b["ms"] += s["total_ms"]
                b["flops"] += s["flops"]
                b["bytes"] += s["bytes"]
                b["launches"] += s["launches"]
            sym, d = max(by_symbol.items(), key=lambda kv: kv[1]["ms"])
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            # HBM-side bytes per launch of that kernel, from the committed PMC passes (rocprofv3
            # cannot run inside this process); null when no measurement is on file.
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    traffic = round(json.load(f)["kernels"][sym]["hbm_bytes_per_launch"])
            except Exception:
                pass
            result["roofline"] = {
                "kernel": sym, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "flops_per_launch": d["flops"] / d["launches"],
            }
            if all_stats:  # one extra untimed step with every launch bracketed
                result["kernels_one_step"] = {
                    s["kind"]: {"ms": round(s["total_ms"], 2), "launches": s["launches"],
                                "tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 2) if s["flops"] else None,
                                "gbs": round(s["bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1)}
                    for s in all_stats if s["launches"]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, tensors)
            result["speedup_vs_cpu_baseline"] = round(value / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
