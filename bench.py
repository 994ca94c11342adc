#!/usr/bin/env python3
"""Headline benchmark: sentences/s, minilm-l6-v2 batch encode (seq 128, fp32).

    python bench.py --gpus N --steps K --warmup W [--workload embed|rerank]

One process per GPU.  The driver launches N > 1 through torch.distributed.run; a plain
`python bench.py --gpus N` spawns its N ranks itself (the parent never touches the GPU).

--workload embed (default; BASELINE.json configs[1], weak scaling): a "step" is one pass of
the hot path over this rank's 65 536 sentences x 128 tokens -- ids -> embeddings+LN -> 6
encoder layers -> mean-pool -> L2 (get_hidden_states_batch_from_ids + encode_batch_flat),
inputs resident in HBM -- followed, when N > 1, by the RCCL all-gather of the [65 536, 384]
slabs (kjarni_amd.distributed.sharded_embed), so every rank holds all N*65 536 vectors.

--workload rerank (BASELINE.json configs[2], STRONG scaling): 100 000 pre-tokenised
(query, doc) pairs x 128 tokens in total, balanced row blocks per rank through the
cross-encoder (kjarni_amd.distributed.sharded_rerank_scores), all-gather of the [100 000]
scores, and the host's stable descending sort (cross_encoder/model.rs:251-252) -- all
inside the timed region.

--in-process (with --gpus N, launched plainly): ONE process drives the N devices through the library's own
EncoderGroup (kjarni_hip_group_embed_allgather / _logits_allgather: a host thread + stream per device, ncclAllGather
on a ncclCommInitAll communicator) -- the arrangement a C# / Go caller of the C ABI gets; `config.arrangement` says which
of the two produced `value`.

`value` is measured with inputs and outputs RESIDENT IN HBM (device pointers in, device pointers out; `config.io`);
the PCIe-inclusive rate of the same workload is `value_host_ptrs`.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  steps_ms      wall time of every timed step (HIP events on the launch stream, resolved after the region's final
                synchronisation: no extra sync inside the region)
  roofline.frac_of_peak_at_that_clock / e2e_frac_of_peak_at_that_clock
                the same fractions against the matrix-core peak at the clock the chip held UNDER THE WORK (mean of
                clock_ghz; 157.3 TFLOP/s is the 2.4 GHz figure).  f32 matrix-core kernels run the board at its power cap and
                the clock settles well below 2.4 GHz there; `frac` stays the contract's figure against the fixed peak
  clock_ghz     the shader clock under load, one value per step of a pass of up to five UNTIMED steps right in front of the
                timed region: a one-wave kernel on a stream of its own beside the
                launch stream (kjarni_hip_clock_trace, asleep between its reads) stamps shader cycles against the 100 MHz
                counter over 16 consecutive windows per step (beside the timed steps themselves it cost 1.2 % of `value`:
                profiles/r06r_instrument_ab.log); `clock_ghz_min` / `clock_ghz_max` over all windows;
                `clock_ghz_idle`: the same reading with nothing else running, before the region (what a probe BETWEEN
                two kernels of the launch stream reads: rounds 1-5 quoted that, 2.3-2.4 GHz).
                `gpu_sensors`: board power / temperature / sclk sampled from sysfs during the region
                (+ rocm-smi before and after) -- what tells a slower box from a power-limited long run
  scan / scan_1e7  (N = 1, embed) the other half of the hot path, R14: cosine search (scan + top-10) of 1 and of 64 queries
                over a [1 000 000, 384] and a [10 000 000, 384] corpus resident in HBM, each with its own roofline (hbm /
                mfma) and an oracle check of the TIMED call's own output (scores, order, no miss over a random subset)
  whisper / llm_decode  (N = 1, embed) BASELINE.json configs[3] / configs[4]: Whisper-base shape transcribe of 30 s, Llama-3.2-1B
                shape greedy decode -- each with its own roofline (hbm), oracle check after its clock, cpu_baseline
  scan_sharded  (N > 1, embed) the cosine search over a corpus sharded by rows: local search, one all-gather of the candidate
                lists, merge on the host
  roofline      the dominant kernel, timed live with HIP events on its launch stream over
                the timed region (libkjarni_ffi's profiler)
  rerank        (embed workload) the 100 000-pair STRONG-scaling rerank leg run after the embed region, same timing rules:
                {"pairs_per_s", "ms_per_step", "n_gpus", "scaling": "strong", ...}
  collective    (N > 1) the start-up smoke of the communicator: backend, ranks seen, RCCL version
  max_abs_err_vs_oracle  64 sampled rows of the last timed step's output against the CPU oracle (checked after the
                timed region; the run fails above 1e-4)
  cpu_baseline  (N = 1, embed) the port of the reference's CPU path (oracle/
                kjarni_cpu_baseline.c) timed on this host: calls of 32 and of 256, with the
                reference's serial row loops and with them parallelised
  value_host_ptrs / value_ragged  (N = 1, embed) the same workload through host pointers
                (H2D of ids/mask + D2H of the embeddings inside the timed region) and with
                ragged lengths U{16..128}.
  value_by_call_size  (N = 1, embed) one host-pointer call at a time of 1 / 8 / 32 / 64 / 256 sentences.
  value_f32_on_bf16   (N = 1, embed) the headline workload in the opt-in mode kjarni_hip_set_f32_on_bf16 (default off;
                never part of `value`), with the largest difference of its embeddings from the default path's.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SENTENCES_PER_GPU = 65536
RERANK_PAIRS = 100000
SEQ = 128
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def flops_per_sentence(H=384, L=6, I=1536, S=128):
    """GEMMs only, 2*M*N*K."""
    per_layer = 2 * S * H * 3 * H + 2 * 2 * S * S * H + 2 * S * H * H + 2 * 2 * S * H * I
    return L * per_layer


def host_description():
    """lscpu model, physical cores and SMT state of the box the CPU leg runs on."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        import psutil
        info["physical_cores"] = psutil.cpu_count(logical=False)
    except Exception:
        pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                info["model"] = v
            elif k == "Thread(s) per core":
                info["threads_per_core"] = int(v)
            elif k == "Socket(s)":
                info["sockets"] = int(v)
    except Exception:
        pass
    try:
        with open("/sys/devices/system/cpu/smt/active") as f:
            info["smt_active"] = f.read().strip() == "1"
    except Exception:
        pass
    # What this process may actually use: the affinity mask and the cgroup CPU quota (a container on a 128-core
    # host is typically given a fraction of it; threads beyond the quota only fight each other for time slices).
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info["affinity_cpus"] = usable
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            info["cgroup_cpu_quota"] = round(int(quota) / int(period), 2)
            usable = min(usable, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    info["usable_cpus"] = usable
    return info


def cpus_granted():
    """CPUs this process may use: the affinity mask, cut to the cgroup's CPU quota when there is one."""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            usable = min(usable, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, usable)


def cpu_baseline(cfg, tensors, budget_s=60.0, cap_sentences=4096):
    """The reference's CPU path as oracle/kjarni_cpu_baseline.c ports it (fused QKV, 64-row / 4x3 AVX2 GEMM
    blocks, per-(b,h) attention GEMMs, persistent buffers), one thread per physical core (the reference pins its
    rayon pool that way, kjarni-ffi/src/lib.rs:37-40) -- of the cores this process is allowed to use: the GPU
    box runs under a cgroup CPU quota, and 128 threads on a 16-CPU quota measured 10x slower than 16.  Four figures: calls of 32 (the
    Indexer default, kjarni-ffi/src/indexer.rs:132) and of 256 sentences, with the reference's serial row loops
    (softmax, LayerNorm, mask, residual) and with those loops parallelised.  Each figure: one warm-up call, then
    the median of 3 passes over up to `cap_sentences` sentences, passes bounded so the whole leg stays near
    `budget_s`."""
    import numpy as np
    from oracle import cpu_baseline as CB
    from tests import synth
    host = host_description()
    cores = int(min(host.get("physical_cores") or os.cpu_count() or 1, host["usable_cpus"]))
    CB.lib().kb_set_num_threads(cores)
    model = CB.BaselineModel(tensors, cfg, max_batch=256, max_seq=SEQ)
    ids, mask = synth.synthetic_ids(cap_sentences, SEQ, seed=0)
    per_pass = budget_s / (4 * 3.5)
    variants = []
    for B in (32, 256):
        for par in (False, True):
            t0 = time.perf_counter()
            model.embed_batch(ids[:B], mask[:B], par)  # warm-up: pages in the weights, spins up the threads
            est = max(time.perf_counter() - t0, 1e-4) / B
            n = int(min(cap_sentences, max(B, (per_pass / est) // B * B)))
            rates = []
            for _ in range(3):
                t0 = time.perf_counter()
                for s in range(0, n, B):
                    model.embed_batch(ids[s:s + B], mask[s:s + B], par)
                rates.append(n / (time.perf_counter() - t0))
            variants.append({"call_size": B, "row_loops": "parallel" if par else "serial (as the reference)",
                             "sentences_per_pass": n, "value": round(float(np.median(rates)), 2),
                             "passes": [round(r, 2) for r in rates]})
    head = variants[0]
    return {"value": head["value"], "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": f"calls of {head['call_size']} sentences x {SEQ} tokens, {head['sentences_per_pass']} sentences "
                      "per pass, median of 3 passes after 1 warm-up; oracle/kjarni_cpu_baseline.c = the reference's "
                      "no-alloc path (fused QKV, 64-row / 4x3 AVX2 GEMM blocks, per-(b,h) attention GEMMs, serial "
                      "softmax / LayerNorm loops)",
            "variants": variants, "host": host}


class GpuSensors:
    """Board power / temperature / sclk of one GPU, sampled from sysfs (hwmon) by a host thread every `period` seconds while
    a timed region runs (file reads only: nothing is enqueued, nothing synchronises), plus `rocm-smi` snapshots taken by
    snapshot() OUTSIDE the region.  Every source is optional: what the box does not expose is left out."""

    def __init__(self, pci_bus_id=None, period=0.5):
        """pci_bus_id: "0000:bb:dd.f" of the device (a box shows every GPU of the host under /sys/class/drm, whatever the
        container may use); None = the first AMD card."""
        import glob
        import threading
        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        self.files, self.card = {}, None
        amd = []
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(c, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            if pci_bus_id is None or os.path.basename(os.path.realpath(c)).lower() == pci_bus_id.lower():
                amd.append(c)
        if amd:
            self.card = os.path.basename(os.path.realpath(amd[0]))
            for hw in glob.glob(os.path.join(amd[0], "hwmon", "hwmon*")):
                for key, names in (("power_w", ("power1_average", "power1_input")), ("temp_c", ("temp2_input", "temp1_input")),
                                   ("sclk_mhz", ("freq1_input",)), ("power_cap_w", ("power1_cap",))):
                    for nm in names:
                        f = os.path.join(hw, nm)
                        if key not in self.files and os.path.exists(f):
                            self.files[key] = f
        self._scale = {"power_w": 1e-6, "temp_c": 1e-3, "sclk_mhz": 1e-6, "power_cap_w": 1e-6}

    def read(self):
        out = {}
        for k, f in self.files.items():
            try:
                out[k] = round(int(open(f).read().strip()) * self._scale[k], 1)
            except (OSError, ValueError):
                pass
        return out

    def start(self):
        import threading
        if not self.files:
            return
        t0 = time.perf_counter()

        def run():
            while not self._stop.is_set():
                r = self.read()
                if r:
                    r["t_s"] = round(time.perf_counter() - t0, 2)
                    self.samples.append(r)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=2)
        out = {"source": f"sysfs hwmon of {self.card}" if self.files else None, "samples": len(self.samples)}
        for k in ("power_w", "temp_c", "sclk_mhz"):
            v = [x[k] for x in self.samples if k in x]
            if v:
                q = max(1, len(v) // 4)
                out[k] = {"first_quarter_mean": round(sum(v[:q]) / q, 1), "last_quarter_mean": round(sum(v[-q:]) / q, 1),
                          "min": min(v), "max": max(v)}
        cap = self.read().get("power_cap_w")
        if cap:
            out["power_cap_w"] = cap
        return out

    @staticmethod
    def snapshot():
        """One `rocm-smi` reading (power, temperature, clocks) as a flat dict; {} when the tool is missing or slow."""
        try:
            p = subprocess.run(["rocm-smi", "--showpower", "--showtemp", "--showclocks", "--json"], capture_output=True, text=True,
                               timeout=20)
            card = next(iter(json.loads(p.stdout).values()))
            keep = {}
            for k, v in card.items():
                kl = k.lower()
                if any(w in kl for w in ("power", "temperature", "sclk", "mclk")):
                    keep[k] = v
            return keep
        except Exception:
            return {}


def scan_leg(torch, np, dev, n_docs, check_rows=50000, dim=384, k=10):
    """R14, the other half of the hot path (kjarni-search/src/vector.rs:131-166, kjarni-rag/src/segment.rs:307-371): cosine
    search = scan + top-k in ONE call (kjarni_hip_cosine_search) of 1 query, of 8 and of 64 queries over a unit-norm Gaussian
    corpus [n_docs, 384] resident in HBM.  One query streams the corpus once: HBM-bound, dim x 4 algorithmic bytes per
    document.  8 / 64 queries: a bf16 filter pass over the same bytes (HBM-bound) + the exact f32 cosines of the few hundred
    documents per query it lets through.  Each entry carries its own roofline; the output of the TIMED call itself is held to the CPU oracle
    after the clock has stopped (every returned score to 1e-6, the reference's order, and no missed document over a random
    subset of `check_rows` corpus rows)."""
    from kjarni_amd import _ffi
    from oracle import oracle as O
    L = _ffi.lib()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(2)
    corpus = torch.randn((n_docs, dim), generator=g, device=dev, dtype=torch.float32)
    corpus /= torch.linalg.vector_norm(corpus, dim=1, keepdim=True)
    out = {"corpus": f"[{n_docs}, {dim}] unit-norm Gaussian rows resident in HBM, k = {k}, Segment semantics",
           "unit": "ms per search call (scan + top-k in one call, device pointers)"}
    for nq in (1, 8, 64):
        q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, k), dtype=torch.float32, device=dev)

        ws = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, n_docs, dim, k), dtype=torch.uint8, device=dev)

        def call():
            _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n_docs, dim, 1, k, ws.data_ptr(),
                                                        idx.data_ptr(), sc.data_ptr(), stream))
        for _ in range(5):
            call()
        reps = 50 if n_docs <= 2_000_000 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg_bytes, flop = n_docs * dim * 4, 2.0 * nq * n_docs * dim
        if nq == 1:
            roof = {"kernel": "cosine_search_stream_kernel", "bound": "hbm", "achieved": round(alg_bytes / ms / 1e6, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg_bytes / ms / 1e6 / PEAK_HBM_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_call": alg_bytes}
        else:
            # the bf16 filter pass streams the corpus once (HBM-bound) and the exact f32 cosines are taken of what it lets through:
            # the call is held to the corpus bytes; beside it the rate in f32 products the call delivers (what round 5's f32
            # matrix-core scan was held to: 2 nq dim flop per document against the f32 MFMA peak)
            roof = {"kernel": "cosine_filter_bf16_kernel (+ sample, exact rescoring, selection: the whole call)", "bound": "hbm",
                    "achieved": round(alg_bytes / ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(alg_bytes / ms / 1e6 / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes_per_call": alg_bytes,
                    "f32_products_tflops": round(flop / ms / 1e9, 2),
                    "f32_products_frac_of_f32_mfma_peak": round(flop / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)}
        # Oracle check of the TIMED call's own output (the last of the `reps` calls over all n_docs rows), after the clock has
        # stopped: (a) the oracle's cosine of every returned document equals the returned score, and the list is in the
        # reference's order (score descending, ties by ascending index); (b) no miss: over a random subset of `check_rows`
        # corpus rows the oracle finds no document that beats the k-th returned score without being in the returned list, and
        # every returned document of the subset is in the subset's own oracle top-k.
        got_i, got_s = idx.cpu().numpy(), sc.cpu().numpy()
        qh = q.cpu().numpy()
        sub = np.sort(np.random.default_rng(5 + nq).choice(n_docs, min(check_rows, n_docs), replace=False))
        host_sub = corpus[torch.from_numpy(sub).to(dev)].cpu().numpy()
        # the CPU side of this row: the oracle's restatement of the reference's search (vector.rs:150-166: one thread, a scalar dot
        # + norm per document, then a sort) over the same subset, one query; a reported baseline, not the target
        cpu_t = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.search(qh[0], host_sub, k, mode=1)
            cpu_t.append(time.perf_counter() - t0)
        cpu_dqs = len(sub) / sorted(cpu_t)[1]
        same, worst, checked = True, 0.0, 0
        for j in range(nq if nq == 1 else 8):   # every 8th query of the 64
            jj = j * (nq // 8) if nq > 1 else 0
            ri, rs = got_i[jj], got_s[jj]
            rows = corpus[torch.from_numpy(ri).to(dev)].cpu().numpy()
            exact = O.cosine_scan(qh[jj], rows, mode=1)                      # (a) the oracle's score of each returned document
            worst = max(worst, float(np.abs(exact - rs).max()))
            in_order = all(rs[t] > rs[t + 1] or (rs[t] == rs[t + 1] and ri[t] < ri[t + 1]) for t in range(k - 1))
            si, ss = O.search(qh[jj], host_sub, k, mode=1)                   # (b) the subset's own top-k
            beat = [int(sub[a]) for a, v in zip(si, ss) if v > rs[-1] + 1e-6]
            missed = [d_ for d_ in beat if d_ not in set(ri.tolist())]
            ret_in_sub = [t for t, d_ in enumerate(ri) if sub[min(np.searchsorted(sub, d_), len(sub) - 1)] == d_]
            not_top = [int(ri[t]) for t in ret_in_sub if exact[t] < ss[-1] - 1e-6]   # (a returned row of the subset is among the subset's best k)
            same = same and in_order and not missed and not not_top
            checked += 1
        del host_sub
        assert same, f"scan leg: the timed call's top-{k} of {nq} quer{'y' if nq == 1 else 'ies'} is not the oracle's"
        assert worst < 1e-6, f"scan leg: scores differ from the oracle by {worst}"
        out[f"queries_{nq}"] = {"ms_per_call": round(ms, 4), "doc_queries_per_s": round(nq * n_docs / ms * 1e3, 0), "roofline": roof,
                               "cpu_baseline": {"value": round(cpu_dqs, 0), "unit": "doc-queries/s", "cores": 1, "kind": "port",
                                                "sample": f"oracle ko_search (the reference's one-thread scan + sort), 1 query over {len(sub)} of the "
                                                          "corpus rows, median of 3"},
                               "speedup_vs_cpu_baseline": round(nq * n_docs / ms * 1e3 / cpu_dqs, 1),
                               "timed_output_equals_oracle": same, "max_abs_score_err_vs_oracle": worst,
                               "queries_checked": checked, "subset_rows_checked_for_misses": int(len(sub))}
        del ws, idx, sc, q
    del corpus
    torch.cuda.empty_cache()
    return out


def whisper_leg(np, tmp):
    """BASELINE.json configs[3]: Whisper-base shape (d 512, 6 + 6 layers, 8 heads, ffn 2048, vocabulary 51 865; random init),
    30 s of synthetic audio through log-mel (audio/mel.rs:60-135) -> conv stem + encoder -> greedy decode of 448 tokens
    (models/whisper/*).  `value` = seconds of audio per second of wall time.  The decode step streams the decoder's weights
    and the cross-attention K / V once per token: HBM-bound, algorithmic bytes per token in `roofline`.  After the clock has
    stopped the encoder output and three decoder steps are held to the oracle (1e-4); `cpu_baseline` = the oracle timed on
    this host (one encoder pass + 16 decoder steps, extrapolated to the same token count)."""
    import kjarni_amd
    from oracle import oracle as O
    from oracle import whisper_oracle as WO
    from tests import synth
    d = os.path.join(tmp, "whisper-base")
    cfg_w, t_w = synth.whisper_model(d, seed=0, base=True)
    wm = kjarni_amd.HipWhisper(d)
    audio = synth.synthetic_audio(30.0, seed=1)
    n_tok, prompt = 448, [50258, 50259, 50359, 50363]
    wm.encode_audio(audio, fetch=False)
    wm.greedy(prompt, False, 8)                                                    # warm-up (graph capture)
    t0 = time.perf_counter()
    for _ in range(5):
        wm.encode_audio(audio, fetch=False)                                        # mel + stem + encoder, synchronised
    t_enc = (time.perf_counter() - t0) / 5
    t_dec, ids = None, None
    for _ in range(3):                                                             # best of three decodes (each 448 dependent steps)
        t0 = time.perf_counter()
        ids = wm.greedy(prompt, False, n_tok)
        dt = time.perf_counter() - t0
        t_dec = dt if t_dec is None else min(t_dec, dt)
    H, L_, I, S_, V = 512, 6, 2048, 1500, 51865
    enc_flops = L_ * (2 * S_ * H * 3 * H + 4 * S_ * S_ * H + 2 * S_ * H * H + 4 * S_ * H * I) + 2 * 3000 * 512 * 240 + 2 * 1500 * 512 * 1536
    # decoder step: 6 layers x (8 HxH + 2 HxI) weights + the vocabulary head + the cross-attention K / V of 1 500 frames, f32
    dec_bytes = 4 * (L_ * (8 * H * H + 2 * H * I) + V * H + L_ * 2 * S_ * H)
    gbs = dec_bytes * len(ids) / t_dec / 1e9
    res = {"workload": "BASELINE.json configs[3]: whisper-base shape (d=512, 6+6 layers, vocab 51865), random init, 30 s synthetic "
                       f"audio, {len(ids)} generated tokens (EOS is never the argmax with random weights), f32",
           "value": round(30.0 / (t_enc + t_dec), 1), "unit": "x real time",
           "ms_mel_stem_encoder": round(t_enc * 1e3, 3), "ms_decode": round(t_dec * 1e3, 2),
           "ms_per_token": round(t_dec * 1e3 / len(ids), 4), "tokens_per_s": round(len(ids) / t_dec, 1),
           "encoder_tflops": round(enc_flops / t_enc / 1e12, 2),
           "roofline": {"kernel": "decoder step (graph-replayed GEMV / attention launches)", "bound": "hbm", "achieved": round(gbs, 1),
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_token": dec_bytes}}
    # oracle check, after the clock: the encoder on the SAME mel, then three decoder steps on the oracle's encoder output
    cores = cpus_granted()
    O.lib().ko_set_num_threads(int(cores))
    orc = WO.WhisperOracle(t_w, cfg_w)
    mel = WO.log_mel(audio)
    t0 = time.perf_counter()
    enc = orc.encode_mel(mel)
    c_enc = time.perf_counter() - t0
    got = wm.encode_mel(mel)
    err_enc = float(np.abs(got - enc[0]).max())
    wm.decode_begin()
    cross = orc.precompute_cross_kv(enc)
    cache = [None] * len(orc.dec_layers)
    err_dec = 0.0
    for step_ids in (prompt, [int(ids[0])], [int(ids[1])]):
        ref_h = orc.decoder_forward(np.asarray([step_ids], np.uint32), enc, cache, cross)[0]
        h, logits = wm.decode_forward(step_ids)
        err_dec = max(err_dec, float(np.abs(h - ref_h).max()), float(np.abs(logits - orc.logits(ref_h[None, -1:, :])[0, 0]).max()))
    res["max_abs_err_vs_oracle"] = {"encoder_output": err_enc, "decoder_hidden_and_logits_3_steps": err_dec, "tolerance": 1e-4}
    assert err_enc < 1e-4 and err_dec < 1e-4, f"whisper leg differs from the oracle: {res['max_abs_err_vs_oracle']}"
    t0 = time.perf_counter()
    orc.decode_chunk_ids(enc, max_tokens=15)
    c_dec16 = time.perf_counter() - t0
    c_total = c_enc + c_dec16 / 16 * len(ids)
    res["cpu_baseline"] = {"value": round(30.0 / c_total, 3), "unit": "x real time", "cores": int(cores), "kind": "port",
                           "sample": f"oracle/whisper_oracle.py: encoder {c_enc:.2f} s, 16 decoder steps {c_dec16:.2f} s extrapolated "
                                     f"to {len(ids)} tokens (log-mel not counted)"}
    return res


def llm_decode_leg(np, tmp):
    """BASELINE.json configs[4]: Llama-3.2-1B geometry (2048 hidden, 16 layers, 32 / 8 heads of 64, inner 8192, vocabulary
    128 256; random init), bf16 weights and the KV cache resident in HBM, batch 1: prefill of a 128-token prompt and greedy
    decode of 256 tokens (decoder/generator.rs:228-383).  `value` = decode tokens/s.  A decode step streams every weight
    once: HBM-bound; `roofline.achieved` = (weight bytes + the average KV-cache read) per token / time per token.  After the
    clock: a 16-token prompt and three single-token steps against oracle/llm_oracle.py on the SAME bf16-rounded weights
    (hidden rows and logits, 1e-4 relative to max(1, max |reference|)); `cpu_baseline` = the oracle's decode step timed here."""
    import kjarni_amd
    from oracle import llm_oracle as LO
    from oracle import oracle as O
    from tests import synth
    d = os.path.join(tmp, "llama-1b")
    cfg_l, t_l = synth.llm_model(d, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=4096, eos_token_id=[])
    dec = kjarni_amd.HipDecoder(d, max_context=2048)
    prompt = np.random.default_rng(0).integers(1000, 100000, 128).tolist()
    n_new = 256
    dec.generate(prompt, 8)                                                        # warm-up (graph capture)
    t_prefill = None
    for _ in range(3):
        t0 = time.perf_counter()
        dec.reset()
        dec.forward(prompt, fetch=False)
        dt = time.perf_counter() - t0
        t_prefill = dt if t_prefill is None else min(t_prefill, dt)
    t_total, out = None, None
    for _ in range(3):                                                             # best of three (each 256 dependent steps)
        t0 = time.perf_counter()
        out = dec.generate(prompt, n_new)
        dt = time.perf_counter() - t0
        t_total = dt if t_total is None else min(t_total, dt)
    t_dec = t_total - t_prefill
    kv_dim = cfg_l["num_key_value_heads"] * 64
    kv_bytes = 2 * cfg_l["num_hidden_layers"] * kv_dim * 4 * (128 + n_new / 2)     # average cache read per step (f32 cache)
    per_tok = dec.weight_bytes + kv_bytes
    gbs = per_tok * len(out) / t_dec / 1e9
    res = {"workload": "BASELINE.json configs[4]: Llama-3.2-1B geometry (2048 hidden, 16 layers, 32/8 heads, vocab 128256), random "
                       f"init, bf16 weights, f32 activations / accumulation / KV cache, 128-token prompt, {len(out)} generated tokens",
           "value": round(len(out) / t_dec, 1), "unit": "tokens/s", "ms_per_token": round(t_dec * 1e3 / len(out), 4),
           "ms_prefill_128": round(t_prefill * 1e3, 2), "weight_bytes": dec.weight_bytes,
           "roofline": {"kernel": "decode step (graph-replayed weight-streaming GEMV launches + decode attention)", "bound": "hbm",
                        "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                        "traffic": None, "algorithmic_bytes_per_token": int(per_tok)}}
    cores = cpus_granted()
    O.lib().ko_set_num_threads(int(cores))
    orc = LO.LlmOracle(t_l, cfg_l)
    cache = orc.new_cache()
    dec.reset()
    worst_h = worst_l = 0.0
    c_dec = []
    for step_ids in (prompt[:16], [int(out[0])], [int(out[1])], [int(out[2])]):
        t0 = time.perf_counter()
        ref_h = orc.forward(step_ids, cache)[0]
        ref_l = orc.logits(ref_h[-1])
        if len(step_ids) == 1:
            c_dec.append(time.perf_counter() - t0)
        h, logits = dec.forward(step_ids)
        k = (len(step_ids) - 1) % 8 + 1
        worst_h = max(worst_h, float(np.abs(h[-k:] - ref_h[-k:]).max()) / max(1.0, float(np.abs(ref_h).max())))
        worst_l = max(worst_l, float(np.abs(logits - ref_l).max()) / max(1.0, float(np.abs(ref_l).max())))
    res["max_rel_err_vs_oracle"] = {"hidden": worst_h, "logits": worst_l, "tolerance": 1e-4,
                                    "note": "|gpu - oracle| / max(1, max |oracle|): a 16-token prompt and 3 single-token steps"}
    assert worst_h < 1e-4 and worst_l < 1e-4, f"llm leg differs from the oracle: {res['max_rel_err_vs_oracle']}"
    c = sum(c_dec) / len(c_dec)
    res["cpu_baseline"] = {"value": round(1.0 / c, 2), "unit": "tokens/s", "cores": int(cores), "kind": "port",
                           "sample": f"oracle/llm_oracle.py: {len(c_dec)} single-token steps at {c * 1e3:.0f} ms each (f32 arithmetic "
                                     "on the same bf16-rounded weights)"}
    return res


def sharded_scan_leg(torch, np, dist, D, dev, rank, world, docs_per_gpu, dry, steps=10, dim=384, k=10):
    """SURVEY.md section 8(e), the third sharded path: the corpus sharded by rows over the ranks (docs_per_gpu unit-norm rows
    on EVERY rank: weak scaling, as the embed leg), each rank runs the one-call cosine search over its shard
    (kjarni_hip_cosine_search: scan + local top-k), ONE all-gather of the [queries, k] candidate lists
    (kjarni_amd.distributed.sharded_cosine_topk_batch) and the merge on the host by (score descending, global index
    ascending) -- all inside the timed region, bracketed by barrier + synchronize, max over ranks.  1 and 64 queries."""
    out = {"corpus": f"[{world} x {docs_per_gpu}, {dim}] unit-norm Gaussian rows, {docs_per_gpu} per GPU resident in HBM, k = {k}",
           "scaling": "weak", "n_gpus": world, "unit": "ms per search over the whole sharded corpus (local search + all-gather + merge)"}
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    corpus = torch.randn((docs_per_gpu, dim), generator=g, device=dev, dtype=torch.float32)
    corpus /= torch.linalg.vector_norm(corpus, dim=1, keepdim=True)
    if not dry:
        from kjarni_amd import _ffi
        L = _ffi.lib()
        stream = torch.cuda.current_stream().cuda_stream

    def sync():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for nq in (1, 64):
        qg = torch.Generator(device=dev).manual_seed(7 + nq)                   # the same queries on every rank
        q = torch.randn((nq, dim), generator=qg, device=dev, dtype=torch.float32)
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        if not dry:
            ws = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, docs_per_gpu, dim, k), dtype=torch.uint8, device=dev)
        held = {}

        def step():
            if dry:   # host stub: the same result by plain torch (stable order: score descending, index ascending)
                scores = (q @ corpus.T) / (torch.linalg.vector_norm(q, dim=1, keepdim=True) * torch.linalg.vector_norm(corpus, dim=1)[None, :])
                o = torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :k]
                idx.copy_(o)
                sc.copy_(scores.gather(1, o))
            else:
                _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), docs_per_gpu, dim, 1, k,
                                                            ws.data_ptr(), idx.data_ptr(), sc.data_ptr(), stream))
            held["merged"] = D.sharded_cosine_topk_batch(idx, sc, rank * docs_per_gpu, k)

        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        mi, ms_ = held["merged"]
        assert mi.shape == (nq, k) and bool(((mi >= 0) & (mi < world * docs_per_gpu)).all()), "sharded scan: indices out of range"
        assert bool((ms_[:, :-1] >= ms_[:, 1:]).all()), "sharded scan: merged scores are not descending"
        # every merged hit that lives in this rank's shard is one of this rank's own local hits, with its score
        mine = (mi >= rank * docs_per_gpu) & (mi < (rank + 1) * docs_per_gpu)
        li, ls = idx.cpu(), sc.cpu()
        for j in range(nq):
            for gi, gs in zip(mi[j][mine[j]].tolist(), ms_[j][mine[j]].tolist()):
                at = (li[j] == gi - rank * docs_per_gpu).nonzero()
                assert at.numel() == 1 and float(ls[j][at[0, 0]]) == gs, "sharded scan: a merged hit is not this rank's local hit"
        ms = dt / steps * 1e3
        out[f"queries_{nq}"] = {"ms_per_search": round(ms, 4), "doc_queries_per_s": round(nq * world * docs_per_gpu / ms * 1e3, 0),
                               "steps": steps}
        del idx, sc, q
    del corpus
    if not dry:
        torch.cuda.empty_cache()
    return out


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when a communicator is made; on a pipe that buffer is flushed at process
    exit, i.e. AFTER the JSON line.  Flushing it right after the communicator exists keeps the JSON line the last line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children.  Nothing in this process has
    touched torch.cuda or HIP, and it never execs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a dead rank would leave the others waiting in the collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class HostStubEncoder:
    """--dry-run-cpu only: stands in for HipEncoder so the launcher / sharding / collective / JSON plumbing can be
    exercised on a box without a GPU (gloo).  Its line is marked data = "dry-run" and carries no roofline."""
    hidden_size, num_labels, num_layers = 384, 1, 6

    @staticmethod
    def _view(ptr, rows, cols, ctype, dtype):
        import ctypes as C

        import numpy as np
        return np.ctypeslib.as_array((ctype * (rows * cols)).from_address(ptr)).reshape(rows, cols).view(dtype)

    def embed_dev(self, ids_ptr, mask_ptr, batch, seq, out_ptr, type_ptr=0, pooling=0, normalize=True, fill=0,
                  stream=0):
        import ctypes as C

        import numpy as np
        ids = self._view(ids_ptr, batch, seq, C.c_uint32, np.uint32).astype(np.float64)
        out = self._view(out_ptr, batch, self.hidden_size, C.c_float, np.float32)
        x = np.sin(ids.sum(1, keepdims=True) * 1e-4 + np.arange(self.hidden_size)[None, :])
        out[:] = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    def logits_dev(self, ids_ptr, mask_ptr, type_ptr, batch, seq, out_ptr, fill=0, stream=0):
        import ctypes as C

        import numpy as np
        ids = self._view(ids_ptr, batch, seq, C.c_uint32, np.uint32).astype(np.float64)
        self._view(out_ptr, batch, 1, C.c_float, np.float32)[:, 0] = np.cos(ids.sum(1) * 1e-4)


class HostStubGroup:
    """--dry-run-cpu --in-process: stands in for HipEncoderGroup (host buffers instead of device buffers)."""
    hidden_size, num_labels, transport = 384, 1, "dry-run (host)"

    def __init__(self, n):
        self.size, self.devices, self._enc = n, list(range(n)), HostStubEncoder()

    def shard(self, rows, i):
        from kjarni_amd import distributed as D
        return D.shard_rows(rows, self.size, i)

    def _gather(self, run, batch_total, cols, out_ptrs):
        import ctypes as C

        import numpy as np
        outs = [HostStubEncoder._view(p, batch_total, cols, C.c_float, np.float32) for p in out_ptrs]
        for i in range(self.size):
            start, count = self.shard(batch_total, i)
            if count:
                run(i, count, outs[i][start:start + count].ctypes.data)
            for j in range(self.size):
                if j != i:
                    outs[j][start:start + count] = outs[i][start:start + count]

    def embed_allgather(self, ids_ptrs, mask_ptrs, batch_total, seq, out_ptrs, **kw):
        self._gather(lambda i, c, o: self._enc.embed_dev(ids_ptrs[i], mask_ptrs[i], c, seq, o), batch_total, self.hidden_size, out_ptrs)

    def logits_allgather(self, ids_ptrs, mask_ptrs, type_ptrs, batch_total, seq, out_ptrs, **kw):
        self._gather(lambda i, c, o: self._enc.logits_dev(ids_ptrs[i], mask_ptrs[i], 0, c, seq, o), batch_total, 1, out_ptrs)

    def close(self):
        pass


def main_in_process(args):
    """ONE process, all --gpus devices: the library's own multi-device arrangement (csrc/group.cpp; what a C# / Go caller of
    kjarni_embedder_* / kjarni_reranker_* gets, here on device pointers).  A step = kjarni_hip_group_embed_allgather (every
    replica encodes its row block on its own host thread + stream, then ONE ncclAllGather on the ncclCommInitAll communicator
    leaves all rows in every device's buffer) -- synchronous, so steps are timed with the host clock.  Same workloads, same
    JSON contract as the one-process-per-GPU arrangement; `config.arrangement` names this one."""
    import numpy as np
    import torch

    from kjarni_amd import distributed as D
    from tests import synth
    dry, n, S = args.dry_run_cpu, args.gpus, SEQ
    torch.set_num_threads(max(1, cpus_granted()))
    if not dry:
        import kjarni_amd
        if not torch.cuda.is_available() or torch.cuda.device_count() < n or kjarni_amd.device_count() < n:
            print(f"bench.py --in-process: {n} AMD GPUs needed (there is no CPU fallback for the product path)", file=sys.stderr)
            return 1

    def on(i):
        return torch.device("cpu") if dry else torch.device("cuda", i)

    def to(i, a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(on(i))

    def load(make, seed):
        with tempfile.TemporaryDirectory(prefix="kjarni_bench_grp_") as tmp:
            cfg, tensors = make(tmp, seed=seed, family=args.weights)
            grp = HostStubGroup(n) if dry else kjarni_amd.HipEncoderGroup(tmp, devices=list(range(n)))
        return cfg, tensors, grp

    def sync_all():
        if not dry:
            for i in range(n):
                torch.cuda.synchronize(i)

    def say(msg):
        print(f"bench.py --in-process: {msg}", file=sys.stderr, flush=True)

    def run_leg(rerank, n_total, steps, warmup):
        say(f"{'rerank' if rerank else 'embed'} leg: {n_total} rows over {n} device(s), {warmup} + {steps} steps")
        cfg, tensors, grp = load(synth.minilm_cross_encoder if rerank else synth.minilm_embedder, 1 if rerank else 0)
        cols = 1 if rerank else grp.hidden_size
        blocks, ids_d, mask_d, types_d, outs = [], [], [], [], []
        for i in range(n):
            start, count = grp.shard(n_total, i)
            if rerank:   # every replica builds only its rows of the one fixed pair set
                a, b, c = synth.synthetic_pairs_rows(start, count, S, seed=1)
                types_d.append(to(i, c))
            else:        # weak scaling: replica i's block is the block rank i of the other arrangement encodes
                a, b = synth.synthetic_ids(count, S, seed=i)
            blocks.append((start, count, a, b, types_d[-1] if rerank else None))
            ids_d.append(to(i, a))
            mask_d.append(to(i, b))
            outs.append(torch.empty((n_total, cols), dtype=torch.float32, device=on(i)))
        ptr = lambda ts: [t.data_ptr() for t in ts]  # noqa: E731
        held = {}

        def step():
            if rerank:
                grp.logits_allgather(ptr(ids_d), ptr(mask_d), ptr(types_d), n_total, S, ptr(outs))
                held["order"] = D.rerank_order_arrays(outs[0][:, 0])   # D2H + the host's stable descending sort
            else:
                grp.embed_allgather(ptr(ids_d), ptr(mask_d), n_total, S, ptr(outs))
        for _ in range(warmup):
            step()
        sync_all()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            per_step.append(round((time.perf_counter() - t1) * 1e3, 2))
        sync_all()
        elapsed = time.perf_counter() - t0
        first = outs[0].cpu()
        assert bool(torch.isfinite(first).all()), "outputs are not finite"
        for i in range(1, n):
            assert bool(torch.equal(outs[i].cpu(), first)), f"device {i}'s buffer differs from device 0's after the all-gather"
        if rerank:
            order = held["order"]
            assert bool((order[1][:-1] >= order[1][1:]).all()), "order is not descending"
        parity = {}
        if not dry and not args.no_parity_check:
            from oracle import oracle as O
            orc = O.OracleModel(tensors, cfg, blocked_gemm=True)
            last = n - 1   # rows of the LAST replica: they reached device 0 through the collective
            start, count, a, b, _ = blocks[last]
            sel = np.unique(np.concatenate([np.array([0, count - 1]), np.random.default_rng(5).choice(count, min(32, count), replace=False)]))[:32]
            if rerank:
                c = types_d[last].cpu().numpy().view(np.uint32)
                ref = orc.rerank_scores(*(np.ascontiguousarray(x[sel]) for x in (a, b, c)))
                got = first[start + torch.from_numpy(sel), 0].numpy()
            else:
                ref = orc.embed_batch(np.ascontiguousarray(a[sel]), np.ascontiguousarray(b[sel]))
                got = first[start + torch.from_numpy(sel)].numpy()
            parity = {"max_abs_err_vs_oracle": float(np.abs(got - ref).max()), "rows_checked_vs_oracle": int(len(sel)),
                      "parity_tolerance": 1e-4}
            assert parity["max_abs_err_vs_oracle"] < 1e-4, f"in-process output differs from the oracle: {parity}"
        transport = grp.transport
        flush_c_stdio()
        say(f"leg done in {elapsed:.2f} s, transport {transport}")
        fps = flops_per_sentence(grp.hidden_size, cfg["num_hidden_layers"], cfg["intermediate_size"], S)
        grp.close()
        return elapsed, per_step, parity, transport, fps, blocks[0][1]

    rerank = args.workload == "rerank"
    n_total = args.pairs if rerank else n * args.sentences
    elapsed, per_step, parity, transport, fps, rows0 = run_leg(rerank, n_total, args.steps, args.warmup)
    value = n_total * args.steps / elapsed
    what = ("pairs/sec minilm-l6-v2-cross-encoder rerank (seq=128)", "pairs/s") if rerank else \
        ("sentences/sec minilm-l6-v2 batch encode (seq=128)", "sentences/s")
    result = {"metric": what[0], "value": round(value, 1), "unit": what[1], "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
              "scaling": "strong" if rerank else "weak", "vs_baseline": None, "dtype": "f32",
              "data": "dry-run" if dry else "synthetic",
              "config": {"workload": (f"minilm-l6-v2-cross-encoder Reranker over {n_total} synthetic query-doc pairs x {S} tokens in total "
                                      "(BASELINE.json configs[2])" if rerank else
                                      f"minilm-l6-v2 Embedder: {args.sentences} synthetic sentences per GPU, seq_len=128, fp32 "
                                      "(BASELINE.json configs[1])") + "; random weights of that architecture",
                         "rows_per_step": n_total, "rows_per_gpu": rows0, "seq_len": S,
                         "sharding": f"rows x{n}, all-gather of the outputs into every device's buffer",
                         "weights": f"tests/synth.py family '{args.weights}', random (no checkpoints offline)",
                         "io": "ids / mask and the output vectors resident in HBM (device pointers, one block per device)",
                         "arrangement": f"in-process group: one process, a host thread + stream per device (kjarni_hip_group_*_allgather), "
                                        f"transport {transport}"},
              "steps_ms": per_step}
    if not dry:
        result["e2e_frac_fp32_mfma_peak"] = round(value * fps / 1e12 / (PEAK_FP32_MFMA_TFLOPS * n), 4)
    result.update(parity)
    if not rerank and not args.no_rerank_leg:
        r_steps = args.rerank_steps or max(1, min(args.steps, 5))
        r_el, r_ms, r_par, _, r_fps, r_rows0 = run_leg(True, args.pairs, r_steps, 1 if args.warmup else 0)
        leg = {"pairs_per_s": round(args.pairs * r_steps / r_el, 1), "ms_per_step": round(r_el / r_steps * 1e3, 3), "n_gpus": n,
               "scaling": "strong", "steps": r_steps, "pairs_per_step": args.pairs, "pairs_per_gpu": r_rows0, "unit": "pairs/s",
               "steps_ms": r_ms}
        leg.update(r_par)
        result["rerank"] = leg
    flush_c_stdio()
    print(json.dumps(result), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # the driver's form (--steps 20 --warmup 5): builder and driver
    ap.add_argument("--warmup", type=int, default=5)    # numbers are the same experiment
    ap.add_argument("--workload", choices=("embed", "rerank"), default="embed")
    ap.add_argument("--sentences", type=int, default=SENTENCES_PER_GPU, help="embed: sentences per GPU per step")
    ap.add_argument("--pairs", type=int, default=RERANK_PAIRS, help="rerank: pairs per step over ALL GPUs")
    ap.add_argument("--chunk-tokens", type=int, default=0, help="override the encoder's chunk size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-pointer and ragged legs")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle check of sampled output rows")
    ap.add_argument("--no-rerank-leg", action="store_true",
                    help="embed workload: skip the 100 000-pair strong-scaling rerank leg that follows the embed region")
    ap.add_argument("--rerank-steps", type=int, default=0, help="timed steps of the rerank leg (default min(steps, 5))")
    ap.add_argument("--no-scan", action="store_true", help="embed workload: skip the cosine-search (R14) leg")
    ap.add_argument("--scan-docs", type=int, default=1_000_000, help="corpus rows of the scan leg (10 000 000 on request)")
    ap.add_argument("--no-scan-1e7", action="store_true", help="embed workload: skip the second scan leg over 10 000 000 rows")
    ap.add_argument("--no-models", action="store_true",
                    help="embed workload: skip the Whisper-base (configs[3]) and Llama-1B decode (configs[4]) legs")
    ap.add_argument("--no-sensors", action="store_true", help="do not sample sysfs / rocm-smi around the timed region")
    ap.add_argument("--no-instrument", action="store_true", help="no per-step events, clock probes or sensors at all")
    ap.add_argument("--no-clock-trace", action="store_true", help="no clock-trace kernel beside the timed steps (events and sensors stay)")
    ap.add_argument("--weights", choices=("trained", "init"), default="trained",
                    help="tests/synth.py weight family: trained-checkpoint statistics (default) or N(0, 0.02) initialisation")
    ap.add_argument("--in-process", action="store_true",
                    help="one process drives all --gpus devices through the library's EncoderGroup (kjarni_hip_group_*_allgather)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="plumbing self-test without a GPU: gloo + a host stub instead of the HIP encoder")
    args = ap.parse_args()

    if args.in_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            print("bench.py: --in-process is ONE process for all devices; do not launch it through torch.distributed.run",
                  file=sys.stderr)
            sys.exit(2)
        sys.exit(main_in_process(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch  # imported before libkjarni_ffi.so so both share torch's HIP runtime
    import torch.distributed as dist

    # Host threads: torch's intra-op pool defaults to every logical CPU of the box (256 on the GPU host) while the job's cgroup
    # may grant far fewer (16 there), and N ranks share that grant: each rank takes its share for the host-side sort / copies.
    host_cpus = cpus_granted()
    torch.set_num_threads(max(1, host_cpus // max(1, world)))

    from kjarni_amd import distributed as D
    from tests import synth

    dry = args.dry_run_cpu
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        import kjarni_amd
        if not torch.cuda.is_available() or kjarni_amd.device_count() < 1:
            print("bench.py needs an AMD GPU (there is no CPU fallback for the product path)", file=sys.stderr)
            sys.exit(1)
        if torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible", file=sys.stderr)
            sys.exit(1)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
    comm_info = None
    if world > 1:
        # Start-up smoke of the collective layer: the communicator must span exactly the ranks --gpus asked for, and a
        # sum of ones over it must come back as that count (on the GPU path this is an RCCL all-reduce over xGMI).
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world, f"all-reduce over {world} ranks returned {ones.item()}"
        version = None
        if not dry:
            try:
                version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        comm_info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "rccl_version": version,
                     "allreduce_of_ones": int(ones.item())}
        flush_c_stdio()
        if rank == 0:
            print(f"bench.py: {comm_info['backend']} communicator over {comm_info['ranks']} ranks"
                  f" (RCCL {version}); all-reduce of ones = {comm_info['allreduce_of_ones']}", file=sys.stderr, flush=True)

    S = SEQ
    rerank = args.workload == "rerank"
    with tempfile.TemporaryDirectory(prefix=f"kjarni_bench_r{rank}_") as tmp:
        # random-init weights of the named architecture (no network for checkpoints)
        # (--weights trained: trained-checkpoint statistics -- peaked softmax, LayerNorm gain outliers, GELU tails -- so the
        # post-clock oracle check below is a check in the regime users run, not in the uniform-softmax regime of N(0, 0.02))
        cfg, tensors = (synth.minilm_cross_encoder if rerank else synth.minilm_embedder)(tmp, seed=1 if rerank else 0,
                                                                                         family=args.weights)
        enc = HostStubEncoder() if dry else kjarni_amd.HipEncoder(tmp, local_rank)
    if args.chunk_tokens and not dry:
        enc.set_chunk_tokens(args.chunk_tokens)
    H = enc.hidden_size

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)

    if rerank:
        n_total = args.pairs
        start, n_local = D.shard_rows(n_total, world, rank)
        # this rank's rows of the one fixed pair set (the same 100 000 pairs at every world size; no rank builds the others' rows)
        ids_np, mask_np, types_np = synth.synthetic_pairs_rows(start, n_local, S, seed=1)
        ids, mask, types = (to_dev(a) for a in (ids_np, mask_np, types_np))
    else:
        n_local = args.sentences
        n_total = world * n_local
        ids_np, mask_np = synth.synthetic_ids(n_local, S, seed=rank)
        ids, mask = to_dev(ids_np), to_dev(mask_np)
    result_holder = {}

    def step():
        if rerank:
            scores = D.sharded_rerank_scores(enc, ids, mask, types, n_total=n_total)
            # D2H + stable descending sort on the host, by the rank that hands the ranking to the caller (rank 0); the gathered
            # scores are on every rank
            result_holder["order"] = D.rerank_order_arrays(scores) if rank == 0 else None
            result_holder["scores"] = scores
        else:
            result_holder["emb"] = D.sharded_embed(enc, ids, mask, n_total=n_total)

    def sync():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # The matrix-core GEMMs hold ~85 % of the time, so the dominant kernel is one of them: only
    # their launches are bracketed by HIP events inside the timed region (events on every one of
    # the launches of a chunk cost ~2.5 % throughput).  The full per-kernel table comes from one
    # extra, untimed step below.
    GEMM_KINDS = ("gemm_qkv", "gemm_out_proj", "gemm_fc1", "gemm_fc2")
    profile = not args.no_profile and not dry   # (begun right in front of the timed region, below)
    # Per-step wall time, without a synchronisation inside the region: an event on the launch stream after every timed step.
    # The shader clock UNDER LOAD is read in a pass of its own, up to five untimed steps of the same workload right in front of the
    # timed region: beside each step, on the library's own non-blocking stream, a one-wave kernel stamps shader cycles against the
    # 100 MHz counter over 16 windows of ~1/16 step (asleep in between).  Round 6 measured that trace kernel BESIDE the timed steps
    # at 1.2 % of `value` (profiles/r06r_instrument_ab.log: 45 663 with it, 46 183 without, 46 223 with no instrumentation at
    # all; the events, the sysfs sampler and the GEMM events of `roofline` cost nothing) -- so it no longer runs there.  (A probe
    # on the launch stream between two steps reads an IDLE chip: the power controller lets the clock back up within microseconds
    # -- that is `clock_ghz_idle`.  And not a torch side stream: the first torch.cuda.Stream() of a process creates torch's pool
    # of 32 streams, with which the library's own streams share hardware queues -- a 64-sentence call, three parts on three
    # streams, went 1.77 -> 2.02 ms.)
    sensors = smi_before = None
    timing_errors = []
    instrument = not dry and rank == 0 and not args.no_instrument
    if instrument:
        from kjarni_amd import ops as _probe_ops
        main_stream = torch.cuda.current_stream()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        TRACE_WINDOWS = 16
        n_trace = 0 if args.no_clock_trace else max(1, min(args.steps, 5))
        probes = torch.zeros((max(1, n_trace), TRACE_WINDOWS, 2), dtype=torch.int64, device=dev)
        idle_probe = torch.zeros((2,), dtype=torch.int64, device=dev)
        side_stream = 0 if args.no_clock_trace else _probe_ops.measurement_stream()
        _probe_ops.clock_probe(idle_probe.data_ptr(), 200, main_stream.cuda_stream)  # (the idle reading: nothing else is running)
    # One more untimed step on EVERY rank (a step holds the output all-gather at N > 1: all ranks or none), timed on the host: the
    # length of a clock-trace window.
    est_us = 0.0
    if not args.no_instrument:  # (--dry-run-cpu included: the gloo tests then cover that every rank makes the same collectives)
        sync()
        t_est = time.perf_counter()
        step()
        sync()
        est_us = (time.perf_counter() - t_est) * 1e6
    if instrument:
        window_us = int(min(10_000_000 // TRACE_WINDOWS - 1, max(10, 0.97 * est_us / TRACE_WINDOWS)))  # (the call rejects > 10 s in all)
        if not args.no_sensors:
            smi_before = GpuSensors.snapshot()
            try:
                pr_ = torch.cuda.get_device_properties(local_rank)
                bus = f"{pr_.pci_domain_id:04x}:{pr_.pci_bus_id:02x}:{pr_.pci_device_id:02x}.0"
            except Exception:
                bus = None
            # (no PCI address -> no sensors: the first AMD card of /sys/class/drm may be another GPU of the host)
            sensors = GpuSensors(bus) if bus else None
    # The clock-trace pass: up to five untimed steps on EVERY rank (a step holds the all-gather at N > 1), rank 0's with the trace
    # kernel beside them.
    trace_steps = 0 if (args.no_instrument or args.no_clock_trace) else max(1, min(args.steps, 5))
    if trace_steps:
        sync()
        for i in range(trace_steps):
            if instrument and side_stream:
                try:
                    _probe_ops.clock_trace(probes[i].data_ptr(), TRACE_WINDOWS, window_us, side_stream)
                except Exception as e:   # (diagnostics only: never at the price of the line)
                    timing_errors.append(repr(e))
                    side_stream = 0
            step()
        sync()   # (torch.cuda.synchronize: the side stream's last trace too)
        if instrument:
            try:
                _probe_ops.measurement_stream_release()  # (or one of the encoder's lane streams may share a hardware queue with it)
            except Exception as e:
                timing_errors.append(repr(e))
    sync()
    if profile:
        enc.profile_begin(GEMM_KINDS)   # (again: the dominant kernel's events are those of the timed region only)
    if sensors:
        sensors.start()
    t0 = time.perf_counter()
    if instrument:
        evs[0].record(main_stream)
    for i in range(args.steps):
        step()
        if instrument:
            evs[i + 1].record(main_stream)
    sync()
    elapsed = time.perf_counter() - t0
    timing = {}
    if instrument:
        try:
            timing["steps_ms"] = [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(args.steps)]
            pr = probes[:max(1, trace_steps)].cpu().numpy().astype(np.float64)
            win = np.where(pr[..., 1] > 0, pr[..., 0] / np.maximum(pr[..., 1], 1) / 10.0, np.nan)  # GHz per window
            if np.isfinite(win).any():
                timing["clock_ghz"] = [round(float(np.nanmean(w)), 3) if np.isfinite(w).any() else None for w in win]
                timing["clock_ghz_min"], timing["clock_ghz_max"] = round(float(np.nanmin(win)), 3), round(float(np.nanmax(win)), 3)
            ip = idle_probe.cpu().numpy().astype(np.float64)
            timing["clock_ghz_idle"] = round(float(ip[0] / ip[1] / 10.0), 3) if ip[1] > 0 else None
            timing["clock_note"] = (f"clock_ghz[i]: shader cycles per 10 ns tick, mean of {TRACE_WINDOWS} windows of {window_us} us read by a one-wave "
                                    f"kernel on a stream of its own while untimed step i of {trace_steps} runs right in front of the timed region "
                                    "(the clock UNDER the work; beside the timed steps themselves the trace kernel cost 1.2 % of `value`); "
                                    "clock_ghz_idle: the same reading over 200 us with nothing else running")
            if sensors:
                timing["gpu_sensors"] = sensors.stop()
                timing["gpu_sensors"]["rocm_smi_before"] = smi_before
                timing["gpu_sensors"]["rocm_smi_after"] = GpuSensors.snapshot()
        except Exception as e:  # (diagnostics only: never at the price of the line)
            timing_errors.append(repr(e))
        if timing_errors:
            timing["instrumentation_error"] = "; ".join(timing_errors)
    stats = enc.profile_end() if profile else []
    all_stats = []
    if profile and rank == 0 and world == 1:
        enc.profile_begin()
        step()
        all_stats = enc.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on what the step produced
    if rerank:
        sc, order = result_holder["scores"], result_holder["order"]
        assert sc.shape == (n_total,) and bool(torch.isfinite(sc).all()), "rerank scores are not finite"
        if rank == 0:
            assert order[0].shape == (n_total,) and bool((order[1][:-1] >= order[1][1:]).all()), "order is not descending"
            assert bool(torch.equal(torch.sort(order[0]).values, torch.arange(n_total))), "order is not a permutation"
    else:
        emb = result_holder["emb"]
        assert emb.shape == (n_total, H)
        norms = torch.linalg.vector_norm(emb[:: max(1, n_total // 4096)], dim=1)
        assert torch.allclose(norms, torch.ones_like(norms), atol=1e-4), "embeddings are not L2-normalised"

    # Parity of what the timed region produced: 64 sampled rows of the LAST timed step's output (this rank's block; the
    # first and last row of the block and of its first 2 048-sentence chunk included) against the CPU oracle on the same
    # ids.  The oracle is the checker here, after the clock has stopped; it never produces anything that is reported
    # as throughput.
    parity = {}
    if rank == 0 and not dry and not args.no_parity_check:
        from oracle import oracle as O
        orc = O.OracleModel(tensors, cfg, blocked_gemm=True)
        prng = np.random.default_rng(123)
        edge = [0, n_local - 1, min(2047, n_local - 1), min(2048, n_local - 1)]
        sel = np.unique(np.concatenate([np.array(edge), prng.choice(n_local, min(64, n_local), replace=False)]))[:64]
        sel_t = torch.from_numpy(sel).to(dev)
        take = lambda a: np.ascontiguousarray(a[sel_t].cpu().numpy().view(np.uint32))
        if rerank:
            start, _ = D.shard_rows(n_total, world, rank)
            got = result_holder["scores"][start + sel_t].cpu().numpy()
            ref = orc.rerank_scores(take(ids), take(mask), take(types))
        else:
            got = result_holder["emb"][rank * n_local + sel_t].cpu().numpy()
            ref = orc.embed_batch(take(ids), take(mask))
        parity = {"max_abs_err_vs_oracle": float(np.abs(got - ref).max()), "rows_checked_vs_oracle": int(len(sel)),
                  "parity_tolerance": 1e-4}
        assert parity["max_abs_err_vs_oracle"] < 1e-4, f"timed output differs from the oracle: {parity}"

    # The north star's scaling target (>= 6x from 1 to 8 GPUs) is on the sharded RERANK path, and the driver only ever
    # runs the default workload: so the default run carries that leg too -- the same 100 000 pairs in total at every N
    # (strong scaling), balanced row blocks, all-gather of the scores, the host's stable descending sort, timed the
    # same way (barrier + synchronize either side, max over ranks).
    rerank_leg = None
    if not rerank and not args.no_rerank_leg:
        with tempfile.TemporaryDirectory(prefix=f"kjarni_bench_ce_r{rank}_") as tmp:
            ce_cfg, ce_tensors = synth.minilm_cross_encoder(tmp, seed=1, family=args.weights)
            ce = HostStubEncoder() if dry else kjarni_amd.HipEncoder(tmp, local_rank)
        if args.chunk_tokens and not dry:
            ce.set_chunk_tokens(args.chunk_tokens)
        p_total = args.pairs
        p_start, p_local = D.shard_rows(p_total, world, rank)
        pi, pm, pt = synth.synthetic_pairs_rows(p_start, p_local, S, seed=1)  # this rank's rows only
        pi_d, pm_d, pt_d = (to_dev(a) for a in (pi, pm, pt))
        r_steps = args.rerank_steps or max(1, min(args.steps, 5))
        held = {}

        def rerank_step():
            held["scores"] = D.sharded_rerank_scores(ce, pi_d, pm_d, pt_d, n_total=p_total)
            held["order"] = D.rerank_order_arrays(held["scores"]) if rank == 0 else None  # (rank 0 hands the ranking to the caller)

        for _ in range(1 if args.warmup else 0):
            rerank_step()
        sync()
        t1 = time.perf_counter()
        for _ in range(r_steps):
            rerank_step()
        sync()
        r_elapsed = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([r_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r_elapsed = float(t.item())
        sc, order = held["scores"], held["order"]
        assert sc.shape == (p_total,) and bool(torch.isfinite(sc).all()), "rerank scores are not finite"
        assert rank != 0 or bool((order[1][:-1] >= order[1][1:]).all()), "rerank order is not descending"
        rerank_leg = {"pairs_per_s": round(p_total * r_steps / r_elapsed, 1), "ms_per_step": round(r_elapsed / r_steps * 1e3, 3),
                      "n_gpus": world, "scaling": "strong", "steps": r_steps, "pairs_per_step": p_total,
                      "pairs_per_gpu": p_local, "unit": "pairs/s",
                      "workload": f"minilm-l6-v2-cross-encoder Reranker over {p_total} synthetic query-doc pairs x {S} tokens in "
                                  f"total, row blocks over {world} GPU(s), all-gather of the scores + rank 0's stable "
                                  "descending sort on the host inside the timed region (BASELINE.json configs[2])"}
        if rank == 0 and not dry and not args.no_parity_check:
            from oracle import oracle as O
            prng = np.random.default_rng(321)
            sel = np.unique(np.concatenate([np.array([0, p_local - 1]), prng.choice(p_local, min(32, p_local), replace=False)]))[:32]
            ref = O.OracleModel(ce_tensors, ce_cfg, blocked_gemm=True).rerank_scores(
                *(np.ascontiguousarray(a[sel]) for a in (pi, pm, pt)))
            got = sc[torch.from_numpy(p_start + sel).to(sc.device)].cpu().numpy()
            rerank_leg["max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
            rerank_leg["rows_checked_vs_oracle"] = int(len(sel))
            assert rerank_leg["max_abs_err_vs_oracle"] < 1e-4, f"rerank scores differ from the oracle: {rerank_leg}"
        if not dry:
            fr = flops_per_sentence(H, ce.num_layers, ce_cfg["intermediate_size"], S) + 2 * H * H + 2 * H
            rerank_leg["e2e_frac_fp32_mfma_peak"] = round(rerank_leg["pairs_per_s"] * fr / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4)
            ce.close()
        del pi, pm, pt, pi_d, pm_d, pt_d, held

    # The third sharded path of SURVEY.md section 8(e): the cosine scan over a corpus sharded by rows (N > 1 only: at N = 1 the
    # `scan` legs below time the same call with the oracle check).
    scan_sharded = None
    if world > 1 and not rerank and not args.no_scan:
        scan_sharded = sharded_scan_leg(torch, np, dist, D, dev, rank, world, min(args.scan_docs, 2000) if dry else args.scan_docs, dry,
                                        steps=max(1, min(args.steps, 10)))

    extras = {}
    if rank == 0 and world == 1 and not rerank and not dry and not args.no_extras:
        # PCIe-inclusive rate: the same 65 536 x 128 workload through host pointers (kjarni_hip_encoder_embed_host:
        # H2D of ids/mask, D2H of the [N,384] embeddings inside the timed region); and ragged lengths U{16..128}.
        enc.embed(ids_np[:1024], mask_np[:1024])
        t1 = time.perf_counter()
        for _ in range(2):
            enc.embed(ids_np, mask_np)
        extras["value_host_ptrs"] = round(2 * n_local / (time.perf_counter() - t1), 1)
        rid, rmask = synth.synthetic_ids(n_local, S, seed=7, ragged=True)
        rid_d, rmask_d = to_dev(rid), to_dev(rmask)
        out = torch.empty((n_local, H), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        # (packed rows on DEVICE pointers are opt-in, kjarni_hip_encoder_set_packing(enc, 2): the call then synchronises its stream once)
        enc.set_packing(2)
        enc.embed_dev(rid_d.data_ptr(), rmask_d.data_ptr(), 1024, S, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            enc.embed_dev(rid_d.data_ptr(), rmask_d.data_ptr(), n_local, S, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        extras["value_ragged"] = round(2 * n_local / (time.perf_counter() - t1), 1)
        extras["ragged_kept_token_fraction"] = round(float(rmask.sum()) / rmask.size, 4)
        enc.set_packing(0)   # the same ragged batch on the padded layout (every [PAD] row computed, as the reference does)
        enc.embed_dev(rid_d.data_ptr(), rmask_d.data_ptr(), 1024, S, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc.embed_dev(rid_d.data_ptr(), rmask_d.data_ptr(), n_local, S, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        extras["value_ragged_padded_layout"] = round(n_local / (time.perf_counter() - t1), 1)
        enc.set_packing(1)
        # the sizes callers make: one call at a time of 1 / 8 / 32 / 64 / 256 sentences through host pointers (the reference's default
        # batch is 32, crates/kjarni-ffi/src/embedder.rs); 1 and 32 take the few-rows / mid-size GEMM routes
        by_call = {}
        for b in (1, 8, 32, 64, 256):
            reps = max(8, 2048 // b)
            for _ in range(reps):   # (untimed: after a pause the first ~40 ms of calls run below the steady clock -- 64 x 128 tokens
                enc.embed(ids_np[:b], mask_np[:b])   # 2.2 -> 1.75 ms over twenty calls, tools/seq_probe.py)
            passes = []   # (the median of three passes: a pass is 30-60 ms, and a busy host core shows in a single one)
            for _ in range(3):
                t1 = time.perf_counter()
                for _ in range(reps):
                    enc.embed(ids_np[:b], mask_np[:b])
                passes.append((time.perf_counter() - t1) / reps)
            dtc = sorted(passes)[1]
            by_call[str(b)] = {"ms_per_call": round(dtc * 1e3, 4), "sentences_per_s": round(b / dtc, 1),
                               "passes_ms": [round(v * 1e3, 4) for v in passes]}
        extras["value_by_call_size"] = by_call
        # Opt-in mode (kjarni_hip_set_f32_on_bf16, default OFF and never part of `value`): the same workload with the
        # large-batch projections' f32 products computed on the bf16 matrix cores from three exact bf16 pieces per operand
        # (six cross products, f32 accumulation: DESIGN.md section 3); its output against the default path's, all rows.
        from kjarni_amd import ops as _ops
        ref_emb = result_holder["emb"]
        before = _ops.set_f32_on_bf16(True)
        try:
            enc.embed_dev(ids.data_ptr(), mask.data_ptr(), 2048, S, out.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                enc.embed_dev(ids.data_ptr(), mask.data_ptr(), n_local, S, out.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            extras["value_f32_on_bf16"] = round(2 * n_local / (time.perf_counter() - t1), 1)
            extras["f32_on_bf16_max_abs_diff_vs_default"] = float((out - ref_emb).abs().max().item())
            enc.embed(ids_np[:32], mask_np[:32])   # the reference's default call size in the mode (host pointers, as value_by_call_size)
            t1 = time.perf_counter()
            for _ in range(64):
                enc.embed(ids_np[:32], mask_np[:32])
            extras["f32_on_bf16_ms_per_call_of_32"] = round((time.perf_counter() - t1) / 64 * 1e3, 4)
        finally:
            _ops.set_f32_on_bf16(before)
        extras["extras_note"] = ("value_host_ptrs: ids/mask handed over as host buffers, embeddings returned to the "
                                 "host (PCIe inclusive); value_ragged: lengths U{16..128} right-padded to 128, run over the kept tokens only "
                                 "(packed rows); value_ragged_padded_layout: the same batch with every [PAD] row computed; 2 steps each; "
                                 "value_by_call_size: one host-pointer call at a time of 1 / 8 / 32 / 64 / 256 sentences x 128 tokens; "
                                 "value_f32_on_bf16: the headline workload in the opt-in mode kjarni_hip_set_f32_on_bf16 (f32 products "
                                 "from three exact bf16 pieces per operand on the bf16 matrix cores; off by default, not part of `value`)")

    scan = None
    if rank == 0 and world == 1 and not rerank and not dry and not args.no_scan:
        scan = {}
        for key, n_docs in (("scan", args.scan_docs), ("scan_1e7", 10_000_000)):   # (10^7 x 384 f32 = 15.4 GB of the 288)
            if key == "scan_1e7" and (args.no_scan_1e7 or args.scan_docs >= 10_000_000):
                continue
            try:
                scan[key] = scan_leg(torch, np, dev, n_docs)
            except Exception as e:  # (an auxiliary leg must not take the headline line down with it; the failure is in the line)
                print(f"bench.py: {key} leg failed: {e!r}", file=sys.stderr, flush=True)
                scan[key] = {"error": repr(e)}

    # BASELINE.json configs[3] and configs[4]: one compact object each, with its own roofline
    # (HBM: a decode step streams the weights once per token), oracle check after its clock has stopped, and CPU baseline.
    models = {}
    if rank == 0 and world == 1 and not rerank and not dry and not args.no_models:
        enc.close()          # (the encoder's 4 GB of workspaces are not needed any more; `enc` attributes read below are host-side)
        for name, leg in (("whisper", whisper_leg), ("llm_decode", llm_decode_leg)):
            try:
                with tempfile.TemporaryDirectory(prefix=f"kjarni_bench_{name}_") as tmp:
                    models[name] = leg(np, tmp)
            except Exception as e:  # (an auxiliary leg must not take the headline line down with it; the failure is in the line)
                print(f"bench.py: {name} leg failed: {e!r}", file=sys.stderr, flush=True)
                models[name] = {"error": repr(e)}

    if rank == 0:
        total = n_total * args.steps
        value = total / elapsed
        fps = flops_per_sentence(H, enc.num_layers, cfg["intermediate_size"], S)
        if rerank:
            fps += 2 * H * H + 2 * H  # pooler + classifier
            metric, unit = "pairs/sec minilm-l6-v2-cross-encoder rerank (seq=128)", "pairs/s"
            workload = (f"minilm-l6-v2-cross-encoder Reranker over {n_total} synthetic query-doc pairs, seq_len=128, fp32, "
                        f"batch-sharded across {world} GPU(s) with an all-gather of the scores and the host's stable "
                        "descending sort inside the timed region (BASELINE.json configs[2]); random-init weights of that "
                        "architecture")
            shard = f"rows x{world}" + (" + RCCL all-gather of [N] scores" if world > 1 else "")
        else:
            metric, unit = "sentences/sec minilm-l6-v2 batch encode (seq=128)", "sentences/s"
            workload = ("minilm-l6-v2 Embedder: 65 536 synthetic sentences per GPU, seq_len=128, fp32 "
                        "(BASELINE.json configs[1]); random weights of that architecture")
            shard = f"rows x{world}" + (" + RCCL all-gather of [N,384] outputs" if world > 1 else "")
        result = {
            "metric": metric,
            "value": round(value, 1),
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if rerank else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "dry-run" if dry else "synthetic",
            "config": {"workload": workload, "rows_per_step": n_total, "rows_per_gpu": n_local, "seq_len": S,
                       "sharding": shard, "weights": f"tests/synth.py family '{args.weights}', random (no checkpoints offline)",
                       "io": "ids / mask and the output vectors resident in HBM (device pointers); the PCIe-inclusive rate of the "
                             "same workload is value_host_ptrs",
                       "arrangement": "one process per GPU (kjarni_amd.distributed over torch.distributed; RCCL when N > 1)"},
        }
        if not dry:
            result["e2e_tflops"] = round(value * fps / 1e12, 2)
            result["e2e_frac_fp32_mfma_peak"] = round(value * fps / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4)
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
            traffic, traffic_source = None, "not measured (no committed PMC pass names this kernel)"
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                traffic = round(pmc["kernels"][sym]["hbm_bytes_per_launch"])
                traffic_source = ("replayed from the committed file profiles/pmc_traffic.json (" + pmc.get("round", "an earlier run")
                                  + "), NOT measured in this run: " + pmc.get("source", ""))
            except Exception:
                pass
            result["roofline"] = {
                "kernel": sym, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "flops_per_launch": d["flops"] / d["launches"],
            }
            # The same fraction against the peak AT THE CLOCK THE CHIP HELD UNDER THIS WORK (the matrix-core peak scales with the
            # shader clock; 157.3 TFLOP/s is the 2.4 GHz figure): the board runs these kernels at its power cap and the clock it
            # settles at there is what the kernels' rate is a fraction of -- `frac` stays the contract's figure against the fixed peak.
            ghz = [g for g in (timing.get("clock_ghz") or []) if g]
            if ghz:
                mean_ghz = sum(ghz) / len(ghz)
                result["roofline"]["clock_ghz_mean"] = round(mean_ghz, 3)
                result["roofline"]["frac_of_peak_at_that_clock"] = round(achieved / (PEAK_FP32_MFMA_TFLOPS * mean_ghz / 2.4), 4)
                result["e2e_frac_of_peak_at_that_clock"] = round(result["e2e_tflops"] / (PEAK_FP32_MFMA_TFLOPS * world * mean_ghz / 2.4), 4)
            if all_stats:  # one extra untimed step with every launch bracketed
                result["kernels_one_step"] = {
                    s["kind"]: {"ms": round(s["total_ms"], 2), "launches": s["launches"],
                                "tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 2) if s["flops"] else None,
                                "gbs": round(s["bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1)}
                    for s in all_stats if s["launches"]}
        result.update(timing)
        result.update(parity)
        if rerank_leg:
            result["rerank"] = rerank_leg
        if scan:
            result.update(scan)
        if scan_sharded:
            result["scan_sharded"] = scan_sharded
        result.update(models)
        if comm_info:
            result["collective"] = comm_info
        result["host_threads_per_rank"] = torch.get_num_threads()  # (the rank's share of the CPUs the cgroup grants)
        result.update(extras)
        if world == 1 and not rerank and not dry and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, tensors)
            result["speedup_vs_cpu_baseline"] = round(value / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
