"""Turns the PMC passes of tools/pmc_bench.sh into profiles/pmc_traffic.json:
per kernel symbol, HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request of a wide coalesced stream, hence the factor 2 on the read side
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM; WRITE_SIZE is exact for 16-B-per-lane stores).
Infinity-Cache hits are included in both counters (they are fabric-side)."""
import collections, csv, glob, json, os, sys

root, out = sys.argv[1], sys.argv[2]
round_label = sys.argv[3] if len(sys.argv) > 3 else "unlabelled passes"   # e.g. "round 6, passes r06c"
EPI = {0: "EPI_BIAS", 1: "EPI_BIAS_GELU", 2: "EPI_BIAS_GELU_NEW", 3: "EPI_BIAS_RELU", 4: "EPI_BIAS_TANH",
       5: "EPI_BIAS_RESIDUAL"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = r["Kernel_Name"].replace("void kjarni::(anonymous namespace)::", "").replace(
            "kjarni::(anonymous namespace)::", "").split("(")[0]
        if name.startswith("gemm_nt_f32_mfma_ln"):
            name = "gemm_nt_f32_mfma_ln"
        elif name.startswith(("gemm_nt_f32_mfma<", "gemm_nt_f32_stream<")):
            epi = int(name.split("<")[1].split(",")[0].rstrip(">"))
            name = f"{name.split('<')[0]}<{EPI.get(epi, epi)}>"
        name = name.split("<")[0] if name.startswith(("attention", "pool")) else name
        if not name.startswith(("gemm_", "attention", "embed_", "pool", "layernorm", "cosine", "mid_", "small_linear")):
            continue  # (torch's fills / reductions around the step, the clock probes)
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        res[k] = {"fetch_size_kib_raw": f, "write_size_kib": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                  "launches_sampled": len(c["FETCH_SIZE"])}
json.dump({"round": round_label, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                     "`python bench.py --steps 1 --warmup 1 --sentences 16384 --no-extras` (chunks of 2 048 sentences = "
                     "262 144 tokens, the launch shape of the full run)", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
