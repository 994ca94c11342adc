"""Aggregates rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        short = name.replace("void kjarni::(anonymous namespace)::", "").replace("kjarni::(anonymous namespace)::", "").split("(")[0][:60]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in sorted(agg):
    if not any(t in k for t in ("gemm", "attention", "layernorm", "cosine", "pool", "embed", "topk")):
        continue
    d = dur[k]
    print(f"{k}   dispatches={len(d)} mean_us={sum(d)/len(d):.1f}")
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    for n in sorted(c):
        print(f"    {n:28s} mean={c[n]:.4g}")
    if "GRBM_GUI_ACTIVE" in c and d:
        us = sum(d) / len(d)
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9
        print(f"    -> effective clock {clk:.2f} GHz (GRBM_GUI_ACTIVE/8/duration)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"    -> MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.1f}% of SIMD-cycles")
    if "FETCH_SIZE" in c:
        print(f"    -> FETCH_SIZE {c['FETCH_SIZE']/1024:.1f} MB raw (x2 for 16-byte coalesced streams on gfx950), WRITE_SIZE {c.get('WRITE_SIZE',0)/1024:.1f} MB")
