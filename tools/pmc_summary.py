"""Aggregates rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        short = name.replace("void kjarni::(anonymous namespace)::", "").replace("kjarni::(anonymous namespace)::", "").split("(")[0][:60]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "gemm" not in k and "attention" not in k and "layernorm" not in k and "cosine" not in k:
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
