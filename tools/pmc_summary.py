"""Summarise rocprofv3 --pmc CSVs under a directory: per kernel, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "render" not in k and "trace" not in k and "shade" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} n={len(v)} mean={sum(v)/len(v):.6g}")
