"""Copies the newest gpurun_out results into profiles/ (round-1 names) and rebuilds profiles/traffic.json."""
import collections, csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
newest = lambda pat: max(glob.glob(os.path.join(R, pat), recursive=True), key=os.path.getmtime)
shutil.copy(newest("gpurun_out/prof/**/*_kernel_stats.csv"), os.path.join(R, "profiles/r01_front_bunny_kernel_stats.csv"))
shutil.copy(os.path.join(R, "gpurun_out/bench.json"), os.path.join(R, "profiles/r01_front_bench.json"))
os.makedirs(os.path.join(R, "profiles/r01_pmc_front"), exist_ok=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    src = newest(f"gpurun_out/pmc_r01front/{f}/**/*counter_collection.csv")
    shutil.copy(src, os.path.join(R, f"profiles/r01_pmc_front/{f}_counter_collection.csv"))
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"]
        k = 'shade' if 'front_shade' in k else 'trace' if 'front_trace' in k else 'resolve' if 'resolve' in k else None
        if not k or r["Counter_Name"] != c: continue
        agg[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": calls[k] += 1
frames = 2
out = {k: {"launches_per_frame": calls[k] / frames, "FETCH_SIZE_KB_per_frame": agg[k]["FETCH_SIZE"] / frames,
           "WRITE_SIZE_KB_per_frame": agg[k]["WRITE_SIZE"] / frames} for k in agg}
t = out["trace"]
hb = (2 * t["FETCH_SIZE_KB_per_frame"] + t["WRITE_SIZE_KB_per_frame"]) * 1024 / t["launches_per_frame"]
tp = os.path.join(R, "profiles/traffic.json")
j = json.load(open(tp))
j["bunny_1200x1200_20spp_n1"] = {"kernel": "rtf::front_trace_kernel<false, true>", "hbm_bytes_per_launch": hb, "per_kernel": out,
    "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (profiles/r01_pmc_front), summed over the 11 trace launches of a frame and divided by 11; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads; uncalibrated for 8-16 B gathers, so an upper-side estimate); the shade kernel's traffic (path state and ray queues) is listed beside it"}
json.dump(j, open(tp, "w"), indent=1)
print(json.dumps(out, indent=1)); print("trace hbm bytes/launch", hb)
# the bench line was printed before this traffic figure existed: put the matching figure into the saved copy
bp = os.path.join(R, "profiles/r01_front_bench.json")
b = json.load(open(bp)); b["roofline"]["traffic"] = hb
json.dump(b, open(bp, "w"))
