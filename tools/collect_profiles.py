"""Copy one run of tools/gpu_profiles.sh (gpurun_out/<tag>/) into profiles/ under round-prefixed names and derive
profiles/traffic.json from the PMC passes.  Copies only: bench lines and rocprofv3 tables land byte for byte.

    python tools/collect_profiles.py <tag> <round-prefix>       e.g.  r02b r02
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
SRC, DST = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
FRAMES_PER_PASS = 2          # tools/prof_render.py <workload> <spp> 2


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern))
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


def copy(src_pattern, dst_name):
    shutil.copyfile(one(src_pattern), os.path.join(DST, dst_name))


for name in ("bench_bunny", "bench_bunny_streams1", "bench_soup", "bench_simple_sphere", "bench_multisphere", "bench_suzanne", "prof_bunny",
             "prof_bunny_streams1", "prof_soup"):
    copy(name + ".json", f"{pre}_{name}.json")
for name in ("prof_bunny", "prof_bunny_streams1", "prof_soup"):
    copy(f"{name}/*/*_kernel_stats.csv", f"{pre}_{name}_kernel_stats.csv")
os.makedirs(os.path.join(DST, f"{pre}_pmc"), exist_ok=True)
for wl in ("bunny", "soup"):
    for p in ("fetch", "write", "sq1", "sq2"):
        copy(f"pmc_{wl}_{p}/*/*_counter_collection.csv", f"{pre}_pmc/{wl}_{p}_counter_collection.csv")
for name in ("scenes.log", "shares.log", "shares_soup.log", "shares_multisphere.log", "sphere_screen.log", "diag_bunny.log", "launches_w8.log", "rounds_inlane0.log", "rounds_inlane24.log", "steps.log"):
    copy(name, f"{pre}_{name}")


def per_kernel(path, counter):
    """sum of <counter> and number of dispatches per kernel family (shade / trace / resolve) over the pass"""
    tot, cnt = {}, {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"]
        fam = "trace" if "front_trace_kernel" in k else "shade" if "front_shade_kernel" in k else "resolve" if "front_resolve_kernel" in k else None
        if fam is None:
            continue
        tot[fam] = tot.get(fam, 0.0) + float(row["Counter_Value"])
        cnt[fam] = cnt.get(fam, 0) + 1
        tot.setdefault("name_" + fam, k)
    return tot, cnt


tpath = os.path.join(DST, "traffic.json")
traffic = json.load(open(tpath))
for wl, key, spp_measured in (("bunny", "bunny_1200x1200_20spp_n1", 20), ("soup", "soup_3840x3840_4spp_n1", 4)):
    f, fc = per_kernel(os.path.join(DST, f"{pre}_pmc/{wl}_fetch_counter_collection.csv"), "FETCH_SIZE")
    w, wc = per_kernel(os.path.join(DST, f"{pre}_pmc/{wl}_write_counter_collection.csv"), "WRITE_SIZE")
    assert fc == wc, (fc, wc)
    per = {}
    for fam in ("shade", "trace", "resolve"):
        per[fam] = {"launches_per_frame": fc[fam] / FRAMES_PER_PASS,
                    "FETCH_SIZE_KB_per_frame": f[fam] / FRAMES_PER_PASS, "WRITE_SIZE_KB_per_frame": w[fam] / FRAMES_PER_PASS}
    # Both counters are in KiB.  FETCH_SIZE tallies 64 B per fabric read request.  The trace kernel's reads are per-lane gathers of
    # 8-16 bytes at scattered lines: 64-byte requests, counted in full (calibrated: tools/fetch_calib.hip, profiles/r03_fetch_calib --
    # 64 B per load at one load per random line, 2 x 58 B for both halves of a line).  Only wide coalesced streams (the shade
    # kernel's state) are 128-byte requests counted as 64, i.e. to be doubled, as MI355X_MICROARCH.md says.
    trace_bytes = (1.0 * f["trace"] + w["trace"]) * 1024.0
    traffic[key] = {
        "kernel": f["name_trace"], "round": pre, "measured_at_spp": spp_measured,
        "hbm_bytes_per_launch": trace_bytes / fc["trace"],
        "trace_launches_in_pass": fc["trace"], "frames_in_pass": FRAMES_PER_PASS, "per_kernel": per,
        "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{pre}_pmc/{wl}_fetch|write_counter_collection.csv), "
                f"{FRAMES_PER_PASS} frames of tools/prof_render.py {wl} {spp_measured}; (FETCH_SIZE + WRITE_SIZE) KiB summed over every "
                "front_trace_kernel dispatch of the pass and divided by the number of those dispatches (two chunks in flight, so a frame has "
                "launches_per_frame of them); FETCH_SIZE is NOT doubled for this kernel: its reads are 8-16-byte gathers, which the round-3 "
                "calibration (tools/fetch_calib.hip, profiles/r03_fetch_calib) shows to be 64-byte requests counted in full -- the doubling of "
                "MI355X_MICROARCH.md applies to wide coalesced reads (128-byte requests tallied at 64), i.e. to the shade kernel's FETCH_SIZE "
                "listed beside it (raw, undoubled KiB)"
                + ("" if spp_measured == 20 else "; measured at 4 spp because a 64-spp frame does not fit a counter pass: bench.py --workload soup "
                   "--spp 64 finds no key of its own and reports traffic null"),
    }
json.dump(traffic, open(tpath, "w"), indent=1)
print("profiles/ updated from", SRC)
for k in ("bunny_1200x1200_20spp_n1", "soup_3840x3840_4spp_n1"):
    print(k, "trace bytes/launch = %.1f MB" % (traffic[k]["hbm_bytes_per_launch"] / 1e6))
