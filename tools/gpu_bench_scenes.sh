set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 200 python -u bench.py --workload simple_sphere --spp 1 --steps 20 --warmup 3 > $O/r03_bench_simple_sphere.json 2> $O/r03_bench_simple_sphere.err || { tail -5 $O/r03_bench_simple_sphere.err; exit 1; }
for w in multisphere suzanne bunny; do
  timeout -k 10 250 python -u bench.py --workload $w --steps 10 --warmup 2 > $O/r03_bench_$w.json 2> $O/r03_bench_$w.err || { tail -5 $O/r03_bench_$w.err; exit 1; }
done
python - <<'PY'
import json,os
for w in ("simple_sphere","multisphere","suzanne","bunny"):
    d=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out",f"r03_bench_{w}.json")))
    r=d["roofline"]; print(w, round(d["ms_per_step"],3),"ms", round(d["value"]),"Mrays/s", r["kernel"], "frac",round(r["frac"],3), "iso", r["frac_isolated"] and round(r["frac_isolated"],3), "valu", round(d["valu"]["frac"],4), "cpu", round(d["cpu_baseline"]["value"],1), "x", round(d["speedup_vs_cpu_baseline"]))
PY
