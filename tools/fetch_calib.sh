set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03_fetch_calib; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $R/tools/fetch_calib > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
cat $O/fetch.log | grep -v "^W\|^I\|^E2" | tail -8
python3 - <<PY
import csv,glob
for f in glob.glob("$O/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"], "(KB) =", float(r["Counter_Value"])*1024, "bytes")
PY
