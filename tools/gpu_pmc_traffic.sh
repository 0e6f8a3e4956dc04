#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE in separate runs).  Usage: gpu_pmc_traffic.sh <tag> [workload] [spp]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
W=${2:-bunny}; SPP=${3:-20}
export TMPDIR=/tmp; cd /tmp
run() { n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/tools/prof_render.py $W $SPP 2 > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE
