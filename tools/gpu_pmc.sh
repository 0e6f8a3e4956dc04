#!/bin/bash
# PMC passes (each its own run, no tracing domains besides kernel-trace). Usage: gpu_pmc.sh <tag> [workload] [spp]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT
W=${2:-bunny}; SPP=${3:-20}
export TMPDIR=/tmp; cd /tmp
run() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/tools/prof_render.py $W $SPP 2 > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
find $OUT -name '*counter_collection.csv' | head -20
