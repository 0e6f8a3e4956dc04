#!/bin/bash
# Memory-pipeline PMC passes (TA/TCP/TD).  Usage: gpu_pmc_mem.sh <tag> [workload] [spp]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmcmem_$1; mkdir -p $OUT
W=${2:-bunny}; SPP=${3:-20}
export TMPDIR=/tmp; cd /tmp
run() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/tools/prof_render.py $W $SPP 2 > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }
}
run m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE &&
run m2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum &&
run m3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES
find $OUT -name '*counter_collection.csv' | head
