#!/bin/bash
# GPU-box script: is the trace kernel bound by the texture-addresser / L1 path?  Usage: gpu_pmc_ta.sh <tag> [workload spp]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-ta}; WL=${2:-bunny}; SPP=${3:-20}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run_pmc() { d=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -- python3 $R/tools/prof_render.py $WL $SPP 2 > $OUT/$d.log 2>&1 || { echo "pass $d failed"; tail -5 $OUT/$d.log; exit 1; }
}
run_pmc ta1 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum || exit 1
run_pmc ta2 GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || exit 1
run_pmc ta3 GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum || exit 1
python3 $R/tools/pmc_summary.py $OUT
