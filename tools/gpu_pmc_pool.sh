# SQ counters of the two trace kernels (one ray per lane / LDS pools) on the bunny frame, one chunk in flight
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; cd /tmp
export RTM_FRONT_STREAMS=1
for pool in 0 1; do
  export RTM_FRONT_POOL=$pool
  O=$R/gpurun_out/r03_pmc_pool$pool; mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -- python3 $R/tools/prof_render.py bunny 20 2 > $O/sq1.log 2>&1 || { tail -5 $O/sq1.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/prof_render.py bunny 20 2 > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
  echo "== pool $pool"; python3 $R/tools/pmc_summary.py $O | grep -A20 "trace_kernel"
done
