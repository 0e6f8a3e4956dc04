#!/bin/bash
# GPU-box script: the round's evidence.  Usage: gpu_profiles.sh <tag>   (outputs under gpurun_out/<tag>/)
# Every GPU step has its own time limit; the script stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
step() { echo "== $*" | tee -a $OUT/steps.log; }
step bench bunny
timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 > $OUT/bench_bunny.json 2> $OUT/bench_bunny.err || { tail -5 $OUT/bench_bunny.err; exit 1; }
step bench bunny, one chunk in flight
timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline > $OUT/bench_bunny_streams1.json 2> $OUT/bench_bunny_streams1.err || exit 1
export TMPDIR=/tmp; cd /tmp
step rocprofv3 kernel stats, bunny
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bunny -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-isolated > $OUT/prof_bunny.json 2> $OUT/prof_bunny.err || { tail -5 $OUT/prof_bunny.err; exit 1; }
step rocprofv3 kernel stats, bunny, one chunk in flight
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bunny_streams1 -- python3 $R/bench.py --steps 5 --warmup 1 --streams 1 --no-cpu-baseline > $OUT/prof_bunny_streams1.json 2> $OUT/prof_bunny_streams1.err || exit 1
run_pmc() { d=$1; w=$2; spp=$3; shift 3
  timeout -s KILL 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -- python3 $R/tools/prof_render.py $w $spp 2 > $OUT/$d.log 2>&1 || { echo "pass $d failed"; tail -5 $OUT/$d.log; exit 1; }
}
step PMC traffic passes, bunny
run_pmc pmc_bunny_fetch bunny 20 FETCH_SIZE && run_pmc pmc_bunny_write bunny 20 WRITE_SIZE || exit 1
step PMC stall passes, bunny
run_pmc pmc_bunny_sq1 bunny 20 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS || exit 1
run_pmc pmc_bunny_sq2 bunny 20 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU || exit 1
step PMC traffic and stall passes, soup at 4 spp
run_pmc pmc_soup_fetch soup 4 FETCH_SIZE && run_pmc pmc_soup_write soup 4 WRITE_SIZE || exit 1
run_pmc pmc_soup_sq1 soup 4 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS || exit 1
run_pmc pmc_soup_sq2 soup 4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU || exit 1
step rocprofv3 kernel stats, soup 3840x3840 at 8 spp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_soup -- python3 $R/bench.py --workload soup --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated > $OUT/prof_soup.json 2> $OUT/prof_soup.err || { tail -5 $OUT/prof_soup.err; exit 1; }
cd $R
step bench soup 3840x3840 at 64 spp
timeout -k 10 600 python -u bench.py --workload soup --spp 64 --steps 1 --warmup 0 > $OUT/bench_soup.json 2> $OUT/bench_soup.err || { tail -5 $OUT/bench_soup.err; exit 1; }
step bench lines of BASELINE configs 1-3
timeout -k 10 200 python -u bench.py --workload simple_sphere --spp 1 --steps 20 --warmup 3 > $OUT/bench_simple_sphere.json 2> $OUT/bench_simple_sphere.err || { tail -5 $OUT/bench_simple_sphere.err; exit 1; }
for w in multisphere suzanne; do
  timeout -k 10 250 python -u bench.py --workload $w --steps 10 --warmup 2 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { tail -5 $OUT/bench_$w.err; exit 1; }
done
step other scenes and rank shares
timeout -k 10 300 python -u tools/timing.py scenes > $OUT/scenes.log 2>&1 || exit 1
timeout -k 10 300 python -u tools/timing.py shares --streams 1,2,3 > $OUT/shares.log 2>&1 || exit 1
timeout -k 10 400 python -u tools/timing.py shares --workload soup --spp 16 > $OUT/shares_soup.log 2>&1 || exit 1
timeout -k 10 200 python -u tools/timing.py shares --workload multisphere > $OUT/shares_multisphere.log 2>&1 || exit 1
timeout -k 10 200 python -u tools/timing.py spheres > $OUT/sphere_screen.log 2>&1 || exit 1
step lane diagnostics and per-launch durations
RTM_DIAG=1 RTM_DIAG_KEEP_CULL=1 RTM_FRONT_STREAMS=1 timeout -k 10 120 python -u tools/diag.py bunny 20 > $OUT/diag_bunny.log 2>&1 || exit 1
timeout -k 10 120 python -u tools/prof_launches.py 8 > $OUT/launches_w8.log 2>&1 || exit 1
step paths alive and rays traced per round, without and with the in-lane advance
RTM_FRONT_INLANE=0 timeout -k 10 120 python -u tools/diag_rounds.py 1 > $OUT/rounds_inlane0.log 2>&1 || exit 1
timeout -k 10 120 python -u tools/diag_rounds.py 1 > $OUT/rounds_inlane24.log 2>&1 || exit 1
step done
