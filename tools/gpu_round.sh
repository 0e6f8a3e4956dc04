#!/bin/bash
# GPU-box script: tests, bench, rocprofv3 kernel stats.  Each GPU step has its own time limit; stop at first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -5 $OUT/pytest_gpu.log
timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err || { echo "rocprof failed"; tail -20 $OUT/prof.err; exit 1; }
find $OUT/prof -name '*kernel_stats*' | head
