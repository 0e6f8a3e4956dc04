#!/bin/bash
# Knob sweep (env settings are read at rtm_init: one process per setting): rank shares of the bunny frame, soup at 4 spp.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/sweep.log; : > $O
timeout -k 10 400 python -u tools/time_share_variants.py default \
  default@RTM_FRONT_CLAIM=64 default@RTM_FRONT_CLAIM=256 default@RTM_FRONT_STEAL=0 default@RTM_FRONT_STEAL=24 \
  default@RTM_FRONT_REFILL=16 default@RTM_FRONT_REFILL=32 default@RTM_TEST_FRAC=36 default@RTM_TEST_FRAC=52 \
  default@RTM_FRONT_TRACE_WGS=4 default@RTM_FRONT_TRACE_WGS=2 default@RTM_FRONT_SHADE_WAVES=16 default@RTM_FRONT_SHADE_WAVES=32 \
  default@RTM_FRONT_MAX_PATHS=8388608 default@RTM_FRONT_MAX_PATHS=16777216 >> $O 2>&1
timeout -k 10 500 python -u tools/time_variants.py soup 4 default \
  default@RTM_TEST_FRAC=24 default@RTM_TEST_FRAC=32 default@RTM_TEST_FRAC=56 default@RTM_FRONT_REFILL=12 default@RTM_FRONT_REFILL=32 \
  default@RTM_FRONT_REFILL=44 default@RTM_FRONT_CLAIM=64 default@RTM_FRONT_CLAIM=256 default@RTM_FRONT_TRACE_WGS=3 \
  default@RTM_FRONT_STEAL=0 default@RTM_FRONT_STEAL=24 default@RTM_FRONT_STREAMS=1 default@RTM_FRONT_STREAMS=3 \
  default@RTM_FRONT_SHADE_WAVES=16 default@RTM_FRONT_MAX_PATHS=8388608 >> $O 2>&1
echo done >> $O
