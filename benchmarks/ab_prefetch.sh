#!/bin/bash
# the per-timestep path: pipelined (default) | prefetch only (NAF_STEP_PIPELINE=0) | neither (NAF_STEP_PREFETCH=0), A/B/C on one
# box; with "delay": the same with a slower environment (NAF_BENCH_ENV_DELAY_US of busy waiting per env.step).
# Usage (GPU box): bash benchmarks/ab_prefetch.sh [delay] > gpurun_out/ab_prefetch.txt
set -e
run() { echo "== B=$1 NAF_STEP_PIPELINE=$2 NAF_STEP_PREFETCH=$3 env delay ${4:-0} us"
        NAF_BENCH_ENV_DELAY_US=${4:-0} NAF_STEP_PIPELINE=$2 NAF_STEP_PREFETCH=$3 python benchmarks/host_api_breakdown.py $1 2>/dev/null | head -7; }
if [ "$1" = "delay" ]; then
  for d in 0 10 30 100; do run 256 1 1 $d; run 256 0 1 $d; run 256 0 0 $d; done
  exit 0
fi
for b in 64 256; do
  for rep in 1 2; do run $b 1 1; run $b 0 1; run $b 0 0; done
done
