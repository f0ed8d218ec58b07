#!/bin/bash
# the per-timestep path: pipelined (default) | prefetch | fused, A/B/C on one box (NAF_STEP_FORM); with "delay": the same with a
# slower environment (NAF_BENCH_ENV_DELAY_US of busy waiting per env.step).
# Usage (GPU box): bash benchmarks/ab_prefetch.sh [delay] > gpurun_out/ab_prefetch.txt
set -e
run() { echo "== B=$1 NAF_STEP_FORM=$2 env delay ${3:-0} us"
        NAF_BENCH_ENV_DELAY_US=${3:-0} NAF_STEP_FORM=$2 python benchmarks/host_api_breakdown.py $1 2>/dev/null | head -7; }
if [ "$1" = "delay" ]; then
  for d in 0 10 30 100; do run 256 pipelined $d; run 256 prefetch $d; run 256 fused $d; done
  exit 0
fi
for b in 64 256; do
  for rep in 1 2; do run $b pipelined; run $b prefetch; run $b fused; done
done
