#!/bin/bash
# the per-timestep path with and without the prefetch of the next timestep's minibatch (NAF_STEP_PREFETCH), A/B/A/B on one box;
# with "delay": the same with a slower environment (NAF_BENCH_ENV_DELAY_US of busy waiting per env.step) — does the launch call's
# cost depend on how long ago the previous graph finished?
# Usage (GPU box): bash benchmarks/ab_prefetch.sh [delay] > gpurun_out/ab_prefetch.txt
set -e
if [ "$1" = "delay" ]; then
  for d in 0 10 30; do
    for pf in 1 0; do
      echo "== B=256 NAF_STEP_PREFETCH=$pf env delay $d us"
      NAF_BENCH_ENV_DELAY_US=$d NAF_STEP_PREFETCH=$pf python benchmarks/host_api_breakdown.py 256 2>/dev/null | head -7
    done
  done
  exit 0
fi
for b in 64 256; do
  for pf in 1 0 1 0; do
    echo "== B=$b NAF_STEP_PREFETCH=$pf"
    NAF_STEP_PREFETCH=$pf python benchmarks/host_api_breakdown.py $b 2>/dev/null | head -7
  done
done
