#!/bin/bash
# A/B of build variants of the per-timestep launches on ONE box: rocprofv3 kernel stats of the reference-API loop (B given) for
# each set of NAF_BUILD_DEFINES. Usage: benchmarks/ab_step_prep.sh <batch> "<defines A>" "<defines B>" ...
batch=$1; shift
export TMPDIR=/tmp
i=0
for defs in "$@"; do
  i=$((i+1))
  export NAF_BUILD_DEFINES="$defs"
  d=/tmp/prof_ab_$i; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ab -- python3 benchmarks/host_api_steps.py $batch > /tmp/ab_$i.out 2>&1 || { tail -5 /tmp/ab_$i.out; exit 1; }
  echo "== defines: '$defs'  $(grep timesteps/s /tmp/ab_$i.out)"
  python3 benchmarks/stats_summary.py $(find $d -name "*kernel_stats.csv") --top 8 2>/dev/null | grep "step_prep\|adam_act"
done
