#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py configuration, digested. Usage (on the GPU box, from the repo root):
#   benchmarks/prof_bench.sh <tag> <steps> <warmup> [bench.py args...]
# writes gpurun_out/<tag>_kernel_stats.csv (raw rocprofv3 stats) and gpurun_out/<tag>_digest.csv
set -e
tag=$1; steps=$2; warm=$3; shift 3
export TMPDIR=/tmp
d=/tmp/prof_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o $tag -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-extras "$@" > /tmp/prof_$tag.out 2>&1 || { tail -20 /tmp/prof_$tag.out; exit 1; }
cp $(find $d -name "*kernel_stats.csv") gpurun_out/${tag}_kernel_stats.csv
# updates = (steps + warmup + graph-capture warm-ups: 2 chunk warm-ups) * 64
python3 benchmarks/stats_summary.py gpurun_out/${tag}_kernel_stats.csv --updates $(( (steps + warm + 2) * 64 )) --top 22 --out gpurun_out/${tag}_digest.csv
cat gpurun_out/${tag}_digest.csv
grep -o '"value": [0-9.]*' /tmp/prof_$tag.out | head -1
# the bench line of THIS (profiled) process: its roofline.avg_launch_ms is the HIP-event view of the very launches the CSV averages
grep '^{' /tmp/prof_$tag.out | tail -1 > gpurun_out/${tag}_line_under_rocprof.json
