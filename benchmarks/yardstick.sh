#!/bin/bash
# Vendor-GEMM yardstick (benchmarks/vendor_gemm_yardstick.py) plain and under rocprofv3 --kernel-trace --stats.
# Usage (GPU box, repo root): benchmarks/yardstick.sh <tag>   -> gpurun_out/<tag>_vendor_gemm_yardstick.{md,json}, <tag>_yardstick_kernel_stats.csv
set -e
tag=${1:-r05}
export TMPDIR=/tmp
python3 benchmarks/vendor_gemm_yardstick.py --out gpurun_out/${tag}_vendor_gemm_yardstick.md > /tmp/yard_$tag.out 2>&1 || { tail -30 /tmp/yard_$tag.out; exit 1; }
tail -25 /tmp/yard_$tag.out
d=/tmp/prof_yard_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o yard -- python3 benchmarks/vendor_gemm_yardstick.py --out /tmp/yard_prof_$tag.md > /tmp/yard_prof_$tag.out 2>&1 || { tail -20 /tmp/yard_prof_$tag.out; exit 1; }
cp $(find $d -name "*kernel_stats.csv") gpurun_out/${tag}_yardstick_kernel_stats.csv
python3 benchmarks/stats_summary.py gpurun_out/${tag}_yardstick_kernel_stats.csv --top 30 --out gpurun_out/${tag}_yardstick_digest.csv
cat gpurun_out/${tag}_yardstick_digest.csv
