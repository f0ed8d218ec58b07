#!/bin/bash
# The reference-API path (one host env: NAFAgent.act -> env.step -> NAFAgent.step per timestep) under rocprofv3: how many
# launches a timestep costs and what each takes. Usage (GPU box, repo root): benchmarks/prof_api_path.sh <tag> [batch]
#   -> gpurun_out/<tag>_api_path_kernel_stats.csv, <tag>_api_path_digest.csv (us and launches per timestep)
set -e
tag=${1:-r05}; batch=${2:-64}
export TMPDIR=/tmp
d=/tmp/prof_api_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o api -- python3 benchmarks/host_api_steps.py $batch > /tmp/prof_api_$tag.out 2>&1 || { tail -20 /tmp/prof_api_$tag.out; exit 1; }
tail -3 /tmp/prof_api_$tag.out
cp $(find $d -name "*kernel_stats.csv") gpurun_out/${tag}_api_path_kernel_stats.csv
python3 benchmarks/stats_summary.py gpurun_out/${tag}_api_path_kernel_stats.csv --updates 3300 --top 24 --out gpurun_out/${tag}_api_path_digest.csv
cat gpurun_out/${tag}_api_path_digest.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_api_path_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows)
print(f"launches in the process: {calls}; per timestep (3300 timesteps incl. warm-up): {calls / 3300:.2f}")
PY
