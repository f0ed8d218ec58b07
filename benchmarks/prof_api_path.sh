#!/bin/bash
# The reference-API path (one host env: NAFAgent.act -> env.step -> NAFAgent.step per timestep) under rocprofv3: how many
# launches a timestep costs and what each takes. Usage (GPU box, repo root): benchmarks/prof_api_path.sh <tag> [batch] [joints] [layer_size]
#   -> gpurun_out/<tag>_api_path_kernel_stats.csv, <tag>_api_path_digest.csv (us and launches per timestep)
set -e
tag=${1:-r06}; batch=${2:-64}; joints=${3:-6}; layer=${4:-256}
export TMPDIR=/tmp
d=/tmp/prof_api_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o api -- python3 benchmarks/host_api_steps.py $batch $joints $layer > /tmp/prof_api_$tag.out 2>&1 || { tail -20 /tmp/prof_api_$tag.out; exit 1; }
tail -3 /tmp/prof_api_$tag.out
cp $(find $d -name "*kernel_stats.csv") gpurun_out/${tag}_api_path_kernel_stats.csv
# timesteps of the process: warm-up max(300, 4 B + 60) + 3000 timed
steps=$(python3 -c "print(max(300, 4 * $batch + 60) + 3000)")
python3 benchmarks/stats_summary.py gpurun_out/${tag}_api_path_kernel_stats.csv --updates $steps --top 24 --out gpurun_out/${tag}_api_path_digest.csv
cat gpurun_out/${tag}_api_path_digest.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_api_path_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows)
own = sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("step_prep", "step_prefetch", "bb_layer1", "bb_linear", "bb_layer2", "gemm_bundle", "adam_act", "policy_act", "replay_", "counter_add", "bb_moments", "adam_polyak")))
print(f"launches in the process: {calls} ({own} of the path's own kernels); per timestep ($steps timesteps incl. warm-up): {own / $steps:.2f}")
PY
