#!/bin/bash
# Round-6 evidence, from the repo root on the GPU box:  bash benchmarks/refresh_profiles.sh [part ...]
# parts: line | stats | rings | small | sweep | pmc | timeline | api   (default: all). Everything lands under gpurun_out/; the summaries quoted in
# DESIGN.md are then copied to profiles/ (r06_*). rocprofv3 always gets the program itself behind `--`.
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
parts=${@:-line stats rings small sweep pmc timeline api}
for part in $parts; do
case $part in
line)      # the bench line as the driver runs it (defaults) — with every other_configs entry
  python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_stderr.txt; tail -c 600 gpurun_out/r06_bench_line.json ;;
stats)     # rocprofv3 kernel-trace stats + digest of the bench at configs[1] and at the other row-split batch sizes
  benchmarks/prof_bench.sh r06_bench 300 40 > gpurun_out/prof_bench.log 2>&1; tail -3 gpurun_out/prof_bench.log
  benchmarks/prof_bench.sh r06_b512 200 30 --batch 512 > gpurun_out/prof_b512.log 2>&1; tail -2 gpurun_out/prof_b512.log
  benchmarks/prof_bench.sh r06_b1024 150 20 --batch 1024 --robot xarm6_robot --obstacle-jitter 0.1 > gpurun_out/prof_b1024.log 2>&1; tail -2 gpurun_out/prof_b1024.log
  benchmarks/prof_bench.sh r06_b2048 100 15 --batch 2048 --robot panda --buffer 4000000 > gpurun_out/prof_b2048.log 2>&1; tail -2 gpurun_out/prof_b2048.log ;;
rings)     # the bulk gather on EACH ring in a process of its own: the stats CSV alone gives each ring's fraction
  benchmarks/prof_bench.sh r06_ring4e6 6 2 --roofline-ring 4000000 --roofline-hbm-ring 0 > gpurun_out/prof_ring4e6.log 2>&1; tail -2 gpurun_out/prof_ring4e6.log
  benchmarks/prof_bench.sh r06_ring16e6 6 2 --roofline-ring 16000000 --roofline-hbm-ring 0 > gpurun_out/prof_ring16e6.log 2>&1; tail -2 gpurun_out/prof_ring16e6.log ;;
small)     # other batch sizes: configs[0]'s batch, the reference's default, sizes that are not whole 16-row groups, and B = 4096
  benchmarks/prof_bench.sh r06_b64 300 40 --batch 64 --buffer 100000 > gpurun_out/prof_b64.log 2>&1; tail -2 gpurun_out/prof_b64.log
  benchmarks/prof_bench.sh r06_b128 300 40 --batch 128 --buffer 100000 > gpurun_out/prof_b128.log 2>&1; tail -2 gpurun_out/prof_b128.log
  benchmarks/prof_bench.sh r06_b100 300 40 --batch 100 --buffer 100000 > gpurun_out/prof_b100.log 2>&1; tail -2 gpurun_out/prof_b100.log
  benchmarks/prof_bench.sh r06_b1000 150 20 --batch 1000 > gpurun_out/prof_b1000.log 2>&1; tail -2 gpurun_out/prof_b1000.log
  benchmarks/prof_bench.sh r06_b4096 60 10 --batch 4096 > gpurun_out/prof_b4096.log 2>&1; tail -2 gpurun_out/prof_b4096.log ;;
sweep)     # SURVEY 8d bulk sweep of the streaming kernels, HIP events (the table) AND rocprofv3 kernel stats of the same process
  d=/tmp/prof_sweep; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r06_sweep -- python3 benchmarks/roofline_sweep.py > gpurun_out/r06_roofline_sweep.md 2> gpurun_out/sweep.err
  cp $(find $d -name "*kernel_stats.csv") gpurun_out/r06_sweep_kernel_stats.csv; cat gpurun_out/r06_roofline_sweep.md ;;
pmc)       # HBM traffic of the bulk gather: separate passes per counter AND per ring (a process each), kernel trace only (never
           # combined with other trace domains)
  for ring in 4000000 16000000; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc_${ring}_$c; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o g -- python3 bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-extras --roofline-ring $ring --roofline-hbm-ring 0 > /tmp/pmc_${ring}_$c.out 2>&1
    python benchmarks/pmc_gather.py $d gpurun_out/r06_gather_pmc_ring${ring}_$c.csv
  done
  done ;;
timeline)  # phases inside the kernels, gaps between them (no profiler attached); needs its own build
  export NAF_BUILD_DEFINES=-DNAF_TIMELINE
  python benchmarks/kernel_timeline.py --batch 256 --out gpurun_out/r06_timeline_b256.json > gpurun_out/r06_timeline_b256.txt
  python benchmarks/kernel_timeline.py --batch 1024 --out gpurun_out/r06_timeline_b1024.json > gpurun_out/r06_timeline_b1024.txt
  python benchmarks/kernel_timeline.py --batch 2048 --robot panda --out gpurun_out/r06_timeline_b2048.json > gpurun_out/r06_timeline_b2048.txt
  unset NAF_BUILD_DEFINES
  head -30 gpurun_out/r06_timeline_b256.txt ;;
api)       # the per-timestep path (NAFAgent.act -> env.step -> NAFAgent.step, one host env): launches per timestep under rocprofv3,
           # host-side breakdown, the three forms A/B/C (pipelined | prefetch | fused: NAF_STEP_FORM) with and without a slower
           # environment, in-kernel timeline of the pipelined graph + the prefetch launch beside it (its own build), the probes
  benchmarks/prof_api_path.sh r06 64 > gpurun_out/prof_api_b64.log 2>&1; tail -3 gpurun_out/prof_api_b64.log
  benchmarks/prof_api_path.sh r06_b256 256 > gpurun_out/prof_api_b256.log 2>&1; tail -3 gpurun_out/prof_api_b256.log
  for b in 64 256; do python3 benchmarks/host_api_breakdown.py $b 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_api_breakdown.txt
  cat gpurun_out/r06_api_breakdown.txt
  bash benchmarks/ab_prefetch.sh > gpurun_out/r06_ab_pipeline.txt 2>&1; grep "==\|timesteps/s" gpurun_out/r06_ab_pipeline.txt
  bash benchmarks/ab_prefetch.sh delay > gpurun_out/r06_ab_pipeline_env_delay.txt 2>&1; grep "==\|timesteps/s" gpurun_out/r06_ab_pipeline_env_delay.txt
  (cd benchmarks/probe && hipcc -O2 --offload-arch=gfx950 -o graph_branch graph_branch.hip 2>/dev/null; ./graph_branch 6 7 5 13 9; ./graph_branch 6 7 5 13 30) > gpurun_out/r06_graph_branch_probe.txt; cat gpurun_out/r06_graph_branch_probe.txt
  python3 benchmarks/host_vector_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_host_vector.txt; cat gpurun_out/r06_host_vector.txt
  export NAF_BUILD_DEFINES=-DNAF_TIMELINE
  python3 benchmarks/step_timeline.py 64 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_step_timeline_b64.txt
  python3 benchmarks/step_timeline.py 256 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_step_timeline_b256.txt
  unset NAF_BUILD_DEFINES
  head -16 gpurun_out/r06_step_timeline_b64.txt ;;
esac
done
