#!/bin/bash
# Round-4 evidence, from the repo root on the GPU box:  bash benchmarks/refresh_profiles.sh [part ...]
# parts: line | stats | small | sweep | pmc | timeline   (default: all). Everything lands under gpurun_out/; the summaries quoted in
# DESIGN.md are then copied to profiles/ (r04_*). rocprofv3 always gets the program itself behind `--`.
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
parts=${@:-line stats small sweep pmc timeline}
for part in $parts; do
case $part in
line)      # the bench line as the driver runs it (defaults) — with every other_configs entry
  python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_stderr.txt; tail -c 600 gpurun_out/r04_bench_line.json ;;
stats)     # rocprofv3 kernel-trace stats + digest of the bench at configs[1] and at the other row-split batch sizes
  benchmarks/prof_bench.sh r04_bench 300 40 > gpurun_out/prof_bench.log 2>&1; tail -3 gpurun_out/prof_bench.log
  benchmarks/prof_bench.sh r04_b512 200 30 --batch 512 > gpurun_out/prof_b512.log 2>&1; tail -2 gpurun_out/prof_b512.log
  benchmarks/prof_bench.sh r04_b1024 150 20 --batch 1024 --robot xarm6_robot --obstacle-jitter 0.1 > gpurun_out/prof_b1024.log 2>&1; tail -2 gpurun_out/prof_b1024.log
  benchmarks/prof_bench.sh r04_b2048 100 15 --batch 2048 --robot panda --buffer 4000000 > gpurun_out/prof_b2048.log 2>&1; tail -2 gpurun_out/prof_b2048.log ;;
small)     # other batch sizes: configs[0]'s batch, the reference's default, sizes that are not whole 16-row groups, and B = 4096
  benchmarks/prof_bench.sh r04_b64 300 40 --batch 64 --buffer 100000 > gpurun_out/prof_b64.log 2>&1; tail -2 gpurun_out/prof_b64.log
  benchmarks/prof_bench.sh r04_b128 300 40 --batch 128 --buffer 100000 > gpurun_out/prof_b128.log 2>&1; tail -2 gpurun_out/prof_b128.log
  benchmarks/prof_bench.sh r04_b100 300 40 --batch 100 --buffer 100000 > gpurun_out/prof_b100.log 2>&1; tail -2 gpurun_out/prof_b100.log
  benchmarks/prof_bench.sh r04_b1000 150 20 --batch 1000 > gpurun_out/prof_b1000.log 2>&1; tail -2 gpurun_out/prof_b1000.log
  benchmarks/prof_bench.sh r04_b4096 60 10 --batch 4096 > gpurun_out/prof_b4096.log 2>&1; tail -2 gpurun_out/prof_b4096.log ;;
sweep)     # SURVEY 8d bulk sweep of the streaming kernels, HIP events (the table) AND rocprofv3 kernel stats of the same process
  d=/tmp/prof_sweep; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r04_sweep -- python3 benchmarks/roofline_sweep.py > gpurun_out/r04_roofline_sweep.md 2> gpurun_out/sweep.err
  cp $(find $d -name "*kernel_stats.csv") gpurun_out/r04_sweep_kernel_stats.csv; cat gpurun_out/r04_roofline_sweep.md ;;
pmc)       # HBM traffic of the bulk gather: separate passes per counter, kernel trace only (never combined with other domains)
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc_$c; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o g -- python3 bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-extras > /tmp/pmc_$c.out 2>&1
    python benchmarks/pmc_gather.py $d gpurun_out/r04_gather_pmc_$c.csv
  done ;;
timeline)  # phases inside the kernels, gaps between them (no profiler attached); needs its own build
  export NAF_BUILD_DEFINES=-DNAF_TIMELINE
  python benchmarks/kernel_timeline.py --batch 256 --out gpurun_out/r04_timeline_b256.json > gpurun_out/r04_timeline_b256.txt
  python benchmarks/kernel_timeline.py --batch 1024 --out gpurun_out/r04_timeline_b1024.json > gpurun_out/r04_timeline_b1024.txt
  python benchmarks/kernel_timeline.py --batch 2048 --robot panda --out gpurun_out/r04_timeline_b2048.json > gpurun_out/r04_timeline_b2048.txt
  unset NAF_BUILD_DEFINES
  head -30 gpurun_out/r04_timeline_b256.txt ;;
esac
done
