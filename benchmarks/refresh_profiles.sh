set -x
python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_stderr.txt; tail -c 1500 gpurun_out/r02_bench_line.json
benchmarks/prof_bench.sh r02_bench 300 40 > gpurun_out/prof_bench.log 2>&1; tail -3 gpurun_out/prof_bench.log
benchmarks/prof_bench.sh r02_b1024 150 20 --batch 1024 > gpurun_out/prof_b1024.log 2>&1; tail -3 gpurun_out/prof_b1024.log
benchmarks/prof_bench.sh r02_b2048 100 15 --batch 2048 --robot panda > gpurun_out/prof_b2048.log 2>&1; tail -3 gpurun_out/prof_b2048.log
benchmarks/prof_bench.sh r02_b512 200 30 --batch 512 > gpurun_out/prof_b512.log 2>&1; tail -3 gpurun_out/prof_b512.log
export NAF_BUILD_DEFINES=-DNAF_TIMELINE
python benchmarks/kernel_timeline.py --batch 256 --out gpurun_out/r02_timeline_b256.json > gpurun_out/r02_timeline_b256.txt
python benchmarks/kernel_timeline.py --batch 1024 --out gpurun_out/r02_timeline_b1024.json > gpurun_out/r02_timeline_b1024.txt
python benchmarks/kernel_timeline.py --batch 2048 --robot panda --out gpurun_out/r02_timeline_b2048.json > gpurun_out/r02_timeline_b2048.txt
head -30 gpurun_out/r02_timeline_b256.txt
