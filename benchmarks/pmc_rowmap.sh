#!/bin/bash
# Memory-side traffic of the GEMM bundle with its dA1 blocks placed by block column (NAF_GB_ROWMAP=0) and by block row (1) on the
# XCDs: rocprofv3 --pmc passes (one counter set per pass, kernel trace only) over eager launches of a few vector steps, averaged
# per launch by benchmarks/pmc_aggregate.py. Usage (GPU box, repo root): benchmarks/pmc_rowmap.sh [bench.py args...]
#   -> gpurun_out/r02_pmc_rowmap{0,1}_${TAG:-b256}_<counter>.csv   (TAG names the batch size in the file names)
set -e
export TMPDIR=/tmp
for m in 0 1; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    d=/tmp/pmc_rowmap_${m}_$tag
    rm -rf $d
    NAF_GB_ROWMAP=$m rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --roofline-ring 0 "$@" > /tmp/pmc_rowmap.out 2>&1 || { tail -5 /tmp/pmc_rowmap.out; exit 1; }
    python3 benchmarks/pmc_aggregate.py $d gpurun_out/r02_pmc_rowmap${m}_${TAG:-b256}_$tag.csv
    grep "gemm_bundle\|kernel," gpurun_out/r02_pmc_rowmap${m}_${TAG:-b256}_$tag.csv
  done
done
