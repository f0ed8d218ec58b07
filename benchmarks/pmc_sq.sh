#!/bin/bash
# SQ counters per launch of the row-split chain's kernels at B = 2048 (two 8-counter passes, kernel trace only):
#   bash benchmarks/pmc_sq.sh <tag>   ->  gpurun_out/<tag>_p{1,2}.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1
P1="SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=1
for P in "$P1" "$P2"; do
  d=/tmp/sq_${tag}_$i; rm -rf $d
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $d -o sq -- python3 bench.py --batch 2048 --robot panda --buffer 4000000 --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-extras --roofline-ring 0 > /tmp/sq_${tag}_$i.out 2>&1 || tail -5 /tmp/sq_${tag}_$i.out
  python3 benchmarks/pmc_aggregate.py $d gpurun_out/${tag}_p$i.csv
  i=$((i+1))
done
head -4 gpurun_out/${tag}_p1.csv | cut -c1-400
