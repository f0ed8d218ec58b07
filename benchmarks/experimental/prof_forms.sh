# rocprofv3 digests of the row-split chain with the two forms of the backward GEMM launch (NAF_GEMM_FORM = 1: 32 x 32 blocks, 2: LDS-DMA ring)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for form in ${FORMS:-1 2}; do
export NAF_GEMM_FORM=$form
benchmarks/prof_bench.sh r04_b2048_form$form 100 15 --batch 2048 --robot panda --buffer 4000000 > gpurun_out/prof_x.log 2>&1; head -8 gpurun_out/r04_b2048_form${form}_digest.csv
benchmarks/prof_bench.sh r04_b1024_form$form 150 20 --batch 1024 --robot xarm6_robot --obstacle-jitter 0.1 > gpurun_out/prof_x.log 2>&1; head -8 gpurun_out/r04_b1024_form${form}_digest.csv
done
