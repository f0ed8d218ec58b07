# shared-GPU rehearsal (every rank on cuda:0: functional, not a measurement of xGMI): the gradient exchange inside the finish launch
# (NAF_DP_EXCHANGE=merged) against the all-reduce as a launch of its own behind it (0); W = 2 and 4
cd $GRAFT_REPO_ROOT
export NAF_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
for rep in 1 2; do
for w in 2 4; do
for m in 0 1; do
  NAF_DP_EXCHANGE=$( [ $m = 1 ] && echo merged || echo oneshot ) python bench.py --gpus $w --steps 200 --warmup 30 --buffer 100000 --roofline-ring 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('W=$w merge=$m', o['value'], o['us_per_update'], o['sanity']['replicas_identical'], o['sanity'].get('xgmi_timed_out_waits'), o['sanity']['fold_fallbacks'])"
done; done; done
