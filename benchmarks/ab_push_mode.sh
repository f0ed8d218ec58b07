cd $GRAFT_REPO_ROOT
export NAF_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 NAF_DP_EXCHANGE=merged
for mode in 0 1 2; do
  export NAF_BUILD_DEFINES="-DBB_PUSH_MODE=$mode"
  python -c "from robotic_manipulator_rloa_amd import _lib; _lib.build_library()" 
  for w in 2; do
  python bench.py --gpus $w --steps 200 --warmup 30 --buffer 100000 --roofline-ring 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('mode $mode W=$w', o['value'], o['us_per_update'], o['sanity']['replicas_identical'], o['sanity'].get('xgmi_timed_out_waits'), o['sanity']['fold_fallbacks'])"
  done
done
unset NAF_BUILD_DEFINES
python -c "from robotic_manipulator_rloa_amd import _lib; _lib.build_library(force=True)"
