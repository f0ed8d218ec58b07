cd $GRAFT_REPO_ROOT
for cols in 32 8 16 32 8 16; do
  export NAF_BUILD_DEFINES="-DGB_FOLD_COLS=$cols"
  for cfg in "--robot panda --batch 2048 --buffer 4000000" "--robot xarm6_robot --batch 1024 --obstacle-jitter 0.1" "--batch 256"; do
    python bench.py $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-extras --roofline-ring 0 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('cols $cols', o['config']['workload'][:40], o['updates_per_s'], o['us_per_update'], o['sanity']['params_finite'], o['sanity']['fold_fallbacks'])"
  done
done
unset NAF_BUILD_DEFINES
python -c "from robotic_manipulator_rloa_amd import _lib; _lib.build_library(force=True)"
