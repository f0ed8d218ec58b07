set -e
cd $GRAFT_REPO_ROOT
for form in 1 2 1 2; do
  for cfg in "--robot panda --batch 2048 --buffer 4000000" "--robot xarm6_robot --batch 1024 --obstacle-jitter 0.1"; do
    NAF_GEMM_FORM=$form python bench.py $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-extras --roofline-ring 0 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('form $form', o['config']['workload'][:40], o['updates_per_s'], o['us_per_update'], o['sanity']['params_finite'], o['sanity']['fold_fallbacks'])"
  done
done
