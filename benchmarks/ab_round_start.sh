#!/bin/bash
# A/B/A/B on one box: the tree at the start of the round (a checkout under _ab_old/, built there) against HEAD, bench.py at three
# batch sizes. Usage (GPU box): bash benchmarks/ab_round_start.sh > gpurun_out/ab_round_start.txt
one() {  # dir batch ring
  (cd $1 && python bench.py --steps 1500 --warmup 50 --batch $2 --buffer $3 --no-extras --no-cpu-baseline 2>/dev/null) | D=$1 B=$2 python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s B = %5s  %8.1f env-steps/s' % ('old' if os.environ['D']!='.' else 'HEAD', os.environ['B'], d['value']))"
}
for rep in 1 2; do
  for b in 256 64 1024; do
    ring=1000000; [ "$b" = 64 ] && ring=100000
    one _ab_old $b $ring
    one . $b $ring
  done
done
