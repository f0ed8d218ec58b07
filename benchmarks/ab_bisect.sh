#!/bin/bash
# bench.py from several checkouts (_ab_<commit>/, each built in place) and from HEAD, round-robin on one box
# usage: bash benchmarks/ab_bisect.sh "<dirs>" "<batches>" [reps]
dirs=${1:-"_ab_old ."}; batches=${2:-256}; reps=${3:-2}
one() {   # "dir" or "dir:-Dflag" (the flags the tree's library was built with), batch
  t=$1; d=${t%%:*}; def=""; [ "$t" != "$d" ] && def=${t#*:}
  ring=1000000; [ "$2" -le 128 ] && ring=100000
  (cd $d && NAF_BUILD_DEFINES=$def python bench.py --steps 1500 --warmup 50 --batch $2 --buffer $ring --no-extras --no-cpu-baseline 2>/dev/null) | D=$1 B=$2 python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s B = %5s %8.1f env-steps/s  %6.2f us' % (os.environ['D'], os.environ['B'], d['value'], 1e6/d['value']))"
}
for rep in $(seq $reps); do for b in $batches; do for d in $dirs; do one $d $b; done; done; done
