#!/bin/bash
# updates/s of batch sizes that are not whole 16-row groups (round 4: the row-split chain with a partial last workgroup) beside
# their neighbours that are. Usage (GPU box): bash benchmarks/ab_odd_batches.sh > gpurun_out/ab_odd_batches.txt
set -e
for b in 64 65 100 112 127 128 250 256 500 512 1000 1024 2000 2047 2048; do
  ring=100000; [ "$b" -ge 256 ] && ring=1000000
  python bench.py --steps 300 --warmup 30 --batch $b --buffer $ring --no-extras --no-cpu-baseline 2>/dev/null | B=$b python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B = %5s  %8.1f updates/s  %6.2f us/update  chain %s' % (os.environ['B'], d['value'], 1e6/d['value'], d['config'].get('chain','?')))"
done
