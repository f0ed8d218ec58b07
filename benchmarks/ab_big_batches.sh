#!/bin/bash
# updates/s beyond B = 2048 (round 4: the row-split chain up to 4096) against the unfused chain, and odd sizes below it.
# Usage (GPU box): bash benchmarks/ab_big_batches.sh > gpurun_out/ab_big_batches.txt
set -e
for b in 1000 2000 2500 3072 4096; do
  for fuse in default unfused; do
    NAF_FUSE=$fuse python bench.py --steps 200 --warmup 20 --batch $b --buffer 1000000 --no-extras --no-cpu-baseline 2>/dev/null | B=$b F=$fuse python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B = %5s  NAF_FUSE=%-8s %8.1f updates/s  %6.2f us/update  chain %s' % (os.environ['B'], os.environ['F'], d['value'], 1e6/d['value'], d['config'].get('chain','?')))"
  done
done
