#!/bin/bash
# GEMM 2 with 64 x 16 tiles up to B = BB_MAX16 (default 512) against 1024 / 2048: updates/s, library rebuilt on the box per setting
for rep in 1 2; do
for m in 512 1024 2048; do
  export NAF_BUILD_DEFINES=-DBB_MAX16=$m
  for b in 768 1024 1536 2048; do
    python bench.py --steps 400 --warmup 30 --batch $b --no-extras --no-cpu-baseline 2>/dev/null | M=$m B=$b python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BB_MAX16 = %4s  B = %5s  %8.1f updates/s' % (os.environ['M'], os.environ['B'], d['value']))"
  done
done; done
