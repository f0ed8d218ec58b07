# batch sizes that are multiples of 16 but not of 64: the row-split chain with a partial last block (round 4) against the chain they
# took before (column tiles up to 512, unfused beyond)
cd $GRAFT_REPO_ROOT
for cfg in "96 columns" "96 rows" "160 columns" "160 rows" "1200 unfused" "1200 rows" "2000 unfused" "2000 rows" "1008 rows" "1024 rows"; do
  set -- $cfg
  NAF_FUSE=$2 python -W ignore bench.py --batch $1 --buffer 100000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --roofline-ring 0 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('B=$1 fuse=$2', o['config']['chain'], o['updates_per_s'], o['us_per_update'], o['sanity']['params_finite'])"
done
