#!/bin/bash
# per-kernel rocprofv3 averages of bench.py (B = 256; BENCH_ARGS for another shape) from several checkouts on one box (TREES)
for rep in 1 2 3; do
for t in ${TREES:-_ab_old .}; do
  d=${t%%:*}; def=""; [ "$t" != "$d" ] && def=${t##*:}      # "dir" or "dir:-DDEFINE" (the tree's library was built with it)
  mkdir -p $d/gpurun_out
  (cd $d && NAF_BUILD_DEFINES=$def bash benchmarks/prof_bench.sh t256 300 40 ${BENCH_ARGS:-} > /tmp/t.log 2>&1)
  echo "== $d"; python3 - $d/gpurun_out/t256_digest.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
want=['gemm_bundle','layer2_head','bb_layer1_kernel<6, true','linear_stats16_kernel<true','bwd_finish']
out=[]
for w in want:
    for r in rows:
        if w in r['kernel']:
            out.append('%s %.3f/%.2f' % (w[:12], float(r['avg_us']), float(r['min_us']))); break
print('  '.join(out))
PY
done; done
