#!/bin/bash
# in-kernel timelines (B = 256) of two checkouts on one box; each builds its own -DNAF_TIMELINE library
export NAF_BUILD_DEFINES=-DNAF_TIMELINE
for rep in 1 2; do
for d in _ab_old .; do
  echo "== $d (rep $rep)"
  (cd $d && python benchmarks/kernel_timeline.py --batch ${TLB:-256} --reps 400 --out /tmp/tl.json 2>/dev/null | head -19)
done; done
