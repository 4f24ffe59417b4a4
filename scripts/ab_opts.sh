#!/bin/bash
# Developer A/B of library options on one box: bench.py (no CPU baseline, no extras) once per GPHIP_OPTIONS setting, interleaved.
#   bash scripts/ab_opts.sh "supertile=0" "supertile=2" ...
for rep in 1 2; do
  for o in "$@"; do
    GPHIP_OPTIONS="$o" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s ms/step %7.2f  syrk in-run %.3f  alone %.3f  kbuild %.3f' % ('$o', b['ms_per_step'], b['roofline']['frac'], b.get('roofline_syrk_alone',{}).get('frac',0), b['roofline_kbuild']['frac']))"
  done
done
