#!/bin/bash
# like ab_opts.sh, with the other configs: cfg 2 / 4 / 5 and N = 16384 / 24576 single-theta evaluations
for rep in 1 2; do
  for o in "$@"; do
    echo "== $o"
    GPHIP_OPTIONS="$o" python scripts/gpu_sizes.py 16384 24576 2>/dev/null | grep "N="
    GPHIP_OPTIONS="$o" python scripts/gpu_cfg5.py 2>/dev/null | grep "^fit\|fp32 N"
  done
done
