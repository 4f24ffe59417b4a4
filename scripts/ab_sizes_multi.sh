#!/bin/bash
# same-box A/B of several library builds on the single-theta latency table:
#   bash scripts/ab_sizes_multi.sh "default noacq sc1" 512 2048 4096 ...
VARS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in $VARS; do
    if [ "$v" = default ]; then unset GPHIP_LIB; else export GPHIP_LIB=$R/bayesianinference_amd/lib/variants/libgphip_$v.so; fi
    echo "== $v"; python scripts/gpu_sizes.py ${@:-512 1024 2048 4096 8192} 2>/dev/null | grep N=
  done
done
