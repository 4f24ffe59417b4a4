#!/bin/bash
# same-box A/B of library builds on the single-theta latency table: bash scripts/ab_sizes_lib.sh base default [sizes...]
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $A $B; do
    if [ "$v" = default ]; then unset GPHIP_LIB; else export GPHIP_LIB=$R/bayesianinference_amd/lib/variants/libgphip_$v.so; fi
    echo "== $v"; python scripts/gpu_sizes.py ${@:-512 1024 2048 4096 8192} 2>/dev/null | grep N=
  done
done
