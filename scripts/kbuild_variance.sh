#!/bin/bash
# kernel-build HBM fraction as bench.py reports it, under different step / warm-up counts, on ONE box
for cfg in "5 1" "8 2" "20 2" "5 1" "8 2" "20 2"; do
  set -- $cfg
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['roofline_kbuild']
print('steps %2d warmup %d: ms/step %.2f kbuild frac %.3f avg %.3f ms (%d launches)' % (b['steps'], b['warmup'], b['ms_per_step'], k['frac'], k['avg_launch_ms'], k['launches']))"
done
