#!/bin/bash
# A/B of two library builds on the agent workloads of bench.py (c4: 1000 x 256^2 with 4 greedy agents and world flags; c3; c5)
#   usage: tools/c4_ab.sh <other lib.so> [outfile]
OTHER=$1; OUT=${2:-/dev/stdout}
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', '$3', 'ms/step', round(d['ms_per_step'],5), 'cell-updates/s %.4g' % d['value'])"; }
for w in c4 c3 c5; do for m in exact fast; do
  for arm in base new base new; do
    if [ $arm = base ]; then export DW_LIB=$OTHER; else unset DW_LIB; fi
    python3 bench.py --workload $w --precision $m --steps 128 --warmup 16 --no-cpu-baseline --no-modes --no-workloads --preheat-s 0.5 2>/dev/null | line $w $m $arm >> $OUT
  done
done; done
