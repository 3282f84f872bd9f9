#!/bin/bash
# A/B of the small-world episode kernels through bench.py's c4_dim8 workload (1000 worlds of 8x8, 4 greedy agents):
# DW_NO_EPISODE_WAVE=1 (episode_small, the round-3 kernel) against the default (episode_wave).  usage: tools/ep_ab.sh [outfile]
OUT=${1:-/dev/stdout}
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'ms/step', d['ms_per_step'], 'cell-updates/s', d['value'])"; }
for m in exact fast; do
  DW_NO_EPISODE_WAVE=1 python3 bench.py --workload c4_dim8 --precision $m --steps 512 --warmup 64 --no-cpu-baseline --no-modes --preheat-s 0.5 2>/dev/null | line episode_small $m >> $OUT
  python3 bench.py --workload c4_dim8 --precision $m --steps 512 --warmup 64 --no-cpu-baseline --no-modes --preheat-s 0.5 2>/dev/null | line episode_wave $m >> $OUT
done
