#!/bin/bash
# Board power and engine clock while the step-pair kernels run at the north-star shape (1024 x 4096^2), sampled from rocm-smi
# next to a running bench.py; then the same for C2 (short launches).  usage (GPU box, repo root): bash tools/power_trace.sh <outfile>
OUT=${1:-gpurun_out/power_trace.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
: > "$OUT"
sample() {   # <label> <pid>
  while kill -0 "$2" 2>/dev/null; do
    p=$(rocm-smi --showpower 2>/dev/null | grep -m1 "(W)" | sed 's/.*: *//')
    c=$(rocm-smi --showclocks 2>/dev/null | grep -i -m1 "sclk" | sed 's/.*: *//')
    echo "$1 $(date +%s.%N | cut -c1-14) power: $p | sclk: $c" >> "$OUT"
    sleep 0.4
  done
}
for spec in "target fast 240" "target exact 180" "c2 fast 512" "c2 exact 512" "c2 fast 512" "c2 exact 512"; do
  set -- $spec
  python3 "$R/bench.py" --workload $1 --precision $2 --steps $3 --warmup 4 --no-cpu-baseline --no-modes --no-workloads --no-power --preheat-s 0.3 \
      > "$OUT.$1.$2.json" 2> /dev/null &
  pid=$!
  sample "$1/$2" $pid
  wait $pid
  python3 -c "import json,sys; d=json.loads(open('$OUT.$1.$2.json').read().strip().splitlines()[-1]); print('$1/$2 bench: ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'])" >> "$OUT"
done
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^=\|^$" >> "$OUT"
