#!/bin/bash
# Every fuzzer once, with the given seed (on the GPU box, repo root):  bash tools/fuzz_all.sh [seed]
# One line per fuzzer; non-zero exit if any case differs.
SEED=${1:-1}
rc=0
for cmd in "fuzz_dropin.py 150" "fuzz_harness.py 40" "fuzz_fitness.py 60" "fuzz_engine.py 100" "fuzz_pairs.py 100" \
           "fuzz_exact.py 150" "fuzz_fast_consistency.py 40"; do
  out=$(timeout -k 10 600 python3 tools/$cmd $SEED 2>&1 | grep -v amdgpu.ids | tail -1)
  echo "tools/$cmd $SEED: $out"
  case "$out" in *"$(echo $cmd | cut -d' ' -f2)/$(echo $cmd | cut -d' ' -f2) cases"*) ;; *) rc=1 ;; esac
done
exit $rc
