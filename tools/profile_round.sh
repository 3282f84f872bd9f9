#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats and the HBM-traffic PMC passes.
#   usage (from the repo root, on the box):  bash tools/profile_round.sh <tag>      e.g. r01e
# Outputs under gpurun_out/prof_<tag>/ ; tools/traffic_from_pmc.py turns the PMC CSVs into profiles/traffic_*.json.
# (counters in their own passes with --kernel-trace only; the program itself follows `--`.)
# The --stats pass runs bench.py's default 512 steps after 64 of warm-up, so that its per-kernel average is
# comparable with the bench line (shorter runs see clocks that are still rising: +8..20 % per launch).
set -e -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
for prec in fast exact; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$prec" -o s --output-format csv -- \
      python3 "$R/bench.py" --steps 512 --warmup 64 --precision $prec --no-cpu-baseline --no-modes \
      > "$OUT/stats_$prec.json" 2> "$OUT/stats_$prec.err"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/pmc_${prec}_$ctr" -o p --output-format csv -- \
        python3 "$R/bench.py" --steps 13 --warmup 3 --precision $prec --no-cpu-baseline --no-modes \
        > "$OUT/pmc_${prec}_$ctr.json" 2> "$OUT/pmc_${prec}_$ctr.err"
  done
  echo "profiled $prec"
done
cat "$OUT/bench_default.json"
