#!/bin/bash
# Round profile on the GPU box for ONE workload of bench.py: bench line, rocprofv3 kernel stats, the HBM-traffic
# PMC passes and the SQ (VALU) counter passes, both arithmetic modes.
#   usage (from the repo root, on the box):  bash tools/profile_round.sh <tag> <workload> [steps]     e.g. r02a target 21
# Outputs under gpurun_out/prof_<tag>_<workload>/ ; tools/profile_summarise.py turns them into profiles/*.
# Counters run in their own passes with --kernel-trace only, and the program itself follows `--` (python3, no
# wrapper): see the task statement's rules for rocprofv3 on this pool.
set -e -o pipefail
TAG=${1:-rXX}
WL=${2:-target}
STEPS=${3:-21}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--workload $WL --warmup 4 --no-cpu-baseline --no-modes --no-workloads --no-power --preheat-s 0.3"
for prec in exact fast; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$prec" -o s --output-format csv -- \
      python3 "$R/bench.py" $COMMON --steps $((STEPS * 3)) --preheat-s 1.5 --precision $prec \
      > "$OUT/stats_$prec.json" 2> "$OUT/stats_$prec.err"
  echo "stats $prec done"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    name=$(echo $set | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc_${prec}_$name" -o p --output-format csv -- \
        python3 "$R/bench.py" $COMMON --steps $STEPS --precision $prec \
        > "$OUT/pmc_${prec}_$name.json" 2> "$OUT/pmc_${prec}_$name.err"
    echo "pmc $prec $name done"
  done
done
