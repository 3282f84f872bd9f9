#!/bin/bash
# SQ counters of the fused kernels (VALU instruction count and busy time) -> gpurun_out/valu_<tag>/ ; summarised by
# tools/valu_from_pmc.py into profiles/<tag>_valu_pmc.json.   usage (GPU box, repo root): bash tools/valu_pmc.sh <tag>
# (counters in their own passes with --kernel-trace only; the program itself follows `--`.)
set -e -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/valu_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for prec in fast exact; do
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    name=$(echo $set | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $set -d "$OUT/${prec}_$name" -o p --output-format csv -- \
        python3 "$R/bench.py" --steps 41 --warmup 7 --precision $prec --no-cpu-baseline --no-modes \
        > "$OUT/${prec}_$name.json" 2> "$OUT/${prec}_$name.err"
  done
  echo "counters $prec done"
done
