#!/bin/bash
# Quick SQ-counter comparison of library builds on one shape (tuning aid; the judged profiles come from
# tools/profile_round.sh).  usage (on the GPU box, from the repo root):
#   bash tools/pmc_quick.sh <outdir> <B> <G> <precision> name=libpath[@ENV=V] [name=libpath ...]
# Every counter set is its own rocprofv3 pass with --kernel-trace only, and the program itself follows `--`.
set -e -o pipefail
OUT=$1; B=$2; G=$3; PREC=$4; shift 4
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
# counter sets (one rocprofv3 pass each); PMC_SETS="A B|C D" overrides, PMC_STALL=1 adds the stall / memory-pipe sets
SETS=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
if [ -n "$PMC_STALL" ]; then
  SETS+=("SQ_WAIT_ANY SQ_INST_LEVEL_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" "SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_MISC SQ_IFETCH" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT")
fi
if [ -n "$PMC_SETS" ]; then IFS='|' read -r -a SETS <<< "$PMC_SETS"; fi
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  name=${arm%%=*}
  for set in "${SETS[@]}"; do
    tag=$(echo $set | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $set -d "$R/$OUT/${name}_$tag" -o p --output-format csv -- \
        python3 "$R/tools/kbench.py" $B $G $PREC --libs "$arm" --steps 20 --rounds 1 \
        > "$R/$OUT/${name}_$tag.txt" 2> "$R/$OUT/${name}_$tag.err" || echo "pass $name $tag failed"
  done
  echo "$name done"
done
python3 "$R/tools/pmc_quick_summarise.py" "$R/$OUT" $B $G > "$R/$OUT/summary.txt"
cat "$R/$OUT/summary.txt"
