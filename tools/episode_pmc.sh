#!/bin/bash
# rocprofv3 evidence for the small-world episode kernels (episode_wave: bench.py's c4_dim8 workload, 1000 worlds of 8x8 with 4
# greedy agents; episode_mlp_wave: tools/es_fitness_bench.py, 2048 worlds of 16x16 with 4 MLP agents): kernel statistics and SQ
# counters, every counter set its own pass with --kernel-trace only, the program itself after `--`.
#   usage (GPU box, repo root):  bash tools/episode_pmc.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/episode_pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C4="--workload c4_dim8 --steps 512 --warmup 64 --no-cpu-baseline --no-modes --preheat-s 0.5"
for prec in exact fast; do
  rocprofv3 --kernel-trace --stats -d "$OUT/c4dim8_stats_$prec" -o s --output-format csv -- \
      python3 "$R/bench.py" $C4 --precision $prec > "$OUT/c4dim8_stats_$prec.json" 2> "$OUT/c4dim8_stats_$prec.err"
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    name=$(echo $set | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $set -d "$OUT/c4dim8_pmc_${prec}_$name" -o p --output-format csv -- \
        python3 "$R/bench.py" $C4 --precision $prec > /dev/null 2> "$OUT/c4dim8_pmc_${prec}_$name.err"
  done
  echo "c4_dim8 $prec done"
done
rocprofv3 --kernel-trace --stats -d "$OUT/es_stats" -o s --output-format csv -- \
    python3 "$R/tools/es_fitness_bench.py" 64 32 16 768 64 > "$OUT/es_stats.txt" 2> "$OUT/es_stats.err"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/es_pmc_$name" -o p --output-format csv -- \
      python3 "$R/tools/es_fitness_bench.py" 64 32 16 768 64 > /dev/null 2> "$OUT/es_pmc_$name.err"
done
echo "es done"
python3 "$R/tools/episode_pmc_summarise.py" "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
