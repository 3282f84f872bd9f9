#!/bin/bash
# Evidence of a round's FINAL tree, in one session on the GPU box (repo root):  bash tools/final_evidence.sh <tag> [fuzz seed]
# Writes gpurun_out/final_<tag>/…; copy what is to be judged into profiles/ afterwards (names as in profiles/README.md).
#   bench_default.json            the driver's command, python bench.py --gpus 1 --steps 20 --warmup 5
#   bench_default_profiled.json   the same command under rocprofv3 --kernel-trace --stats (+ its kernel_stats csv)
#   soak_exact.txt                tools/soak_exact.py over five shapes (ring, packed, overlapped strips, odd width)
#   kbench_small_ensembles.txt    tools/kbench.py on ensembles of reference-sized worlds and on small jobs
#   fuzz_all.txt                  the seven fuzzers, one seed
#   dropin_latency.txt, es_fitness_bench.txt
set -o pipefail
TAG=${1:-rXX}
SEED=${2:-53}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_$TAG
mkdir -p "$OUT"
cd "$R"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_default" -o s --output-format csv -- \
    python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default_profiled.json" 2> "$OUT/bench_default_profiled.err" ) || exit 1
cp "$(find "$OUT/prof_default" -name '*kernel_stats.csv' | head -n 1)" "$OUT/bench_default_kernel_stats.csv"
echo "profiled bench done"
: > "$OUT/soak_exact.txt"
for cfg in "256 256 512 4" "16 1024 256 4" "4 4096 64 2" "2048 64 128 4" "3 520 200 4"; do
  timeout -k 10 400 python3 tools/soak_exact.py $cfg >> "$OUT/soak_exact.txt" 2>&1 || exit 1
done
echo "soak done"
: > "$OUT/kbench_small_ensembles.txt"
for cfg in "4096 64" "65536 16" "2048 128" "4096 96" "32 256" "256 256"; do
  python3 tools/kbench.py $cfg fast,exact --steps 200 --rounds 7 2>&1 | grep -v amdgpu.ids >> "$OUT/kbench_small_ensembles.txt" || exit 1
done
echo "kbench done"
bash tools/fuzz_all.sh $SEED > "$OUT/fuzz_all.txt" 2>&1 || { cat "$OUT/fuzz_all.txt"; exit 1; }
echo "fuzz done"
python3 tools/dropin_latency.py > "$OUT/dropin_latency.txt" 2>&1 || exit 1
python3 tools/es_fitness_bench.py 64 32 16 768 64 > "$OUT/es_fitness_bench.txt" 2>&1 || exit 1
echo "all done"
