#!/bin/bash
# bench.py over the named workloads, both arithmetic modes (one JSON line each).
#   usage (on the GPU box, repo root): bash tools/round_table.sh <tag>
set -e -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/table_$TAG
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-workloads"
python3 bench.py --workload target --steps 20 --warmup 5 $COMMON > $OUT/target.json 2> $OUT/target.err; echo "target done"
python3 bench.py --workload c2 --steps 512 --warmup 64 $COMMON > $OUT/c2.json 2> $OUT/c2.err; echo "c2 done"
for w in c3 c5; do
  python3 bench.py --workload $w --steps 128 --warmup 16 $COMMON > $OUT/$w.json 2> $OUT/$w.err; echo "$w done"
done
python3 bench.py --workload target --worlds 64 --steps 41 --warmup 7 $COMMON > $OUT/target64.json 2> $OUT/target64.err
python3 bench.py --workload c4 --steps 256 --warmup 64 $COMMON > $OUT/c4.json 2> $OUT/c4.err
python3 bench.py --workload c1 --steps 500 --warmup 10 --precision exact --no-modes $COMMON > $OUT/c1.json 2> $OUT/c1.err
DW_NO_FUSE=1 python3 bench.py --workload c2 --steps 512 --warmup 64 $COMMON > $OUT/c2_nofuse.json 2> $OUT/c2_nofuse.err
OUT=$OUT python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    m = d.get("modes", {})
    other = "; ".join(f"{k}: {v['ms_per_step']:.4f} ms {v['value']:.3e} frac {v['roofline']['frac']:.3f}" for k, v in m.items())
    print(f"{os.path.basename(f):18s} {d['config']['precision'][:5]:5s} {d['ms_per_step']:.4f} ms  {d['value']:.3e}  "
          f"frac {d['roofline']['frac']:.3f} ({d['roofline']['bound']}) | {other}")
PY
