#!/bin/bash
# The README's lifespan sweep (ref README.md:59-84; notebooks/greedy_longevity_abatement.ipynb cell 2) on one GPU:
# 1000 worlds of 8x8, 4 agents, the five policies, both albedo settings, the reference's seeding (seed 13, host
# legacy-RNG reset).  Writes gpurun_out/<tag>/sweep_{default,neutral}.txt and profiles/<tag>_readme_sweep.json.
#   usage (GPU box, repo root): bash tools/readme_sweep.sh r03
set -e -o pipefail
TAG=${1:-rXX}
mkdir -p gpurun_out/$TAG
for alb in default neutral; do
  python3 tools/lifespan_sweep.py --worlds 1000 --dim 8 --albedos $alb > gpurun_out/$TAG/sweep_$alb.txt 2>&1
done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
ref = {  # /root/reference/README.md:59-80, mean +/- standard error over 1000 simulations
    "default": {"greedy": (382.983, 0.722, 199.024, 5.319), "antigreedy": (447.099, 0.266, 359.436, 4.228),
                "random": (416.836, 0.259, 408.401, 0.583), "half_random": (376.665, 0.372, 380.428, 0.515),
                "no_agents_act": (489.000, 0.000, None, None)},
    # the README shows the neutral-albedo setting only as a figure (assets/neutral_biosphere_longevity.png): no numbers
    "neutral": {k: (None, None, None, None) for k in ("greedy", "antigreedy", "random", "half_random", "no_agents_act")}}
out = {"sweep": "README lifespan sweep, 1000 worlds of 8x8, 4 agents, seed 13 (tools/readme_sweep.sh)",
       "reference": "README.md:59-80 of riveSunder/therldaisyworld (numbers quoted there, other RNG state / numpy version)",
       "settings": {}}
for alb in ("default", "neutral"):
    line = [ln for ln in open(f"gpurun_out/{tag}/sweep_{alb}.txt") if ln.startswith("{")][-1]
    d = json.loads(line)
    rows = []
    for r in d["rows"]:
        b, bs, a, as_ = ref[alb][r["policy"]]
        n = r["worlds"] ** 0.5
        rows.append({"policy": r["policy"], "biosphere_lifespan_mean": r["biosphere_lifespan_mean"],
                     "biosphere_lifespan_se": r["biosphere_lifespan_std"] / n,
                     "agent_lifespan_mean": r["agent_lifespan_mean"], "agent_lifespan_se": r["agent_lifespan_std"] / (n * 2),
                     "readme_biosphere": b, "readme_biosphere_se": bs, "readme_agent": a, "readme_agent_se": as_,
                     "steps_run": r["steps_run"], "wall_s": r["wall_s"]})
    out["settings"][alb] = {"precision": d["precision"], "rows": rows}
json.dump(out, open(f"profiles/{tag}_readme_sweep.json", "w"), indent=1)
for alb, s in out["settings"].items():
    for r in s["rows"]:
        print(f"{alb:8s} {r['policy']:14s} biosphere {r['biosphere_lifespan_mean']:8.3f} (README {r['readme_biosphere']})  "
              f"agents {r['agent_lifespan_mean']:8.3f} (README {r['readme_agent']})  {r['wall_s']:.3f} s")
PY
