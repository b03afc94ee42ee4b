#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier.pkl
for G in 4 2 1; do
OMC_SUB_GUARD=$G timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_guard$G.json
done
python - <<'PY'
import json
for n in ("guard4","guard2","guard1"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), round(c["nodes_per_s_all"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3)) for q in ("colprox","cone","cone_sub","global","small")}, d["roofline"]["subspace"])
PY
