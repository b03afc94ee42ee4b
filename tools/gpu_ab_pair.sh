#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier2.pkl
rm -f $F
OMC_NO_COLPROX_PAIR=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null      # the frontier both runs relax (built by the one-column kernel)
timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_pair1.json
OMC_NO_COLPROX_PAIR=1 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_pair0.json
python - <<'PY'
import json
for n in ("pair1","pair0"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), round(c["nodes_per_s_all"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3),k[q]["launches"]) for q in ("colprox","cone","cone_sub","global","small")})
PY
