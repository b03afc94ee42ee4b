#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier.pkl
export OMC_SUB_GUARD=2
for T in 1 8 0; do
OMC_TIMING_STRIDE=$T timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_stride$T.json
done
OMC_TIMING_STRIDE=0 OMC_NO_WS_SPLIT=1 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_stride0ns.json
OMC_TIMING_STRIDE=0 OMC_GRAPH_MAX=0 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_stride0ng.json
python - <<'PY'
import json
for n in ("stride1","stride8","stride0","stride0ns","stride0ng"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), round(c["nodes_per_s_all"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3),k[q]["launches"]) for q in ("colprox","cone","cone_sub","global","small")})
PY
