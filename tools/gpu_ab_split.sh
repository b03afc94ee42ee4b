#!/bin/bash
# A/B on the GPU box: warm chain vs parents-only, split launch of the full eigen-kernel on/off (bench without extras, frontier cached)
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier.pkl
timeout -k 10 300 python bench.py --extras 0 --frontier-file $F --warm 2 > gpurun_out/ab_warm2_split.json
OMC_NO_WS_SPLIT=1 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F --warm 2 > gpurun_out/ab_warm2_nosplit.json
timeout -k 10 300 python bench.py --extras 0 --frontier-file $F --warm 1 > gpurun_out/ab_warm1_split.json
OMC_NO_WS_SPLIT=1 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F --warm 1 > gpurun_out/ab_warm1_nosplit.json
python - <<'PY'
import json
for n in ("warm2_split","warm2_nosplit","warm1_split","warm1_nosplit"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), round(c["nodes_per_s_all"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:round(k[q]["avg_launch_ms"],3) for q in ("colprox","cone","cone_sub","global","small")})
PY
