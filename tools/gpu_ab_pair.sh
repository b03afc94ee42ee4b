#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matches_oracle or determinism or full_size_config2" > gpurun_out/t_pair.log 2>&1 || { tail -30 gpurun_out/t_pair.log; exit 1; }
tail -2 gpurun_out/t_pair.log
F=/tmp/frontier2.pkl
rm -f $F
timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_pair1.json
OMC_NO_COLPROX_PAIR=1 timeout -k 10 300 python bench.py --extras 0 --frontier-file $F > gpurun_out/ab_pair0.json
python - <<'PY'
import json
for n in ("pair1","pair0"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), round(c["nodes_per_s_all"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3),k[q]["launches"]) for q in ("colprox","cone","cone_sub","global","small")})
PY
