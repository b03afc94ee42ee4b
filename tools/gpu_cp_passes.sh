#!/bin/bash
# timing experiment (results are NOT the algorithm's for MAXPASS < 60): how much of k_colprox_pair is the secular passes
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier3.pkl; rm -f $F
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null
for MP in 60 1 0; do
OMC_STREAMS=1 OMC_CP_MAXPASS=$MP timeout -k 10 300 python bench.py --extras 0 --pipeline 0 --frontier-file $F > gpurun_out/cp_pass$MP.json
done
python - <<'PY'
import json
for n in ("60","1","0"):
    d=json.loads(open(f"gpurun_out/cp_pass{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print("maxpass", n, round(d["value"],1), c["status_counts"], c["iters_median"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3),k[q]["launches"]) for q in ("colprox","cone","cone_sub","global","small")})
PY
