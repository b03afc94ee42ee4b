#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "relaxation_matches_oracle or colprox_pair or determinism or warm" > gpurun_out/t_yx.log 2>&1 || { tail -30 gpurun_out/t_yx.log; exit 1; }
tail -2 gpurun_out/t_yx.log
F=/tmp/frontier5.pkl; rm -f $F
OMC_NO_YX=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null
timeout -k 10 300 python bench.py --extras 0 --pipeline 0 --frontier-file $F > gpurun_out/ab_yx1.json
OMC_NO_YX=1 timeout -k 10 300 python bench.py --extras 0 --pipeline 0 --frontier-file $F > gpurun_out/ab_yx0.json
OMC_STREAMS=1 timeout -k 10 300 python bench.py --extras 0 --pipeline 0 --frontier-file $F > gpurun_out/ab_yx1s.json
OMC_STREAMS=1 OMC_NO_YX=1 timeout -k 10 300 python bench.py --extras 0 --pipeline 0 --frontier-file $F > gpurun_out/ab_yx0s.json
python - <<'PY'
import json
for n in ("yx1","yx0","yx1s","yx0s"):
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1]); c=d["config"]; k=d["roofline"]["kernel_ms"]
    print(n, round(d["value"],1), c["status_counts"], c["iters_median"], c["iters_max"], round(d["ms_per_step"],1), {q:(round(k[q]["avg_launch_ms"],3)) for q in ("colprox","cone","cone_sub","global","small")})
PY
