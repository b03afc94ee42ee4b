#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "relaxation_matches_oracle or colprox_pair or determinism or order_200 or n150" > gpurun_out/t_wide.log 2>&1 || { tail -30 gpurun_out/t_wide.log; exit 1; }
tail -2 gpurun_out/t_wide.log
F=/tmp/frontier_c3.pkl; rm -f $F
OMC_NO_COLPROX_WIDE=1 timeout -k 10 600 python bench.py --config 3 --depth 9 --slots 256 --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null 2>&1
for V in 0 1; do
if [ $V = 1 ]; then export OMC_NO_COLPROX_WIDE=1; else unset OMC_NO_COLPROX_WIDE; fi
timeout -k 10 600 python bench.py --config 3 --depth 9 --slots 256 --steps 1 --warmup 0 --extras 0 --frontier-file $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['roofline']['kernel_ms']
print('no_wide=$V', round(d['value'],1), round(c['nodes_per_s_all'],1), c['status_counts'], c['iters_median'], c['iters_max'], round(d['ms_per_step'],1), {q:round(k[q]['avg_launch_ms'],3) for q in ('colprox','cone','cone_sub','global','small')})"
done
unset OMC_NO_COLPROX_WIDE
timeout -k 10 300 python bench.py --extras 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['roofline']['kernel_ms']
print('config2', round(d['value'],1), c['status_counts'], round(d['ms_per_step'],1), {q:round(k[q]['avg_launch_ms'],3) for q in ('colprox','cone','cone_sub','global','small')})"
