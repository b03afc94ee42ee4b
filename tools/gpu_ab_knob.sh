#!/bin/bash
# A/B of one tuning knob on a common frontier: KNOB=name VALUES="a b c" (first value = reference)
set -e
cd "$GRAFT_REPO_ROOT"
F=/tmp/frontier_knob.pkl; rm -f $F
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null 2>&1
for V in $VALUES; do
env $KNOB=$V timeout -k 10 300 python bench.py --extras 0 --steps ${STEPS:-2} --frontier-file $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['roofline']['kernel_ms']; sb=d['roofline']['subspace']
print('$KNOB=$V', round(d['value'],1), round(c['nodes_per_s_all'],1), c['status_counts'], c['iters_median'], c['iters_max'], round(d['ms_per_step'],1), {q:round(k[q]['avg_launch_ms'],3) for q in ('colprox','cone','cone_sub','global','small')}, 'steps/call', round(sb['power_steps']/max(1,sb['calls']),2), 'fallbacks', sb['fallbacks'])"
done
