#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; cd $R
rm -rf /tmp/tl /tmp/fr.pkl
export OMC_BENCH_MARKERS=1
python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file /tmp/fr.pkl > /dev/null 2>&1
OMC_STREAMS=${STREAMS:-3} OMC_TIMING_STRIDE=${STRIDE:-1} rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 2 --warmup 1 --extras 0 --frontier-file /tmp/fr.pkl > gpurun_out/tl_bench.json 2> gpurun_out/tl_bench.err
python3 tools/trace_timeline.py /tmp/tl k_eval_objective gpurun_out/r03_timeline.txt gpurun_out/tl_trace.csv.gz
