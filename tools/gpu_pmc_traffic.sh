#!/bin/bash
# HBM traffic of the bench command at its own regime (1024 slots), per unit of work: separate FETCH_SIZE / WRITE_SIZE passes (never with other trace domains),
# FETCH x2 (gfx950 under-reports wide coalesced reads, MI355X_MICROARCH.md); plus the two timelines (streams concurrent / serialised) cut to the timed steps
set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; cd $R
F=/tmp/r03_frontier.pkl; rm -rf $F /tmp/pf /tmp/pw /tmp/tl
python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > gpurun_out/r03_pmc_fetch.log 2>&1
python3 tools/summarize_prof.py pmc /tmp/pf gpurun_out/r03_fetch_pmc.txt "rocprofv3 --kernel-trace --pmc FETCH_SIZE (KB per launch as reported; gfx950 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md); the bench command itself (2048 warm-started nodes through 1024 slots, ancestor relaxations included), per-launch averages; workgroups_total = workgroups over all launches (one per live slot for the per-slot kernels)" > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file $F > gpurun_out/r03_pmc_write.log 2>&1
python3 tools/summarize_prof.py pmc /tmp/pw gpurun_out/r03_write_pmc.txt "rocprofv3 --kernel-trace --pmc WRITE_SIZE (KB per launch); same command" > /dev/null
python3 - <<'PY'
import re, json
def parse(f):
    out = {}
    for line in open(f).read().splitlines()[1:]:
        m = re.match(r"(.*?) launches=(\d+) last_grid=\d+ workgroups_total=(\d+) (\w+)=([\d.e+]+)", line)
        if m: out[m.group(1)] = (int(m.group(2)), int(m.group(3)), float(m.group(5)))
    return out
fe, wr = parse("gpurun_out/r03_fetch_pmc.txt"), parse("gpurun_out/r03_write_pmc.txt")
res = {"regime": "the bench command itself: 2048 warm-started config-2 nodes streamed through 1024 slots (~0.9 GB of state: beyond the 256 MiB Infinity Cache), ancestor relaxations included; separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE; FETCH x2 (gfx950 under-reports wide coalesced reads, MI355X_MICROARCH.md); per unit = per workgroup (one per live slot and iteration)"}
alg = {"k_cone_sub<0>": 188672.0, "k_global<true>": 960000.0, "k_small<true>": None, "k_colprox_pair": None}
for k in alg:
    if k in fe and k in wr:
        n, wg, f = fe[k]; _, wg2, w_ = wr[k]
        b = (2.0 * f * n / wg + w_ * wr[k][0] / wg2) * 1024.0
        res[k] = {"fetch_KB_per_launch_reported": f, "write_KB_per_launch": w_, "launches": n, "workgroups_total": wg, "hbm_bytes_per_unit": b, "algorithmic_bytes_per_unit": alg[k], "ratio": (b / alg[k]) if alg[k] else None}
c = res.get("k_cone_sub<0>", {})
res["bytes_per_projection"] = c.get("hbm_bytes_per_unit"); res["algorithmic_bytes_per_projection"] = 188672.0; res["ratio"] = c.get("ratio")
res["note"] = "k_cone_sub: the power-step products of a call re-read M from L2 / HBM once the slots no longer fit the Infinity Cache; k_colprox_pair's unit is a workgroup of four waves = eight columns"
json.dump(res, open("gpurun_out/r03_cone_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
export OMC_BENCH_MARKERS=1
OMC_TIMING_STRIDE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 2 --warmup 1 --extras 0 --pipeline 0 --frontier-file $F > /dev/null 2> gpurun_out/r03_tl.err
python3 tools/trace_timeline.py /tmp/tl k_eval_objective gpurun_out/r03_timeline_concurrent.txt > /dev/null
rm -rf /tmp/tl; OMC_STREAMS=1 OMC_TIMING_STRIDE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 2 --warmup 1 --extras 0 --pipeline 0 --frontier-file $F > /dev/null 2> gpurun_out/r03_tl.err
python3 tools/trace_timeline.py /tmp/tl k_eval_objective gpurun_out/r03_timeline_serial.txt > /dev/null
rm -rf /tmp/tl /tmp/pf /tmp/pw $F
head -3 gpurun_out/r03_timeline_concurrent.txt
