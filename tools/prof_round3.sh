set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cd $R
export OMC_BENCH_MARKERS=1
# the frontier is built (and cached) by an unprofiled run; the profiled command then relaxes the ancestors level by level (untimed: it fills the
# warm-start pool), one warm-up step and two timed steps (two batches handed over as one stream).  The timed steps are bracketed by two k_eval_objective launches (OMC_BENCH_MARKERS) and the
# per-kernel summary is cut to them, so that rocprofv3's averages and the HIP events inside bench.py describe the same launches.
python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file gpurun_out/r03_frontier.pkl > gpurun_out/r03_bench_unprofiled.json 2> gpurun_out/r03_bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_stats -- python3 bench.py --steps 2 --warmup 1 --extras 0 --frontier-file gpurun_out/r03_frontier.pkl > gpurun_out/r03_bench_prof.json 2> gpurun_out/r03_bench_prof.err
python3 tools/summarize_prof.py stats-between gpurun_out/r03_stats gpurun_out/r03_kernel_stats.csv k_eval_objective
python3 tools/summarize_prof.py stats gpurun_out/r03_stats gpurun_out/r03_kernel_stats_whole_command.csv > /dev/null
# HBM traffic of the same command at the bench's own regime (1024 slots, ~0.9 GB of state): separate passes for FETCH_SIZE and WRITE_SIZE
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file gpurun_out/r03_frontier.pkl > gpurun_out/r03_pmc_fetch.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r03_pmc_fetch gpurun_out/r03_fetch_pmc.txt "rocprofv3 --kernel-trace --pmc FETCH_SIZE (KB per launch as reported; gfx950 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md); the bench command itself (2048 warm-started nodes through 1024 slots, priming pass included), per-launch averages"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_pmc_write -- python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file gpurun_out/r03_frontier.pkl > gpurun_out/r03_pmc_write.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r03_pmc_write gpurun_out/r03_write_pmc.txt "rocprofv3 --kernel-trace --pmc WRITE_SIZE (KB per launch); same command"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r03_pmc_sq -- python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file gpurun_out/r03_frontier.pkl > gpurun_out/r03_pmc_sq.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r03_pmc_sq gpurun_out/r03_sq_pmc.txt "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; same command; per-launch averages"
# Shor mode: BASELINE config 3 with its 632 732 class-4 minors, 16 root copies, 300 iterations
unset OMC_BENCH_MARKERS
B=16 ITERS=300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_shor_stats -- python3 tools/gpu_shor_cfg3.py > gpurun_out/r03_shor_cfg3_prof.json 2> gpurun_out/r03_shor_cfg3_prof.err
python3 tools/summarize_prof.py stats gpurun_out/r03_shor_stats gpurun_out/r03_shor_config3_kernel_stats.csv
rm -rf gpurun_out/r03_stats gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_pmc_sq gpurun_out/r03_shor_stats
export OMC_BENCH_MARKERS=1
# timeline of the same command: iteration period by live slots, kernel durations in the saturated regime, streams concurrent and serialised
rm -rf /tmp/tl; OMC_TIMING_STRIDE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 2 --warmup 1 --extras 0 --pipeline 0 --frontier-file gpurun_out/r03_frontier.pkl > /dev/null 2> gpurun_out/r03_tl.err
python3 tools/trace_timeline.py /tmp/tl k_eval_objective gpurun_out/r03_timeline_concurrent.txt > /dev/null
rm -rf /tmp/tl; OMC_STREAMS=1 OMC_TIMING_STRIDE=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 2 --warmup 1 --extras 0 --pipeline 0 --frontier-file gpurun_out/r03_frontier.pkl > /dev/null 2> gpurun_out/r03_tl.err
python3 tools/trace_timeline.py /tmp/tl k_eval_objective gpurun_out/r03_timeline_serial.txt > /dev/null
rm -rf /tmp/tl gpurun_out/r03_frontier.pkl
tail -1 gpurun_out/r03_bench_prof.json | cut -c1-300
