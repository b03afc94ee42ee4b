set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cd $R
# the frontier is built (and cached) by an unprofiled run, so that the profiled command launches the kernels of its 1 + 2 steps only and
# the per-kernel averages of rocprofv3 and of the HIP events inside bench.py describe the same launches
python3 bench.py --steps 1 --warmup 0 --extras 0 --frontier-file gpurun_out/r02_frontier.pkl > gpurun_out/r02_bench_unprofiled.json 2> gpurun_out/r02_bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 2 --warmup 1 --extras 0 --frontier-file gpurun_out/r02_frontier.pkl > gpurun_out/r02_bench_prof.json 2> gpurun_out/r02_bench_prof.err
python3 tools/summarize_prof.py stats gpurun_out/r02_stats gpurun_out/r02_kernel_stats.csv
export DEPTH=7 ITERS=400
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/r02_pmc_mfma -- python3 tools/gpu_prof_small.py > gpurun_out/r02_pmc_mfma.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r02_pmc_mfma gpurun_out/r02_mfma_pmc.txt "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU; tools/gpu_prof_small.py (config 2, 128 depth-7 nodes, cap 400 iterations); per-launch averages"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r02_pmc_sq -- python3 tools/gpu_prof_small.py > gpurun_out/r02_pmc_sq.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r02_pmc_sq gpurun_out/r02_sq_wait_lds_pmc.txt "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; same workload; per-launch averages"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_fetch -- python3 tools/gpu_prof_small.py > gpurun_out/r02_pmc_fetch.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r02_pmc_fetch gpurun_out/r02_fetch_pmc.txt "rocprofv3 --kernel-trace --pmc FETCH_SIZE (KB per launch as reported; gfx950 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md); same workload: 128 workgroups / matrices per launch"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_pmc_write -- python3 tools/gpu_prof_small.py > gpurun_out/r02_pmc_write.log 2>&1
python3 tools/summarize_prof.py pmc gpurun_out/r02_pmc_write gpurun_out/r02_write_pmc.txt "rocprofv3 --kernel-trace --pmc WRITE_SIZE (KB per launch); same workload"
rm -rf gpurun_out/r02_frontier.pkl gpurun_out/r02_stats gpurun_out/r02_pmc_mfma gpurun_out/r02_pmc_sq gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
tail -1 gpurun_out/r02_bench_prof.json | cut -c1-300
