"""Copy the last bench lines and the rocprofv3 kernel statistics (rocpd sqlite output) into profiles/ (tracked)."""
import json, sqlite3, csv, sys
tag = sys.argv[1]          # e.g. r01h
for f, dst in ((f"gpurun_out/bench_{tag}.json", "profiles/r01_final_bench_default.json"), (f"gpurun_out/bench_{tag}_serial.json", "profiles/r01_final_bench_default_serial_streams.json")):
    line = open(f).read().strip().splitlines()[-1]
    d = json.loads(line)
    open(dst, "w").write(line + "\n")
    print(f, round(d["value"], 1), round(d["ms_per_step"]), d["config"]["status_counts"], "frac", round(d["roofline"]["frac"], 4), "ach", round(d["roofline"]["achieved"], 3),
          round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["matrices_per_launch"], 1), {k: round(v / d["steps"]) for k, v in d["roofline"]["kernel_ms"].items()},
          (d["cpu_baseline"] or {}).get("value"), d["time_to_gap"]["median_seconds"], d["config"]["iters_median"])
db = sqlite3.connect(f"gpurun_out/prof_{tag}/{tag}_results.db"); cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open("profiles/r01_final_bench_default_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100.0 * r[2] / tot, 3), r[4], r[5]])
rows = list(cur.execute("select name, grid_x, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels where (name like '%k_cone_ws%' and grid_x = 2048*512) or (name like 'k_colprox%' and grid_x = 2048*100/4*256) or (name like '%k_small%' and grid_x=2048*256) or (name like '%k_global%' and grid_x=2048*512) group by name order by 4 desc"))
with open("profiles/r01_final_bench_default_kernel_stats_2048slot_launches.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name (launches over all 2048 slots only: warm-up + timed steps of bench.py; k_cone_ws also serves the certificate checks)", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[2], r[3], round(r[4], 3), r[5], r[6]]); print(r[0][:40], r[2], round(r[4] / 1e3, 1), "us")
