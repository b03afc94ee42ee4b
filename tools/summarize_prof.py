"""Summaries of rocprofv3 CSV output for profiles/ (tracked):
   summarize_prof.py stats <dir> <out.csv>        kernel-trace: calls / total / average / min / max per kernel
   summarize_prof.py pmc <dir> <out.txt> <title>  counter collection: per-launch averages per kernel and counter"""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    return name.replace("(OmcWS)", "").replace("(OmcWS, int)", "").replace("void ", "").strip()

mode, d, out = sys.argv[1], sys.argv[2], sys.argv[3]
if mode in ("stats", "stats-between"):
    # stats-between <marker>: only the launches between the first and the second launch of the marker kernel (bench.py with
    # OMC_BENCH_MARKERS=1 brackets its timed steps with k_eval_objective), so that the averages describe the launches the HIP events time
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0, 0, 10**18, 0])
    rows_ = list(csv.DictReader(open(f)))
    if mode == "stats-between":
        marker = sys.argv[4]
        rows_.sort(key=lambda r: int(r["Start_Timestamp"]))
        marks = [int(r["Start_Timestamp"]) for r in rows_ if marker in r["Kernel_Name"]]
        assert len(marks) >= 2, "marker kernel launched fewer than twice"
        rows_ = [r for r in rows_ if marks[0] < int(r["Start_Timestamp"]) < marks[1]]
    for r in rows_:
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a = acc[r["Kernel_Name"]]
        a[0] += 1; a[1] += t; a[2] = min(a[2], t); a[3] = max(a[3], t)
    tot = sum(a[1] for a in acc.values())
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k_, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k_, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), a[2], a[3]])
            if a[1] > 0.005 * tot: print("%-60s calls %7d avg %9.1f us  %5.1f %%" % (short(k_)[:60], a[0], a[1] / a[0] / 1e3, 100.0 * a[1] / tot))
else:
    title = sys.argv[4] if len(sys.argv) > 4 else ""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0])); grid = {}; wgsum = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); grid[r["Kernel_Name"]] = r["Grid_Size"]
        wgsum[r["Kernel_Name"]][r["Counter_Name"]] += int(r["Grid_Size"]) // max(1, int(r.get("Workgroup_Size", 1) or 1))      # workgroups launched: one per live slot for the per-slot kernels
    with open(out, "w") as fo:
        fo.write(title + "\n")
        for k_, cs in sorted(acc.items()):
            if k_.startswith("__amd") or not any(v[1] for v in cs.values()): continue
            n_ = max(v[0] for v in cs.values())
            line = "%s launches=%d last_grid=%s workgroups_total=%d " % (short(k_), n_, grid[k_], max(wgsum[k_].values())) + " ".join("%s=%.4g" % (c, v[1] / v[0]) for c, v in sorted(cs.items()))
            fo.write(line + "\n"); print(line)
