"""Timeline of a rocprofv3 kernel trace between two marker launches: wall time, union of busy intervals, idle gaps, and the iteration
period split into its phases (k_global end -> first kernel of the next iteration -> last concurrent kernel end -> k_global end).
usage: trace_timeline.py <dir with *_kernel_trace.csv> <marker kernel> <out.txt>"""
import csv, glob, os, sys
import numpy as np
d, marker, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
if len(sys.argv) > 4:      # compact copy of the trace for offline analysis
    import gzip
    with gzip.open(sys.argv[4], 'wt') as fh:
        for r in rows:
            fh.write(f'{r[0]},{r[1]},{r[2]},{r[3]},{r[4]}\n')
mk = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = (mk[0] + 1, mk[-1]) if len(mk) >= 2 else (0, len(rows))
rows = rows[lo:hi]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
for s, e, *_ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = np.array(gaps, float) / 1e3
L = []
L.append(f"launches {len(rows)}  wall {(t1 - t0) / 1e6:.1f} ms  busy-union {busy / 1e6:.1f} ms ({busy / (t1 - t0):.3f})  idle gaps {len(gaps)}: total {gaps.sum() / 1e3:.1f} ms, median {np.median(gaps):.1f} us, p90 {np.percentile(gaps, 90):.1f} us, max {gaps.max():.0f} us")
big = gaps[gaps > 100]
L.append(f"gaps > 100 us: {len(big)} totalling {big.sum() / 1e3:.1f} ms")
# iteration structure: one k_global per iteration
gl = [r for r in rows if r[2].startswith("k_global")]
per = np.diff([g[1] for g in gl]) / 1e3
L.append(f"iterations {len(gl)}  period (k_global end to end): median {np.median(per):.1f} us  mean {per.mean():.1f} us  p90 {np.percentile(per, 90):.1f}")
names = sorted(set(r[2] for r in rows))
for nm in names:
    du = np.array([r[1] - r[0] for r in rows if r[2] == nm], float) / 1e3
    L.append(f"  {nm:40s} n {len(du):6d}  mean {du.mean():8.1f} us  median {np.median(du):8.1f}  p90 {np.percentile(du, 90):8.1f}  max {du.max():8.0f}  total {du.sum() / 1e3:8.1f} ms")
# phases inside a typical iteration: relative to the previous k_global end
ph = {}
gi = 0; ends = [g[1] for g in gl]; starts = [g[0] for g in gl]
import bisect
for s, e, nm, *_ in rows:
    i = bisect.bisect_right(ends, s) - 1          # iteration after global i
    if i < 0 or i + 1 >= len(gl) or nm.startswith("k_global"):
        continue
    if s > starts[i + 1]:
        continue
    a = ph.setdefault(nm, [[], []])
    a[0].append((s - ends[i]) / 1e3); a[1].append((e - ends[i]) / 1e3)
L.append("start / end offsets after the previous k_global end (median us):")
for nm, (a, b) in sorted(ph.items(), key=lambda kv: np.median(kv[1][0])):
    L.append(f"  {nm:40s} start {np.median(a):8.1f}  end {np.median(b):8.1f}  (n {len(a)})")
# time by live-slot count (grid of k_global = live slots of that iteration)
live = np.array([g[4] for g in gl][1:]); edges = [0, 16, 64, 128, 256, 512, 768, 1025]
L.append("iteration period by live slots:")
for a, b in zip(edges[:-1], edges[1:]):
    sel = (live > a) & (live <= b)
    if sel.any():
        L.append(f"  live {a + 1:5d}..{b:5d}: iterations {int(sel.sum()):5d}  total {per[sel].sum() / 1e3:8.1f} ms  mean period {per[sel].mean():8.1f} us  per slot-iteration {per[sel].sum() / live[sel].sum():6.2f} us")
# kernel durations in the saturated regime (their own grid says how many slots they ran over)
L.append("kernel durations at more than 768 live slots (grid / slots-per-workgroup taken from k_global's grid):")
big_t = [(gl[i][1], gl[i + 1][1]) for i in range(len(gl) - 1) if gl[i + 1][4] > 768]
if big_t:
    import bisect as _b
    bs = [a for a, _ in big_t]
    acc = {}
    for s_, e_, nm, _, g_ in rows:
        j = _b.bisect_right(bs, s_) - 1
        if j >= 0 and s_ < big_t[j][1]:
            acc.setdefault(nm, []).append((e_ - s_) / 1e3)
    for nm, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        L.append(f"  {nm:40s} n {len(v):5d}  mean {np.mean(v):8.1f} us  total {sum(v) / 1e3:8.1f} ms  per iteration {sum(v) / len(big_t):8.1f} us")
L.append(f"slot-iterations {int(live.sum())}  mean live {live.mean():.1f}")
st = np.array([(starts[i + 1] - ends[i]) / 1e3 for i in range(len(gl) - 1)])
L.append(f"next k_global start offset: median {np.median(st):.1f} us; its duration median {np.median([(g[1] - g[0]) / 1e3 for g in gl]):.1f} us")
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
