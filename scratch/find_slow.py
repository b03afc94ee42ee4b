import sys, os, time, pickle
sys.path.insert(0, "/root/repo/oracle")
import numpy as np, omc_oracle as orc
A, mask = orc.make_instance(20, 24, 1, seed=11, kind="readme")
inst = orc.Instance(A, mask, 80.0, 1)
rng = np.random.default_rng(0)
P = orc.RelaxParams(rho_scale=16.0, max_iters=1500)
found = []
t0 = time.time()
for trial in range(40):
    cuts = []
    for d in range(6):
        r = orc.sdp_relaxation(inst, cuts, "linear", params=P, want_certificate=False)
        if r["termination_status"] == 3: break
        if r["iters"] >= 1500 and len(cuts) > 0:
            gap = (r["objective"] - r["dual_bound"]) / abs(r["objective"])
            found.append((list(cuts), r["iters"], gap, r["rp"], r["rd"]))
            print("SLOW depth", len(cuts), "gap %.1e rp %.1e rd %.1e" % (gap, r["rp"], r["rd"]), "t", time.time() - t0, flush=True)
        x, ev = orc.breakpoint_vector(r["Y"], r["U"])
        dr = orc.child_directions("linear", 1)[int(rng.integers(2))]
        cuts = cuts + [(x, r["U"].copy(), dr)]
    if len(found) >= 4 or time.time() - t0 > 240: break
np.save("/root/repo/scratch/slow_nodes.npy", np.array([dict(cuts=f[0]) for f in found], dtype=object), allow_pickle=True)
print(len(found))
