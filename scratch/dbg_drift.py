import sys; sys.path.insert(0, 'oracle')
import numpy as np, omc_oracle as orc
n, m, k = 20, 25, 1
A, mask = orc.make_instance(n, m, k, seed=1, kind="readme")
inst = orc.Instance(A, mask, 80.0, k)
cuts = []
dirs_all = orc.child_directions("linear", k)
for d in range(2):
    r = orc.sdp_relaxation(inst, cuts, "linear", want_certificate=False)
    x, ev = orc.breakpoint_vector(r["Y"], r["U"])
    cuts = cuts + [(x, r["U"].copy(), dirs_all[d % 2])]
    print("cut", d, "vhat", r["U"].T @ x, dirs_all[d % 2], "|U|", np.linalg.norm(r["U"]))
p = orc.RelaxParams(max_iters=1000, relax=1.0)
r1 = orc.sdp_relaxation(inst, cuts, "linear", params=p, want_certificate=False)
p2 = orc.RelaxParams(max_iters=50, relax=1.0)
r2 = orc.sdp_relaxation(inst, cuts, "linear", params=p2, warm=r1['warm'], want_certificate=False)
dY = r2['Y'] - r1['Y']; dU = r2['U'] - r1['U']
print("dY", np.linalg.norm(dY), "dU", np.linalg.norm(dU), "obj", r1['objective'], r2['objective'])
X = np.stack([c[0] for c in cuts], 1)
print("dU along cuts", X.T @ dU[:, 0], "U along cuts", X.T @ r2['U'][:, 0], "|U|", np.linalg.norm(r2['U']))
w, V = np.linalg.eigh(r2['Y']); print("eig Y top", w[-5:], "tr", w.sum())
print("dY eig", np.linalg.eigvalsh(dY)[[0, 1, -2, -1]])
print("x'Yx", [x_ @ r2['Y'] @ x_ for x_ in X.T], "lam", r2['lam'])
S = r2['Y'] - r2['U'] @ r2['U'].T; print("eig Y-uu'", np.linalg.eigvalsh(S)[:3])
