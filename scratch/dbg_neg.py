import sys; sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np, omc_oracle as orc
from test_gpu_parity import oracle_path
n, m, k, kind, cut_type, rho_scale, depth = 24, 30, 1, "lowrank", "linear2", 4.0, 3
A, mask = orc.make_instance(n, m, k, seed=21, kind=kind, n_indices=int(0.35 * n * m))
inst = orc.Instance(A, mask, 80.0, k)
nodes = oracle_path(orc, inst, cut_type, depth, rho_scale, seed=3)
for b, c in enumerate(nodes):
    r = orc.sdp_relaxation(inst, c, cut_type, params=orc.RelaxParams(rho_scale=rho_scale), want_certificate=False)
    print(b, "status", r['termination_status'], "iters", r['iters'], "obj", r['objective'], "lb", r['dual_bound'], "dirs", [q[2] for q in c], "vhat", [float(q[1].T @ q[0]) for q in c])
    if r['objective'] < 0 or r['iters'] > 2000:
        for h in r['hist'][::10]: print("   it %d obj %.6f lb %.6f rp %.2e rd %.2e rho %.3g" % h)
        print("  eig Y", np.linalg.eigvalsh(r['Y'])[[0, 1, -2, -1]], "lam", r['lam'])
