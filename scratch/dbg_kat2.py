import sys; sys.path.insert(0, 'oracle')
import numpy as np, omc_oracle as orc
rng = np.random.default_rng(0); n, m, k, g = 12, 15, 2, 80.0
A = rng.standard_normal((n, m)); mask = np.ones((n, m), bool)
inst = orc.Instance(A, mask, g, k)
for kw in [dict(), dict(adapt=0), dict(adapt=0, rho_scale=30.0), dict(rho_scale=30.0)]:
    r = orc.sdp_relaxation(inst, params=orc.RelaxParams(max_iters=600, **kw), want_certificate=False)
    print(kw, "iters", r['iters'], "status", r['termination_status'])
    for h in r['hist'][::3]: print("   it %d obj %.8f lb %.8f rp %.2e rd %.2e rho %.3g" % h)
