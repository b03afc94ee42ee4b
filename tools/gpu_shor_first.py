"""First GPU run of the Shor-mode relaxation against the oracle (development probe; the parity tests live in tests/test_gpu_shor.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_amd
import omc_oracle as orc, omc_oracle_shor as sh

def case(n, m, nidx, seed, noise, classes, depth=0, eps=1e-5, kind="lowrank", max_iters=6000):
    A, mask = orc.make_instance(n, m, 1, n_indices=nidx, seed=seed, noise=noise, kind=kind)
    inst = orc.Instance(A, mask, 80.0, 1)
    minors, soc = sh.driver_shor_lists(mask, classes)
    cuts = []
    for d in range(depth):
        r0 = orc.sdp_relaxation(inst, cuts=cuts)
        x, _ = orc.breakpoint_vector(r0["Y"], r0["U"])
        cuts = cuts + [(x, r0["U"], ["left" if d % 2 == 0 else "right"])]
    t0 = time.time()
    ro = sh.sdp_relaxation_shor(inst, minors, soc, cuts=cuts, params=sh.ShorParams(eps_gap=eps, max_iters=max_iters))
    to = time.time() - t0
    eng = omc_amd.Engine(A, mask, 80.0, 1)
    p = omc_amd.default_params(eps_gap=eps, max_iters=max_iters)
    t0 = time.time()
    rg = eng.matrix_completion_SDP_relaxation([cuts], "linear", p, add_Shor_valid_inequalities=True, shor_info=[(minors, None)], want_Theta=True)[0]
    tg = time.time() - t0
    print(f"{n}x{m} nq={len(minors)} depth={depth}: oracle obj {ro['objective']:.9f} lb {ro['dual_bound']:.9f} it {ro['iters']} st {ro['termination_status']} ({to:.1f}s) | "
          f"gpu obj {rg['objective']:.9f} lb {rg['dual_bound']:.9f} it {rg['iters']} st {rg['status_code']} ({tg:.2f}s) | rel diff {abs(rg['objective']-ro['objective'])/abs(ro['objective']):.2e}", flush=True)
    st = ro["structure"]
    X, W, Th, Y, U = rg["X"], rg["W"], rg["Theta"], rg["Y"], rg["U"]
    ref = orc.compute_SDP_relaxation_objective(X, Th, A, mask, 80.0, W=W)
    big = np.block([[Y, X], [X.T, Th]])
    print(f"   reference formula on GPU point {ref:.9f}; theta_diag {np.abs(np.diag(Th) - W.sum(0)).max():.1e} Wmin {W.min():.1e} soc {max(0, (X*X-W)[st.soc_list].max()) if st.soc_list.any() else 0:.1e} psd {max(0,-np.linalg.eigvalsh(0.5*(big+big.T))[0]):.1e}")
    print("   kernels:", {k_: round(v["ms"], 1) for k_, v in eng.kernel_stats().items() if v["launches"]})
    eng.close()

case(10, 12, 60, 1, 0.3, [])
case(12, 14, 70, 2, 0.1, [4])
case(12, 14, 70, 2, 0.1, [4], depth=2)
case(20, 24, 150, 5, 0.1, [4])
