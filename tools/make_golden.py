"""Generate tests/golden/*.npz: inputs (A, mask, gamma, k, cut lists) and the oracle's outputs.
The reference (Julia + Mosek) cannot run in this environment and ships no fixtures, so these vectors come from the
in-repo oracle (oracle/omc_oracle.py), each certified by its duality gap; they pin oracle and HIP path against
regressions and against each other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import omc_oracle as orc

CASES = [
    dict(name="readme_20x24_k1_linear", n=20, m=24, k=1, kind="readme", seed=11, cut_type="linear", rho_scale=16.0, depth=4),
    dict(name="lowrank_24x28_k1_linear2", n=24, m=28, k=1, kind="lowrank", seed=12, cut_type="linear2", rho_scale=4.0, depth=4),
    dict(name="lowrank_16x20_k2_linear3", n=16, m=20, k=2, kind="lowrank", seed=13, cut_type="linear3", rho_scale=4.0, depth=3),
]

def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for c in CASES:
        A, mask = orc.make_instance(c["n"], c["m"], c["k"], seed=c["seed"], kind=c["kind"],
                                    n_indices=None if c["kind"] == "readme" else int(0.4 * c["n"] * c["m"]))
        inst = orc.Instance(A, mask, 80.0, c["k"])
        rng = np.random.default_rng(c["seed"])
        dirs = orc.child_directions(c["cut_type"], c["k"])
        cuts = []; recs = []
        for d in range(c["depth"] + 1):
            r = orc.sdp_relaxation(inst, cuts, c["cut_type"], params=orc.RelaxParams(rho_scale=c["rho_scale"]))
            x, ev = orc.breakpoint_vector(r["Y"], r["U"])
            recs.append(dict(L=len(cuts), objective=r["objective"], dual_bound=r["dual_bound"], status=r["termination_status"],
                             iters=r["iters"], lmin=ev[0], eval_obj=orc.evaluate_objective(r["X"], A, mask, 80.0)))
            print(c["name"], d, recs[-1])
            # avoid the degenerate inner pieces when the parent's v-hat is ~0 (interval [-a, a] with a ~ 0: no Slater point)
            vhat = r["U"].T @ x
            ok = [d_ for d_ in dirs if all((abs(vhat[j]) > 0.05) or (d_[j] in ("left", "right")) for j in range(c["k"]))]
            dr = ok[int(rng.integers(len(ok)))]
            cuts = cuts + [(x, r["U"].copy(), dr)]
        L = len(cuts)
        np.savez_compressed(os.path.join(out_dir, c["name"] + ".npz"),
                            A=A, mask=mask, gamma=80.0, k=c["k"], cut_type=c["cut_type"], rho_scale=c["rho_scale"],
                            cut_x=np.stack([q[0] for q in cuts]), cut_U=np.stack([q[1] for q in cuts]),
                            cut_dir=np.array([[orc.DIR_CODES[s] for s in q[2]] for q in cuts], dtype=np.int8),
                            node_L=np.array([r["L"] for r in recs]), objective=np.array([r["objective"] for r in recs]),
                            dual_bound=np.array([r["dual_bound"] for r in recs]), status=np.array([r["status"] for r in recs]),
                            iters=np.array([r["iters"] for r in recs]), lmin=np.array([r["lmin"] for r in recs]),
                            eval_obj=np.array([r["eval_obj"] for r in recs]))

def make_shor_golden():
    """Shor-minor fixtures (integer work): mask, X and the oracle's index lists / top violated minors."""
    out_dir = os.path.join(ROOT, "tests", "golden")
    rng = np.random.default_rng(77)
    n, m, k = 9, 13, 2
    mask = rng.random((n, m)) < 0.45
    mask[rng.integers(0, n, m), np.arange(m)] = True; mask[np.arange(n), rng.integers(0, m, n)] = True
    X3 = np.round(rng.standard_normal((k, n, m)) * 4) / 4          # quarter-integers: exact products, many tied scores
    rec = dict(mask=mask, X3=X3, A=rng.standard_normal((n, m)))
    for p in (4, 3, 2, 1, 0):
        rec["idx_%d" % p] = np.array(orc.shor_constraints_indexes(mask, [p]), dtype=np.int64).reshape(-1, 4)
    rec["idx_2_4"] = np.array(orc.shor_constraints_indexes(mask, [2, 4]), dtype=np.int64).reshape(-1, 4)
    first = orc.violated_shor_minors(X3, mask, [4, 3], [], 12)
    existing = np.array([t for _, t in first[:6]], dtype=np.int64)
    second = orc.violated_shor_minors(X3, mask, [4, 3], [tuple(int(v) for v in t) for t in existing], 12)
    rec["existing"] = existing
    for nm, lst in (("first", first), ("second", second)):
        rec[nm + "_scores"] = np.array([s for s, _ in lst]); rec[nm + "_minors"] = np.array([t for _, t in lst], dtype=np.int64)
    np.savez_compressed(os.path.join(out_dir, "shor_9x13_k2.npz"), **rec)
    print("shor fixture:", {k_: v.shape for k_, v in rec.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "shor":
        make_shor_golden()
    else:
        main(); make_shor_golden()
