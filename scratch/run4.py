import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
n, m, k, frac, kind = 20, 25, 1, 0.5, "noise"
A, mask = make_instance(n, m, k, frac, 0, kind=kind)
I = Inst(A, mask, 80.0, k)
out = admm(I, rho_f=1000, rho_c=1000, iters=1000, tol=1e-10)
print("root", out['hist'][-1], dual_bound(I, out))
x, w = separation(out['Y'], out['U'])
cuts = [(x, out['U'].copy(), ["left"])]
o2 = admm(I, cuts=cuts, rho_f=1000, rho_c=1000, iters=3000, tol=1e-10)
print(o2['hist'][-1], dual_bound(I, o2))
lam = o2['lam']; Psi = o2['Psi']; rows = o2['rows']
cU = sum(lam[r] * rows[r][1] for r in range(len(rows)))
print("lam", lam, "cU norm", np.linalg.norm(cU), "2Psi12 norm", np.linalg.norm(2 * Psi[:n, n:]), "diff", np.linalg.norm(cU - 2 * Psi[:n, n:]))
print("U", np.linalg.norm(o2['U']), "v", x @ o2['U'], "vhat", x @ out['U'], "trPsi22", Psi[n, n])
print("W1-Z resid; eig D1", np.linalg.eigvalsh(o2['D1'])[[0, -1]])
print([r[3] for r in rows])
Y=o2['Y']; U=o2['U']
for r,(CY,CU,rhs,kind) in enumerate(rows):
    val = (0 if CY is None else (CY*Y).sum()) + (CU*U).sum()
    print(kind, "val", val, "rhs", rhs, "lam", lam[r])
print("LB fixed", dual_bound(I,o2))
R=len(rows); AY=np.zeros((R,n*n)); AU=np.zeros((R,n*k)); b=np.zeros(R)
for r,(CY,CU,rhs,_) in enumerate(rows):
    if CY is not None: AY[r]=CY.ravel()
    AU[r]=CU.ravel(); b[r]=rhs
print("A xi - b", AY@Y.ravel()+AU@U.ravel()-b)
G=(AY/o2['wY'].ravel())@AY.T+(AU/o2['wU'].ravel())@AU.T
print(G)
