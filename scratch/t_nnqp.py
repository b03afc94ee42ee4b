import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
rng = np.random.default_rng(1)
for sc in [1.0, 1e-3]:
    Ar = rng.standard_normal((5, 30)) * np.sqrt(sc); G = Ar @ Ar.T; c = rng.standard_normal(5) * sc
    lam = nnqp(G, c, None)
    grad = G @ lam - c
    print("sc", sc, "lam", lam, "grad", grad, "kkt viol", np.minimum(grad, 0).min(), (lam * grad))
# singular test: parallel rows
Ar = rng.standard_normal((3, 10)); Ar = np.vstack([Ar, -Ar[1]]); G = Ar @ Ar.T
p = rng.standard_normal(10); b = np.array([0.1, -0.5, 0.2, 1.0]); c = Ar @ p - b
lam = nnqp(G, c); grad = G @ lam - c
print(lam, grad, lam * grad)
