import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *
# KAT-2 check: fully observed root
n, m, k, g = 12, 15, 2, 80.0
rng = np.random.default_rng(0)
A = rng.standard_normal((n, m)); mask = np.ones((n, m), bool)
I = Inst(A, mask, g, k)
sv = np.linalg.svd(A, compute_uv=False)
# waterfill: min sum .5 s^2/(1+g y) 0<=y<=1 sum y<=k -> with sv distinct y=(1,1,0..)?
from scipy.optimize import minimize
obj = lambda y: (0.5 * sv**2 / (1 + g * y)).sum()
res = minimize(obj, np.full(n, k / n), bounds=[(0, 1)] * n, constraints=[dict(type='ineq', fun=lambda y: k - y.sum())], method='SLSQP', options=dict(ftol=1e-15, maxiter=500))
print("waterfill", res.fun, res.x.round(4))
t = time.time(); out = admm(I, rho_f=1.0, rho_c=1.0, iters=3000, tol=1e-10, verbose=True, fstar=res.fun); print(time.time() - t, out['iters'])
