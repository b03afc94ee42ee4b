"""Prototype of the reduced relaxation + ADMM (scratch, numpy)."""
import numpy as np, time
from scipy.optimize import nnls

def make_instance(n, m, k, frac, seed, noise=0.01, kind="lowrank"):
    rng = np.random.default_rng(seed)
    if kind == "lowrank":
        L = rng.standard_normal((n, k)); R = rng.standard_normal((k, m)); E = rng.standard_normal((n, m))
        A = L @ R + noise * E
        nidx = int(round(frac * n * m))
        for _ in range(101):
            perm = rng.permutation(n * m)[:nidx]
            mask = np.zeros(n * m, bool); mask[perm] = True
            mask = mask.reshape((n, m), order="F")
            if mask.any(0).all() and mask.any(1).all(): break
    else:
        A = rng.standard_normal((n, m)); mask = rng.integers(0, 2, (n, m)).astype(bool)
    return A, mask

class Inst:
    def __init__(s, A, mask, gamma, k):
        s.A, s.mask, s.g, s.k = A, mask, gamma, k
        s.n, s.m = A.shape
        s.cols = [np.flatnonzero(mask[:, j]) for j in range(s.m)]
        s.a = [A[s.cols[j], j] for j in range(s.m)]
        Mf = mask.astype(float)
        s.N = Mf @ Mf.T  # multiplicity of (i,i') among column blocks

def f_and_grad(I, Y):
    g = I.g; val = 0.0; G = np.zeros_like(Y); al = []
    for j in range(I.m):
        o = I.cols[j]
        if len(o) == 0: al.append(np.zeros(0)); continue
        B = np.eye(len(o)) + g * Y[np.ix_(o, o)]
        a = np.linalg.solve(B, I.a[j]); al.append(a)
        val += 0.5 * I.a[j] @ a
        G[np.ix_(o, o)] -= 0.5 * g * np.outer(a, a)
    return val, G, al

def prox_col(B, a, cp, s0=None):
    """solve s = ||(B + cp*s I)^-1 a||^2 ; return alpha."""
    b, Q = np.linalg.eigh(B); qa2 = (Q.T @ a) ** 2
    smin = max(0.0, -b.min() / cp) if b.min() <= 0 else 0.0
    lo = smin; hi = max(smin * 2 + 1.0, 1.0)
    phi = lambda s: (qa2 / (b + cp * s) ** 2).sum() - s
    while phi(hi) > 0: hi *= 2
    # safeguarded newton
    s = hi if s0 is None or not (lo < s0 < hi) else s0
    for it in range(100):
        d = b + cp * s
        ph = (qa2 / d ** 2).sum() - s
        dph = -2 * cp * (qa2 / d ** 3).sum() - 1
        if ph > 0: lo = s
        else: hi = s
        sn = s - ph / dph
        if not (lo < sn < hi): sn = 0.5 * (lo + hi)
        if abs(sn - s) <= 1e-15 * max(1, abs(s)): s = sn; break
        s = sn
    alpha = Q @ ((Q.T @ a) / (b + cp * s))
    return alpha, s

def cut_coeffs(ctype, d, vhat, q1=True):
    """return lo, hi, alpha, beta for piece d of cut type."""
    a = abs(vhat)
    if ctype == "linear":
        if d == "left": return -1.0, vhat, vhat - 1.0, vhat
        if d == "right": return vhat, 1.0, 1.0 + vhat, -vhat
    if ctype == "linear2":
        if d == "left": return -1.0, -a, -(1 + a), -a
        if d == "middle": return -a, a, 0.0, vhat ** 2
        if d == "right": return a, 1.0, 1 + a, -a
    if ctype == "linear3":
        if d == "left": return -1.0, -a, -(1 + a), -a
        if d == "inner_left": return -a, 0.0, -a, 0.0
        if d == "inner_right": return 0.0, a, a, 0.0
        if d == "right": return (a, 1.0, a, 0.0) if q1 else (a, 1.0, 1 + a, -a)
    raise ValueError

def nnqp(G, c, nonneg=None, lam0=None, tol=1e-13):
    """min 1/2 l'Gl - c'l s.t. l>=0, G psd possibly singular. Lawson-Hanson active set in QP form."""
    R = len(c); lam = np.zeros(R); P = np.zeros(R, bool)
    scale = max(np.abs(c).max(), 1e-300)
    for outer in range(10 * R + 10):
        w = c - G @ lam
        w[P] = -np.inf
        t = int(np.argmax(w))
        if w[t] <= tol * scale: break
        P[t] = True
        for inner in range(10 * R + 10):
            idx = np.flatnonzero(P)
            s = np.zeros(R)
            s[idx] = np.linalg.lstsq(G[np.ix_(idx, idx)], c[idx], rcond=None)[0]
            if s[idx].min() > 0: lam = s; break
            neg = idx[s[idx] <= 0]
            al = np.min(lam[neg] / (lam[neg] - s[neg]))
            lam = lam + al * (s - lam)
            P &= lam > 1e-300 * 0 + 0  # keep
            P[neg[np.isclose(lam[neg], 0, atol=1e-18 * scale) | (lam[neg] <= 0)]] = False
            lam[~P] = 0
    return lam

def build_rows(I, cuts, ctype, q1=True):
    """rows: list of (CY (n x n sym or None), CU (n x k), rhs). all '<='."""
    n, k = I.n, I.k
    rows = []
    rows.append((np.eye(n), np.zeros((n, k)), float(k), "trace"))
    for i in range(k):
        for r in range(n - k + i, n):
            CU = np.zeros((n, k)); CU[r, i] = -1.0
            rows.append((None, CU, 0.0, "sign"))
    for (x, Uh, dirs) in cuts:
        vh = Uh.T @ x
        CUc = np.zeros((n, k)); rhs = 0.0
        for j in range(k):
            lo, hi, al, be = cut_coeffs(ctype, dirs[j], vh[j], q1)
            CUc[:, j] = -al * x; rhs += be
            CU = np.zeros((n, k)); CU[:, j] = x; rows.append((None, CU, hi, "hi"))
            CU = np.zeros((n, k)); CU[:, j] = -x; rows.append((None, CU, -lo, "lo"))
        rows.append((np.outer(x, x), CUc, rhs, "cut"))
    return rows

def psd_proj(M):
    w, V = np.linalg.eigh(M)
    wp = np.maximum(w, 0)
    return (V * wp) @ V.T, w, V

def admm(I, cuts=(), ctype="linear", rho_f=1.0, rho_c=1.0, iters=2000, tol=1e-9, verbose=False, q1=True, Y0=None, U0=None, fstar=None):
    n, m, k, g = I.n, I.m, I.k, I.g
    rows = build_rows(I, cuts, ctype, q1)
    R = len(rows)
    wY = rho_f * I.N + rho_c * (2 if k > 1 else 1)
    wU = 2 * rho_c * np.ones((n, k))
    # row matrices
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); b = np.zeros(R)
    for r, (CY, CU, rhs, _) in enumerate(rows):
        if CY is not None: AY[r] = CY.ravel()
        AU[r] = CU.ravel(); b[r] = rhs
    G = (AY / wY.ravel()) @ AY.T + (AU / wU.ravel()) @ AU.T
    Y = np.eye(n) * (k / n) if Y0 is None else Y0.copy()
    U = np.zeros((n, k)) if U0 is None else U0.copy()
    Yp = Y.copy()
    alpha = [np.zeros(len(I.cols[j])) for j in range(m)]
    svals = [None] * m
    D1 = np.zeros((n + k, n + k)); D2 = np.zeros((n, n))
    cp = g * g / (2 * rho_f)
    hist = []
    lam = np.zeros(R)
    for it in range(iters):
        # F block
        LL = np.zeros((n, n))
        Yx = 2 * Y - Yp
        for j in range(m):
            o = I.cols[j]
            if len(o) == 0: continue
            Zj = Yx[np.ix_(o, o)] - (g / (2 * rho_f)) * np.outer(alpha[j], alpha[j])
            B = np.eye(len(o)) + g * Zj
            alpha[j], svals[j] = prox_col(B, I.a[j], cp, svals[j])
            LL[np.ix_(o, o)] += np.outer(alpha[j], alpha[j])
        # cone block 1
        Zg = np.block([[Y, U], [U.T, np.eye(k)]])
        W1, ev1, _ = psd_proj(Zg - D1)
        H1 = W1 + D1
        tY = rho_f * (I.N * Y + (g / (2 * rho_f)) * LL) + rho_c * H1[:n, :n]
        if k > 1:
            w2, V2 = np.linalg.eigh(Y - D2)
            W2 = (V2 * np.minimum(w2, 1.0)) @ V2.T
            H2 = W2 + D2
            tY += rho_c * H2
        tY /= wY
        tU = H1[:n, n:]
        # global projection onto P
        c = AY @ tY.ravel() + AU @ tU.ravel() - b
        lam = nnqp(G, c, None)
        Yn = tY - ((AY.T @ lam) / wY.ravel()).reshape(n, n)
        Un = tU - ((AU.T @ lam) / wU.ravel()).reshape(n, k)
        # duals
        Zn = np.block([[Yn, Un], [Un.T, np.eye(k)]])
        D1 = D1 + W1 - Zn
        if k > 1: D2 = D2 + W2 - Yn
        rprim = np.sqrt(np.linalg.norm(W1 - Zn) ** 2)
        rdual = np.linalg.norm(Yn - Y)
        Yp = Y; Y = Yn; U = Un
        if it % 10 == 0 or it == iters - 1:
            fv, _, _ = f_and_grad(I, Y)
            hist.append((it, fv, rprim, rdual))
            if verbose and it % 50 == 0:
                print(it, "f=%.10f" % fv, "rp=%.2e rd=%.2e" % (rprim, rdual), "" if fstar is None else "err=%.2e" % (abs(fv - fstar) / abs(fstar)))
            if max(rprim, rdual) < tol: break
    Psi = psd_proj(rho_c * D1)[0]
    Psi2 = psd_proj(-rho_c * D2)[0] if k > 1 else None
    return dict(Y=Y, U=U, alpha=alpha, D1=D1, D2=D2, lam=lam, rows=rows, hist=hist, wY=wY, wU=wU, iters=it + 1, Psi=Psi, Psi2=Psi2)

def dual_bound(I, out):
    """certified lower bound from ADMM multipliers (see notes)."""
    n, m, k, g = I.n, I.m, I.k, I.g
    # Lambda from alpha
    c0 = 0.0; LL = np.zeros((n, n))
    for j in range(m):
        o = I.cols[j]; a = out['alpha'][j]
        if len(o) == 0: continue
        c0 += I.a[j] @ a - 0.5 * a @ a
        LL[np.ix_(o, o)] += np.outer(a, a)
    Gm = -0.5 * g * LL
    rows = out['rows']; lam = out['lam']
    M = Gm.copy(); cU = np.zeros((n, k)); const = 0.0
    for r, (CY, CU, rhs, kind) in enumerate(rows):
        if kind == "trace": continue  # handled by the simple set
        if CY is not None: M += lam[r] * CY
        cU += lam[r] * CU; const -= lam[r] * rhs
    # Psi from cone dual: scaled dual D1 -> y = rho_c * D1 ; Psi = PSD part of (-rho D1)
    Psi = out['Psi']
    M -= Psi[:n, :n]; cU -= 2 * Psi[:n, n:]; const -= np.trace(Psi[n:, n:])
    if k > 1:
        Ph2 = out['Psi2']  # multiplier for I - Y >= 0 : -<Ph2, I - Y> = -tr Ph2 + <Ph2,Y>
        M += Ph2; const -= np.trace(Ph2)
    ev = np.linalg.eigvalsh(M)
    # simple set for Y: 0<=Y<=I (if k>1 we dualized Y<=I; keep it anyway, valid), trY<=k
    lb = c0 + np.minimum(ev[:k], 0).sum() - np.linalg.norm(cU, axis=0).sum() + const
    return lb

def separation(Y, U, nev=1):
    S = U @ U.T - Y
    w, V = np.linalg.eigh(S)
    if nev == 1 or not (w[1] < -1e-10):
        return V[:, 0].copy(), w[:2]
    wt = np.abs(w[:2]) / np.sqrt((w[:2] ** 2).sum())
    return wt[0] * V[:, 0] + wt[1] * V[:, 1], w[:2]
