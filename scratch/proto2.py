"""Vectorised ADMM prototype with adaptive rho / over-relaxation options."""
import sys; sys.path.insert(0, '/root/repo/scratch')
from proto import *

class Cols:
    def __init__(s, I):
        s.I = I; m = I.m
        cs = np.array([len(o) for o in I.cols])
        s.groups = []
        for c in np.unique(cs):
            if c == 0: continue
            js = np.flatnonzero(cs == c)
            idx = np.stack([I.cols[j] for j in js])          # (g, c)
            a = np.stack([I.a[j] for j in js])                # (g, c)
            s.groups.append((c, js, idx, a))

def prox_cols(Cg, Yx, alpha, sv, g, rho_f):
    """alpha: list per group (g,c); returns new alpha list and LL (n x n)"""
    n = Yx.shape[0]; LL = np.zeros((n, n)); cp = g * g / (2 * rho_f); out = []; svo = []
    for gi, (c, js, idx, a) in enumerate(Cg.groups):
        al = alpha[gi]
        Z = Yx[idx[:, :, None], idx[:, None, :]] - (g / (2 * rho_f)) * al[:, :, None] * al[:, None, :]
        B = np.eye(c)[None] + g * Z
        b, Q = np.linalg.eigh(B)
        qa = np.einsum('gij,gi->gj', Q, a); qa2 = qa ** 2
        lo = np.maximum(0.0, -b[:, 0] / cp) ; 
        s = sv[gi].copy() if sv[gi] is not None else np.full(len(js), 0.0)
        # bracket hi
        hi = np.maximum(2 * lo + 1.0, 1.0)
        def phi(s_):
            d = b + cp * s_[:, None]
            return (qa2 / d ** 2).sum(1) - s_, -2 * cp * (qa2 / d ** 3).sum(1) - 1
        for _ in range(200):
            ph, _d = phi(hi); bad = ph > 0
            if not bad.any(): break
            hi = np.where(bad, hi * 2, hi)
        s = np.where((s > lo) & (s < hi), s, hi)
        for it in range(100):
            ph, dph = phi(s)
            lo = np.where(ph > 0, s, lo); hi = np.where(ph <= 0, s, hi)
            sn = s - ph / dph
            sn = np.where((sn > lo) & (sn < hi), sn, 0.5 * (lo + hi))
            done = np.abs(sn - s) <= 1e-15 * np.maximum(1, np.abs(s))
            s = sn
            if done.all(): break
        y = qa / (b + cp * s[:, None])
        aln = np.einsum('gij,gj->gi', Q, y)
        out.append(aln); svo.append(s)
        np.add.at(LL, (idx[:, :, None], idx[:, None, :]), aln[:, :, None] * aln[:, None, :])
    return out, svo, LL

def fval(Cg, Y, g):
    v = 0.0
    for (c, js, idx, a) in Cg.groups:
        B = np.eye(c)[None] + g * Y[idx[:, :, None], idx[:, None, :]]
        v += 0.5 * (a * np.linalg.solve(B, a[:, :, None])[:, :, 0]).sum()
    return v

def admm2(I, cuts=(), ctype="linear", rho=None, iters=2000, tol=1e-9, relax=1.0, adapt=0, q1=True, Y0=None, U0=None, verbose=False, fstar=None, rf_ratio=1.0):
    n, m, k, g = I.n, I.m, I.k, I.g
    Cg = Cols(I)
    rows = build_rows(I, cuts, ctype, q1); R = len(rows)
    if rho is None: rho = 0.5 * g * (I.A[I.mask] ** 2).sum() / m
    ncone = 2 if k > 1 else 1
    w1Y = rf_ratio * I.N + ncone; w1U = 2 * np.ones((n, k))   # weights for rho=1
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); b = np.zeros(R)
    for r, (CY, CU, rhs, _) in enumerate(rows):
        if CY is not None: AY[r] = CY.ravel()
        AU[r] = CU.ravel(); b[r] = rhs
    G1 = (AY / w1Y.ravel()) @ AY.T + (AU / w1U.ravel()) @ AU.T
    Y = np.eye(n) * (k / n) if Y0 is None else Y0.copy()
    U = np.zeros((n, k)) if U0 is None else U0.copy()
    Yp = Y.copy()
    alpha = [np.zeros_like(a) for (_, _, _, a) in Cg.groups]; sv = [None] * len(Cg.groups)
    D1 = np.zeros((n + k, n + k)); D2 = np.zeros((n, n))
    hist = []; lam = np.zeros(R)
    Ik = np.eye(k)
    for it in range(iters):
        rho_f = rho * rf_ratio; rho_c = rho
        alpha, sv, LL = prox_cols(Cg, 2 * Y - Yp, alpha, sv, g, rho_f)
        Zg = np.block([[Y, U], [U.T, Ik]])
        W1, ev1, _ = psd_proj(Zg - D1)
        W1r = relax * W1 + (1 - relax) * Zg
        H1 = W1r + D1
        tY = rho_f * (I.N * Y + (g / (2 * rho_f)) * LL) + rho_c * H1[:n, :n]
        if k > 1:
            w2, V2 = np.linalg.eigh(Y - D2)
            W2 = (V2 * np.minimum(w2, 1.0)) @ V2.T
            W2r = relax * W2 + (1 - relax) * Y
            tY += rho_c * (W2r + D2)
        tY /= (rho * w1Y)
        tU = H1[:n, n:]
        c = AY @ tY.ravel() + AU @ tU.ravel() - b
        lam = nnqp(G1 / rho, c)
        Yn = tY - ((AY.T @ lam) / (rho * w1Y.ravel())).reshape(n, n)
        Un = tU - ((AU.T @ lam) / (rho * w1U.ravel())).reshape(n, k)
        Zn = np.block([[Yn, Un], [Un.T, Ik]])
        D1 = D1 + W1r - Zn
        if k > 1: D2 = D2 + W2r - Yn
        rp = np.linalg.norm(W1 - Zn); rd = rho * np.linalg.norm(Zn - Zg)
        Yp = Y; Y = Yn; U = Un
        if adapt and it > 0 and it % adapt == 0:
            if rp * rho > 10 * rd: fac = 2.0
            elif rd > 10 * rp * rho: fac = 0.5
            else: fac = 1.0
            if fac != 1.0:
                rho *= fac; D1 /= fac; D2 /= fac
        if it % 10 == 0 or it == iters - 1:
            fv = fval(Cg, Y, g); hist.append((it, fv, rp, rd / rho, rho))
            if verbose and it % 50 == 0: print(it, "f=%.10f rp=%.2e rd=%.2e rho=%.1f" % (fv, rp, rd / rho, rho), "" if fstar is None else "err=%.2e" % (abs(fv - fstar) / abs(fstar)))
            if max(rp, rd / rho) < tol: break
    Psi = psd_proj(rho * D1)[0]; Psi2 = psd_proj(-rho * D2)[0] if k > 1 else None
    # expand alpha to per-column list for dual_bound
    al_list = [np.zeros(0)] * m
    for gi, (c, js, idx, a) in enumerate(Cg.groups):
        for t, j in enumerate(js): al_list[j] = alpha[gi][t]
    return dict(Y=Y, U=U, alpha=al_list, lam=lam, rows=rows, hist=hist, iters=it + 1, Psi=Psi, Psi2=Psi2, rho=rho)

if __name__ == "__main__":
    for (n, m, k, frac, kind) in [(50, 50, 1, 0.5, "noise"), (100, 100, 1, 0.2, "lowrank")]:
        A, mask = make_instance(n, m, k, frac, 0, kind=kind)
        I = Inst(A, mask, 80.0, k)
        ref = admm2(I, iters=6000, tol=1e-11, adapt=50)
        fs = ref['hist'][-1][1]; print("ref f*", fs, ref['iters'], "LB", dual_bound(I, ref), "rho", ref['rho'])
        for kw in [dict(), dict(relax=1.6), dict(adapt=25), dict(adapt=25, relax=1.6), dict(adapt=25, relax=1.8), dict(adapt=25, relax=1.6, rf_ratio=0.3), dict(adapt=25, relax=1.6, rf_ratio=3.0)]:
            t = time.time(); o = admm2(I, iters=1500, tol=1e-9, **kw); el = time.time() - t
            h = o['hist']
            def itacc(e): return next((it for i_, (it, f_, rp, rd, r_) in enumerate(h) if all(abs(x[1] - fs) / abs(fs) < e for x in h[i_:])), None)
            print(n, kw, "iters", o['iters'], "it1e-5", itacc(1e-5), "it1e-6", itacc(1e-6), "it1e-7", itacc(1e-7), "rho", o['rho'], "%.1fs" % el, flush=True)
