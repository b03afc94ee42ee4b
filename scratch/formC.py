"""Formulation C prototype: U restricted to span(Q) with small cone [Q'YQ V; V' I] >= 0."""
import sys; sys.path.insert(0, 'oracle')
import numpy as np, math, time
import omc_oracle as orc
from omc_oracle import *

def relaxC(inst, cuts=(), cut_type="linear", params=None, verbose=False, adapt=0, rho_mult=1.0):
    p = params or RelaxParams()
    n, m, k, g = inst.n, inst.m, inst.k, inst.gamma
    rows = build_rows(inst, cuts, cut_type, None, None, p.reference_quirk_q1)
    R = len(rows)
    # subspace of U functionals
    Xext = [rows.CU[r] for r in range(R) if rows.kinds[r] != "trace"]
    cols = []
    for CU in Xext:
        for j in range(k):
            if np.abs(CU[:, j]).max() > 0: cols.append(CU[:, j])
    if cols:
        Xe = np.stack(cols, 1)
        Uq, sq, _ = np.linalg.svd(Xe, full_matrices=False)
        Q = Uq[:, sq > 1e-10 * sq[0]]
    else:
        Q = np.zeros((n, 0))
    r = Q.shape[1]
    rho = rho_mult * p.rho_scale * 0.5 * g * inst.sumA2 / m; rho_f = rho * p.rho_f_ratio
    wY = rho_f * inst.N + 2 * rho
    wU = 2.0 * rho * np.ones((n, k))
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); b = np.array(rows.rhs)
    for rr in range(R):
        if rows.kinds[rr] == "trace": AY[rr] = np.eye(n).ravel()
        elif rows.kinds[rr] == "cut": AY[rr] = np.outer(rows.xs[rr], rows.xs[rr]).ravel()
        AU[rr] = rows.CU[rr].ravel()
    G = (AY / wY.ravel()) @ AY.T + (AU / wU.ravel()) @ AU.T
    Y = np.eye(n) * (k / n); U = np.zeros((n, k)); Yp = Y.copy()
    alpha = [np.zeros_like(a) for (_, _, _, a) in inst.groups]; svals = [None] * len(inst.groups)
    D1 = np.zeros((n, n)); D3 = np.zeros((n, n)); D3V = np.zeros((r, k)); D3T = np.zeros((k, k))
    Ik = np.eye(k); lam = np.zeros(R); lb = -math.inf; hist = []; status = 1
    for it in range(1, p.max_iters + 1):
        alpha, svals, LL = orc._prox_columns(inst, 2.0 * Y - Yp, alpha, svals, rho_f)
        # C1: 0 <= Y <= I
        w1, V1 = np.linalg.eigh(Y - D1)
        W1 = (V1 * np.clip(w1, 0.0, 1.0)) @ V1.T
        W1r = p.relax * W1 + (1 - p.relax) * Y
        # C3: small cone
        Min = Y - D3
        S_in = Q.T @ Min @ Q; V_in = Q.T @ U - D3V
        M3 = np.block([[S_in, V_in], [V_in.T, Ik - D3T]])
        w3, V3 = np.linalg.eigh(0.5 * (M3 + M3.T))
        P3 = (V3 * np.maximum(w3, 0)) @ V3.T
        dS = P3[:r, :r] - S_in
        W3Y = Min + Q @ dS @ Q.T; W3V = P3[:r, r:]; W3T = P3[r:, r:]
        W3Yr = p.relax * W3Y + (1 - p.relax) * Y
        W3Vr = p.relax * W3V + (1 - p.relax) * (Q.T @ U)
        W3Tr = p.relax * W3T + (1 - p.relax) * Ik
        tY = (rho_f * (inst.N * Y) + 0.5 * g * LL + rho * (W1r + D1) + rho * (W3Yr + D3)) / wY
        tU = Q @ (W3Vr + D3V)
        c = AY @ tY.ravel() + AU @ tU.ravel() - b
        lam = nnqp(G, c)
        Yn = tY - ((AY.T @ lam) / wY.ravel()).reshape(n, n); Yn = 0.5 * (Yn + Yn.T)
        Un = tU - ((AU.T @ lam) / wU.ravel()).reshape(n, k)
        Un = Q @ (Q.T @ Un)
        D1 = D1 + W1r - Yn
        D3 = D3 + W3Yr - Yn; D3V = D3V + W3Vr - Q.T @ Un; D3T = D3T + W3Tr - Ik
        rp = math.sqrt(np.linalg.norm(W1 - Yn) ** 2 + np.linalg.norm(W3Y - Yn) ** 2 + 2 * np.linalg.norm(W3V - Q.T @ Un) ** 2 + np.linalg.norm(W3T - Ik) ** 2)
        rd = math.sqrt(np.linalg.norm(Yn - Y) ** 2 + 2 * np.linalg.norm(Un - U) ** 2)
        Yp = Y; Y = Yn; U = Un
        if adapt and it % adapt == 0 and it < 2000:
            zn = math.sqrt(np.linalg.norm(Y) ** 2 + 2 * np.linalg.norm(U) ** 2 + k)
            dn = math.sqrt(np.linalg.norm(D1) ** 2 + np.linalg.norm(D3) ** 2 + 2 * np.linalg.norm(D3V) ** 2 + np.linalg.norm(D3T) ** 2) + 1e-300
            ratio = (rp / zn) / (rd / dn + 1e-300)
            if ratio > 5 or ratio < 0.2:
                fac = min(10.0, max(0.1, math.sqrt(ratio)))
                rho *= fac; rho_f *= fac; wY = wY * fac; wU = wU * fac; G = G / fac
                D1 /= fac; D3 /= fac; D3V /= fac; D3T /= fac
                if verbose: print("   rho ->", rho, "ratio", ratio)
        if it % p.check_every == 0 or it == p.max_iters:
            obj, Lam = inst.f_value(Y, want=True)
            # dual bound
            Q3 = P3 - M3  # >= 0 (negative part magnitude)
            Psi = rho * Q3
            c0 = float((inst.A * Lam).sum()) - 0.5 * float((Lam * Lam).sum())
            M = -0.5 * g * (Lam @ Lam.T); cU = np.zeros((n, k)); const = 0.0
            for rr in range(R):
                if rows.kinds[rr] == "trace" or lam[rr] == 0: continue
                if rows.kinds[rr] == "cut": M += lam[rr] * np.outer(rows.xs[rr], rows.xs[rr])
                cU += lam[rr] * rows.CU[rr]; const -= lam[rr] * rows.rhs[rr]
            M -= Q @ Psi[:r, :r] @ Q.T
            cV = Q.T @ cU - 2 * Psi[:r, r:]; const -= np.trace(Psi[r:, r:])
            ev = np.linalg.eigvalsh(0.5 * (M + M.T))
            lbn = c0 + np.minimum(ev[:k], 0).sum() - np.linalg.norm(cV, axis=0).sum() + const
            lb = max(lb, lbn)
            hist.append((it, obj, lb, rp, rd))
            if verbose: print(it, "obj %.9f lb %.9f gap %.2e rp %.2e rd %.2e" % (obj, lb, (obj - lb) / abs(obj), rp, rd))
            if (obj - lb) <= p.eps_gap * max(1, abs(obj)) and rp <= p.eps_feas * math.sqrt(n + k):
                status = 0; break
    obj, Lam = inst.f_value(Y, want=True)
    # recover U' = Y Q (Q'YQ)^+ Vt
    S = Q.T @ Y @ Q
    Uout = Y @ Q @ np.linalg.pinv(S, rcond=1e-12) @ (Q.T @ U) if r > 0 else np.zeros((n, k))
    return dict(objective=obj, dual_bound=lb, Y=Y, U=Uout, Usub=U, iters=it, hist=hist, status=status, lam=lam, rows=rows, r=r, rho=rho)

