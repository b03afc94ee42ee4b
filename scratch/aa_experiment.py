"""Experiment: acceleration of the consensus ADMM on a crawling node (oracle-side, numpy)."""
import sys, os, time, math
sys.path.insert(0, "/root/repo/oracle")
import numpy as np, omc_oracle as orc
from omc_oracle import _prox_columns, _psd_split, nnqp, build_rows, row_subspace, initial_rho, dual_bound_from

A, mask = orc.make_instance(20, 24, 1, seed=11, kind="readme")
inst = orc.Instance(A, mask, 80.0, 1)
cuts = list(np.load("/root/repo/scratch/slow_nodes.npy", allow_pickle=True)[0]["cuts"])

class Solver:
    def __init__(self, inst, cuts, rho_scale=16.0, relax=1.6):
        self.inst = inst; p = orc.RelaxParams(rho_scale=rho_scale); self.p = p
        n, m, k, g = inst.n, inst.m, inst.k, inst.gamma
        self.rows = build_rows(inst, cuts, "linear", None, None, True); R = len(self.rows)
        self.Q = row_subspace(self.rows, n, k); r = self.Q.shape[1]; self.r = r
        self.rho = initial_rho(inst, p); self.rx = relax
        self.wY1 = p.rho_f_ratio * inst.N + 2.0
        AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); self.b = np.array(self.rows.rhs)
        for rr in range(R):
            if self.rows.kinds[rr] == "trace": AY[rr] = np.eye(n).ravel()
            elif self.rows.kinds[rr] == "cut": AY[rr] = np.outer(self.rows.xs[rr], self.rows.xs[rr]).ravel()
            AU[rr] = self.rows.CU[rr].ravel()
        self.AY, self.AU = AY, AU
        self.G1 = (AY / self.wY1.ravel()) @ AY.T + (AU / 2.0) @ AU.T
        self.svals = [None] * len(inst.groups)
        self.shapes = None
    def init_state(self):
        n, k, r = self.inst.n, self.inst.k, self.r
        Y = np.eye(n) * (k / n)
        return dict(Y=Y, Yp=Y.copy(), Vt=np.zeros((r, k)), D1=np.zeros((n, n)), D3=np.zeros((n, n)), D3V=np.zeros((r, k)), D3T=np.zeros((k, k)),
                    alpha=[np.zeros_like(a) for (_, _, _, a) in self.inst.groups])
    def pack(self, s):
        parts = [s["Y"].ravel(), s["Yp"].ravel(), s["Vt"].ravel(), s["D1"].ravel(), s["D3"].ravel(), s["D3V"].ravel(), s["D3T"].ravel()] + [a.ravel() for a in s["alpha"]]
        return np.concatenate(parts)
    def unpack(self, z, like):
        out = {}; o = 0
        for key in ("Y", "Yp", "Vt", "D1", "D3", "D3V", "D3T"):
            sz = like[key].size; out[key] = z[o:o + sz].reshape(like[key].shape).copy(); o += sz
        al = []
        for a in like["alpha"]:
            al.append(z[o:o + a.size].reshape(a.shape).copy()); o += a.size
        out["alpha"] = al
        return out
    def step(self, s):
        inst = self.inst; n, k, g = inst.n, inst.k, inst.gamma; r = self.r; Q = self.Q; rho = self.rho; rx = self.rx
        Y, Yp, Vt, D1, D3, D3V, D3T = s["Y"], s["Yp"], s["Vt"], s["D1"], s["D3"], s["D3V"], s["D3T"]
        rho_f = rho * self.p.rho_f_ratio
        alpha, self.svals, LL = _prox_columns(inst, 2.0 * Y - Yp, s["alpha"], self.svals, rho_f)
        w1, V1 = _psd_split(Y - D1); W1 = (V1 * np.clip(w1, 0.0, 1.0)) @ V1.T
        Min = Y - D3; Ik = np.eye(k)
        M3 = np.block([[Q.T @ Min @ Q, Vt - D3V], [(Vt - D3V).T, Ik - D3T]])
        w3, V3 = _psd_split(M3); P3 = (V3 * np.maximum(w3, 0.0)) @ V3.T; Q3 = P3 - 0.5 * (M3 + M3.T)
        dS = Q3[:r, :r]; W3V = P3[:r, r:]; W3T = P3[r:, r:]
        tY = (rho_f * (inst.N * Y) + 0.5 * g * LL + rho * (rx * W1 + (1.0 - rx) * Y + D1) + rho * (Y + (1.0 - rx) * D3 + rx * (Q @ dS @ Q.T))) / (rho * self.wY1)
        tV = rx * W3V + (1.0 - rx) * Vt + D3V
        c = self.AY @ tY.ravel() + self.AU @ (Q @ tV).ravel() - self.b
        mu = nnqp(self.G1, c); self.lam = rho * mu; self.Q3 = Q3
        Yn = tY - ((self.AY.T @ mu) / self.wY1.ravel()).reshape(n, n); Yn = 0.5 * (Yn + Yn.T)
        Vn = tV - Q.T @ ((self.AU.T @ mu) / 2.0).reshape(n, k)
        nD1 = D1 + rx * W1 + (1.0 - rx) * Y - Yn
        nD3 = (1.0 - rx) * D3 + Y + rx * (Q @ dS @ Q.T) - Yn
        nD3V = D3V + rx * W3V + (1.0 - rx) * Vt - Vn
        nD3T = D3T + rx * (W3T - Ik)
        self.rp = math.sqrt(float(np.linalg.norm(W1 - Yn) ** 2 + np.linalg.norm(Min + Q @ dS @ Q.T - Yn) ** 2 + 2.0 * np.linalg.norm(W3V - Vn) ** 2 + np.linalg.norm(W3T - Ik) ** 2))
        self.rd = math.sqrt(float(np.linalg.norm(Yn - Y) ** 2 + 2.0 * np.linalg.norm(Vn - Vt) ** 2))
        return dict(Y=Yn, Yp=Y, Vt=Vn, D1=nD1, D3=nD3, D3V=nD3V, D3T=nD3T, alpha=alpha)
    def certificate(self, s):
        obj, Lam = self.inst.f_value(s["Y"], want=True)
        lb = dual_bound_from(self.inst, Lam, self.rows, self.lam, self.Q, self.rho * self.Q3)
        return obj, lb

def run(mode, max_iters=3000, mem=8, every=1, relax=1.6, rho_scale=16.0, verbose=False):
    S = Solver(inst, cuts, rho_scale=rho_scale, relax=relax)
    s = S.init_state(); best_lb = -1e300
    Zs, Gs = [], []          # iterates and their images
    t0 = time.time(); naa = 0; nrej = 0
    fprev = None
    for it in range(1, max_iters + 1):
        g_ = S.step(s)
        if mode == "plain" or it < 50:
            s = g_
        else:
            z = S.pack(s); gz = S.pack(g_)
            Zs.append(z); Gs.append(gz)
            if len(Zs) > mem + 1: Zs.pop(0); Gs.pop(0)
            if len(Zs) >= 3 and it % every == 0:
                F = np.array([gg - zz for gg, zz in zip(Gs, Zs)])       # residuals f_i
                dF = (F[1:] - F[:-1]).T; dG = (np.array(Gs)[1:] - np.array(Gs)[:-1]).T
                gam, *_ = np.linalg.lstsq(dF, F[-1], rcond=1e-10)
                zaa = gz - dG @ gam
                # safeguard: accept if the AA point's own residual is not much larger
                saa = S.unpack(zaa, g_)
                sv = list(S.svals)
                g2 = S.step(saa)
                fa = np.linalg.norm(S.pack(g2) - zaa); fn = np.linalg.norm(F[-1])
                if fa <= 1.0 * fn:
                    s = g2; naa += 1          # we already paid for the step: take its image
                    Zs.append(zaa); Gs.append(S.pack(g2))
                    if len(Zs) > mem + 1: Zs.pop(0); Gs.pop(0)
                else:
                    S.svals = sv; s = g_; nrej += 1; Zs, Gs = [], []
            else:
                s = g_
        if it % 25 == 0:
            obj, lb = S.certificate(s); best_lb = max(best_lb, lb)
            gap = (obj - best_lb) / abs(obj)
            if verbose and it % 250 == 0: print(mode, it, "gap %.2e rp %.1e rd %.1e" % (gap, S.rp, S.rd), "aa", naa, "rej", nrej, flush=True)
            if gap <= 1e-6 and S.rp <= 1e-7 * math.sqrt(inst.n + inst.k):
                break
    print("%-10s mem %2d every %2d relax %.2f: iters %4d  gap %.2e  rp %.1e rd %.1e  accepted %d rejected %d  (%.0fs)" % (mode, mem, every, relax, it, gap, S.rp, S.rd, naa, nrej, time.time() - t0), flush=True)

if __name__ == "__main__":
    run("plain")
    run("plain", relax=1.8)
    run("aa", mem=5, every=1)
    run("aa", mem=10, every=5)
