"""CPU oracle, Shor mode of matrix_completion_SDP_relaxation  --  TEST INFRASTRUCTURE ONLY (see omc_oracle.py header).

PARITY UNPINNED: as for omc_oracle.py, nothing produced by the reference (Julia + Mosek, not runnable here, no
fixtures) pins this file.  What pins it: the closed-form / sandwich KATs of tests/test_oracle_shor.py and the
two-sided certificate computed below (primal residuals of every cone of the reference's program on the returned
point, and a Lagrangian dual bound that is valid for ANY multipliers).

What is restated (OMC.jl = /root/reference/src/OptimalMatrixCompletion.jl), rank k = 1:
    variables W >= 0, V1, V2, V3                     OMC.jl:1503-1525
    cones [Y X;X' Th], [Y U;U' I], I - Y, tr Y <= k  OMC.jl:1554-1558
    rotated cones W_ij >= X_ij^2 on the SOC list     OMC.jl:1757-1762
    Th_jj = sum_i W_ij                               OMC.jl:1763-1767
    one order-5 PSD block per minor (i1,i2,j1,j2)    OMC.jl:1768-1779
    objective 1/2 sum_Omega (A^2 - 2AX + W) + tr(Th)/(2 gamma)   OMC.jl:1838-1846, 1960-1967
and rank k > 1 (Xt, Wt, H, per-layer order-5 blocks, one order-(k+1) block per coordinate, W = sum Wt + 2 sum H on the minor coordinates:
OMC.jl:1491-1494, 1526-1551, 1780-1827).  REFERENCE QUIRK Q5: for k > 1 the minors do not constrain (X, W) at all.  With s >= 0 free,
    Xt_1 = X, Xt_t = 0;  Wt_1 = X^2 + e + s, Wt_t = s/(k-1);  H_(1,t) = -s/(k-1), H_(t,t') = 0;  V^1 = the products of X (V3 = their mean), V^t = 0
satisfies W = sum Wt + 2 sum H = X^2 + e, the order-(k+1) block (its Schur complement is [[e + s, -b 1'], [-b 1, b I]], b = s/(k-1): PSD for e >= 0), the
layers t >= 2 trivially, and the layer-1 block of every minor as soon as s >= |X_{i1j2} X_{i2j1} - X_{i1j1} X_{i2j2}| / 2 at its four coordinates --
the slack s that H cancels in W costs nothing.  So every (X, W) with W >= X^2 on the minor coordinates extends to a feasible point: the k > 1
program has the value of the same program WITHOUT its minors (W >= X^2 kept on their coordinates, which the order-(k+1) block implies).
`sdp_relaxation_shor` solves that and `complete_shor_rank_k` builds the extension; `shor_rank_k_residuals` checks the reference's full program on it.

How it is solved (ours; the reference hands the program to Mosek).  Consensus ADMM with X explicit (Shor mode):
  * W is kept only on the coordinates that occur in a minor (set C).  Off C the program only needs the column sums:
    with S_j = SOC rows of column j outside C,  theta_j = Th_jj,  t_j = theta_j - sum_{i in C_j} W_ij,  the constraints
    W_ij >= X_ij^2 (i in S_j), W >= 0 and Th_jj = sum_i W_ij are equivalent to ONE paraboloid per column,
    t_j >= ||X[S_j, j]||^2, the slack e_j = t_j - ||X[S_j,j]||^2 being carried by the cheapest entry outside C:
        type 0 (an unobserved entry outside C exists): slack costs nothing beyond theta_j / (2 gamma); the observed SOC entries get
               their 1/2 X^2 back as a quadratic objective term (1/2 (A - X)^2: the strong convexity of the base program);
        type 1 (entries outside C exist, all observed): slack costs 1/2 per unit -> linear terms 1/2 theta_j - 1/2 sum_C W;
        type 2 (the whole column is in C): theta_j = sum_i W_ij stays an equality.
  * blocks: (B0) [Y X;X' Th] >= 0 (order n+m, Th off-diagonal free), (B1) 0 <= Y <= I, (B3) the small cone of the rows as in the base
    solver, (B4) one order-5 cone per minor, (B5) one paraboloid per column; penalties rho, rho, rho, r4 rho, r5 rho.
  * global step: weighted averages shifted by the objective, with the column couplings and the node's linear rows exact.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass

import numpy as np

import omc_oracle as base
from omc_oracle import (OMC_INFEASIBLE, OMC_OPTIMAL, OMC_SLOW_PROGRESS, OMC_TIME_LIMIT, build_rows, nnqp, recover_U,
                        row_subspace)


@dataclass
class ShorParams:
    eps_gap: float = 1e-5        # two-sided: |objective - dual_bound| <= eps_gap * max(1, |objective|)   (1e-5 at the Shor configs)
    eps_feas: float = 1e-7       # and rp <= eps_feas * sqrt(n + m + k)   (scaled variables)
    max_iters: int = 6000
    check_every: int = 25
    rho: float = 0.05            # penalty of the cone blocks, in the scaled variables
    r4: float = 0.0              # penalty of the minor blocks relative to rho; 0 = automatic: 75 n m / (4 nq) clamped to [0.25, 40]
    r5: float = 2.0              # penalty of the column paraboloids relative to rho
    relax: float = 1.6
    time_limit: float = 3600.0
    reference_quirk_q1: bool = True
    stall_checks: int = 40
    bump_max: int = 1            # rho *= bump_factor when rp > bump_ratio * rd at a check (as the base solver)
    bump_factor: float = 4.0
    bump_ratio: float = 4.0
    bump_after: int = 100
    bump_window: int = 4
    early_stop_after: int = 400
    early_stop_factor: float = 0.0       # off: the bound of a Shor node lags the primal value for the first ~1000 iterations
    scale: float = 0.0           # 0: automatic.  A is multiplied by this (the program is homogeneous of degree 2 in A)
    verbose: int = 0


def shor_scale(inst):
    """A -> scale * A so that the scaled entries of X are O(1/sqrt m): scale^2 ||A_Omega||^2 = min(n, m) * |Omega| / (n m)."""
    n, m = inst.n, inst.m
    return math.sqrt(inst.indices.mean() * min(n, m) / max(inst.sumA2, 1e-300))


def _proj_psd_batch(M):
    w, V = np.linalg.eigh(M)
    return np.einsum("qij,qj,qkj->qik", V, np.maximum(w, 0.0), V)


def proj_paraboloid(xi, t):
    """Projection of (xi, t) onto {t >= ||xi||^2}: xi / (1 + 2 nu), t + nu with nu >= 0 the root of
    s / (1 + 2 nu)^2 = t + nu, s = ||xi||^2 (decreasing minus increasing: bisection)."""
    s = float(xi @ xi)
    if t >= s:
        return xi, t, 0.0
    lo = max(0.0, -t); hi = max(1.0, 2.0 * lo + 1.0)
    f = lambda nu: s / (1.0 + 2.0 * nu) ** 2 - t - nu
    while f(hi) > 0.0:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid) > 0.0:
            lo = mid
        else:
            hi = mid
        if hi - lo <= 1e-16 * max(1.0, hi):
            break
    nu = 0.5 * (lo + hi)
    return xi / (1.0 + 2.0 * nu), t + nu, nu


class ShorStructure:
    """Index structure of a node's Shor lists (k = 1 form)."""

    def __init__(self, n, m, shor_idx, soc_idx, mask):
        q = np.asarray(shor_idx, np.int64).reshape(-1, 4) - 1          # 0-based (i1, i2, j1, j2)
        self.nq = len(q)
        self.n, self.m = n, m
        i1, i2, j1, j2 = (q[:, 0], q[:, 1], q[:, 2], q[:, 3]) if self.nq else (np.zeros(0, np.int64),) * 4
        if self.nq and not ((0 <= i1).all() and (i1 < i2).all() and (i2 < n).all() and (0 <= j1).all() and (j1 < j2).all() and (j2 < m).all()):
            raise ValueError("Shor minors must satisfy 1 <= i1 < i2 <= n, 1 <= j1 < j2 <= m (OMC.jl:2556-2603)")
        if self.nq and len(np.unique(q, axis=0)) != self.nq:
            raise ValueError("duplicate Shor minor")
        # coordinates of the four entries, in the block order of OMC.jl:1772: (i1,j1), (i1,j2), (i2,j1), (i2,j2)
        self.ci = np.stack([i1, i1, i2, i2], 1)
        self.cj = np.stack([j1, j2, j1, j2], 1)
        # V1[i,(j1,j2)] at (1,2) for i1 and (3,4) for i2 ; V2[(i1,i2),j] at (1,3) for j1 and (2,4) for j2 ; V3 at (1,4) and (2,3)
        def keys(*cols):
            return np.stack(cols, 1) if self.nq else np.zeros((0, len(cols)), np.int64)
        if self.nq:
            u1, inv1 = np.unique(np.concatenate([keys(i1, j1, j2), keys(i2, j1, j2)]), axis=0, return_inverse=True)
            u2, inv2 = np.unique(np.concatenate([keys(i1, i2, j1), keys(i1, i2, j2)]), axis=0, return_inverse=True)
        else:
            u1 = u2 = np.zeros((0, 3), np.int64); inv1 = inv2 = np.zeros(0, np.int64)
        inv1 = np.asarray(inv1).ravel(); inv2 = np.asarray(inv2).ravel()
        self.nv1, self.nv2 = len(u1), len(u2)
        self.k12 = inv1[:self.nq]; self.k34 = inv1[self.nq:]
        self.k13 = inv2[:self.nq]; self.k24 = inv2[self.nq:]
        self.cnt1 = np.bincount(inv1, minlength=self.nv1).astype(float)
        self.cnt2 = np.bincount(inv2, minlength=self.nv2).astype(float)
        self.cntC = np.zeros((n, m))                                   # number of minors containing each matrix coordinate
        if self.nq:
            np.add.at(self.cntC, (self.ci.ravel(), self.cj.ravel()), 1.0)
        s = np.asarray(soc_idx, np.int64).reshape(-1, 2) - 1
        if len(s) and not ((0 <= s[:, 0]).all() and (s[:, 0] < n).all() and (0 <= s[:, 1]).all() and (s[:, 1] < m).all()):
            raise ValueError("SOC coordinates out of range")
        soc = np.zeros((n, m), bool)
        if len(s):
            soc[s[:, 0], s[:, 1]] = True
        self.soc_list = soc
        self.inC = self.cntC > 0
        self.inS = soc & ~self.inC          # W >= X^2 is implied by the order-5 block on C
        nonC = ~self.inC
        has_unobs = (nonC & ~mask).any(0)
        has_nonC = nonC.any(0)
        self.ctype = np.where(has_unobs, 0, np.where(has_nonC, 1, 2))


def assemble_minor_blocks(st, X, W, V1, V2, V3):
    """The order-5 matrices of OMC.jl:1771-1777, one per minor (nq x 5 x 5)."""
    nq = st.nq
    M = np.zeros((nq, 5, 5))
    M[:, 0, 0] = 1.0
    xs = X[st.ci, st.cj]; ws = W[st.ci, st.cj]
    for p in range(4):
        M[:, 0, p + 1] = M[:, p + 1, 0] = xs[:, p]
        M[:, p + 1, p + 1] = ws[:, p]
    M[:, 1, 2] = M[:, 2, 1] = V1[st.k12]
    M[:, 3, 4] = M[:, 4, 3] = V1[st.k34]
    M[:, 1, 3] = M[:, 3, 1] = V2[st.k13]
    M[:, 2, 4] = M[:, 4, 2] = V2[st.k24]
    M[:, 1, 4] = M[:, 4, 1] = V3
    M[:, 2, 3] = M[:, 3, 2] = V3
    return M


def complete_W(inst, st, X, Wc, theta):
    """The full W of the reference's program from (X, W on C, theta): X^2 on the SOC entries, 0 on the others, and the slack
    e_j = theta_j - sum_C W - ||X_S||^2 on the cheapest entry of the column outside C (first unobserved one, else the first one)."""
    m = inst.m
    W = np.where(st.inC, Wc, np.where(st.inS, X * X, 0.0))
    for j in range(m):
        if st.ctype[j] == 2:
            continue
        e = theta[j] - W[:, j].sum()
        nonC = np.flatnonzero(~st.inC[:, j])
        un = [i for i in nonC if not inst.indices[i, j]]
        W[(un[0] if un else nonC[0]), j] += e
    return W


def shor_primal_residuals(inst, st, rows, X, W, V1, V2, V3, Theta, Y, U):
    """Violation of the cones / rows of the reference's Shor-mode program (OMC.jl:1554-1561, 1564-1685, 1757-1779, 1831-1835)
    on a primal point (0 = feasible)."""
    res = {}
    res["theta_diag"] = float(np.abs(np.diag(Theta) - W.sum(0)).max())                        # 1763-1767
    res["W_nonneg"] = max(0.0, -float(W.min()))                                              # 1504
    res["soc"] = max(0.0, float((X * X - W)[st.soc_list].max())) if st.soc_list.any() else 0.0   # 1757-1762
    if st.nq:
        M = assemble_minor_blocks(st, X, W, V1, V2, V3)
        res["minors"] = max(0.0, -float(np.linalg.eigvalsh(M)[:, 0].min()))                  # 1768-1779
    else:
        res["minors"] = 0.0
    res.update(base.primal_residuals(inst, rows, Y, U, X, Theta))
    res["max"] = max(v for kk, v in res.items() if kk != "max")
    return res


def sdp_relaxation_shor(inst, shor_idx, soc_idx, cuts=(), cut_type="linear", U_lower=None, U_upper=None, params=None):
    """matrix_completion_SDP_relaxation with add_Shor_valid_inequalities = true, k = 1 (OMC.jl:1503-1525, 1755-1779,
    1838-1846).  `shor_idx`: (nq, 4) 1-based (i1, i2, j1, j2) = node.Shor_info.constraints_indexes; `soc_idx`: (ns, 2)
    1-based = node.Shor_info.SOC_constraints_indexes.  Returns the reference's keys (objective, Y, U, X, Theta, W, V1, V2, V3,
    termination_status, feasible, solve_time) plus dual_bound, iters, residuals."""
    if cut_type not in base.CUT_TYPES:
        raise ValueError("Invalid input for disjunctive cuts type (OMC.jl:1456-1462)")
    p = params or ShorParams()
    t0 = time.time()
    n, m, k, g = inst.n, inst.m, inst.k, inst.gamma
    N = n + m
    mask = inst.indices
    st_full = ShorStructure(n, m, shor_idx, soc_idx, mask)
    if k > 1:
        # quirk Q5 (header): the minors of the k > 1 form are vacuous; their coordinates keep W >= X^2 (implied by the order-(k+1) block)
        keep = st_full.soc_list | st_full.inC
        soc_k = [(i + 1, j + 1) for j in range(m) for i in range(n) if keep[i, j]]
        st = ShorStructure(n, m, [], soc_k, mask)
    else:
        st = st_full
    nq = st.nq
    rows = build_rows(inst, cuts, cut_type, U_lower, U_upper, p.reference_quirk_q1)
    R = len(rows)
    Q = row_subspace(rows, n, k)
    r = Q.shape[1]
    sc = p.scale if p.scale > 0 else shor_scale(inst)
    s2 = sc * sc
    Ah = inst.A * sc
    inC, inS, ctype = st.inC, st.inS, st.ctype
    t01 = ctype != 2
    qX = np.where(mask & inS & (ctype == 0)[None, :], 1.0, 0.0)       # quadratic objective coefficient of X
    cX = np.where(mask, -Ah, 0.0)
    cW = np.where(mask & inC, 0.5, 0.0) - np.where(inC & (ctype == 1)[None, :], 0.5, 0.0)
    cT = 1.0 / (2.0 * g) + np.where(ctype == 1, 0.5, 0.0)
    const0 = 0.5 * float((Ah[mask] ** 2).sum())
    rho = p.rho; rx = p.relax; r5 = p.r5
    r4 = p.r4 if p.r4 > 0 else min(40.0, max(0.25, 75.0 * n * m / (4.0 * max(nq, 1))))
    wY = 3.0                                          # big cone, clip, small cone
    wX = 2.0 + 2.0 * r4 * st.cntC + r5 * inS          # big cone (two symmetric positions), minors (two each), paraboloid copy
    wW = r4 * st.cntC
    iwW = np.where(inC, 1.0 / np.where(inC, wW, 1.0), 0.0)
    siw = 1.0 + iwW.sum(0)                            # 1/w_theta + sum_C 1/w_W   (w_theta = 1)
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); b = np.array(rows.rhs)
    for rr in range(R):
        if rows.kinds[rr] == "trace":
            AY[rr] = np.eye(n).ravel()
        elif rows.kinds[rr] == "cut":
            AY[rr] = np.outer(rows.xs[rr], rows.xs[rr]).ravel()
        AU[rr] = rows.CU[rr].ravel()
    G1 = (AY / wY) @ AY.T + (AU / 2.0) @ AU.T
    # state
    Y = np.eye(n) * (k / n); X = np.zeros((n, m)); Th = np.zeros((m, m)); W = np.zeros((n, m))
    V1 = np.zeros(st.nv1); V2 = np.zeros(st.nv2); V3 = np.zeros(nq); Vt = np.zeros((r, k))
    D0 = np.zeros((N, N)); D1 = np.zeros((n, n)); D3 = np.zeros((n, n)); D3V = np.zeros((r, k)); D3T = np.zeros((k, k))
    Dq = np.zeros((nq, 5, 5)); D5x = np.zeros((n, m)); D5t = np.zeros(m)
    Ik = np.eye(k)
    lam = np.zeros(R)
    status = OMC_SLOW_PROGRESS
    obj = math.inf; lb = -math.inf; rp = rd = math.inf
    hist = []; stall = 0; obj_prev = math.inf; lb_prev = -math.inf
    n_bumps = 0; last_bump = 0
    gap_prev = 1e300; gap_rate = 1.0; slow_votes = 0
    it = 0
    for it in range(1, p.max_iters + 1):
        # (B0) big cone
        G = np.block([[Y, X], [X.T, Th]])
        In0 = G - D0
        w0, E0 = np.linalg.eigh(0.5 * (In0 + In0.T))
        P0 = (E0 * np.maximum(w0, 0.0)) @ E0.T
        # (B1) clip
        w1, E1 = np.linalg.eigh(Y - D1)
        P1 = (E1 * np.clip(w1, 0.0, 1.0)) @ E1.T
        # (B3) small cone
        Min = Y - D3
        S_in = Q.T @ Min @ Q; V_in = Vt - D3V
        M3 = np.block([[S_in, V_in], [V_in.T, Ik - D3T]])
        w3, E3 = np.linalg.eigh(0.5 * (M3 + M3.T))
        P3 = (E3 * np.maximum(w3, 0.0)) @ E3.T
        Q3 = P3 - 0.5 * (M3 + M3.T)
        dS = Q3[:r, :r]; W3V = P3[:r, r:]; W3T = P3[r:, r:]
        # (B4) minors
        if nq:
            Mq = assemble_minor_blocks(st, X, W, V1, V2, V3)
            Inq = Mq - Dq
            Pq = _proj_psd_batch(Inq)
        # (B5) one paraboloid per column of type 0 / 1
        tcur = np.diag(Th) - W.sum(0)                 # W is zero off C
        P5x = np.zeros((n, m)); P5t = np.zeros(m); nu5 = np.zeros(m)
        for j in range(m):
            if ctype[j] == 2:
                continue
            Sj = inS[:, j]
            xi, tt, nu = proj_paraboloid((X[:, j] - D5x[:, j])[Sj], tcur[j] - D5t[j])
            P5x[Sj, j] = xi; P5t[j] = tt; nu5[j] = nu
        # ---- global step ----
        T0 = rx * P0 + (1.0 - rx) * G + D0
        tY = (T0[:n, :n] + (rx * P1 + (1.0 - rx) * Y + D1) + (Y + (1.0 - rx) * D3 + rx * (Q @ dS @ Q.T))) / wY
        T5x = np.where(inS, rx * P5x + (1.0 - rx) * X + D5x, 0.0)
        T5t = rx * P5t + (1.0 - rx) * tcur + D5t
        tX = 2.0 * T0[:n, n:] + r5 * T5x
        tW = np.zeros((n, m))
        tTh = T0[n:, n:]
        if nq:
            Tq = rx * Pq + (1.0 - rx) * Mq + Dq
            np.add.at(tX, (st.ci, st.cj), 2.0 * r4 * Tq[:, 0, 1:5])
            np.add.at(tW, (st.ci, st.cj), r4 * Tq[:, [1, 2, 3, 4], [1, 2, 3, 4]])
            s1 = np.zeros(st.nv1); s2_ = np.zeros(st.nv2)
            np.add.at(s1, st.k12, Tq[:, 1, 2]); np.add.at(s1, st.k34, Tq[:, 3, 4])
            np.add.at(s2_, st.k13, Tq[:, 1, 3]); np.add.at(s2_, st.k24, Tq[:, 2, 4])
            V1n = s1 / st.cnt1; V2n = s2_ / st.cnt2
            V3n = 0.5 * (Tq[:, 1, 4] + Tq[:, 2, 3])
        else:
            V1n, V2n, V3n = V1, V2, V3
        Xn = (tX - cX / rho) / (wX + qX / rho)
        # theta_j and W on C: weighted averages shifted by the objective, coupled per column through the paraboloid copy of
        # t_j = theta_j - sum_C W (types 0 / 1) or through the equality theta_j = sum_C W (type 2)
        thbar = np.diag(tTh) - cT / rho
        wbar = (tW * iwW) - np.where(inC, cW * iwW, 0.0) / rho
        base_ = thbar - wbar.sum(0)
        u = np.where(t01, (base_ - T5t) / (1.0 + r5 * siw), base_ / siw)      # type 2: u = tau (multiplier / rho)
        cpl = np.where(t01, r5 * u, u)
        thn = thbar - cpl
        Wn = np.where(inC, wbar + cpl[None, :] * iwW, 0.0)
        Thn = 0.5 * (tTh + tTh.T)
        Thn[np.arange(m), np.arange(m)] = thn
        # Y, Vt: rows
        tV = rx * W3V + (1.0 - rx) * Vt + D3V
        tU = Q @ tV
        c = AY @ tY.ravel() + AU @ tU.ravel() - b
        mu = nnqp(G1, c)
        lam = rho * mu
        Yn = tY - ((AY.T @ mu) / wY).reshape(n, n)
        Yn = 0.5 * (Yn + Yn.T)
        Vn = tV - Q.T @ ((AU.T @ mu) / 2.0).reshape(n, k)
        # ---- duals ----
        Gn = np.block([[Yn, Xn], [Xn.T, Thn]])
        D0 = T0 - Gn
        D1 = D1 + rx * P1 + (1.0 - rx) * Y - Yn
        D3 = (1.0 - rx) * D3 + Y + rx * (Q @ dS @ Q.T) - Yn
        D3V = D3V + rx * W3V + (1.0 - rx) * Vt - Vn
        D3T = D3T + rx * (W3T - Ik)
        tnew = thn - Wn.sum(0)
        D5x = np.where(inS, T5x - Xn, 0.0)
        D5t = np.where(t01, T5t - tnew, 0.0)
        rp2 = float(np.linalg.norm(P0 - Gn) ** 2 + np.linalg.norm(P1 - Yn) ** 2 + np.linalg.norm(Min + Q @ dS @ Q.T - Yn) ** 2
                    + 2.0 * np.linalg.norm(W3V - Vn) ** 2 + np.linalg.norm(W3T - Ik) ** 2
                    + np.linalg.norm(np.where(inS, P5x - Xn, 0.0)) ** 2 + np.linalg.norm(np.where(t01, P5t - tnew, 0.0)) ** 2)
        rd2 = float(np.linalg.norm(Gn - G) ** 2 + 2.0 * np.linalg.norm(Vn - Vt) ** 2 + np.linalg.norm(Wn - W) ** 2)
        if nq:
            Mqn = assemble_minor_blocks(st, Xn, Wn, V1n, V2n, V3n)
            Dq = Tq - Mqn
            rp2 += float(np.linalg.norm(Pq - Mqn) ** 2)
        rp = math.sqrt(rp2); rd = math.sqrt(rd2)
        Y, X, Th, W, V1, V2, V3, Vt = Yn, Xn, Thn, Wn, V1n, V2n, V3n, Vn
        if it % p.check_every == 0 or it == p.max_iters:
            obj_s = const0 + float((cX * X).sum() + 0.5 * (qX * X * X).sum() + (cW * W).sum() + (cT * np.diag(Th)).sum())
            Gam = rho * r4 * (Pq - Inq) if nq else np.zeros((0, 5, 5))
            lb_s = shor_dual_bound(inst, st, Ah, rows, lam, Q, rho * Q3, Gam, rho * r5 * nu5, P5x, X, qX, cX, cW, cT)
            obj = obj_s / s2; lb_new = lb_s / s2
            lb = max(lb, lb_new)
            hist.append((it, obj, lb_new, rp, rd, rho))
            if p.verbose:
                print(f"it {it:6d} obj {obj:.9g} lb {lb_new:.9g} rp {rp:.2e} rd {rd:.2e} rho {rho:g}")
            if abs(obj - lb) <= p.eps_gap * max(1.0, abs(obj)) and rp <= p.eps_feas * math.sqrt(N + k):
                status = OMC_OPTIMAL
                break
            if lb > 0.5 * inst.sumA2 * (1.0 + 1e-9) + 1e-9:       # optimum <= 1/2||A_Omega||^2 (X = W = Theta = 0 with any feasible (Y, U))
                status = OMC_INFEASIBLE
                break
            if abs(obj - obj_prev) <= 1e-7 * max(1.0, abs(obj)) and lb_new <= lb_prev + 1e-7 * max(1.0, abs(obj)):
                stall += 1
            else:
                stall = 0
            obj_prev = obj; lb_prev = lb
            if stall >= p.stall_checks:
                if abs(obj - lb) <= p.eps_gap * max(1.0, abs(obj)) and rp <= 10.0 * p.eps_feas * math.sqrt(N + k):
                    status = OMC_OPTIMAL
                break
            if time.time() - t0 > p.time_limit:
                status = OMC_TIME_LIMIT
                break
            if it >= p.max_iters:
                break
            if p.early_stop_factor > 0.0:
                target = p.eps_gap * max(1.0, abs(obj))
                gnow = obj - lb
                qq = 0.5 * gap_rate + 0.5 * min(gnow / gap_prev, 2.0) if (gap_prev < 1e299 and gap_prev > 0.0 and gnow > 0.0) else 1.0
                gap_prev = gnow; gap_rate = qq
                left = (p.max_iters - it) / float(p.check_every)
                need = math.log(max(gnow, target) / target) / -math.log(qq) if qq < 1.0 else 1e300
                hopeless = it >= p.early_stop_after and gnow > target and need > p.early_stop_factor * left
                slow_votes = slow_votes + 1 if hopeless else 0
                if slow_votes >= 8:
                    break
            if (p.bump_max and it >= p.bump_after and n_bumps < p.bump_max and it - last_bump >= p.check_every * p.bump_window
                    and rp > p.bump_ratio * rd):
                fac = p.bump_factor
                rho *= fac
                D0 /= fac; D1 /= fac; D3 /= fac; D3V /= fac; D3T /= fac; Dq /= fac; D5x /= fac; D5t /= fac
                n_bumps += 1; last_bump = it
                slow_votes = 0; gap_rate = 1.0
    Xo = X / sc; Tho = Th / s2
    Wo = complete_W(inst, st, Xo, W / s2, np.diag(Tho))
    U = recover_U(Y, Q, Vt)
    out = dict(objective=obj, dual_bound=lb, Y=Y, U=U, X=Xo, Theta=Tho, W=Wo, V1=V1 / s2, V2=V2 / s2, V3=V3 / s2,
               termination_status=status, feasible=status != OMC_INFEASIBLE, solve_time=time.time() - t0, iters=it,
               hist=hist, rp=rp, rd=rd, rho=rho, structure=st, rows=rows, scale=sc)
    out["objective_reference_formula"] = base.compute_SDP_relaxation_objective(Xo, Tho, inst.A, inst.indices, g, W=Wo)
    out["residuals"] = shor_primal_residuals(inst, st, rows, Xo, Wo, out["V1"], out["V2"], out["V3"], Tho, Y, U)
    if k > 1:
        out["structure_full"] = st_full
        out.update(complete_shor_rank_k(k, st_full, Xo, Wo))          # Xt, Wt, H, V1, V2, V3 of OMC.jl:1915-1917, 1909-1911
        out["residuals_rank_k"] = shor_rank_k_residuals(k, st_full, Xo, Wo, out)
        out["residuals"]["max"] = max(out["residuals"]["max"], out["residuals_rank_k"]["max"])
    return out


def shor_dual_bound(inst, st, Ah, rows, lam, Q, Psi3, Gam, zeta_adm, xi_at, Xbar, qX, cX, cW, cT):
    """Lower bound on the optimum of the (scaled) Shor-mode program, valid for ANY inputs with Gam_q >= 0 (order 5), zeta_adm >= 0,
    lam >= 0, Psi3 >= 0.  Lagrangian with multipliers
        Gam_q                            on the order-5 blocks,
        zeta_j                           on the paraboloid of column j, linearised at xi_j:  t_j - 2 xi_j' x_S + ||xi_j||^2 >= 0
                                         (on the equality theta_j = sum_C W of a type-2 column: free sign),
        nu_ij >= 0                       on W_ij >= 0, (i, j) in C (implied by the order-5 blocks, hence valid),
        [[Phi_j, phi_j], [phi_j', mu_j]] on [[Y, X_j], [X_j', Th_jj]] >= 0  (the big cone with its free off-diagonal eliminated),
    Stationarity in theta fixes mu_j = cT_j - zeta_j; in W it asks zeta_j >= max_{i in C_j} (sum_q Gam_q[p,p] - cW_ij), so zeta_j is raised to
    that; in X it fixes phi (where X has a quadratic objective term the residual -q Xbar is left to it: -1/2 q Xbar^2); in V1, V2, V3 the block
    entries must cancel -- they are made to cancel exactly by spreading each residual over the blocks that share the variable, every block
    being shifted by the Frobenius norm of its correction (so it stays PSD).  What is left is minimised over the compact set
    {0 <= Y <= I, tr Y <= k} x {||Vt_j|| <= 1}."""
    n, m, k = inst.n, inst.m, inst.k
    mask = inst.indices
    nq = st.nq
    r = Q.shape[1]
    inC, inS, ctype = st.inC, st.inS, st.ctype
    const = 0.5 * float((Ah[mask] ** 2).sum())
    gdiag = np.zeros((n, m)); g0 = np.zeros((n, m))
    if nq:
        Gm = 0.5 * (Gam + np.transpose(Gam, (0, 2, 1)))
        r1 = np.zeros(st.nv1); r2 = np.zeros(st.nv2)
        np.add.at(r1, st.k12, Gm[:, 1, 2]); np.add.at(r1, st.k34, Gm[:, 3, 4])
        np.add.at(r2, st.k13, Gm[:, 1, 3]); np.add.at(r2, st.k24, Gm[:, 2, 4])
        e1 = r1 / st.cnt1; e2 = r2 / st.cnt2
        e3 = 0.5 * (Gm[:, 1, 4] + Gm[:, 2, 3])
        E = np.zeros_like(Gm)
        E[:, 1, 2] = E[:, 2, 1] = e1[st.k12]; E[:, 3, 4] = E[:, 4, 3] = e1[st.k34]
        E[:, 1, 3] = E[:, 3, 1] = e2[st.k13]; E[:, 2, 4] = E[:, 4, 2] = e2[st.k24]
        E[:, 1, 4] = E[:, 4, 1] = e3; E[:, 2, 3] = E[:, 3, 2] = e3
        shift = np.sqrt((E * E).sum((1, 2)))
        Gm = Gm - E + shift[:, None, None] * np.eye(5)[None]
        const -= float(Gm[:, 0, 0].sum())
        np.add.at(g0, (st.ci, st.cj), Gm[:, 0, 1:5])
        np.add.at(gdiag, (st.ci, st.cj), Gm[:, [1, 2, 3, 4], [1, 2, 3, 4]])
    zmin = np.where(inC, gdiag - cW, -np.inf).max(0)            # W-stationarity on C
    zeta = np.where(ctype == 2, zmin, np.maximum(np.maximum(zeta_adm, 0.0), zmin))
    mu = cT - zeta
    xi = np.where(inS & (ctype != 2)[None, :], xi_at, 0.0)
    const -= float((zeta * (xi * xi).sum(0)).sum())
    # X-stationarity:  cX + q Xbar - 2 phi + 2 zeta xi - 2 g0 = 0 ; the part q Xbar of the gradient stays with the quadratic term
    phi = 0.5 * (cX + qX * Xbar) + zeta[None, :] * xi - g0
    const -= 0.5 * float((qX * Xbar * Xbar).sum())
    bad = mu <= 0.0
    if bad.any():
        if np.abs(phi[:, bad]).max() > 0.0:
            return -math.inf
        mu = np.where(bad, 1.0, mu)
    M = -(phi / mu[None, :]) @ phi.T
    cU = np.zeros((n, k))
    for rr in range(len(rows)):
        if rows.kinds[rr] == "trace" or lam[rr] == 0.0:
            continue
        if rows.kinds[rr] == "cut":
            x = rows.xs[rr]
            M += lam[rr] * np.outer(x, x)
        cU += lam[rr] * rows.CU[rr]
        const -= lam[rr] * rows.rhs[rr]
    M -= Q @ Psi3[:r, :r] @ Q.T
    cV = Q.T @ cU - 2.0 * Psi3[:r, r:]
    const -= float(np.trace(Psi3[r:, r:]))
    ev = np.linalg.eigvalsh(0.5 * (M + M.T))
    return const + float(np.minimum(ev[:k], 0.0).sum()) - float(np.linalg.norm(cV, axis=0).sum())


def driver_shor_lists(indices, num_entries_present=(4,), minors=None):
    """The lists the reference's driver attaches to a node (OMC.jl:646-676): the static minor list (or `minors`), and as SOC list
    every coordinate that occurs in none of them.  1-based."""
    n, m = indices.shape
    if minors is None:
        minors = base.shor_constraints_indexes(indices, list(num_entries_present))
    cov = np.zeros((n, m), bool)
    for (i1, i2, j1, j2) in minors:
        cov[i1 - 1, j1 - 1] = cov[i1 - 1, j2 - 1] = cov[i2 - 1, j1 - 1] = cov[i2 - 1, j2 - 1] = True
    soc = [(i + 1, j + 1) for j in range(m) for i in range(n) if not cov[i, j]]     # column-major order of Iterators.product (663)
    return list(minors), soc


# ----------------------------------------------------------------------------------------------------------
# rank k > 1: the explicit extension of (X, W) to the reference's lifted variables (quirk Q5, header)
# ----------------------------------------------------------------------------------------------------------
def complete_shor_rank_k(k, st, X, W):
    """Xt (k, n, m), Wt (k, n, m; meaningful on the minor coordinates), H (k, k, n, m; H[t1, t2] for t1 < t2), and per layer and minor the five
    lifted products V[t, q] = (V1[i1,(j1,j2)], V1[i2,(j1,j2)], V2[(i1,i2),j1], V2[(i1,i2),j2], V3[(i1,i2),(j1,j2)]) of OMC.jl:1526-1551."""
    n, m = X.shape
    nq = st.nq
    Xt = np.zeros((k, n, m)); Xt[0] = X
    xs = X[st.ci, st.cj] if nq else np.zeros((0, 4))
    e = 0.5 * np.abs(xs[:, 1] * xs[:, 2] - xs[:, 0] * xs[:, 3]) if nq else np.zeros(0)       # |X_{i1j2} X_{i2j1} - X_{i1j1} X_{i2j2}| / 2
    s = np.zeros((n, m))
    for p in range(4):
        if nq:
            np.maximum.at(s, (st.ci[:, p], st.cj[:, p]), e)
    s = np.where(st.inC, s * (1.0 + 1e-12) + 1e-300, 0.0)
    b = s / (k - 1)
    Wt = np.zeros((k, n, m)); H = np.zeros((k, k, n, m))
    slack = np.maximum(W - X * X, 0.0)                       # the column slack a minor coordinate may carry
    Wt[0] = np.where(st.inC, X * X + slack + s, 0.0)
    for t in range(1, k):
        Wt[t] = np.where(st.inC, b, 0.0)
        H[0, t] = np.where(st.inC, -b, 0.0)
    V = np.zeros((k, nq, 5))
    if nq:
        V[0, :, 0] = xs[:, 0] * xs[:, 1]; V[0, :, 1] = xs[:, 2] * xs[:, 3]          # V1[i1,(j1,j2)], V1[i2,(j1,j2)]
        V[0, :, 2] = xs[:, 0] * xs[:, 2]; V[0, :, 3] = xs[:, 1] * xs[:, 3]          # V2[(i1,i2),j1], V2[(i1,i2),j2]
        V[0, :, 4] = 0.5 * (xs[:, 0] * xs[:, 3] + xs[:, 1] * xs[:, 2])              # V3: one value at (1,4) and (2,3)
    return dict(Xt=Xt, Wt=Wt, H=H, V=V)


def shor_rank_k_residuals(k, st, X, W, ext):
    """Violation of the k > 1 Shor constraints of the reference (OMC.jl:1787-1826) on an extended point: X = sum_t Xt, W = sum Wt + 2 sum H on
    the minor coordinates, Wt >= 0, the order-5 block of every (layer, minor), the order-(k+1) block of every minor coordinate."""
    Xt, Wt, H, V = ext["Xt"], ext["Wt"], ext["H"], ext["V"]
    res = {}
    res["X_sum"] = float(np.abs(Xt.sum(0) - X).max())
    Hs = sum(H[t1, t2] for t1 in range(k) for t2 in range(t1 + 1, k))
    C = st.inC
    res["W_sum"] = float(np.abs((Wt.sum(0) + 2.0 * Hs - W)[C]).max()) if C.any() else 0.0            # 1787-1791
    res["Wt_nonneg"] = max(0.0, -float(Wt[:, C].min())) if C.any() else 0.0                           # 1528-1531
    worst = 0.0
    for t in range(k):                                                                                 # 1797-1809
        if st.nq == 0:
            break
        M = np.zeros((st.nq, 5, 5)); M[:, 0, 0] = 1.0
        xs = Xt[t][st.ci, st.cj]; ws = Wt[t][st.ci, st.cj]
        for p in range(4):
            M[:, 0, p + 1] = M[:, p + 1, 0] = xs[:, p]; M[:, p + 1, p + 1] = ws[:, p]
        M[:, 1, 2] = M[:, 2, 1] = V[t, :, 0]; M[:, 3, 4] = M[:, 4, 3] = V[t, :, 1]
        M[:, 1, 3] = M[:, 3, 1] = V[t, :, 2]; M[:, 2, 4] = M[:, 4, 2] = V[t, :, 3]
        M[:, 1, 4] = M[:, 4, 1] = V[t, :, 4]; M[:, 2, 3] = M[:, 3, 2] = V[t, :, 4]
        worst = max(worst, -float(np.linalg.eigvalsh(M)[:, 0].min()))
    res["layer_minors"] = max(0.0, worst)
    worst = 0.0
    ii, jj = np.nonzero(C)                                                                             # 1810-1826
    if len(ii):
        B = np.zeros((len(ii), k + 1, k + 1)); B[:, 0, 0] = 1.0
        for t in range(k):
            B[:, 0, t + 1] = B[:, t + 1, 0] = Xt[t][ii, jj]; B[:, t + 1, t + 1] = Wt[t][ii, jj]
        for t1 in range(k):
            for t2 in range(t1 + 1, k):
                B[:, t1 + 1, t2 + 1] = B[:, t2 + 1, t1 + 1] = H[t1, t2][ii, jj]
        worst = -float(np.linalg.eigvalsh(B)[:, 0].min())
    res["coordinate_blocks"] = max(0.0, worst)
    res["max"] = max(res.values())
    return res
