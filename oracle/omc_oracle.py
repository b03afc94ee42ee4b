"""CPU oracle for the per-node hot path of OptimalMatrixCompletion.jl  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path is the HIP library behind
``include/omc.h``; it never calls into this module and fails loudly when the HIP extension is missing.

PARITY UNPINNED (read this before trusting any number):  the reference (`/root/reference`, Julia) solves
every node with JuMP + the closed-source Mosek interior-point solver and ARPACK; neither Julia nor Mosek
exists in this container, the reference ships no tests, fixtures or golden vectors (`test/runtests.jl` is
empty), so nothing produced by the reference itself pins this oracle.  What pins it instead:

  * closed-form known answers derived from the reference's mathematics (KAT-1..KAT-5, see tests/),
  * a *certificate* for every relaxation solve: primal residuals of every cone / row of the reference's
    conic program (OMC.jl:1554-1561, 1564-1685, 1831-1835) evaluated on the returned (X, Y, Theta, U), and a
    Lagrangian dual bound that is valid for ANY multipliers.  ``objective - dual_bound <= tol`` proves the
    returned objective is within ``tol`` of the unique optimum value of the convex program -- the value
    any exact solver (Mosek included) must return.

Citation shorthand:  ``OMC.jl:L`` = /root/reference/src/OptimalMatrixCompletion.jl line L.

What is restated here (function -> reference lines):
  evaluate_objective                    OMC.jl:2330-2359
  compute_SDP_relaxation_objective      OMC.jl:1945-1977
  compute_MSE                           OMC.jl:2373-2409
  cut_piece (piece table incl. quirk Q1) OMC.jl:1580-1678 and 2056-2091
  default_U_bounds                      OMC.jl:1442-1449, 626-632
  sdp_relaxation (the conic program)    OMC.jl:1491-1499, 1554-1561, 1564-1685, 1831-1857, 1860-1942
  master_feasible                       OMC.jl:1261-1277
  breakpoint_vector (separation oracle) OMC.jl:2466-2477
  child_directions                      OMC.jl:2479-2493
  alternating_minimization              OMC.jl:1979-2279
  svd_rounding glue                     OMC.jl:524, 564, 873, 921
  shor_constraints_indexes              OMC.jl:2545-2612
  violated_shor_minors                  OMC.jl:2614-2640

How the relaxation is solved (the reference delegates to Mosek; its algorithm is not in the repo, so the
*problem* is restated and solved by a method of our own -- documented in DESIGN.md section 3):
  Theta and X are eliminated in closed form (Theta = X' pinv(Y) X, column-wise ridge solve), leaving
      f(Y) = 1/2 sum_j a_j' (I + gamma Y[O_j,O_j])^-1 a_j ,     O_j = observed rows of column j,
  minimised over {[Y U; U' I] >= 0, Y <= I, tr Y <= k, box, cut rows} by a consensus ADMM whose blocks are
  (F) one prox per column (1-D secular equation), (C1) PSD projection of order n+k, (C2, k>1) Y <= I,
  (P) exact weighted projection on the polyhedron of linear rows (active-set NNQP).
All arrays are numpy float64; matrices follow the reference's (n x m, column j = one observed column) shape.
"""
from __future__ import annotations

import itertools
import math
import time
from dataclasses import dataclass, field

import numpy as np

# ----------------------------------------------------------------------------------------------------------
# status codes shared with include/omc.h  (mapping onto the MOI statuses the driver branches on,
# OMC.jl:780-785, 809-812, 841, 1866-1940)
# ----------------------------------------------------------------------------------------------------------
OMC_OPTIMAL = 0          # MOI.OPTIMAL / LOCALLY_SOLVED
OMC_SLOW_PROGRESS = 1    # MOI.SLOW_PROGRESS   (iteration cap hit, values available)
OMC_TIME_LIMIT = 2       # MOI.TIME_LIMIT      (values available)
OMC_INFEASIBLE = 3       # MOI.INFEASIBLE family -> "feasible" = false

CUT_LINEAR, CUT_LINEAR2, CUT_LINEAR3 = 0, 1, 2
CUT_TYPES = {"linear": CUT_LINEAR, "linear2": CUT_LINEAR2, "linear3": CUT_LINEAR3}
# direction codes (wire format of include/omc.h)
DIR_LEFT, DIR_MIDDLE, DIR_RIGHT, DIR_INNER_LEFT, DIR_INNER_RIGHT = 0, 1, 2, 3, 4
DIR_CODES = {"left": DIR_LEFT, "middle": DIR_MIDDLE, "right": DIR_RIGHT,
             "inner_left": DIR_INNER_LEFT, "inner_right": DIR_INNER_RIGHT}
DIR_NAMES = {v: k for k, v in DIR_CODES.items()}
DIRECTIONS_OF = {  # OMC.jl:2481-2491 (order matters: children are enumerated in this order)
    "linear": ["left", "right"],
    "linear2": ["left", "middle", "right"],
    "linear3": ["left", "inner_left", "inner_right", "right"],
}


# ----------------------------------------------------------------------------------------------------------
# plain restatements
# ----------------------------------------------------------------------------------------------------------
def evaluate_objective(X, A, indices, gamma):
    """1/2 sum_Omega (X-A)^2 + ||X||_F^2/(2 gamma)   (OMC.jl:2352-2358)."""
    X = np.asarray(X, float); A = np.asarray(A, float)
    if X.shape != A.shape or A.shape != indices.shape:
        raise ValueError("Dimension mismatch (OMC.jl:2337-2348)")
    R = (X - A)[indices]
    return 0.5 * float(R @ R) + float((X * X).sum()) / (2.0 * gamma)


def compute_SDP_relaxation_objective(X, Theta, A, indices, gamma, W=None):
    """OMC.jl:1960-1976.  Without Shor: 1/2 sum_Omega (A-X)^2 + tr(Theta)/(2 gamma)."""
    if W is not None:
        v = 0.5 * float(((A * A - 2 * A * X + W)[indices]).sum())
    else:
        R = (A - X)[indices]
        v = 0.5 * float(R @ R)
    return v + float(np.trace(Theta)) / (2.0 * gamma)


def compute_MSE(X, A, indices, kind="out"):
    """OMC.jl:2373-2409."""
    D2 = (X - A) ** 2
    if kind == "out":
        cnt = indices.size - int(indices.sum())
        return 0.0 if cnt == 0 else float(D2[~indices].sum()) / cnt
    if kind == "in":
        cnt = int(indices.sum())
        return 0.0 if cnt == 0 else float(D2[indices].sum()) / cnt
    if kind == "all":
        return float(D2.sum()) / indices.size
    raise ValueError('kind must be one of "out", "in", "all" (OMC.jl:2404-2407)')


def default_U_bounds(n, k):
    """U in [-1,1] with symmetry breaking U[n-k+i:n, i] >= 0  (OMC.jl:1442-1449; 0-based here)."""
    lo = -np.ones((n, k)); hi = np.ones((n, k))
    for i in range(k):
        lo[n - k + i:, i] = 0.0
    return lo, hi


def cut_piece(cut_type, direction, vhat, reference_quirk_q1=True):
    """One piece of the piecewise-linear over-estimator of v^2 (OMC.jl:1580-1678).

    Returns (lo, hi, slope, intercept):  lo <= v <= hi  and  g(v) = slope*v + intercept.
    ``reference_quirk_q1``: linear3/right uses g = a*v exactly as OMC.jl:1675 does (it under-estimates
    v^2 on [a,1]); False gives the secant (1+a) v - a that linear2/right has at OMC.jl:1633.
    """
    a = abs(vhat)
    if cut_type == "linear":
        if direction == "left":
            return -1.0, vhat, vhat - 1.0, vhat            # -v + vhat v + vhat       (1591)
        if direction == "right":
            return vhat, 1.0, vhat + 1.0, -vhat            # +v + vhat v - vhat       (1601)
    elif cut_type == "linear2":
        if direction == "left":
            return -1.0, -a, -(1.0 + a), -a                # -v - a v - a             (1613)
        if direction == "middle":
            return -a, a, 0.0, vhat * vhat                 # vhat^2                   (1623)
        if direction == "right":
            return a, 1.0, 1.0 + a, -a                     # +v + a v - a             (1633)
    elif cut_type == "linear3":
        if direction == "left":
            return -1.0, -a, -(1.0 + a), -a                # (1645)
        if direction == "inner_left":
            return -a, 0.0, -a, 0.0                        # -a v                     (1655)
        if direction == "inner_right":
            return 0.0, a, a, 0.0                          # a v                      (1665)
        if direction == "right":
            if reference_quirk_q1:
                return a, 1.0, a, 0.0                      # a v   (quirk Q1)         (1675)
            return a, 1.0, 1.0 + a, -a
    else:
        raise ValueError("Invalid input for disjunctive cuts type (OMC.jl:1456-1462)")
    raise ValueError(f"direction {direction!r} invalid for cut type {cut_type!r}")


def child_directions(cut_type, k):
    """Cartesian product of direction labels, FIRST column varying fastest (Iterators.product,
    OMC.jl:2481-2491).  Child ``ind`` (1-based) gets node_id = counter + ind (OMC.jl:2524)."""
    labels = DIRECTIONS_OF[cut_type]
    out = []
    for combo in itertools.product(*([labels] * k)):      # itertools: LAST varies fastest
        out.append(list(combo[::-1]))                      # reverse -> first varies fastest
    return out


def master_feasible(Y, U, projection_tolerance=1e-6):
    """lambda_min(U U' - Y) >= -1e-6   (disjunctive branch, OMC.jl:1272-1277)."""
    lam = np.linalg.eigvalsh(U @ U.T - Y)
    return bool(lam[0] >= -projection_tolerance), float(lam[0])


def breakpoint_vector(Y, U, breakpoints="smallest_1_eigvec"):
    """Separation oracle (OMC.jl:2466-2477).  Eigenvector sign is arbitrary (ARPACK); we fix it by making
    the largest-magnitude entry positive so CPU and GPU agree.  Returns (x, eigvals[:2])."""
    S = U @ U.T - Y
    S = 0.5 * (S + S.T)
    w, V = np.linalg.eigh(S)

    def canon(v):
        i = int(np.argmax(np.abs(v)))
        return v if v[i] >= 0 else -v

    e1 = canon(V[:, 0])
    if breakpoints == "smallest_1_eigvec":
        return e1, w[:2].copy()
    if breakpoints == "smallest_2_eigvec":
        if w[1] < -1e-10:
            e2 = canon(V[:, 1])
            wt = np.abs(w[:2]) / math.sqrt(float((w[:2] ** 2).sum()))
            return wt[0] * e1 + wt[1] * e2, w[:2].copy()
        return e1, w[:2].copy()
    raise ValueError("Invalid input for disjunctive cuts breakpoints (OMC.jl:2440-2446)")


def svd_rounding(M, k):
    """svd(M).U[:, 1:k]  (OMC.jl:524, 564, 873, 921), sign-canonicalised like breakpoint_vector."""
    Uf, _, _ = np.linalg.svd(M, full_matrices=False)
    Uk = Uf[:, :k].copy()
    for j in range(k):
        i = int(np.argmax(np.abs(Uk[:, j])))
        if Uk[i, j] < 0:
            Uk[:, j] = -Uk[:, j]
    return Uk


# ----------------------------------------------------------------------------------------------------------
# Shor minors (integer work)   OMC.jl:2545-2640
# ----------------------------------------------------------------------------------------------------------
def shor_constraints_indexes(indices, num_entries_present_list):
    """generate_rank1_matrix_completion_Shor_constraints_indexes (OMC.jl:2545-2612).
    Returns a list of 1-based (i1, i2, j1, j2) in the reference's push order."""
    n, m = indices.shape
    out = []
    rows = [indices[i, :] for i in range(n)]
    for p in num_entries_present_list:
        if p == 4:
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    both = np.flatnonzero(rows[i1] & rows[i2])
                    for a in range(len(both)):
                        for b in range(a + 1, len(both)):
                            out.append((i1 + 1, i2 + 1, int(both[a]) + 1, int(both[b]) + 1))
        elif p == 3:
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    both = np.flatnonzero(rows[i1] & rows[i2]); xor = np.flatnonzero(rows[i1] ^ rows[i2])
                    for j1 in both:
                        for j2 in xor:
                            lo_, hi_ = sorted((int(j1), int(j2)))
                            out.append((i1 + 1, i2 + 1, lo_ + 1, hi_ + 1))
        elif p == 2:
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    both = np.flatnonzero(rows[i1] & rows[i2]); none = np.flatnonzero(~(rows[i1] | rows[i2]))
                    for j1 in both:
                        for j2 in none:
                            lo_, hi_ = sorted((int(j1), int(j2)))
                            out.append((i1 + 1, i2 + 1, lo_ + 1, hi_ + 1))
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    xor = np.flatnonzero(rows[i1] ^ rows[i2])
                    for a in range(len(xor)):
                        for b in range(a + 1, len(xor)):
                            out.append((i1 + 1, i2 + 1, int(xor[a]) + 1, int(xor[b]) + 1))
        elif p == 1:
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    none = np.flatnonzero(~(rows[i1] | rows[i2]))
                    for j1 in range(m):
                        if int(rows[i1][j1]) + int(rows[i2][j1]) == 1:
                            for j2 in none:
                                lo_, hi_ = sorted((j1, int(j2)))
                                out.append((i1 + 1, i2 + 1, lo_ + 1, hi_ + 1))
        elif p == 0:
            for i1 in range(n):
                for i2 in range(i1 + 1, n):
                    free = ~(rows[i1] | rows[i2])
                    for j1 in range(m - 1):
                        if free[j1]:
                            for j2 in np.flatnonzero(free[j1 + 1:]):
                                out.append((i1 + 1, i2 + 1, j1 + 1, j1 + 1 + int(j2) + 1))
    return out


def violated_shor_minors(X3, indices, num_entries_present_list, existing, n_minors):
    """generate_violated_Shor_minors (OMC.jl:2614-2640).  X3 has shape (k, n, m).  Returns the list of
    (score, (i1,i2,j1,j2)) sorted by decreasing score, truncated to n_minors."""
    cand = shor_constraints_indexes(indices, num_entries_present_list)
    ex = set(existing)
    seen = set(); uniq = []
    for t in cand:                      # setdiff! keeps first occurrences, drops `existing`
        if t in ex or t in seen:
            continue
        seen.add(t); uniq.append(t)
    scored = []
    for (i1, i2, j1, j2) in uniq:
        terms = np.abs(X3[:, i1 - 1, j1 - 1] * X3[:, i2 - 1, j2 - 1] - X3[:, i1 - 1, j2 - 1] * X3[:, i2 - 1, j1 - 1])
        s = 0.0
        for t in range(terms.shape[0]):     # left-to-right, as Julia's sum over a short vector
            s = s + float(terms[t])
        scored.append((s, (i1, i2, j1, j2)))
    scored.sort(key=lambda t: (t[0], t[1]), reverse=True)   # Julia sorts tuples lexicographically, rev
    return scored[:n_minors] if len(scored) >= n_minors else scored


# ----------------------------------------------------------------------------------------------------------
# small dense helpers
# ----------------------------------------------------------------------------------------------------------
def nnqp(G, c, tol=1e-13):
    """min 1/2 l'Gl - c'l  s.t. l >= 0, G symmetric psd (possibly singular: parallel rows).
    Lawson-Hanson active set written for the QP form.  Exact up to round-off."""
    R = len(c)
    lam = np.zeros(R); P = np.zeros(R, bool)
    if R == 0:
        return lam
    scale = max(float(np.abs(c).max()), 1e-300)
    for _ in range(10 * R + 10):
        w = c - G @ lam
        w[P] = -np.inf
        t = int(np.argmax(w))
        if w[t] <= tol * scale:
            break
        P[t] = True
        for _ in range(10 * R + 10):
            idx = np.flatnonzero(P)
            if len(idx) == 0:                    # every passive index was dropped again (degenerate rows): back to the outer test
                break
            s = np.zeros(R)
            s[idx] = np.linalg.lstsq(G[np.ix_(idx, idx)], c[idx], rcond=None)[0]
            if s[idx].min() > 0:
                lam = s
                break
            neg = idx[s[idx] <= 0]
            al = float(np.min(lam[neg] / (lam[neg] - s[neg])))
            lam = lam + al * (s - lam)
            drop = neg[(lam[neg] <= 1e-18 * scale)]
            if len(drop) == 0:
                drop = neg[[int(np.argmin(lam[neg]))]]
            P[drop] = False
            lam[~P] = 0.0
    return lam


def _psd_split(M):
    w, V = np.linalg.eigh(0.5 * (M + M.T))
    return w, V


# ----------------------------------------------------------------------------------------------------------
# instance
# ----------------------------------------------------------------------------------------------------------
class Instance:
    """(A, indices, gamma, k) plus the per-column index lists the solver gathers with."""

    def __init__(self, A, indices, gamma, k):
        A = np.asarray(A, float); indices = np.asarray(indices, bool)
        if A.shape != indices.shape:
            raise ValueError("Dimension mismatch: A and indices must both be (n, m) (OMC.jl:240-246)")
        n, m = A.shape
        if not n <= m:
            raise ValueError("Input matrix A must have size (n, m) with n <= m (OMC.jl:249-254)")
        self.A, self.indices, self.gamma, self.k, self.n, self.m = A, indices, float(gamma), int(k), n, m
        self.cols = [np.flatnonzero(indices[:, j]) for j in range(m)]
        Mf = indices.astype(float)
        self.N = Mf @ Mf.T                       # N[i,i'] = #columns observing both rows
        self.sumA2 = float((A[indices] ** 2).sum())
        # group columns by count for batched linear algebra
        cs = np.array([len(o) for o in self.cols])
        self.groups = []
        for c in np.unique(cs):
            if c == 0:
                continue
            js = np.flatnonzero(cs == c)
            idx = np.stack([self.cols[j] for j in js])
            a = np.stack([A[self.cols[j], j] for j in js])
            self.groups.append((int(c), js, idx, a))

    # f(Y), X(Y), alpha(Y) --------------------------------------------------------------------------------
    def f_value(self, Y, want=False):
        """f(Y) = 1/2 sum_j a_j'(I + gamma Y_jj)^-1 a_j ; optionally alpha (n x m, zero off-support)."""
        g = self.gamma; v = 0.0
        Lam = np.zeros((self.n, self.m)) if want else None
        for (c, js, idx, a) in self.groups:
            B = np.eye(c)[None] + g * Y[idx[:, :, None], idx[:, None, :]]
            try:
                np.linalg.cholesky(B)          # f is +inf outside {I + gamma Y_jj > 0} (the HIP path reports 1e300)
            except np.linalg.LinAlgError:
                return (math.inf, Lam) if want else math.inf
            al = np.linalg.solve(B, a[:, :, None])[:, :, 0]
            v += 0.5 * float((a * al).sum())
            if want:
                Lam[idx, js[:, None]] = al
        return (v, Lam) if want else v

    def X_of(self, Y, Lam):
        """X[:, j] = gamma * Y[:, O_j] alpha_j  (closed-form minimiser over X of the reference objective)."""
        return self.gamma * (Y @ Lam)


# ----------------------------------------------------------------------------------------------------------
# linear rows of a node  (OMC.jl:1558, 1561, 1564-1685)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class Rows:
    xs: list = field(default_factory=list)     # for "cut" rows: the breakpoint vector (Y part = x x')
    kinds: list = field(default_factory=list)  # trace | box_lo | box_hi | hi | lo | cut
    CU: list = field(default_factory=list)     # (n,k) coefficient on U
    rhs: list = field(default_factory=list)
    meta: list = field(default_factory=list)

    def add(self, kind, x, CU, rhs, meta=None):
        self.kinds.append(kind); self.xs.append(x); self.CU.append(CU); self.rhs.append(float(rhs))
        self.meta.append(meta)

    def __len__(self):
        return len(self.kinds)


def build_rows(inst, cuts, cut_type, U_lower=None, U_upper=None, reference_quirk_q1=True):
    """All '<=' rows of the node in (Y, U):   <CY, Y> + <CU, U> <= rhs.
    trace: tr Y <= k (1558).  box: only entries whose bound is not implied by ||U_j|| <= 1, i.e. lower > -1 or
    upper < 1 (1561).  Per cut l and column j: x'U_j <= hi, -x'U_j <= -lo (1580-1678), and the aggregated row
    x'Yx - sum_j slope_j x'U_j <= sum_j intercept_j (1680-1683)."""
    n, k = inst.n, inst.k
    if U_lower is None or U_upper is None:
        dlo, dhi = default_U_bounds(n, k)
        U_lower = dlo if U_lower is None else U_lower
        U_upper = dhi if U_upper is None else U_upper
    if U_lower.shape != (n, k) or U_upper.shape != (n, k):
        raise ValueError("Dimension mismatch: U_lower/U_upper must be (n, k) (OMC.jl:1465-1477)")
    rows = Rows()
    rows.add("trace", None, np.zeros((n, k)), float(k))
    for j in range(k):
        for i in range(n):
            if U_lower[i, j] > -1.0:
                CU = np.zeros((n, k)); CU[i, j] = -1.0
                rows.add("box_lo", None, CU, -U_lower[i, j], (i, j))
            if U_upper[i, j] < 1.0:
                CU = np.zeros((n, k)); CU[i, j] = 1.0
                rows.add("box_hi", None, CU, U_upper[i, j], (i, j))
    for l, (x, Uhat, dirs) in enumerate(cuts):
        x = np.asarray(x, float); Uhat = np.asarray(Uhat, float)
        if x.shape != (n,) or Uhat.shape != (n, k) or len(dirs) != k:
            raise ValueError("cut must be (x in R^n, Uhat in R^{n x k}, k directions) (OMC.jl:34)")
        vhat = Uhat.T @ x                                           # OMC.jl:1577
        CUc = np.zeros((n, k)); rhs = 0.0
        for j in range(k):
            d = dirs[j] if isinstance(dirs[j], str) else DIR_NAMES[int(dirs[j])]
            lo, hi, sl, ic = cut_piece(cut_type, d, float(vhat[j]), reference_quirk_q1)
            CUc[:, j] = -sl * x; rhs += ic
            CU = np.zeros((n, k)); CU[:, j] = x
            rows.add("hi", None, CU, hi, (l, j))
            rows.add("lo", None, -CU, -lo, (l, j))
        rows.add("cut", x, CUc, rhs, (l,))
    return rows


# ----------------------------------------------------------------------------------------------------------
# the relaxation solve
# ----------------------------------------------------------------------------------------------------------
@dataclass
class RelaxParams:
    eps_gap: float = 1e-6        # stop when (objective - dual_bound) <= eps_gap * max(1,|objective|) ...
    eps_feas: float = 1e-7       # ... and the cone residual rp <= eps_feas * sqrt(n+k)
    max_iters: int = 3000
    check_every: int = 25
    rho_scale: float = 1.0       # rho_0 = rho_scale * gamma/2 * ||A_Omega||^2 / (m (1 + gamma k/n)^2)
    rho_f_ratio: float = 0.1     # rho of the per-column blocks relative to the cone blocks
    relax: float = 1.6           # over-relaxation
    adapt: int = 0               # 1: residual balancing of rho at check points (first adapt_until iterations)
    adapt_until: int = 1500
    stall_checks: int = 8        # stop with SLOW_PROGRESS when the objective is stationary for this many checks
    time_limit: float = 3600.0
    reference_quirk_q1: bool = True
    rho_init: float = 0.0        # > 0: start from this penalty (e.g. the parent's final rho)
    bump: int = 1                # 1: multiply rho by bump_factor when the primal residual exceeds bump_ratio x the dual residual
    bump_factor: float = 4.0
    bump_window: int = 4         # checks between two bumps
    bump_after: int = 100        # first iteration at which a bump may happen
    bump_max: int = 2            # more than two bumps (x16) over-penalise deep nodes: measured on config 2, depth 9 (tools/gpu_bump2.py)
    bump_ratio: float = 4.0
    # Anderson acceleration (type II) of the fixed-point map, mirrored by k_aa in omc_device.hip
    accel: int = 0               # off by default (helps small / ill-conditioned nodes, not config 2)
    aa_mem: int = 10             # residual differences kept
    aa_every: int = 10           # an extrapolated point every this many iterations ...
    aa_start: int = 50           # ... from this iteration on
    aa_reg: float = 1e-10        # Tikhonov weight of the normal equations, relative to the mean diagonal
    aa_safeguard: float = 1.0    # the point is kept when the next fixed-point residual <= this x the last one
    # early SLOW_PROGRESS (mirrors k_check_final): the gap cannot close before max_iters at the rate of the last checks
    early_stop_after: int = 400
    early_stop_factor: float = 1.5   # 0 = off


def _prox_columns(inst, Yx, alpha, svals, rho_f):
    """(F) block.  For every column j solve   alpha = (B_j + c s I)^-1 a_j,  s = ||alpha||^2,
    B_j = I + gamma (Yx[O_j,O_j] - gamma/(2 rho_f) alpha_old alpha_old'),  c = gamma^2/(2 rho_f).
    Returns new alpha (list per group), s values, and LL = sum_j E_j' alpha alpha' E_j."""
    g = inst.gamma; n = inst.n
    cp = g * g / (2.0 * rho_f)
    LL = np.zeros((n, n)); out = []; sout = []
    for gi, (c, js, idx, a) in enumerate(inst.groups):
        al = alpha[gi]
        Z = Yx[idx[:, :, None], idx[:, None, :]] - (g / (2.0 * rho_f)) * al[:, :, None] * al[:, None, :]
        B = np.eye(c)[None] + g * Z
        b, Q = np.linalg.eigh(B)
        qa = np.einsum("gij,gi->gj", Q, a); qa2 = qa * qa
        lo = np.maximum(0.0, -b[:, 0] / cp) * (1.0 + 1e-12)
        hi = np.maximum(2.0 * lo + 1.0, 1.0)

        def phi(s_):
            d = b + cp * s_[:, None]
            return (qa2 / d ** 2).sum(1) - s_, -2.0 * cp * (qa2 / d ** 3).sum(1) - 1.0

        for _ in range(200):
            ph, _d = phi(hi)
            bad = ph > 0
            if not bad.any():
                break
            hi = np.where(bad, hi * 2.0, hi)
        s = svals[gi] if svals[gi] is not None else hi
        s = np.where((s > lo) & (s < hi), s, hi)
        for _ in range(100):
            ph, dph = phi(s)
            lo = np.where(ph > 0, s, lo); hi = np.where(ph <= 0, s, hi)
            sn = s - ph / dph
            sn = np.where((sn > lo) & (sn < hi), sn, 0.5 * (lo + hi))
            done = np.abs(sn - s) <= 1e-15 * np.maximum(1.0, np.abs(s))
            s = sn
            if done.all():
                break
        aln = np.einsum("gij,gj->gi", Q, qa / (b + cp * s[:, None]))
        out.append(aln); sout.append(s)
        np.add.at(LL, (idx[:, :, None], idx[:, None, :]), aln[:, :, None] * aln[:, None, :])
    return out, sout, LL


def row_subspace(rows, n, k, tol=1e-10):
    """Orthonormal basis Q (n x r) of the span of every U-functional used by the rows (cut vectors x_l and
    the coordinate vectors of non-default box entries).  Only v = X'U enters the rows, so the order-(n+k)
    cone [Y U;U' I] >= 0 is equivalent to Y >= 0 plus the order-(r+k) cone [Q'YQ  Q'U; U'Q  I] >= 0
    (Schur complement; U = Y Q (Q'YQ)^+ Q'U recovers a U with U U' <= Y and the same X'U)."""
    cols = []
    for r in range(len(rows)):
        if rows.kinds[r] == "trace":
            continue
        CU = rows.CU[r]
        for j in range(k):
            if np.abs(CU[:, j]).max() > 0:
                cols.append(CU[:, j] / np.linalg.norm(CU[:, j]))
    if not cols:
        return np.zeros((n, 0))
    # modified Gram-Schmidt with re-orthogonalisation, in row order (the HIP host code does the same)
    Q = []
    for c in cols:
        v = c.copy()
        for _ in range(2):
            for q in Q:
                v -= (q @ v) * q
        nv = np.linalg.norm(v)
        if nv > tol:
            Q.append(v / nv)
    return np.stack(Q, 1) if Q else np.zeros((n, 0))


def dual_bound_from(inst, Lam, rows, lam, Q, Psi3):
    """Valid lower bound on the relaxation optimum for ANY Lam (support in Omega), lam >= 0, Psi3 >= 0.

    f(Y) >= <A,Lam> - 1/2||Lam||^2 - gamma/2 <Y, Lam Lam'>        (Fenchel; equality at Lam = alpha(Y))
    L    =  <G,Y> + sum_r lam_r (row_r - rhs_r) - <Psi3, [Q'YQ Vt; Vt' I]>,     Vt = Q'U
    minimised over the simple superset {0<=Y<=I, trY<=k} x {||Vt_j||<=1} (both implied by the constraints):
        sum_{i<=k} min(eig_i(M),0) - sum_j ||c_j|| + const.
    """
    n, k, g = inst.n, inst.k, inst.gamma
    r = Q.shape[1]
    c0 = float((inst.A * Lam).sum()) - 0.5 * float((Lam * Lam).sum())
    M = -0.5 * g * (Lam @ Lam.T)
    cU = np.zeros((n, k)); const = 0.0
    for rr in range(len(rows)):
        if rows.kinds[rr] == "trace" or lam[rr] == 0.0:
            continue                               # trace is kept in the simple set
        if rows.kinds[rr] == "cut":
            x = rows.xs[rr]
            M += lam[rr] * np.outer(x, x)
        cU += lam[rr] * rows.CU[rr]
        const -= lam[rr] * rows.rhs[rr]
    M -= Q @ Psi3[:r, :r] @ Q.T
    cV = Q.T @ cU - 2.0 * Psi3[:r, r:]
    const -= float(np.trace(Psi3[r:, r:]))
    ev = np.linalg.eigvalsh(0.5 * (M + M.T))
    return c0 + float(np.minimum(ev[:k], 0.0).sum()) - float(np.linalg.norm(cV, axis=0).sum()) + const


def primal_residuals(inst, rows, Y, U, X, Theta):
    """Violation of every cone / row of the reference's program on (X, Y, Theta, U) (>= 0, 0 = feasible)."""
    n, m, k = inst.n, inst.m, inst.k
    res = {}
    big = np.block([[Y, X], [X.T, Theta]])
    res["psd_YXTheta"] = max(0.0, -float(np.linalg.eigvalsh(0.5 * (big + big.T))[0]))        # 1554
    yu = np.block([[Y, U], [U.T, np.eye(k)]])
    res["psd_YUI"] = max(0.0, -float(np.linalg.eigvalsh(0.5 * (yu + yu.T))[0]))              # 1555
    res["psd_I_minus_Y"] = max(0.0, float(np.linalg.eigvalsh(0.5 * (Y + Y.T))[-1]) - 1.0)    # 1556
    res["trace"] = max(0.0, float(np.trace(Y)) - k)                                          # 1558
    res["soc_cols"] = max(0.0, float(np.linalg.norm(U, axis=0).max()) - 1.0)                 # 1831-1835
    worst = 0.0
    for r in range(len(rows)):
        if rows.kinds[r] == "trace":
            continue
        v = float((rows.CU[r] * U).sum())
        if rows.kinds[r] == "cut":
            x = rows.xs[r]; v += float(x @ Y @ x)
        worst = max(worst, v - rows.rhs[r])
    res["rows"] = max(0.0, worst)                                                            # 1561-1685
    res["max"] = max(res.values())
    return res


ADAPT_AT = (50, 100, 150, 200, 300, 400, 600, 800, 1200, 1600)


def initial_rho(inst, p):
    g, n, m, k = inst.gamma, inst.n, inst.m, inst.k
    if p.rho_init > 0:
        return p.rho_init
    rho = p.rho_scale * 0.5 * g * inst.sumA2 / (m * (1.0 + g * k / n) ** 2)
    return rho if rho > 0 else 1.0


def recover_U(Y, Q, Vt):
    """U = Y Q (Q'YQ)^+ Vt : satisfies U U' <= Y and Q'U = Vt whenever [Q'YQ Vt; Vt' I] >= 0."""
    n = Y.shape[0]
    if Q.shape[1] == 0:
        return np.zeros((n, Vt.shape[1]))
    S = Q.T @ Y @ Q
    w, V = np.linalg.eigh(0.5 * (S + S.T))
    keep = w > 1e-12 * max(1.0, w[-1])
    Sp = (V[:, keep] / w[keep]) @ V[:, keep].T
    return Y @ (Q @ (Sp @ Vt))


def sdp_relaxation(inst, cuts=(), cut_type="linear", U_lower=None, U_upper=None, params=None,
                   warm=None, want_certificate=True):
    """matrix_completion_SDP_relaxation restated (disjunctive mode, no Shor).  Returns a dict with the
    reference's keys (objective, Y, U, X, Θ->"Theta", feasible, termination_status, solve_time) plus
    dual_bound, iters, residuals and the warm-start state.

    Consensus ADMM on (Y, U = Q Vt):  blocks (F) columns, (C1) 0 <= Y <= I, (C3) [Q'YQ Vt;Vt' I] >= 0,
    global step = weighted projection on the linear rows.  Scaled duals D1, D3 (n x n), D3V (r x k), D3T (k x k).
    """
    if cut_type not in CUT_TYPES:
        raise ValueError("Invalid input for disjunctive cuts type (OMC.jl:1456-1462)")
    p = params or RelaxParams()
    t0 = time.time()
    n, m, k, g = inst.n, inst.m, inst.k, inst.gamma
    rows = build_rows(inst, cuts, cut_type, U_lower, U_upper, p.reference_quirk_q1)
    R = len(rows)
    Q = row_subspace(rows, n, k)
    r = Q.shape[1]
    rho = initial_rho(inst, p)
    wY1 = p.rho_f_ratio * inst.N + 2.0            # weights for rho = 1
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); b = np.array(rows.rhs)
    for rr in range(R):
        if rows.kinds[rr] == "trace":
            AY[rr] = np.eye(n).ravel()
        elif rows.kinds[rr] == "cut":
            AY[rr] = np.outer(rows.xs[rr], rows.xs[rr]).ravel()
        AU[rr] = rows.CU[rr].ravel()
    G1 = (AY / wY1.ravel()) @ AY.T + (AU / 2.0) @ AU.T          # Gram matrix for rho = 1
    if warm is not None:
        # warm start from the parent's final state (round 3; mirrored by k_setup in omc_device.hip): the child keeps its own base penalty,
        # so the parent's scaled duals are rescaled by rho_parent / rho_child (the multipliers rho D stay what they were); Yp = Y;
        # the small-cone duals and the row multipliers start from zero (the child has a row basis of its own: one more cut)
        Y = warm["Y"].copy(); Vt = Q.T @ warm["U"]; Yp = Y.copy()
        alpha = [a.copy() for a in warm["alpha"]]; svals = list(warm["svals"])
        f_w = float(warm.get("rho", rho)) / rho
        D1 = warm["D1"] * f_w; D3 = warm["D3"] * f_w
    else:
        Y = np.eye(n) * (k / n); Vt = np.zeros((r, k)); Yp = Y.copy()
        alpha = [np.zeros_like(a) for (_, _, _, a) in inst.groups]; svals = [None] * len(inst.groups)
        D1 = np.zeros((n, n)); D3 = np.zeros((n, n))
    D3V = np.zeros((r, k)); D3T = np.zeros((k, k))
    Ik = np.eye(k)
    lam = np.zeros(R)
    status = OMC_SLOW_PROGRESS
    obj = math.inf; lb = -math.inf; rp = rd = math.inf
    hist = []
    it = 0; rx = p.relax
    stall = 0; obj_prev = math.inf; lb_prev = -math.inf
    n_bumps = 0; last_bump = 0
    gap_prev = 1e300; gap_rate = 1.0; slow_votes = 0
    Q3 = np.zeros((r + k, r + k))
    # Anderson acceleration: state z = (Y, Yp, D1, D3, Vt, D3V, D3T, alpha); ring of (f, g) pairs; see k_aa
    def aa_pack():
        return np.concatenate([Y.ravel(), Yp.ravel(), D1.ravel(), D3.ravel(), Vt.ravel(), D3V.ravel(), D3T.ravel()] + [a.ravel() for a in alpha])
    def aa_unpack(z):
        o = 0; outm = []
        for shp in ((n, n), (n, n), (n, n), (n, n), (r, k), (r, k), (k, k)):
            sz = shp[0] * shp[1]; outm.append(z[o:o + sz].reshape(shp).copy()); o += sz
        al = []
        for a in alpha:
            al.append(z[o:o + a.size].reshape(a.shape).copy()); o += a.size
        return outm, al
    aa_valid = False; aa_F = []; aa_G = []; aa_zin = None; aa_pending = None; n_aa = n_aa_rej = n_aa_acc = 0; aa_off = False
    for it in range(1, p.max_iters + 1):
        rho_f = rho * p.rho_f_ratio
        alpha, svals, LL = _prox_columns(inst, 2.0 * Y - Yp, alpha, svals, rho_f)
        # (C1) spectral clip of Y - D1 to [0, 1]
        w1, V1 = _psd_split(Y - D1)
        W1 = (V1 * np.clip(w1, 0.0, 1.0)) @ V1.T
        # (C3) small cone
        Min = Y - D3
        S_in = Q.T @ Min @ Q; V_in = Vt - D3V
        M3 = np.block([[S_in, V_in], [V_in.T, Ik - D3T]])
        w3, V3 = _psd_split(M3)
        P3 = (V3 * np.maximum(w3, 0.0)) @ V3.T
        Q3 = P3 - 0.5 * (M3 + M3.T)
        dS = Q3[:r, :r]; W3V = P3[:r, r:]; W3T = P3[r:, r:]
        # global target
        tY = (rho_f * (inst.N * Y) + 0.5 * g * LL + rho * (rx * W1 + (1.0 - rx) * Y + D1)
              + rho * (Y + (1.0 - rx) * D3 + rx * (Q @ dS @ Q.T))) / (rho * wY1)
        tV = rx * W3V + (1.0 - rx) * Vt + D3V
        tU = Q @ tV
        c = AY @ tY.ravel() + AU @ tU.ravel() - b
        mu = nnqp(G1, c)                                  # multipliers / rho
        lam = rho * mu
        Yn = tY - ((AY.T @ mu) / wY1.ravel()).reshape(n, n)
        Yn = 0.5 * (Yn + Yn.T)
        Vn = tV - Q.T @ ((AU.T @ mu) / 2.0).reshape(n, k)
        D1 = D1 + rx * W1 + (1.0 - rx) * Y - Yn
        D3 = (1.0 - rx) * D3 + Y + rx * (Q @ dS @ Q.T) - Yn
        D3V = D3V + rx * W3V + (1.0 - rx) * Vt - Vn
        D3T = D3T + rx * (W3T - Ik)
        W3Y_minus_Yn = Min + Q @ dS @ Q.T - Yn
        rp = math.sqrt(float(np.linalg.norm(W1 - Yn) ** 2 + np.linalg.norm(W3Y_minus_Yn) ** 2
                             + 2.0 * np.linalg.norm(W3V - Vn) ** 2 + np.linalg.norm(W3T - Ik) ** 2))
        rd = math.sqrt(float(np.linalg.norm(Yn - Y) ** 2 + 2.0 * np.linalg.norm(Vn - Vt) ** 2))
        Yp = Y; Y = Yn; Vt = Vn
        if it % p.check_every == 0 or it == p.max_iters:
            obj, Lam = inst.f_value(Y, want=True)
            lb_new = dual_bound_from(inst, Lam, rows, lam, Q, rho * Q3)
            lb = max(lb, lb_new)
            hist.append((it, obj, lb, rp, rd, rho))
            # two-sided: an eps-feasible iterate can sit below the certified bound; that is not a certificate of eps_gap accuracy
            if abs(obj - lb) <= p.eps_gap * max(1.0, abs(obj)) and rp <= p.eps_feas * math.sqrt(n + k):
                status = OMC_OPTIMAL
                break
            if lb > 0.5 * inst.sumA2 * (1.0 + 1e-9) + 1e-9:   # f(Y) <= f(0) = 1/2||A_Omega||^2 for feasible Y
                status = OMC_INFEASIBLE
                break
            if abs(obj - obj_prev) <= 1e-7 * max(1.0, abs(obj)) and lb_new <= lb_prev + 1e-7 * max(1.0, abs(obj)):
                stall += 1
            else:
                stall = 0
            obj_prev = obj; lb_prev = lb
            if stall >= p.stall_checks:
                if abs(obj - lb) <= p.eps_gap * max(1.0, abs(obj)) and rp <= 10.0 * p.eps_feas * math.sqrt(n + k):
                    status = OMC_OPTIMAL                               # certified gap, residual within 10x of its target
                break                                                  # else SLOW_PROGRESS, values available
            if time.time() - t0 > p.time_limit:
                status = OMC_TIME_LIMIT
                break
            if it >= p.max_iters:
                break
            if p.early_stop_factor > 0.0:
                target = p.eps_gap * max(1.0, abs(obj))
                gnow = obj - lb
                q = 0.5 * gap_rate + 0.5 * min(gnow / gap_prev, 2.0) if (gap_prev < 1e299 and gap_prev > 0.0 and gnow > 0.0) else 1.0
                gap_prev = gnow; gap_rate = q
                left = (p.max_iters - it) / float(p.check_every)
                need = math.log(max(gnow, target) / target) / -math.log(q) if q < 1.0 else 1e300
                hopeless = it >= p.early_stop_after and gnow > target and need > p.early_stop_factor * left
                slow_votes = slow_votes + 1 if hopeless else 0
                if slow_votes >= 8:
                    break                                              # SLOW_PROGRESS now: values and the (valid) bound are returned
            # penalty bump: nodes with active cuts want a larger rho than the root-tuned one.  Signal (measured on config 2):
            # crawling nodes sit at rp/rd = 10..200 at iteration 400, healthy ones at 0.4..4
            if (p.bump and it >= p.bump_after and n_bumps < p.bump_max and it - last_bump >= p.check_every * p.bump_window
                    and rp > p.bump_ratio * rd):
                fac = p.bump_factor
                rho *= fac
                D1 /= fac; D3 /= fac; D3V /= fac; D3T /= fac
                n_bumps += 1; last_bump = it
                slow_votes = 0; gap_rate = 1.0                         # a new penalty changes the rate: the prediction starts over
                aa_valid = False                                       # the map changed: restart the history
        # ---- Anderson acceleration (mirrors k_aa: runs after the certificate, on the state the next iteration reads) ----
        if p.accel and it >= p.aa_start - 1 and not aa_off:
            if not aa_valid:
                aa_zin = aa_pack(); aa_F = []; aa_G = []; aa_pending = None; aa_valid = True
                continue
            gz = aa_pack(); f = gz - aa_zin; fn = float(np.linalg.norm(f))
            if aa_pending is not None and not (fn <= p.aa_safeguard * aa_pending[0]):
                (Y, Yp, D1, D3, Vt, D3V, D3T), alpha = aa_unpack(aa_pending[1])          # reject: back to the last plain image
                aa_zin = aa_pending[1].copy(); aa_F = []; aa_G = []; aa_pending = None; n_aa_rej += 1
                if n_aa_rej >= 4 and n_aa_rej > n_aa_acc:           # this node keeps rejecting its points: stop paying for the history
                    aa_off = True
                continue
            if aa_pending is not None:
                n_aa_acc += 1
            aa_pending = None
            aa_F.append(f); aa_G.append(gz)
            if len(aa_F) > p.aa_mem + 1:
                aa_F.pop(0); aa_G.pop(0)
            done_aa = False
            if it % p.aa_every == 0 and len(aa_F) >= 3:
                F_ = np.array(aa_F); G_ = np.array(aa_G)
                dF = F_[1:] - F_[:-1]; dG = G_[1:] - G_[:-1]
                H = dF @ dF.T; rhs = dF @ f
                tr = float(np.trace(H))
                if 0.0 < tr < 1e300:
                    H = H + (p.aa_reg * tr / len(H)) * np.eye(len(H))
                    try:
                        Lc = np.linalg.cholesky(H)
                        gam = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
                    except np.linalg.LinAlgError:
                        gam = None
                    if gam is not None and np.isfinite(gam).all():
                        zaa = gz - dG.T @ gam
                        (Y, Yp, D1, D3, Vt, D3V, D3T), alpha = aa_unpack(zaa)
                        aa_pending = (fn, gz); aa_zin = zaa; done_aa = True; n_aa += 1
            if not done_aa:
                aa_zin = gz
    obj, Lam = inst.f_value(Y, want=True)
    X = inst.X_of(Y, Lam)
    U = recover_U(Y, Q, Vt)
    out = dict(objective=obj, dual_bound=lb, Y=Y, U=U, X=X, termination_status=status,
               feasible=status != OMC_INFEASIBLE, solve_time=time.time() - t0, iters=it, rows=rows, lam=lam,
               hist=hist, rp=rp, rd=rd, rho=rho, Q=Q, Vt=Vt, n_aa=n_aa, n_aa_rejected=n_aa_rej,
               warm=dict(Y=Y, U=Q @ Vt, Yp=Yp, alpha=alpha, svals=svals, D1=D1, D3=D3, rho=rho))
    if want_certificate:
        # Theta = X' pinv(Y) X = gamma * Lam' X is the minimal Theta with [Y X; X' Theta] >= 0
        Theta = g * (Lam.T @ X)
        Theta = 0.5 * (Theta + Theta.T)
        out["Theta"] = Theta
        out["objective_reference_formula"] = compute_SDP_relaxation_objective(X, Theta, inst.A, inst.indices, g)
        out["residuals"] = primal_residuals(inst, rows, Y, U, X, Theta)
    return out


def _proj_psd(M):
    w, V = _psd_split(M)
    return (V * np.maximum(w, 0.0)) @ V.T


# ----------------------------------------------------------------------------------------------------------
# alternating minimisation  (OMC.jl:1979-2279)
# ----------------------------------------------------------------------------------------------------------
AM_MAX_DOUBLINGS = 64      # bracket search of the ball multiplier in the rank-1 U-step (same constant in omc_altmin.hip)
AM_FEAS_TOL = 1e-6         # largest constraint violation accepted from a U-step; beyond it model_U is reported as failed


def altmin_v_step(inst, U):
    """model_V (OMC.jl:2193-2208): unconstrained; per column j
       (sum_{i in O_j} u_i u_i' + U'U/gamma) V_j = sum_{i in O_j} u_i A_ij."""
    k, m, g = inst.k, inst.m, inst.gamma
    UtU = U.T @ U / g
    V = np.zeros((k, m))
    for j in range(m):
        o = inst.cols[j]
        Uo = U[o, :]
        H = Uo.T @ Uo + UtU
        rhs = Uo.T @ inst.A[o, j]
        V[:, j] = np.linalg.lstsq(H, rhs, rcond=None)[0] if k > 1 else (rhs / H[0, 0] if H[0, 0] > 0 else 0.0)
    return V


def altmin_objective(inst, U, V):
    """The objective both JuMP models share (OMC.jl:2193-2207, 2213-2227)."""
    X = U @ V
    R = (X - inst.A)[inst.indices]
    return 0.5 * float(R @ R) + float((X * X).sum()) / (2.0 * inst.gamma)


def _ustep_quadratic(inst, V):
    """Row-separable quadratic of model_U: 1/2 u_i' H_i u_i - g_i' u_i + const."""
    n, k, g = inst.n, inst.k, inst.gamma
    VVt = V @ V.T / g
    H = np.zeros((n, k, k)); gv = np.zeros((n, k))
    for i in range(n):
        o = np.flatnonzero(inst.indices[i, :])
        Vo = V[:, o]
        H[i] = Vo @ Vo.T + VVt
        gv[i] = Vo @ inst.A[i, o]
    const = 0.5 * inst.sumA2
    return H, gv, const


def quadratic_constraint_vectors(k):
    """w_c, r_c of the quadratic constraints sum_i (w_c' u_i)^2 <= r_c of model_U: the k unit balls (OMC.jl:2164-2171), then
    for every pair j1 < j2 the two cones ||U_j1 + U_j2||^2 <= 2, ||U_j1 - U_j2||^2 <= 2 (OMC.jl:2029-2045)."""
    W = []; rad = []
    for j in range(k):
        e = np.zeros(k); e[j] = 1.0; W.append(e); rad.append(1.0)
    for j1 in range(k - 1):
        for j2 in range(j1 + 1, k):
            for sg in (1.0, -1.0):
                e = np.zeros(k); e[j1] = 1.0; e[j2] = sg; W.append(e); rad.append(2.0)
    return np.array(W), np.array(rad)


def _ustep_dual_newton(H, gv, Cm, d, const, tol=1e-12, max_newton=60):
    """Exact dual method for model_U with k > 1 (mirrored by k_altmin on the GPU).
    Multipliers theta_c >= 0 of the k^2 quadratic constraints shift every row Hessian by the same k x k matrix
    S(theta) = 2 sum_c theta_c w_c w_c'; for fixed theta the problem is the row-separable QP with linear rows of the k = 1 case
    (active-set NNQP on the Gram matrix C H(theta)^-1 C').  Outer: semismooth Newton on the complementarity system
    min(theta, -q(theta)) = 0, q_c = sum_i (w_c' u_i)^2 - r_c, Jacobian by forward differences (symmetrised, ridge 1e-12),
    projected step with backtracking on the residual norm."""
    n, k = gv.shape
    R = Cm.shape[0]
    W, rad = quadratic_constraint_vectors(k)
    mq = len(rad)

    def evaluate(theta):
        S = 2.0 * np.einsum("c,ca,cb->ab", theta, W, W)
        Hinv = np.linalg.inv(H + S[None])
        U0 = np.einsum("iab,ib->ia", Hinv, gv)
        mu = np.zeros(R)
        U = U0
        if R:
            T = np.einsum("ria,iab->rib", Cm, Hinv)
            G = np.einsum("rib,sib->rs", T, Cm)
            cc = np.einsum("ria,ia->r", Cm, U0) - d
            mu = nnqp(G, cc)
            U = U0 - np.einsum("r,rib->ib", mu, T)
        q = ((U @ W.T) ** 2).sum(0) - rad
        return U, q, mu

    def dual_value(theta, U, q):
        return 0.5 * float(np.einsum("ia,iab,ib->", U, H, U)) - float((gv * U).sum()) + float(theta @ q)

    theta = np.zeros(mq)
    U, q, mu = evaluate(theta)
    dv = dual_value(theta, U, q)
    res = float(np.abs(np.minimum(theta, -q)).max())
    it = 0
    lm = 1e-10                                   # Levenberg-Marquardt weight relative to the mean diagonal of -J
    while res > tol and it < max_newton:
        it += 1
        act = [c for c in range(mq) if theta[c] > 0.0 or q[c] > 0.0]
        J = np.zeros((len(act), len(act)))
        for a, c in enumerate(act):
            dl = 1e-7 * max(1.0, theta[c])
            th2 = theta.copy(); th2[c] += dl
            _, q2, _ = evaluate(th2)
            J[:, a] = (q2[act] - q[act]) / dl
        P = -0.5 * (J + J.T)
        scale = max(float(np.trace(P)), 1e-300) / len(act)
        accepted = False
        for _try in range(40):                   # maximise the concave dual: damped projected Newton step, ascent test on d
            try:
                Lc = np.linalg.cholesky(P + (lm * scale) * np.eye(len(act)))
                step = np.linalg.solve(Lc.T, np.linalg.solve(Lc, q[act]))
            except np.linalg.LinAlgError:
                lm *= 10.0
                continue
            th_new = theta.copy()
            th_new[act] = np.maximum(theta[act] + step, 0.0)
            Un, qn, mun = evaluate(th_new)
            dn = dual_value(th_new, Un, qn)
            rn = float(np.abs(np.minimum(th_new, -qn)).max())
            if dn >= dv + 1e-4 * float(q @ (th_new - theta)) - 1e-14 * max(1.0, abs(dv)) and (dn > dv or rn < res):
                theta, U, q, mu, res, dv = th_new, Un, qn, mun, rn, dn
                lm = max(lm * 0.1, 1e-12)
                accepted = True
                break
            lm *= 10.0
        if not accepted:
            break
    obj = 0.5 * float(np.einsum("ia,iab,ib->", U, H, U)) - float((gv * U).sum()) + const
    return U, obj, dict(theta=theta, lam=mu, newton_iters=it, kkt_residual=res)


def altmin_u_step(inst, V, cuts=(), cut_type="linear", U_lower=None, U_upper=None, reference_quirk_q1=True, method="dual"):
    """model_U (OMC.jl:2014-2093, 2164-2171, 2213-2228): convex QP with box, ||U_j||<=1,
    ||U_j1 +- U_j2|| <= sqrt 2 and the per-cut bounds on v = U'x (NOT the aggregated cut row).
    k = 1: exact dual method (multiplier of the ball by bracketing + active-set NNQP for the rows).
    k > 1: method="dual" (default): exact dual Newton method, mirrored by the GPU; method="slsqp": scipy SLSQP on the
    same QP (independent implementation used to validate the former; small sizes only)."""
    n, k = inst.n, inst.k
    H, gv, const = _ustep_quadratic(inst, V)
    rows = build_rows(inst, cuts, cut_type, U_lower, U_upper, reference_quirk_q1)
    sel = [r for r in range(len(rows)) if rows.kinds[r] in ("box_lo", "box_hi", "hi", "lo")]
    C = np.array([rows.CU[r].ravel() for r in sel]).reshape(len(sel), n * k)
    d = np.array([rows.rhs[r] for r in sel])
    if k == 1:
        h = H[:, 0, 0]; gg = gv[:, 0]

        def solve_theta(theta):
            hd = h + theta
            u0 = gg / hd
            if len(sel) == 0:
                return u0, np.zeros(0)
            Gm = (C / hd) @ C.T
            cc = C @ u0 - d
            lam = nnqp(Gm, cc)
            return u0 - (C.T @ lam) / hd, lam

        u, lam = solve_theta(0.0)
        theta = 0.0
        if u @ u > 1.0:
            lo_t, hi_t = 0.0, max(1.0, float(np.abs(gg).max()))
            for _dbl in range(AM_MAX_DOUBLINGS + 1):   # capped: an infeasible model_U never enters the ball (k_altmin does the same)
                u, lam = solve_theta(hi_t)
                if u @ u <= 1.0 or _dbl == AM_MAX_DOUBLINGS:
                    break
                hi_t *= 2.0
            for _ in range(200):
                theta = 0.5 * (lo_t + hi_t)
                u, lam = solve_theta(theta)
                if u @ u > 1.0:
                    lo_t = theta
                else:
                    hi_t = theta
                if hi_t - lo_t <= 1e-15 * max(1.0, hi_t):
                    break
            theta = hi_t
            u, lam = solve_theta(theta)
        Un = u.reshape(n, 1)
        obj = 0.5 * float((h * u * u).sum()) - float(gg @ u) + const
        viol = max(float((C @ u - d).max()) if len(sel) else 0.0, float(u @ u) - 1.0)
        return Un, obj, dict(theta=theta, lam=lam, rows=[rows.kinds[r] for r in sel], violation=viol)
    # ---- k > 1 -----------------------------------------------------------------------------------------
    if method == "dual":
        Un, obj, info = _ustep_dual_newton(H, gv, C.reshape(len(sel), n, k), d, const)
        Wq, rad = quadratic_constraint_vectors(k)
        viol = float((((Un @ Wq.T) ** 2).sum(0) - rad).max())
        if len(sel):
            viol = max(viol, float((C @ Un.ravel() - d).max()))
        info["violation"] = viol
        return Un, obj, info
    from scipy.optimize import minimize

    def q(u):
        Um = u.reshape(n, k)
        return 0.5 * float(np.einsum("ia,iab,ib->", Um, H, Um)) - float((gv * Um).sum())

    def dq(u):
        Um = u.reshape(n, k)
        return (np.einsum("iab,ib->ia", H, Um) - gv).ravel()

    cons = []
    if len(sel):
        cons.append(dict(type="ineq", fun=lambda u: d - C @ u, jac=lambda u: -C))
    for j in range(k):
        cons.append(dict(type="ineq", fun=lambda u, j=j: 1.0 - float((u.reshape(n, k)[:, j] ** 2).sum()),
                         jac=lambda u, j=j: _col_jac(u, n, k, [(j, -2.0)])))
    for j1 in range(k - 1):
        for j2 in range(j1 + 1, k):
            for sg in (1.0, -1.0):
                cons.append(dict(type="ineq",
                                 fun=lambda u, a=j1, b_=j2, sg=sg: 2.0 - float(((u.reshape(n, k)[:, a] + sg * u.reshape(n, k)[:, b_]) ** 2).sum()),
                                 jac=lambda u, a=j1, b_=j2, sg=sg: _pm_jac(u, n, k, a, b_, sg)))
    u0 = np.zeros(n * k)
    res = minimize(q, u0, jac=dq, constraints=cons, method="SLSQP", options=dict(ftol=1e-14, maxiter=2000))
    Un = res.x.reshape(n, k)
    return Un, q(res.x) + const, dict(slsqp_status=res.status, message=res.message)


def _col_jac(u, n, k, terms):
    Um = u.reshape(n, k); J = np.zeros((n, k))
    for (j, cf) in terms:
        J[:, j] = cf * Um[:, j]
    return J.ravel()


def _pm_jac(u, n, k, a, b_, sg):
    Um = u.reshape(n, k); J = np.zeros((n, k))
    s = Um[:, a] + sg * Um[:, b_]
    J[:, a] = -2.0 * s; J[:, b_] = -2.0 * sg * s
    return J.ravel()


def alternating_minimization(inst, U_initial, cuts=(), cut_type="linear", U_lower=None, U_upper=None,
                             eps=1e-5, max_iters=100, time_limit=3600.0, reference_quirk_q1=True):
    """OMC.jl:1979-2279 incl. quirk Q3 (divergence is also reported as converged, 2237-2245).
    Returns the reference's keys: converged, U, V, solve_time, n_iters, max_iters, objectives."""
    t0 = time.time()
    n, k, m = inst.n, inst.k, inst.m
    U_current = np.asarray(U_initial, float).reshape(n, k)
    counter = 0
    objective_current = 1e10                                                   # OMC.jl:2012
    objectives = []
    converged = False
    U_new = np.zeros((n, k)); V_new = np.zeros((k, m))
    while counter < max_iters and time.time() - t0 < time_limit:                # OMC.jl:2186-2189
        counter += 1
        V_new = altmin_v_step(inst, U_current)                                 # 2192-2209
        U_new, objective_new, info = altmin_u_step(inst, V_new, cuts, cut_type, U_lower, U_upper,
                                                   reference_quirk_q1)         # 2212-2232
        if not (info.get("violation", 0.0) <= AM_FEAS_TOL):
            break                  # model_U had no solution: the reference's try/catch -> break, converged = false (2231, 2263-2265)
        objectives.append(objective_new)
        objective_diff = abs((objective_new - objective_current) / objective_current)
        if objective_diff < eps:                                               # 2235
            converged = True
        elif len(objectives) > 5 and all(objectives[-1 - i] > objectives[-6] for i in range(5)):  # 2237-2243
            converged = True
        if converged:
            break
        U_current = U_new
        objective_current = objective_new
    return dict(converged=converged, U=U_new, V=V_new, solve_time=time.time() - t0, n_iters=counter,
                max_iters=max_iters, objectives=objectives)


# ----------------------------------------------------------------------------------------------------------
# synthetic instances (distribution of /root/reference/src/utils.jl:3-26, 68-111 and README.md:31-41;
# Julia's MersenneTwister streams are not reproducible outside Julia, so numpy seeds are used and recorded)
# ----------------------------------------------------------------------------------------------------------
def make_instance(n, m, k, n_indices=None, seed=0, noise=0.01, kind="lowrank", max_tries=100):
    """kind="lowrank": A = L R + noise*E, mask = n_indices uniformly random cells redrawn (<=100 tries)
    until every row and column is hit (utils.jl:13-25, 98-103).  Draw order: L, R, E, permutation(s).
    kind="readme": A iid N(0,1), mask iid Bernoulli(1/2)  (README.md:33)."""
    rng = np.random.default_rng(seed)
    if kind == "readme":
        A = rng.standard_normal((n, m))
        mask = rng.integers(0, 2, (n, m)).astype(bool)
        return A, mask
    L = rng.standard_normal((n, k)); Rm = rng.standard_normal((k, m)); E = rng.standard_normal((n, m))
    A = L @ Rm + noise * E
    if n_indices is None:
        n_indices = int(round(0.2 * n * m))
    if n_indices < (n + m) * k:
        raise ValueError("System is under-determined: n_indices must be at least (n + m) * k (utils.jl:85-90)")
    it = 0
    while True:
        perm = rng.permutation(n * m)[:n_indices]
        vec = np.zeros(n * m, bool); vec[perm] = True
        mask = vec.reshape((n, m), order="F")
        if (mask.any(0).all() and mask.any(1).all()) or it >= max_tries:
            return A, mask
        it += 1
