"""Host-side mirror of the reference's per-node interface (same names, argument meaning and error behaviour),
backed by the HIP library.  Reference = /root/reference/src/OptimalMatrixCompletion.jl (OMC.jl).

    Engine(A, indices, gamma, k)                  uploads (A, indices, gamma) once       (OMC.jl:470-471)
    .matrix_completion_SDP_relaxation(nodes, ...) OMC.jl:1431-1943, a batch of nodes; with `shor_info` = one (constraints_indexes,
                                                  SOC_constraints_indexes) pair per node (BBNodeShorInfo, OMC.jl:37-40) the Shor-mode program
                                                  (add_Shor_valid_inequalities = true, rank 1)
    .matrix_completion_master_feasible(Y, U)      OMC.jl:1261-1277
    .breakpoint_vectors(Y, U, breakpoints)        OMC.jl:2466-2477
    .evaluate_objective(X)                        OMC.jl:2330-2359
    .alternating_minimization(U_initial, ...)     OMC.jl:1979-2279
A node is the reference's BBNode reduced to what the relaxation reads: a list of cuts
(breakpoint_vec, U_hat, directions) exactly as BBNodeDisjunctiveCuts.cuts holds them (OMC.jl:33-35).
"""
from __future__ import annotations

import atexit
import weakref
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RelaxParams, OmcError

CUT_TYPES = {"linear": 0, "linear2": 1, "linear3": 2}
DIR_CODES = {"left": 0, "middle": 1, "right": 2, "inner_left": 3, "inner_right": 4}
BREAKPOINTS = {"smallest_1_eigvec": 1, "smallest_2_eigvec": 2}
STATUS_NAMES = {0: "OPTIMAL", 1: "SLOW_PROGRESS", 2: "TIME_LIMIT", 3: "INFEASIBLE"}
KERNEL_CLASSES = ["colprox", "cone", "global", "check", "setup", "small", "accel", "cone_sub", "check_col", "check_build", "harvest",
                  "shor_bigcone", "shor_minors", "shor_cols"]


def default_params(**kw) -> RelaxParams:
    p = RelaxParams()
    _lib.load().omc_relax_params_default(C.byref(p))
    for k_, v in kw.items():
        if not hasattr(p, k_):
            raise TypeError(f"unknown relaxation parameter {k_!r}")
        setattr(p, k_, v)
    return p


def _pack_cuts(nodes, n, k, cut_type):
    if cut_type not in CUT_TYPES:
        raise ValueError("Invalid input for disjunctive cuts type.\nDisjunctive cuts type must be either "
                         f"\"linear\" or \"linear2\" or \"linear3\";\n{cut_type} supplied instead.")   # OMC.jl:1456-1462
    L = np.array([len(c) for c in nodes], dtype=np.int32)
    tot = int(L.sum())
    cx = np.zeros((max(tot, 1), n)); cU = np.zeros((max(tot, 1), n * k)); cd = np.zeros((max(tot, 1), k), dtype=np.int8)
    t = 0
    for cuts in nodes:
        for (x, Uh, dirs) in cuts:
            x = np.asarray(x, float); Uh = np.asarray(Uh, float).reshape(n, k)
            if x.shape != (n,) or len(dirs) != k:
                raise ValueError("Dimension mismatch in cut (OMC.jl:34)")
            cx[t] = x; cU[t] = Uh.ravel(order="F")
            for j, d in enumerate(dirs):
                cd[t, j] = DIR_CODES[d] if isinstance(d, str) else int(d)
            t += 1
    return L, cx, cU, cd


def shor_rank_k_extension(k, X, W, minors):
    """The lifted variables of the reference's rank k > 1 Shor form (Xt, Wt, H, V1, V2, V3: OMC.jl:1526-1551, results at 1909-1917) as an explicit
    extension of the relaxation's (X, W).  Reference quirk Q5: in that form the slack s that H cancels in W = sum Wt + 2 sum H makes every
    per-layer order-5 block satisfiable, so the minors do not constrain (X, W) and the engine solves the program without them; this is the closed form
        Xt_1 = X, Xt_t = 0;  Wt_1 = X^2 + (W - X^2) + s, Wt_t = s/(k-1);  H_(1,t) = -s/(k-1), H_(t,t') = 0;  V^1 = products of X (V3: their mean), V^t = 0,
        s_c = max over the minors that contain c of |X_{i1j2} X_{i2j1} - X_{i1j1} X_{i2j2}| / 2.
    Returns Xt (k, n, m), Wt (k, n, m), H (k, k, n, m) and V (k, nq, 5) with the five products of each minor in the order of omc_relax_fetch_shor_V."""
    X = np.asarray(X, float); W = np.asarray(W, float); n, m = X.shape
    mi = np.asarray(minors, np.int64).reshape(-1, 4) - 1
    nq = len(mi)
    Xt = np.zeros((k, n, m)); Xt[0] = X
    Wt = np.zeros((k, n, m)); H = np.zeros((k, k, n, m)); V = np.zeros((k, nq, 5))
    if nq == 0:
        return dict(Xt=Xt, Wt=Wt, H=H, V=V)
    ci = np.stack([mi[:, 0], mi[:, 0], mi[:, 1], mi[:, 1]], 1); cj = np.stack([mi[:, 2], mi[:, 3], mi[:, 2], mi[:, 3]], 1)
    xs = X[ci, cj]
    e = 0.5 * np.abs(xs[:, 1] * xs[:, 2] - xs[:, 0] * xs[:, 3])
    s = np.zeros((n, m)); inC = np.zeros((n, m), bool)
    for p_ in range(4):
        np.maximum.at(s, (ci[:, p_], cj[:, p_]), e); inC[ci[:, p_], cj[:, p_]] = True
    s = np.where(inC, s * (1.0 + 1e-12) + 1e-300, 0.0)
    b = s / (k - 1)
    Wt[0] = np.where(inC, X * X + np.maximum(W - X * X, 0.0) + s, 0.0)
    for t in range(1, k):
        Wt[t] = np.where(inC, b, 0.0); H[0, t] = np.where(inC, -b, 0.0)
    V[0, :, 0] = xs[:, 0] * xs[:, 1]; V[0, :, 1] = xs[:, 2] * xs[:, 3]; V[0, :, 2] = xs[:, 0] * xs[:, 2]; V[0, :, 3] = xs[:, 1] * xs[:, 3]
    V[0, :, 4] = 0.5 * (xs[:, 0] * xs[:, 3] + xs[:, 1] * xs[:, 2])
    return dict(Xt=Xt, Wt=Wt, H=H, V=V)


_LIVE = weakref.WeakSet()      # engines still holding a device instance: closed before the interpreter (and then the HIP runtime) goes down


def _close_all():
    for e in list(_LIVE):
        try:
            e.close()
        except Exception:
            pass


atexit.register(_close_all)


class Engine:
    """Device-resident instance + the four call sites of the reference driver."""

    def __init__(self, A, indices, gamma, k, device=0):
        A = np.asarray(A, dtype=np.float64); indices = np.asarray(indices)
        if A.ndim != 2 or A.shape != indices.shape:
            raise ValueError("Dimension mismatch.\nInput matrix A must have size (n, m);\nInput matrix indices must have size (n, m).")
        self.n, self.m = A.shape
        self.k = int(k); self.gamma = float(gamma); self.device = int(device)
        self.A = np.asfortranarray(A); self.indices = indices.astype(bool)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        mask = np.asfortranarray(self.indices.astype(np.uint8))
        _lib.check(self._lib.omc_instance_create(self.n, self.m, self.k, _lib.ptr(self.A), _lib.ptr(mask), self.gamma,
                                                 int(device), C.byref(self._h)))
        _LIVE.add(self)

    def close(self):
        _LIVE.discard(self)
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.omc_instance_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tuning knobs ----------------------------------------------------------------------------------
    def tuning_set(self, name, value=None):
        """Set (or, with None, remove) one tuning knob of this instance (omc_tuning_set); the library reads the environment only at creation."""
        _lib.check(self._lib.omc_tuning_set(self._h, name.encode(), None if value is None else str(value).encode()))

    def tuning_reload_env(self):
        """Read the OMC_* knobs from the environment again (omc_tuning_reload_env)."""
        _lib.check(self._lib.omc_tuning_reload_env(self._h))

    # ---- relaxation ------------------------------------------------------------------------------------
    def state_pool_create(self, capacity):
        """Reserve `capacity` final states on the device for warm starts (omc_state_pool_create)."""
        _lib.check(self._lib.omc_state_pool_create(self._h, int(capacity)))
        self.pool_capacity = int(capacity)

    def stage(self, nodes, disjunctive_cuts_type="linear", params=None, U_lower=None, U_upper=None, rho_scales=None, load_from=None, save_to=None):
        n, k = self.n, self.k
        if load_from is not None or save_to is not None:
            lf = None if load_from is None else np.ascontiguousarray(np.asarray(load_from, dtype=np.int32))
            sv = None if save_to is None else np.ascontiguousarray(np.asarray(save_to, dtype=np.int32))
            if (lf is not None and lf.shape != (len(nodes),)) or (sv is not None and sv.shape != (len(nodes),)):
                raise ValueError("load_from / save_to must hold one pool index per node")
            _lib.check(self._lib.omc_relax_set_warm(self._h, len(nodes), _lib.ptr(lf), _lib.ptr(sv)))
        if rho_scales is not None:
            rs = np.ascontiguousarray(np.asarray(rho_scales, dtype=np.float64))
            if rs.shape != (len(nodes),):
                raise ValueError("rho_scales must hold one value per node")
            _lib.check(self._lib.omc_set_node_rho_scales(self._h, len(nodes), _lib.ptr(rs)))
        L, cx, cU, cd = _pack_cuts(nodes, n, k, disjunctive_cuts_type)
        p = params or default_params()
        B = len(nodes)
        lo = hi = None
        if U_lower is not None:
            lo = np.ascontiguousarray(np.stack([np.asfortranarray(u).ravel(order="F") for u in U_lower]))
        if U_upper is not None:
            hi = np.ascontiguousarray(np.stack([np.asfortranarray(u).ravel(order="F") for u in U_upper]))
        self._keep = (L, cx, cU, cd, lo, hi, p)
        _lib.check(self._lib.omc_relax_stage(self._h, B, C.byref(p), CUT_TYPES[disjunctive_cuts_type], _lib.ptr(L), _lib.ptr(cx),
                                             _lib.ptr(cU), _lib.ptr(cd), _lib.ptr(lo), _lib.ptr(hi)))
        self._B = B

    def stage_shor(self, nodes, shor_info, disjunctive_cuts_type="linear", params=None, U_lower=None, U_upper=None, penalties=None, keep_V=False):
        """Stage a batch with add_Shor_valid_inequalities = true (OMC.jl:747-754 with node.Shor_info).  shor_info[b] =
        (constraints_indexes, SOC_constraints_indexes): 1-based (i1, i2, j1, j2) tuples and 1-based (i, j) pairs as the reference
        holds them (OMC.jl:37-40); SOC_constraints_indexes = None means "every coordinate outside the minors" (what the reference's
        driver always builds, OMC.jl:656-673, 2508-2517)."""
        n, k = self.n, self.k
        if len(shor_info) != len(nodes):
            raise ValueError("one Shor_info per node")
        _lib.check(self._lib.omc_set_shor_keep_V(self._h, 1 if keep_V else 0))
        self._nqmax = max([len(np.asarray(mi).reshape(-1, 4)) for (mi, _) in shor_info] + [1])
        if penalties is not None:
            _lib.check(self._lib.omc_set_shor_penalties(self._h, float(penalties[0]), float(penalties[1]), float(penalties[2])))
        L, cx, cU, cd = _pack_cuts(nodes, n, k, disjunctive_cuts_type)
        p = params or default_params()
        B = len(nodes)
        lo = hi = None
        if U_lower is not None:
            lo = np.ascontiguousarray(np.stack([np.asfortranarray(u).ravel(order="F") for u in U_lower]))
        if U_upper is not None:
            hi = np.ascontiguousarray(np.stack([np.asfortranarray(u).ravel(order="F") for u in U_upper]))
        nsh = np.zeros(B, np.int64); nso = np.zeros(B, np.int64); sh_parts = []; so_parts = []
        for b, (minors, soc) in enumerate(shor_info):
            mq = np.asarray(minors, np.int64).reshape(-1, 4)
            nsh[b] = len(mq); sh_parts.append(mq)
            if soc is None:
                nso[b] = -1
            else:
                sq = np.asarray(soc, np.int64).reshape(-1, 2)
                nso[b] = len(sq); so_parts.append(sq)
        shi = np.ascontiguousarray(np.concatenate(sh_parts)) if sh_parts and sum(len(a) for a in sh_parts) else np.zeros((1, 4), np.int64)
        soi = np.ascontiguousarray(np.concatenate(so_parts)) if so_parts and sum(len(a) for a in so_parts) else np.zeros((1, 2), np.int64)
        self._keep = (L, cx, cU, cd, lo, hi, p, nsh, nso, shi, soi)
        _lib.check(self._lib.omc_relax_stage_shor(self._h, B, C.byref(p), CUT_TYPES[disjunctive_cuts_type], _lib.ptr(L), _lib.ptr(cx),
                                                  _lib.ptr(cU), _lib.ptr(cd), _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(nsh), _lib.ptr(shi),
                                                  _lib.ptr(nso), _lib.ptr(soi)))
        self._B = B

    def fetch_shor(self):
        """W of the last Shor-mode batch (results["W"], OMC.jl:1908), one (n, m) matrix per node."""
        W = np.zeros((self._B, self.n * self.m))
        _lib.check(self._lib.omc_relax_fetch_shor(self._h, _lib.ptr(W)))
        return [W[b].reshape((self.n, self.m), order="F") for b in range(self._B)]

    def fetch_shor_V(self):
        """(nq_max, 5) per node: V1[i1,(j1,j2)], V1[i2,(j1,j2)], V2[(i1,i2),j1], V2[(i1,i2),j2], V3 of every minor (needs keep_V at staging)."""
        V = np.zeros((self._B, self._nqmax * 5))
        _lib.check(self._lib.omc_relax_fetch_shor_V(self._h, _lib.ptr(V)))
        return [V[b].reshape(self._nqmax, 5) for b in range(self._B)]

    def reserve(self, extra_nodes, max_cuts):
        """Room for nodes appended to the NEXT staged batch (omc_relax_reserve): extra_nodes more nodes with at most max_cuts cuts each."""
        _lib.check(self._lib.omc_relax_reserve(self._h, int(extra_nodes), int(max_cuts)))

    def append(self, nodes, disjunctive_cuts_type="linear", load_from=None, save_to=None):
        """Add nodes to the staged batch -- also while a submitted solve is running (omc_relax_append): the reference's queue keeps receiving
        the children of relaxed nodes (OMC.jl:700-719, 2520-2542).  Raises OmcError once that solve has ended.  fetch() returns the appended
        nodes behind the staged ones."""
        n, k = self.n, self.k
        L, cx, cU, cd = _pack_cuts(nodes, n, k, disjunctive_cuts_type)
        lf = None if load_from is None else np.ascontiguousarray(np.asarray(load_from, dtype=np.int32))
        sv = None if save_to is None else np.ascontiguousarray(np.asarray(save_to, dtype=np.int32))
        if (lf is not None and lf.shape != (len(nodes),)) or (sv is not None and sv.shape != (len(nodes),)):
            raise ValueError("load_from / save_to must hold one pool index per node")
        _lib.check(self._lib.omc_relax_append(self._h, len(nodes), _lib.ptr(L), _lib.ptr(cx), _lib.ptr(cU), _lib.ptr(cd), _lib.ptr(lf), _lib.ptr(sv)))
        self._B += len(nodes)

    def hold(self, on=True):
        """Keep the submitted solve open when it runs dry (omc_relax_hold): it waits for append() until hold(False)."""
        _lib.check(self._lib.omc_relax_hold(self._h, 1 if on else 0))

    def fetch_done(self, max_nodes=4096, want_Y=False):
        """Results of the nodes finished since the last call (omc_relax_fetch_done), also while the submitted solve is running: a list of dicts
        with the keys of fetch() that a driver needs to branch (node = index among the staged + appended nodes; Y on request)."""
        n, k = self.n, self.k
        key = (int(max_nodes), bool(want_Y))
        if getattr(self, "_fd_key", None) != key:          # the receive buffers are kept between calls (a polling loop calls this every millisecond)
            self._fd_key = key
            self._fd = (np.zeros(max_nodes, np.int32), np.zeros(max_nodes), np.zeros(max_nodes), np.zeros(max_nodes, np.int32), np.zeros(max_nodes, np.int32),
                        np.zeros((max_nodes, n * k)), np.zeros((max_nodes, 2)), np.zeros((max_nodes, n)), np.zeros(1, np.int32),
                        np.zeros((max_nodes, n * n)) if want_Y else None)
        ids, obj, lb, st, it, U, ev, bx, cnt, Y = self._fd
        _lib.check(self._lib.omc_relax_fetch_done(self._h, int(max_nodes), _lib.ptr(ids), _lib.ptr(obj), _lib.ptr(lb), _lib.ptr(st), _lib.ptr(it),
                                                  _lib.ptr(U), _lib.ptr(ev), _lib.ptr(bx), _lib.ptr(Y), _lib.ptr(cnt)))
        out = []
        for i in range(int(cnt[0])):
            d = dict(node=int(ids[i]), objective=float(obj[i]), dual_bound=float(lb[i]), status_code=int(st[i]), termination_status=STATUS_NAMES[int(st[i])],
                     feasible=int(st[i]) != 3, iters=int(it[i]), U=U[i].reshape((n, k), order="F").copy(), lambda_min=ev[i].copy(), breakpoint_vec=bx[i].copy())
            if want_Y:
                d["Y"] = Y[i].reshape((n, n), order="F").copy()
            out.append(d)
        return out

    def solve(self):
        _lib.check(self._lib.omc_relax_solve(self._h))

    def submit(self):
        """Asynchronous solve of the staged batch: returns at once; poll() / wait() follow (omc_relax_submit)."""
        _lib.check(self._lib.omc_relax_submit(self._h))

    def poll(self):
        r = np.zeros(3, np.int32)
        _lib.check(self._lib.omc_relax_poll(self._h, _lib.ptr(r[0:1]), _lib.ptr(r[1:2]), _lib.ptr(r[2:3])))
        return dict(running=bool(r[0]), nodes_done=int(r[1]), nodes_total=int(r[2]))

    def wait(self):
        _lib.check(self._lib.omc_relax_wait(self._h))

    def fetch(self, want_Y=True, want_X=True, want_Theta=False):
        B, n, m, k = self._B, self.n, self.m, self.k
        obj = np.zeros(B); lb = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
        Y = np.zeros((B, n * n)) if want_Y else None
        U = np.zeros((B, n * k))
        X = np.zeros((B, n * m)) if want_X else None
        Th = np.zeros((B, m * m)) if want_Theta else None
        lmin = np.zeros((B, 2)); bx = np.zeros((B, n)); tm = np.zeros(B)
        _lib.check(self._lib.omc_relax_fetch(self._h, _lib.ptr(obj), _lib.ptr(lb), _lib.ptr(st), _lib.ptr(it), _lib.ptr(Y),
                                             _lib.ptr(U), _lib.ptr(X), _lib.ptr(Th), _lib.ptr(lmin), _lib.ptr(bx), _lib.ptr(tm)))
        out = []
        for b in range(B):
            r = {
                "objective": float(obj[b]), "dual_bound": float(lb[b]), "termination_status": STATUS_NAMES[int(st[b])],
                "status_code": int(st[b]), "feasible": int(st[b]) != 3, "iters": int(it[b]), "solve_time": float(tm[b]),
                "U": U[b].reshape((n, k), order="F"), "lambda_min": lmin[b].copy(), "breakpoint_vec": bx[b].copy(),
            }
            if want_Y: r["Y"] = Y[b].reshape((n, n), order="F")
            if want_X: r["X"] = X[b].reshape((n, m), order="F")
            if want_Theta: r["Theta"] = Th[b].reshape((m, m), order="F")
            out.append(r)
        return out

    def matrix_completion_SDP_relaxation(self, nodes, disjunctive_cuts_type="linear", params=None, U_lower=None, U_upper=None,
                                         want_Y=True, want_X=True, want_Theta=False, rho_scales=None, add_Shor_valid_inequalities=False,
                                         shor_info=None, shor_penalties=None, want_V=False, load_from=None, save_to=None):
        """Batch form of OMC.jl:1431-1943 (use_disjunctive_cuts = true).  `nodes` = list of cut lists; with
        add_Shor_valid_inequalities = True, `shor_info` = one (constraints_indexes, SOC_constraints_indexes) pair per node and
        every result also carries "W" (OMC.jl:1907-1908)."""
        if add_Shor_valid_inequalities:
            if shor_info is None:
                raise ValueError("add_Shor_valid_inequalities = true needs node.Shor_info (OMC.jl:1508)")
            self.stage_shor(nodes, shor_info, disjunctive_cuts_type, params, U_lower, U_upper, shor_penalties, keep_V=(want_V and self.k == 1))
            self.solve()
            out = self.fetch(want_Y, want_X, want_Theta)
            for r, Wb in zip(out, self.fetch_shor()):
                r["W"] = Wb
            if want_V and self.k == 1:
                for r, Vb, (mi, _) in zip(out, self.fetch_shor_V(), shor_info):
                    r["V"] = Vb[:len(np.asarray(mi).reshape(-1, 4))]
            elif want_V:                                   # rank k > 1: the lifted variables are an explicit extension of (X, W) (quirk Q5)
                for r, (mi, _) in zip(out, shor_info):
                    r.update(shor_rank_k_extension(self.k, r["X"], r["W"], mi))
            return out
        self.stage(nodes, disjunctive_cuts_type, params, U_lower, U_upper, rho_scales, load_from, save_to)
        self.solve()
        return self.fetch(want_Y, want_X, want_Theta)

    def kernel_stats(self):
        nc = len(KERNEL_CLASSES); la = np.zeros(nc, np.int64); ms = np.zeros(nc); un = np.zeros(nc, np.int64)
        _lib.check(self._lib.omc_last_kernel_stats(self._h, _lib.ptr(la), _lib.ptr(ms), _lib.ptr(un)))
        return {KERNEL_CLASSES[i]: dict(launches=int(la[i]), ms=float(ms[i]), units=int(un[i])) for i in range(nc)}

    def subspace_stats(self):
        """k_cone_sub accounting of the last solve: calls, power steps, fall-backs to the full eigendecomposition, seedings."""
        out = np.zeros(8, np.int64)
        _lib.check(self._lib.omc_last_subspace_stats(self._h, _lib.ptr(out)))
        return dict(calls=int(out[0]), power_steps=int(out[1]), fallbacks=int(out[2]), seeds=int(out[3]), fail_positive=int(out[4]),
                    fail_steps=int(out[5]), fail_cholesky=int(out[6]), ritz_passes=int(out[7]))

    def shor_subspace_stats(self):
        """The same counters for the order-(n+m) cone of the last Shor-mode solve."""
        out = np.zeros(8, np.int64)
        _lib.check(self._lib.omc_last_shor_subspace_stats(self._h, _lib.ptr(out)))
        return dict(calls=int(out[0]), power_steps=int(out[1]), fallbacks=int(out[2]), seeds=int(out[3]), fail_positive=int(out[4]),
                    fail_steps=int(out[5]), fail_cholesky=int(out[6]), ritz_passes=int(out[7]))

    def solver_info(self):
        info = np.zeros(8)
        _lib.check(self._lib.omc_last_solver_info(self._h, _lib.ptr(info)))
        return dict(solve_seconds=info[0], jacobi_sweeps=int(info[1]), rho=info[2], r_max=int(info[3]), cone_lds=bool(info[4]),
                    global_lds=bool(info[5]), small_lds=bool(info[6]), R_max=int(info[7]))

    # ---- multi-GPU exchange behind the C ABI (RCCL) ------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id, created by rank 0 and handed to every rank by the host (torch.distributed, MPI, a file ...)."""
        buf = np.zeros(128, np.uint8)
        _lib.check(_lib.load().omc_comm_unique_id(_lib.ptr(buf)))
        return buf

    def comm_init(self, rank, world_size, unique_id):
        uid = np.ascontiguousarray(np.asarray(unique_id, np.uint8))
        _lib.check(self._lib.omc_comm_init(self._h, int(rank), int(world_size), _lib.ptr(uid)))
        self.rank, self.world_size = int(rank), int(world_size)

    def allreduce_bounds(self, upper_bound, lower_bound):
        """MIN over ranks of {incumbent UB, smallest open LB}; returns (ub, lb, owner rank of the UB)."""
        ub = np.array([upper_bound], np.float64); lb = np.array([lower_bound], np.float64); ow = np.zeros(1, np.int32)
        _lib.check(self._lib.omc_allreduce_bounds(self._h, _lib.ptr(ub), _lib.ptr(lb), _lib.ptr(ow)))
        return float(ub[0]), float(lb[0]), int(ow[0])

    def allgather_records(self, rows, width, capacity_rows):
        """Rows (cnt, width) of every rank, in rank order, through the library's communicator (omc_allgather_records)."""
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, width))
        out = np.zeros((int(capacity_rows), width)); counts = np.zeros(self.world_size, np.int32)
        _lib.check(self._lib.omc_allgather_records(self._h, _lib.ptr(rows) if len(rows) else None, len(rows), int(width), _lib.ptr(out), int(capacity_rows), _lib.ptr(counts)))
        return out[: int(counts.sum())]

    def bcast_incumbent(self, root, X):
        Xf = np.asfortranarray(np.asarray(X, np.float64).reshape(self.n, self.m))
        _lib.check(self._lib.omc_bcast_incumbent(self._h, int(root), _lib.ptr(Xf)))
        return Xf

    # ---- separation / feasibility ----------------------------------------------------------------------
    def breakpoint_vectors(self, Ys, Us, disjunctive_cuts_breakpoints="smallest_1_eigvec"):
        if disjunctive_cuts_breakpoints not in BREAKPOINTS:
            raise ValueError("Invalid input for disjunctive cuts breakpoints.")                     # OMC.jl:2440-2446
        B, n, k = len(Ys), self.n, self.k
        Yb = np.ascontiguousarray(np.stack([np.asfortranarray(y, dtype=np.float64).ravel(order="F") for y in Ys]))
        Ub = np.ascontiguousarray(np.stack([np.asfortranarray(np.asarray(u, float).reshape(n, k)).ravel(order="F") for u in Us]))
        ev = np.zeros((B, 2)); x = np.zeros((B, n)); fe = np.zeros(B, np.int32)
        _lib.check(self._lib.omc_separation_batch(self._h, B, BREAKPOINTS[disjunctive_cuts_breakpoints], _lib.ptr(Yb), _lib.ptr(Ub),
                                                  _lib.ptr(ev), _lib.ptr(x), _lib.ptr(fe)))
        return x, ev, fe.astype(bool)

    def matrix_completion_master_feasible(self, Y, U):
        """OMC.jl:1261-1277 (disjunctive branch): lambda_min(U U' - Y) >= -1e-6."""
        _, ev, fe = self.breakpoint_vectors([Y], [U])
        return bool(fe[0])

    # ---- rounding glue + alternating minimisation ---------------------------------------------------------
    def round_Y(self, Ys):
        """svd(Y).U[:, 1:k] of the relaxed Y (OMC.jl:873), batched; sign: largest-magnitude entry positive."""
        B, n, k = len(Ys), self.n, self.k
        Yb = np.ascontiguousarray(np.stack([np.asfortranarray(y, dtype=np.float64).ravel(order="F") for y in Ys]))
        U = np.zeros((B, n * k))
        _lib.check(self._lib.omc_round_Y_batch(self._h, B, _lib.ptr(Yb), _lib.ptr(U)))
        return [U[b].reshape((n, k), order="F") for b in range(B)]

    def left_singular(self, Xs):
        """svd(X).U[:, 1:k] of n x m matrices (OMC.jl:524, 564, 921), batched on the device (omc_left_singular_batch); sign: largest-magnitude entry positive."""
        B, n, m, k = len(Xs), self.n, self.m, self.k
        Xb = np.ascontiguousarray(np.stack([np.asfortranarray(x, dtype=np.float64).ravel(order="F") for x in Xs]))
        if Xb.shape != (B, n * m):
            raise ValueError("left_singular: every matrix must be n x m")
        U = np.zeros((B, n * k))
        _lib.check(self._lib.omc_left_singular_batch(self._h, B, _lib.ptr(Xb), _lib.ptr(U)))
        return [U[b].reshape((n, k), order="F") for b in range(B)]

    def alternating_minimization(self, U_initials, nodes=None, disjunctive_cuts_type="linear", eps=1e-5, max_iters=100,
                                 time_limit=3600.0, reference_quirk_q1=True):
        """Batch form of OMC.jl:1979-2279 (use_disjunctive_cuts = true).  Returns dicts with the reference's keys
        converged, U, V, solve_time, n_iters, max_iters, objectives (OMC.jl:2249-2257)."""
        n, m, k = self.n, self.m, self.k
        B = len(U_initials)
        nodes = nodes if nodes is not None else [[] for _ in range(B)]
        L, cx, cU, cd = _pack_cuts(nodes, n, k, disjunctive_cuts_type)
        U0 = np.ascontiguousarray(np.stack([np.asfortranarray(np.asarray(u, float).reshape(n, k)).ravel(order="F") for u in U_initials]))
        U = np.zeros((B, n * k)); V = np.zeros((B, k * m)); cv = np.zeros(B, np.int32); ni = np.zeros(B, np.int32)
        obj = np.zeros((B, max_iters)); tm = np.zeros(B)
        _lib.check(self._lib.omc_altmin_batch(self._h, B, CUT_TYPES[disjunctive_cuts_type], int(reference_quirk_q1), _lib.ptr(L),
                                              _lib.ptr(cx), _lib.ptr(cU), _lib.ptr(cd), _lib.ptr(U0), float(eps), int(max_iters),
                                              float(time_limit), _lib.ptr(U), _lib.ptr(V), _lib.ptr(cv), _lib.ptr(ni), _lib.ptr(obj), _lib.ptr(tm)))
        mo = np.full(B, np.nan)
        if time_limit > 0:
            _lib.check(self._lib.omc_altmin_master_objectives(self._h, B, _lib.ptr(mo)))
        # master_objective = evaluate_objective(U V) (OMC.jl:920-927), evaluated on the device from the factors
        return [dict(converged=bool(cv[b]), U=U[b].reshape((n, k), order="F"), V=V[b].reshape((k, m), order="F"),
                     solve_time=float(tm[b]), n_iters=int(ni[b]), max_iters=max_iters, objectives=list(obj[b, :ni[b] - (0 if cv[b] or ni[b] == 0 or not np.isnan(obj[b, ni[b] - 1]) else 1)]),
                     master_objective=float(mo[b])) for b in range(B)]

    # ---- objective -------------------------------------------------------------------------------------
    # ---- Shor minors (OMC.jl:2545-2640) ------------------------------------------------------------------------
    def shor_count(self, num_entries_present_list):
        cl = np.asarray(list(num_entries_present_list), dtype=np.int32)
        out = np.zeros(max(len(cl), 1), dtype=np.int64)
        _lib.check(self._lib.omc_shor_count(self._h, len(cl), _lib.ptr(cl), _lib.ptr(out)))
        return out[:len(cl)]

    def generate_rank1_matrix_completion_Shor_constraints_indexes(self, num_entries_present_list):
        """OMC.jl:2545-2612 on the device.  Returns an int64 array (count, 4) of 1-based (i1, i2, j1, j2) in the
        reference's push order (the instance's `indices` is the mask)."""
        cl = np.asarray(list(num_entries_present_list), dtype=np.int32)
        cnt = np.zeros(1, dtype=np.int64)
        _lib.check(self._lib.omc_shor_indexes(self._h, len(cl), _lib.ptr(cl), 0, None, _lib.ptr(cnt)))
        out = np.zeros((int(cnt[0]), 4), dtype=np.int64)
        if cnt[0] > 0:
            _lib.check(self._lib.omc_shor_indexes(self._h, len(cl), _lib.ptr(cl), int(cnt[0]), _lib.ptr(out), _lib.ptr(cnt)))
        return out

    def generate_violated_Shor_minors(self, X, num_entries_present_list, Shor_constraints_indexes, n_minors):
        """OMC.jl:2614-2640 on the device.  X has shape (k, n, m) as in the reference; returns [(score, (i1, i2, j1, j2)), ...]
        in decreasing (score, tuple) order, at most n_minors of them."""
        X = np.asarray(X, dtype=np.float64)
        if X.shape != (self.k, self.n, self.m):
            raise ValueError("Dimension mismatch.\nInput array X must have size (k, n, m).")
        Xf = np.asfortranarray(X).ravel(order="F")        # Julia Array{Float64,3} layout
        cl = np.asarray(list(num_entries_present_list), dtype=np.int32)
        ex = np.ascontiguousarray(np.asarray(list(Shor_constraints_indexes), dtype=np.int64).reshape(-1, 4))
        K = int(n_minors)
        sc = np.zeros(max(K, 1)); mi = np.zeros((max(K, 1), 4), dtype=np.int64); no = np.zeros(1, dtype=np.int32)
        _lib.check(self._lib.omc_violated_shor_minors(self._h, _lib.ptr(Xf), len(cl), _lib.ptr(cl), len(ex), _lib.ptr(ex) if len(ex) else None,
                                                      K, _lib.ptr(sc), _lib.ptr(mi), _lib.ptr(no)))
        return [(float(sc[r]), tuple(int(v) for v in mi[r])) for r in range(int(no[0]))]

    def shor_last_stats(self):
        ms = np.zeros(1); c = np.zeros(1, dtype=np.int64)
        _lib.check(self._lib.omc_shor_last_stats(self._h, _lib.ptr(ms), _lib.ptr(c)))
        return {"ms": float(ms[0]), "candidates": int(c[0])}

    def evaluate_objective(self, X):
        X = np.asarray(X, dtype=np.float64)
        single = X.ndim == 2
        Xs = [X] if single else list(X)
        for x in Xs:
            if x.shape != (self.n, self.m):
                raise ValueError("Dimension mismatch.\nInput matrix X must have size (n, m).")          # OMC.jl:2337-2348
        Xb = np.ascontiguousarray(np.stack([np.asfortranarray(x).ravel(order="F") for x in Xs]))
        out = np.zeros(len(Xs))
        _lib.check(self._lib.omc_evaluate_objective(self._h, len(Xs), _lib.ptr(Xb), _lib.ptr(out)))
        return float(out[0]) if single else out
