"""ctypes front of oracle/libomc_cpu_ref.so (omc_cpu_ref.cpp): the compiled single-thread restatement of the node relaxation, used by
bench.py's cpu_baseline and checked against the numpy oracle in tests/test_cpu_ref.py.  TEST / BASELINE INFRASTRUCTURE -- the product path
never imports this module.  The rows of a node (OMC.jl:1558-1685) are built by omc_oracle.build_rows / row_subspace and handed to the
library as dense coefficient arrays."""
import ctypes as C
import os
import subprocess

import numpy as np

import omc_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libomc_cpu_ref.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.omc_cpu_relax_nodes.restype = C.c_int
        _LIB.omc_cpu_ref_openmp.restype = C.c_int
    return _LIB


def pack_params(p):
    return np.array([p.eps_gap, p.eps_feas, p.max_iters, p.check_every, p.rho_scale, p.rho_f_ratio, p.relax, p.stall_checks, p.bump, p.bump_factor,
                     p.bump_window, p.bump_after, p.bump_max, p.bump_ratio, p.early_stop_after, p.early_stop_factor], dtype=np.float64)


def pack_node(inst, cuts, cut_type, U_lower=None, U_upper=None, quirk_q1=True):
    n, k = inst.n, inst.k
    rows = orc.build_rows(inst, cuts, cut_type, U_lower, U_upper, quirk_q1)
    R = len(rows)
    Q = np.ascontiguousarray(orc.row_subspace(rows, n, k))
    AY = np.zeros((R, n * n)); AU = np.zeros((R, n * k)); xs = np.zeros((R, n)); kind = np.zeros(R, dtype=np.int32)
    for rr in range(R):
        if rows.kinds[rr] == "trace":
            AY[rr] = np.eye(n).ravel(); kind[rr] = 0
        elif rows.kinds[rr] == "cut":
            AY[rr] = np.outer(rows.xs[rr], rows.xs[rr]).ravel(); xs[rr] = rows.xs[rr]; kind[rr] = 1
        else:
            kind[rr] = 2
        AU[rr] = rows.CU[rr].ravel()
    return dict(R=R, r=Q.shape[1], AY=AY, AU=AU, b=np.array(rows.rhs, dtype=np.float64), kind=kind, xs=xs, Q=Q if Q.size else np.zeros((n, 1)))


def relax_nodes(inst, nodes, cut_type="linear", params=None, threads=1):
    """nodes: list of cut lists.  Returns one dict per node (objective, dual_bound, iters, status_code, rp, rd, rho) and Y of node 0."""
    lib = load()
    p = params or orc.RelaxParams()
    packs = [pack_node(inst, cuts, cut_type, quirk_q1=p.reference_quirk_q1) for cuts in nodes]
    B = len(packs)
    A = np.ascontiguousarray(inst.A, dtype=np.float64); mask = np.ascontiguousarray(inst.indices, dtype=np.uint8)
    R = np.array([q["R"] for q in packs], dtype=np.int32); r = np.array([q["r"] for q in packs], dtype=np.int32)

    def ptrs(key):
        return (C.c_void_p * B)(*[q[key].ctypes.data for q in packs])
    out = np.zeros((B, 7)); Y0 = np.zeros((inst.n, inst.n)); P = pack_params(p)
    rc = lib.omc_cpu_relax_nodes(C.c_int(inst.n), C.c_int(inst.m), C.c_int(inst.k), A.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), C.c_double(inst.gamma),
                                 C.c_int(B), R.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), ptrs("AY"), ptrs("AU"), ptrs("b"), ptrs("kind"), ptrs("xs"), ptrs("Q"),
                                 P.ctypes.data_as(C.c_void_p), C.c_int(int(threads)), out.ctypes.data_as(C.c_void_p), Y0.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"omc_cpu_relax_nodes: {rc}")
    res = [dict(objective=o[0], dual_bound=o[1], rp=o[2], rd=o[3], rho=o[4], iters=int(o[5]), status_code=int(o[6])) for o in out]
    return res, Y0
