"""ctypes binding of libomc_hip.so (the C ABI in include/omc.h).  No fallback: if the library is missing or no
GPU is visible every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMC_AMD_LIB points the loader at another build of the same library (the instrumented `make stamps` build, an A/B variant)
LIB_PATH = os.environ.get("OMC_AMD_LIB") or os.path.join(_HERE, "lib", "libomc_hip.so")

EXPORTS = [
    "omc_relax_params_default", "omc_last_error", "omc_version", "omc_device_count", "omc_instance_create",
    "omc_instance_create_bits", "omc_instance_destroy", "omc_relax_batch", "omc_relax_stage", "omc_relax_solve",
    "omc_relax_fetch", "omc_relax_submit", "omc_relax_poll", "omc_relax_wait", "omc_altmin_batch", "omc_evaluate_objective", "omc_separation_batch", "omc_round_Y_batch",
    "omc_last_kernel_stats", "omc_last_solver_info", "omc_last_subspace_stats", "omc_set_node_rho_scales", "omc_left_singular_batch", "omc_relax_reserve", "omc_relax_append", "omc_relax_fetch_done", "omc_relax_hold", "omc_debug_stamps", "omc_tuning_set", "omc_tuning_reload_env", "omc_debug_residuals", "omc_debug_diag", "omc_debug_aa",
    "omc_shor_count", "omc_shor_indexes", "omc_violated_shor_minors", "omc_shor_last_stats",
    "omc_relax_stage_shor", "omc_relax_fetch_shor", "omc_relax_batch_shor", "omc_set_shor_penalties", "omc_set_shor_keep_V", "omc_relax_fetch_shor_V", "omc_last_shor_subspace_stats", "omc_state_pool_create", "omc_relax_set_warm",
    "omc_altmin_master_objectives", "omc_comm_unique_id", "omc_comm_init", "omc_allreduce_bounds", "omc_bcast_incumbent", "omc_allgather_records", "omc_comm_destroy",
]


class RelaxParams(C.Structure):
    _fields_ = [("eps_gap", C.c_double), ("eps_feas", C.c_double), ("max_iters", C.c_int), ("check_every", C.c_int),
                ("rho_scale", C.c_double), ("rho_f_ratio", C.c_double), ("relax", C.c_double), ("time_limit", C.c_double),
                ("reference_quirk_q1", C.c_int), ("breakpoints", C.c_int), ("stall_checks", C.c_int), ("bump_max", C.c_int), ("bump_ratio", C.c_double),
                ("bump_factor", C.c_double), ("bump_after", C.c_int), ("bump_window", C.c_int), ("slots", C.c_int),
                ("accel", C.c_int), ("aa_mem", C.c_int), ("aa_every", C.c_int), ("aa_start", C.c_int), ("aa_reg", C.c_double), ("aa_safeguard", C.c_double), ("first_wins", C.c_int),
                ("early_stop_after", C.c_int), ("early_stop_factor", C.c_double)]


class OmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libomc_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load the shared library (raises if it has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmcError(-100, f"{LIB_PATH} not found: build it first (__graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int); vp = C.c_void_p
    lib.omc_last_error.restype = C.c_char_p
    lib.omc_version.restype = C.c_int
    lib.omc_device_count.restype = C.c_int
    lib.omc_relax_params_default.argtypes = [C.POINTER(RelaxParams)]
    lib.omc_instance_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_double, C.c_int, C.POINTER(vp)]
    lib.omc_instance_create_bits.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_double, C.c_int, C.POINTER(vp)]
    lib.omc_instance_destroy.argtypes = [vp]
    lib.omc_instance_destroy.restype = None
    lib.omc_relax_batch.argtypes = [vp, C.c_int, C.POINTER(RelaxParams), C.c_int] + [vp] * 17
    lib.omc_relax_stage.argtypes = [vp, C.c_int, C.POINTER(RelaxParams), C.c_int] + [vp] * 6
    lib.omc_relax_solve.argtypes = [vp]
    lib.omc_relax_stage_shor.argtypes = [vp, C.c_int, C.POINTER(RelaxParams), C.c_int] + [vp] * 10
    lib.omc_relax_fetch_shor.argtypes = [vp, vp]
    lib.omc_relax_batch_shor.argtypes = [vp, C.c_int, C.POINTER(RelaxParams), C.c_int] + [vp] * 22
    lib.omc_set_shor_penalties.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    lib.omc_set_shor_keep_V.argtypes = [vp, C.c_int]
    lib.omc_relax_fetch_shor_V.argtypes = [vp, vp]
    lib.omc_last_shor_subspace_stats.argtypes = [vp, vp]
    lib.omc_state_pool_create.argtypes = [vp, C.c_int]
    lib.omc_relax_set_warm.argtypes = [vp, C.c_int, vp, vp]
    lib.omc_relax_submit.argtypes = [vp]
    lib.omc_relax_poll.argtypes = [vp, vp, vp, vp]
    lib.omc_relax_wait.argtypes = [vp]
    lib.omc_relax_fetch.argtypes = [vp] + [vp] * 11
    lib.omc_altmin_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_double, C.c_int, C.c_double,
                                     vp, vp, vp, vp, vp, vp]
    lib.omc_evaluate_objective.argtypes = [vp, C.c_int, vp, vp]
    lib.omc_separation_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.omc_round_Y_batch.argtypes = [vp, C.c_int, vp, vp]
    lib.omc_last_kernel_stats.argtypes = [vp, vp, vp, vp]
    lib.omc_last_solver_info.argtypes = [vp, vp]
    lib.omc_last_subspace_stats.argtypes = [vp, vp]
    lib.omc_set_node_rho_scales.argtypes = [vp, C.c_int, vp]
    lib.omc_debug_stamps.argtypes = [vp, vp]
    lib.omc_left_singular_batch.argtypes = [vp, C.c_int, vp, vp]
    lib.omc_relax_reserve.argtypes = [vp, C.c_int, C.c_int]
    lib.omc_relax_hold.argtypes = [vp, C.c_int]
    lib.omc_relax_fetch_done.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.omc_relax_append.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.omc_tuning_set.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.omc_tuning_reload_env.argtypes = [vp]
    lib.omc_debug_residuals.argtypes = [vp, vp, vp]
    lib.omc_debug_diag.argtypes = [vp, vp]
    lib.omc_debug_aa.argtypes = [vp, vp, vp]
    lib.omc_shor_count.argtypes = [vp, C.c_int, vp, vp]
    lib.omc_shor_indexes.argtypes = [vp, C.c_int, vp, C.c_int64, vp, vp]
    lib.omc_violated_shor_minors.argtypes = [vp, vp, C.c_int, vp, C.c_int64, vp, C.c_int, vp, vp, vp]
    lib.omc_shor_last_stats.argtypes = [vp, vp, vp]
    lib.omc_altmin_master_objectives.argtypes = [vp, C.c_int, vp]
    lib.omc_comm_unique_id.argtypes = [vp]
    lib.omc_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.omc_allreduce_bounds.argtypes = [vp, vp, vp, vp]
    lib.omc_bcast_incumbent.argtypes = [vp, C.c_int, vp]
    lib.omc_allgather_records.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.omc_comm_destroy.argtypes = [vp]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise OmcError(rc, load().omc_last_error().decode("utf-8", "replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["ALIGNED"] + (["F_CONTIGUOUS"] if order == "F" else ["C_CONTIGUOUS"]))
