"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/omc.h declares, and fails
loudly (no CPU fallback) when no GPU is visible.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "omc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(omc_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(omc):
    lib = omc.load()
    names = declared_symbols()
    assert len(names) >= 15
    for s in names:
        assert hasattr(lib, s), f"{s} declared in include/omc.h but not exported"
    assert set(omc.EXPORTS) <= set(names)
    assert lib.omc_version() == 100


def test_params_default_and_struct_layout(omc):
    p = omc.default_params()
    assert (p.eps_gap, p.eps_feas, p.max_iters, p.check_every) == (1e-6, 1e-7, 3000, 25)
    assert (p.rho_scale, p.rho_f_ratio, p.relax) == (1.0, 0.1, 1.6)
    assert p.reference_quirk_q1 == 1 and p.breakpoints == 1 and p.stall_checks == 8
    with pytest.raises(TypeError):
        omc.default_params(not_a_field=1)


def test_no_cpu_fallback(omc):
    """On a box without a GPU the engine must refuse to construct (OMC_ERR_NO_DEVICE), never silently compute on CPU."""
    lib = omc.load()
    if lib.omc_device_count() > 0:
        pytest.skip("GPU present")
    A = np.zeros((4, 5)); mask = np.ones((4, 5), bool)
    with pytest.raises(omc.OmcError) as e:
        omc.Engine(A, mask, 80.0, 1)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_argument_checks_mirror_reference_errors(omc):
    lib = omc.load()
    h = C.c_void_p()
    A = np.zeros((5, 3)); mask = np.ones((5, 3), np.uint8)
    # n <= m required (OMC.jl:249-254) -- checked before any device call
    rc = lib.omc_instance_create(5, 3, 1, A.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 80.0, 0, C.byref(h))
    assert rc == -2 and b"n <= m" in lib.omc_last_error()
    rc = lib.omc_instance_create(3, 5, 1, None, None, 80.0, 0, C.byref(h))
    assert rc == -3
    rc = lib.omc_instance_create(3, 5, 0, A.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 80.0, 0, C.byref(h))
    assert rc == -3
    assert lib.omc_relax_solve(None) == -3
    with pytest.raises(ValueError):
        omc.Engine(np.zeros((3, 4)), np.ones((3, 5), bool), 80.0, 1)       # size(A) == size(indices) (OMC.jl:240-246)


def test_host_side_helpers(omc):
    bnb = omc.pkg.bnb; data = omc.pkg.data
    assert bnb.child_directions("linear", 2) == [["left", "left"], ["right", "left"], ["left", "right"], ["right", "right"]]
    A, mask = data.generate_matrix_completion_data(1, 10, 12, 40, seed=0)
    assert mask.sum() == 40 and mask.any(0).all() and mask.any(1).all()
    A2, mask2 = data.generate_matrix_completion_data(1, 10, 12, 40, seed=0)
    assert data.instance_sha256(A, mask) == data.instance_sha256(A2, mask2)
    with pytest.raises(ValueError):
        data.generate_matrix_completion_data(1, 12, 10, 40, seed=0)
    with pytest.raises(ValueError):
        data.generate_matrix_completion_data(2, 10, 12, 30, seed=0)        # under-determined (utils.jl:85-90)
    nodes = list(range(10))
    assert bnb.shard_nodes(nodes, 1, 4) == [1, 5, 9]
    assert sorted(sum((bnb.shard_nodes(nodes, r, 3) for r in range(3)), [])) == nodes
    x = np.arange(4.0); U = np.ones((4, 1))
    kids = bnb.make_children([("c0",)], dict(breakpoint_vec=x, U=U), "linear", 1)
    assert len(kids) == 2 and kids[0][0] == ("c0",) and kids[1][-1][2] == ["right"]


def test_instance_and_node_log_round_trip(omc, tmp_path):
    """On-disk formats: instance directory (A.npy, Julia-BitMatrix mask chunks, JSON) and node replay log."""
    data = omc.pkg.data
    A, mask = data.generate_matrix_completion_data(1, 9, 13, 60, seed=3)
    info = data.save_instance(str(tmp_path / "inst"), A, mask, 80.0, 1, meta={"config": "test"})
    A2, mask2, info2 = data.load_instance(str(tmp_path / "inst"))
    assert np.array_equal(A, A2) and np.array_equal(mask, mask2) and info2["sha256"] == info["sha256"] and info2["n_indices"] == 60
    chunks = data.pack_mask_bits(mask)
    assert chunks.dtype == np.uint64 and len(chunks) == (9 * 13 + 63) // 64
    assert all(bool((int(chunks[e // 64]) >> (e % 64)) & 1) == bool(mask[e % 9, e // 9]) for e in range(9 * 13))   # column-major, LSB first
    rng = np.random.default_rng(0)
    nodes = [[], [(rng.standard_normal(9), rng.standard_normal((9, 1)), ["left"])],
             [(rng.standard_normal(9), rng.standard_normal((9, 1)), ["right"]), (rng.standard_normal(9), rng.standard_normal((9, 1)), ["left"])]]
    data.save_nodes(str(tmp_path / "nodes.npz"), nodes, 9, 1)
    back = data.load_nodes(str(tmp_path / "nodes.npz"))
    assert [len(c) for c in back] == [0, 1, 2] and back[2][1][2] == ["left"] and np.array_equal(back[2][0][0], nodes[2][0][0])
    assert data.compute_MSE(A, A, mask, "all") == 0.0
    with pytest.raises(ValueError):
        data.compute_MSE(A, A, mask, "bogus")


def test_shor_and_altmin_argument_checks_without_a_handle(omc):
    """NULL handles / pointers are rejected before anything touches the device (no GPU needed)."""
    lib = omc.load()
    cnt = np.zeros(1, dtype=np.int64); cl = np.array([4], dtype=np.int32)
    assert lib.omc_shor_count(None, 1, omc.pkg._lib.ptr(cl), omc.pkg._lib.ptr(cnt)) == -3
    assert lib.omc_shor_indexes(None, 1, omc.pkg._lib.ptr(cl), 0, None, omc.pkg._lib.ptr(cnt)) == -3
    no = np.zeros(1, dtype=np.int32)
    assert lib.omc_violated_shor_minors(None, None, 1, omc.pkg._lib.ptr(cl), 0, None, 10, None, None, omc.pkg._lib.ptr(no)) == -3
    assert lib.omc_shor_last_stats(None, None, None) == -3
    assert lib.omc_altmin_batch(None, 1, 0, 1, None, None, None, None, None, 1e-5, 10, 1.0, None, None, None, None, None, None) == -3
    assert b"NULL" in lib.omc_last_error() or b"handle" in lib.omc_last_error()


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` spawns its ranks itself; asking for more ranks than there are GPUs must fail at once, not hang in a collective."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in (r.stderr + r.stdout)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)
