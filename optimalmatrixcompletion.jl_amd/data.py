"""Synthetic instances with the distribution of the reference's generator (/root/reference/src/utils.jl:3-26,
68-111) and of the README quick-start (/root/reference/README.md:31-41).  Julia's MersenneTwister streams cannot
be reproduced outside Julia, so numpy seeds are used; SHA-256 of (A, mask) identifies an instance."""
from __future__ import annotations

import hashlib

import numpy as np


def generate_matrix_completion_data(k, n, m, n_indices, seed, noise=0.01, max_tries=100):
    """A = L R + noise * E (utils.jl:98-103); mask = n_indices uniformly random cells, redrawn (<= 100 tries) until
    every row and column holds an entry (utils.jl:13-25).  Draw order: L, R, E, permutation(s)."""
    if not n <= m:
        raise ValueError(f"Input matrix A must have size (n, m) with n <= m.\nn = {n}, m = {m} supplied instead.")   # utils.jl:79-84
    if n_indices < (n + m) * k:
        raise ValueError("System is under-determined.\nn_indices must be at least (n + m) * k.")                      # utils.jl:85-90
    if n_indices > n * m:
        raise ValueError("Cannot generate random indices of length more than the size of matrix A.")                 # utils.jl:91-95
    rng = np.random.default_rng(seed)
    L = rng.standard_normal((n, k)); R = rng.standard_normal((k, m)); E = rng.standard_normal((n, m))
    A = L @ R + noise * E
    it = 0
    while True:
        perm = rng.permutation(n * m)[:n_indices]
        vec = np.zeros(n * m, bool); vec[perm] = True
        mask = vec.reshape((n, m), order="F")
        if (mask.any(0).all() and mask.any(1).all()) or it >= max_tries:
            return A, mask
        it += 1


def readme_instance(n, m, seed):
    """README quick-start: A = randn(n, m), indices = rand([0,1], (n, m))  (README.md:33)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, m)), rng.integers(0, 2, (n, m)).astype(bool)


def branching_instance(seed=0, n=100, m=100, noise=0.3, frac=0.1):
    """A 100 x 100 rank-1 instance whose B&B tree really branches (BASELINE config 2's own tree closes at the root: its relaxation
    is tight): the reference's generator (utils.jl:98-103) with noise 0.3 instead of 0.01 and 10 % observed instead of 20 %."""
    return generate_matrix_completion_data(1, n, m, int(round(frac * n * m)), seed, noise=noise)


def instance_sha256(A, mask):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(A, dtype=np.float64).tobytes()); h.update(np.ascontiguousarray(mask, dtype=np.uint8).tobytes())
    return h.hexdigest()


# BASELINE.json configs: (n, m, k, observed fraction, cut type, breakpoints)
CONFIGS = {
    1: dict(n=50, m=50, k=1, kind="readme", cut_type="linear", breakpoints="smallest_1_eigvec"),
    2: dict(n=100, m=100, k=1, frac=0.2, kind="lowrank", cut_type="linear", breakpoints="smallest_1_eigvec"),
    3: dict(n=200, m=200, k=1, frac=0.2, kind="lowrank", cut_type="linear", breakpoints="smallest_1_eigvec"),
    4: dict(n=500, m=500, k=2, frac=0.2, kind="lowrank", cut_type="linear3", breakpoints="smallest_2_eigvec"),
    5: dict(n=1000, m=1000, k=2, frac=0.3, kind="lowrank", cut_type="linear", breakpoints="smallest_1_eigvec"),
}


def config_instance(cfg, seed=0, gamma=80.0):
    c = CONFIGS[cfg]
    if c["kind"] == "readme":
        A, mask = readme_instance(c["n"], c["m"], seed)
    else:
        A, mask = generate_matrix_completion_data(c["k"], c["n"], c["m"], int(round(c["frac"] * c["n"] * c["m"])), seed)
    return A, mask, gamma, c


# ----------------------------------------------------------------------------------------------------------------------
# On-disk formats: an instance is a directory with A.npy (fp64, n x m), mask.bits (Julia BitMatrix
# chunk layout: column-major bit order, LSB first, padded to 64 bits -- what omc_instance_create_bits consumes) and
# instance.json; a node-descriptor replay log is one .npz in the wire format of include/omc.h (cut_x, cut_Uhat, cut_dir).
# ----------------------------------------------------------------------------------------------------------------------
def compute_MSE(X, A, indices, kind="out"):
    """OMC.jl:2373-2409."""
    D2 = (np.asarray(X, float) - np.asarray(A, float)) ** 2
    if kind == "out":
        cnt = indices.size - int(indices.sum())
        return 0.0 if cnt == 0 else float(D2[~indices].sum()) / cnt
    if kind == "in":
        cnt = int(indices.sum())
        return 0.0 if cnt == 0 else float(D2[indices].sum()) / cnt
    if kind == "all":
        return float(D2.sum()) / indices.size
    raise ValueError('Input argument `kind` not recognized!\nMust be one of "out", "in", or "all".')      # OMC.jl:2404-2407


def pack_mask_bits(mask):
    """Julia BitMatrix chunks of a boolean n x m matrix (uint64, column-major bit order, LSB first)."""
    v = np.asarray(mask, bool).ravel(order="F")
    pad = (-len(v)) % 64
    v = np.concatenate([v, np.zeros(pad, bool)])
    return np.packbits(v.reshape(-1, 8), axis=1, bitorder="little").ravel().view(np.uint64)


def unpack_mask_bits(chunks, n, m):
    b = np.unpackbits(np.asarray(chunks, np.uint64).view(np.uint8), bitorder="little")[: n * m]
    return b.astype(bool).reshape((n, m), order="F")


def save_instance(path, A, mask, gamma, k, meta=None):
    import json, os
    os.makedirs(path, exist_ok=True)
    A = np.asarray(A, np.float64); mask = np.asarray(mask, bool)
    np.save(os.path.join(path, "A.npy"), A)
    pack_mask_bits(mask).tofile(os.path.join(path, "mask.bits"))
    info = dict(n=int(A.shape[0]), m=int(A.shape[1]), k=int(k), gamma=float(gamma), n_indices=int(mask.sum()), sha256=instance_sha256(A, mask))
    info.update(meta or {})
    with open(os.path.join(path, "instance.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    return info


def load_instance(path):
    import json, os
    with open(os.path.join(path, "instance.json")) as f:
        info = json.load(f)
    A = np.load(os.path.join(path, "A.npy"), allow_pickle=False)
    mask = unpack_mask_bits(np.fromfile(os.path.join(path, "mask.bits"), dtype=np.uint64), info["n"], info["m"])
    if instance_sha256(A, mask) != info["sha256"]:
        raise ValueError("instance files do not match the recorded SHA-256")
    return A, mask, info


DIR_CODES = {"left": 0, "middle": 1, "right": 2, "inner_left": 3, "inner_right": 4}
DIR_NAMES = {v: k_ for k_, v in DIR_CODES.items()}


def save_nodes(path, nodes, n, k):
    """Replay log of node descriptors: lists of cuts (x, U_hat, directions) -> concatenated wire-format arrays."""
    L = np.array([len(c) for c in nodes], dtype=np.int32); tot = int(L.sum())
    cx = np.zeros((tot, n)); cU = np.zeros((tot, n, k)); cd = np.zeros((tot, k), dtype=np.int8); t = 0
    for cuts in nodes:
        for (x, U, dirs) in cuts:
            cx[t] = x; cU[t] = np.asarray(U).reshape(n, k); cd[t] = [DIR_CODES[d] for d in dirs]; t += 1
    np.savez_compressed(path, L=L, cut_x=cx, cut_Uhat=cU, cut_dir=cd)


def load_nodes(path):
    z = np.load(path, allow_pickle=False)
    nodes = []; t = 0
    for Lb in z["L"]:
        nodes.append([(z["cut_x"][t + l], z["cut_Uhat"][t + l], [DIR_NAMES[int(c)] for c in z["cut_dir"][t + l]]) for l in range(int(Lb))])
        t += int(Lb)
    return nodes
