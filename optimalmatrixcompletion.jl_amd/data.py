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
