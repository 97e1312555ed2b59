"""Utility functions for the RL environments (pde_opt/rl_utils.py), evaluated on the GPU.

``detect_vortices`` keeps the reference's signature and return dictionary
(pde_opt/rl_utils.py:19-84); the phase-circulation census itself runs in a HIP kernel
(``pdeopt_detect_vortices``, csrc/reduce.hip).  For a wavefunction that already lives on the GPU
(a ``VectorPDEEnv`` of GPE environments) call ``engine.detect_vortices(...)`` directly: only the three
counters per environment cross PCIe.
"""

from __future__ import annotations

import numpy as np

from . import _lib as L


def density(psi):
    """|psi|^2 (rl_utils.py:10-11)"""
    return np.abs(np.asarray(psi)) ** 2


def detect_vortices(psi, amp_thresh: float = 0.0, tol: float = 0.5, engine=None):
    """Quantum vortices of a complex (N, M) periodic wavefunction by phase circulation per grid cell.

    Returns the reference's dictionary: ``winding`` (N, M) int32, ``positions`` (K, 2) float32 cell
    centres ``(i + 0.5, j + 0.5)``, ``charges`` (K,), ``num_vortices``, ``total_topological_charge``,
    ``abs_charge_count``."""
    psi = np.asarray(psi)
    if psi.ndim == 3 and psi.shape[-1] == 2 and not np.iscomplexobj(psi):
        pairs = psi
    else:
        if psi.ndim != 2:
            raise ValueError(f"psi must be a complex (N, M) array, got shape {psi.shape}")
        pairs = np.stack([psi.real, psi.imag], axis=-1)
    dtype = np.float64 if pairs.dtype == np.float64 else np.float32
    pairs = np.ascontiguousarray(pairs, dtype=dtype)
    if engine is None:
        from .engine import default_engine

        engine = default_engine()
    nx, ny = pairs.shape[:2]
    engine.configure(equation=L.EQ_GPE, dtype=dtype, nx=nx, ny=ny, batch=1, hx=1.0, hy=1.0)
    engine.set_state(pairs[None])
    counts, winding = engine.detect_vortices(amp_thresh, tol)
    n_int = winding[0]
    idx = np.argwhere(n_int != 0)
    charges = n_int[n_int != 0]
    return {
        "winding": n_int,
        "positions": idx.astype(np.float32) + 0.5,
        "charges": charges,
        "num_vortices": int(counts[0, 0]),
        "total_topological_charge": int(counts[0, 1]),
        "abs_charge_count": int(counts[0, 2]),
    }
