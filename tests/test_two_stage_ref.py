"""CPU checks of the structural claims behind localmd_amd/csrc/sytrd2.hip (see tests/two_stage_ref.py)."""
import numpy as np
import pytest

from tests.two_stage_ref import apply_q1, apply_q2_blocked, panel_qr_hr, stage1, stage2_wavefront, to_band


def test_panel_reconstruction_is_a_product_of_elementary_reflectors():
    rng = np.random.default_rng(0)
    m, b = 40, 6
    P = rng.standard_normal((m, b))
    V, T, Rs = panel_qr_hr(P)
    H = np.eye(m) - V @ T @ V.T
    assert np.abs(H.T @ H - np.eye(m)).max() < 1e-12
    assert np.abs(np.tril(T, -1)).max() < 1e-12
    HP = H.T @ P
    assert np.abs(HP[b:]).max() < 1e-12 and np.abs(HP[:b] - Rs).max() < 1e-12
    assert np.abs(np.triu(V[:b], 1)).max() == 0 and np.abs(np.diag(V) - 1).max() < 1e-12
    # tau_i = T_ii = 2 / |v_i|^2 and T^{-1} = striu(V^T V) + diag(1 / tau): the identity apply_q builds T from
    tau = np.diag(T)
    assert np.abs(tau - 2 / np.sum(V * V, axis=0)).max() < 1e-12
    assert np.abs(np.linalg.inv(T) - (np.triu(V.T @ V, 1) + np.diag(1 / tau))).max() < 1e-10


@pytest.mark.parametrize("n,b,group", [(97, 8, 8), (64, 8, 8), (33, 4, 8), (130, 16, 16), (50, 8, 4)])
def test_two_stage_reduction_and_blocked_back_transformation(n, b, group):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 5))
    A = X @ X.T
    Bd, refl1 = stage1(A, b)
    outside = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > b
    assert np.abs(Bd[outside]).max() < 1e-9 * np.abs(A).max()
    AB = to_band(Bd, b)
    V2 = stage2_wavefront(AB, n, b)
    assert np.abs(AB[:, 2:]).max() < 1e-9 * np.abs(A).max()
    d, e = AB[:, 0], AB[:-1, 1]
    Tm = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w0 = np.linalg.eigvalsh(A)
    w, Z = np.linalg.eigh(Tm)
    assert np.abs(w0 - w).max() < 1e-10 * np.abs(w0).max()
    E = apply_q1(refl1, apply_q2_blocked(V2, Z, n, b, group))
    assert np.abs(A @ E - E * w[None, :]).max() < 1e-10 * np.abs(w).max()
    assert np.abs(E.T @ E - np.eye(n)).max() < 1e-10
