"""oracle/oracle_matrix.c against the reference's OWN radiation_matrix.F90, compiled unmodified into
oracle/_ref/libecrad_refleaf.so: every routine the SPARTACUS solvers use (mat_x_vec, singlemat_x_vec, mat_x_mat with
both sparsity patterns, singlemat_x_mat, mat_x_singlemat, identity_minus_mat_x_mat, solve_vec / solve_mat for
m = 2, 3 and the general LU path, expm for the 6x6 longwave and 9x9 shortwave matrices, fast_expm_exchange_2/3).
The inputs are shaped like the solvers' own: Gamma matrices built from optical depths, two-stream gammas and lateral
transfer rates, albedo/transmittance-like matrices, exchange rates spanning 1e-12 .. 30.  Tolerance 1e-12 relative to
the largest element of each matrix (same double arithmetic, different compilers: flang may contract to FMAs)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.skipif(
    not (os.path.exists(pyoracle.REF_LEAF_PATH) or os.path.isdir("/root/reference")),
    reason="oracle/_ref not built and /root/reference absent")

D = C.POINTER(C.c_double)
TOL = 1e-12
N = 257


def p(a):
    return a.ctypes.data_as(D)


def F(*shape):
    return np.zeros(shape, order="F")


@pytest.fixture(scope="module")
def libs(oracle_lib):
    return C.CDLL(pyoracle.REF_LEAF_PATH), oracle_lib.lib()


def close(a, b, tol=TOL):
    scale = np.abs(b).reshape(b.shape[0], -1).max(axis=1) + 1e-300
    err = np.abs(a - b).reshape(b.shape[0], -1).max(axis=1) / scale
    assert err.max() < tol, (err.max(), int(err.argmax()))


def rand_mats(rng, m, diag_dominant=False):
    A = np.asfortranarray(rng.uniform(0.0, 1.0, (N, m, m)))
    if diag_dominant:
        A = A * 0.3
        for j in range(m):
            A[:, j, j] += 1.0
    return np.asfortranarray(A)


def gamma_matrix(rng, nreg, sw):
    """Gamma_z1 as radiation_spartacus_{sw,lw}.F90 section 3.3a builds it"""
    m = (3 if sw else 2) * nreg
    G = F(N, m, m)
    od = 10.0 ** rng.uniform(-6, 1.2, (N, nreg))
    ssa = rng.uniform(0, 0.999999, (N, nreg))
    g1 = 2.0 - ssa * 1.6
    g2 = ssa * 0.5
    g3 = rng.uniform(0.2, 0.8, (N, nreg))
    mu0 = rng.uniform(0.05, 1.0, N)
    rate = 10.0 ** rng.uniform(-3, 1, (N, nreg, nreg))
    for j in range(nreg):
        G[:, j, j] = od[:, j] * g1[:, j]
        G[:, j + nreg, j] = od[:, j] * g2[:, j]
        if sw:
            G[:, j, j + 2 * nreg] = -od[:, j] * ssa[:, j] * g3[:, j]
            G[:, j + nreg, j + 2 * nreg] = od[:, j] * ssa[:, j] * (1 - g3[:, j])
            G[:, j + 2 * nreg, j + 2 * nreg] = -od[:, j] / mu0
    for j in range(nreg - 1):
        G[:, j, j] += rate[:, j, j + 1]
        G[:, j + 1, j + 1] += rate[:, j + 1, j]
        G[:, j + 1, j] = -rate[:, j, j + 1]
        G[:, j, j + 1] = -rate[:, j + 1, j]
        if sw:
            k = 2 * nreg
            G[:, j + k, j + k] -= rate[:, j, j + 1]
            G[:, j + 1 + k, j + 1 + k] -= rate[:, j + 1, j]
            G[:, j + 1 + k, j + k] = rate[:, j, j + 1]
            G[:, j + k, j + 1 + k] = rate[:, j + 1, j]
    G[:, nreg:2 * nreg, nreg:2 * nreg] = -G[:, :nreg, :nreg]
    G[:, :nreg, nreg:2 * nreg] = -G[:, nreg:2 * nreg, :nreg]
    return np.asfortranarray(G)


@pytest.mark.parametrize("nreg,sw", [(3, True), (2, True), (3, False), (2, False)])
def test_expm(libs, nreg, sw):
    ref, ora = libs
    G = gamma_matrix(np.random.default_rng(5), nreg, sw)
    m = G.shape[1]
    a, b = G.copy(order="F"), G.copy(order="F")
    ref.ref_expm(N, N - 3, m, p(a), 1 if sw else 0)
    ora.om_expm(N, N - 3, m, p(b), 1 if sw else 0)
    close(b[:N - 3], a[:N - 3], 2e-12)
    assert np.array_equal(b[N - 3:], a[N - 3:])          # beyond iend: untouched (still the scaled input is NOT required)


@pytest.mark.parametrize("m", [2, 3, 6, 9])
def test_products_and_solves(libs, m):
    ref, ora = libs
    rng = np.random.default_rng(m)
    A, B = rand_mats(rng, m, True), rand_mats(rng, m)
    S = np.asfortranarray(rng.uniform(0, 1, (m, m)))
    v = np.asfortranarray(rng.uniform(0, 1, (N, m)))
    iend = N - 5
    for pattern in ((0, 1) if m % 3 == 0 else (0,)):
        a, b = F(N, m, m), F(N, m, m)
        ref.ref_mat_x_mat(N, iend, m, p(A), p(B), pattern, p(a))
        ora.om_mat_x_mat(N, iend, m, p(A), p(B), pattern, p(b))
        close(b[:iend], a[:iend])
    for name in ("identity_minus_mat_x_mat", "solve_mat"):
        a, b = F(N, m, m), F(N, m, m)
        getattr(ref, "ref_" + name)(N, iend, m, p(A), p(B), p(a))
        getattr(ora, "om_" + name)(N, iend, m, p(A), p(B), p(b))
        close(b[:iend], a[:iend])
    a, b = F(N, m, m), F(N, m, m)
    ref.ref_singlemat_x_mat(N, iend, m, p(S), p(B), p(a)); ora.om_singlemat_x_mat(N, iend, m, p(S), p(B), p(b))
    close(b[:iend], a[:iend])
    a, b = F(N, m, m), F(N, m, m)
    ref.ref_mat_x_singlemat(N, iend, m, p(A), p(S), p(a)); ora.om_mat_x_singlemat(N, iend, m, p(A), p(S), p(b))
    close(b[:iend], a[:iend])
    for top_left in (0, 1):
        a, b = F(N, m), F(N, m)
        ref.ref_mat_x_vec(N, iend, m, p(A), p(v), top_left, p(a)); ora.om_mat_x_vec(N, iend, m, p(A), p(v), top_left, p(b))
        close(b[:iend], a[:iend])
    a, b = F(N, m), F(N, m)
    ref.ref_singlemat_x_vec(N, iend, m, p(S), p(v), p(a)); ora.om_singlemat_x_vec(N, iend, m, p(S), p(v), p(b))
    close(b[:iend], a[:iend])
    a, b = F(N, m), F(N, m)
    ref.ref_solve_vec(N, iend, m, p(A), p(v), p(a)); ora.om_solve_vec(N, iend, m, p(A), p(v), p(b))
    close(b[:iend], a[:iend])


def test_fast_expm_exchange(libs):
    ref, ora = libs
    rng = np.random.default_rng(9)
    r = [10.0 ** rng.uniform(-12, 1.5, N) for _ in range(4)]
    for k in range(4):
        r[k][k::7] = 0.0                  # the securities for vanishing rates
    r[0][:3] = r[1][:3] = r[2][:3] = r[3][:3] = 0.0
    a, b = F(N, 2, 2), F(N, 2, 2)
    ref.ref_fast_expm_exchange_2(N, N, p(r[0]), p(r[1]), p(a)); ora.om_fast_expm_exchange_2(N, N, p(r[0]), p(r[1]), p(b))
    close(b, a)
    assert np.allclose(b.sum(axis=1), 1.0, atol=1e-12)          # columns of a conservative exchange sum to one
    a, b = F(N, 3, 3), F(N, 3, 3)
    ref.ref_fast_expm_exchange_3(N, N, *[p(x) for x in r], p(a)); ora.om_fast_expm_exchange_3(N, N, *[p(x) for x in r], p(b))
    close(b, a, 1e-9)         # the diagonalisation divides by eigenvalue differences: conditioning, not arithmetic
    assert np.array_equal(b[:3], a[:3]) or np.allclose(b[:3], np.eye(3)[None], atol=1e-9)


def test_expm_is_the_matrix_exponential(libs):
    """known answer: against scipy.linalg.expm to the single precision the Pade-7 routine promises"""
    from scipy.linalg import expm as scipy_expm
    _, ora = libs
    G = gamma_matrix(np.random.default_rng(11), 3, True)
    b = G.copy(order="F")
    ora.om_expm(N, N, 9, p(b), 1)
    for i in range(0, N, 16):
        want = scipy_expm(G[i])
        assert np.abs(b[i] - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), i
