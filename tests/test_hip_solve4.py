"""The FMA build's 4x4 device routines (storm_amd/csrc/mat4.hpp: fast_recip, m4_inverse_fast, lu4_factor / lu4_solve /
m4_solve_n) against numpy, through tools/solve4_check.bin (built by storm_amd/csrc/Makefile; one lane per matrix).

They replace, in the FMA build only, the reference's try_inverse().unwrap() * rhs of the collocation stage systems
(src/stepper.rs:180-186,235-262,300-346) -- same mathematics, different association -- so the bar is the fast build's
tolerance, not bits: relative to cond(A) * eps.  Both branches of m4_solve_n are driven on purpose: diagonally dominant
matrices (the unpivoted LU is accepted) and general ones (a multiplier > 1 sends the lane to the adjugate inverse)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "solve4_check.bin")
pytestmark = pytest.mark.gpu


def run_check(a, b, tmp_path):
    """a, b: (n, 4, 4) row-index-first numpy arrays -> (x, inv_scaled, accepted, ok, recip)."""
    if not os.path.exists(BIN):
        pytest.fail("tools/solve4_check.bin is missing: run __graft_entry__.build()")
    n = a.shape[0]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.int64(n).tobytes())
        rec = np.empty((n, 32))
        rec[:, :16] = a.transpose(0, 2, 1).reshape(n, 16)   # column-major: element (r, c) at c*4 + r
        rec[:, 16:] = b.transpose(0, 2, 1).reshape(n, 16)
        f.write(rec.tobytes())
    subprocess.run([BIN, str(fin), str(fout)], check=True, timeout=120)
    out = np.fromfile(fout).reshape(n, 35)
    x = out[:, :16].reshape(n, 4, 4).transpose(0, 2, 1)
    inv = out[:, 16:32].reshape(n, 4, 4).transpose(0, 2, 1)
    return x, inv, out[:, 32] == 1.0, out[:, 33] == 1.0, out[:, 34]


def rel_err(got, want):
    return np.max(np.abs(got - want), axis=(1, 2)) / np.max(np.abs(want), axis=(1, 2))


def test_dominant_matrices_take_the_lu(tmp_path):
    rng = np.random.default_rng(11)
    n = 4096
    # the shapes of the stage matrices: 12 I + X, S - 4 I, I - c X with small X
    a = rng.uniform(-1.0, 1.0, (n, 4, 4))
    a[: n // 2] += 12.0 * np.eye(4)
    a[n // 2:] -= 4.0 * np.eye(4)
    b = rng.uniform(-3.0, 3.0, (n, 4, 4))
    x, inv, accepted, ok, recip = run_check(a, b, tmp_path)
    assert accepted.all() and ok.all()
    want = np.linalg.solve(a, b)
    cond = np.linalg.cond(a)
    assert np.all(rel_err(x, want) <= 32 * cond * 2.3e-16)
    assert np.all(rel_err(inv, -2.0 * np.linalg.inv(a)) <= 32 * cond * 2.3e-16)
    assert np.all(np.abs(recip * a[:, 0, 0] - 1.0) <= 4.5e-16)


def test_general_matrices_fall_back_to_the_adjugate(tmp_path):
    rng = np.random.default_rng(12)
    n = 4096
    a = rng.uniform(-1.0, 1.0, (n, 4, 4)) * 10.0 ** rng.uniform(-3, 3, (n, 1, 1))
    b = rng.uniform(-3.0, 3.0, (n, 4, 4))
    cond = np.linalg.cond(a)
    keep = cond < 1e6
    a, b, cond = a[keep], b[keep], cond[keep]
    x, inv, accepted, ok, _ = run_check(a, b, tmp_path)
    assert ok.all()
    assert (~accepted).sum() > a.shape[0] // 2            # most random matrices need row exchanges
    want = np.linalg.solve(a, b)
    # the adjugate inverse is not backward stable: its error scales with cond(A)
    assert np.all(rel_err(x, want)[~accepted] <= 64 * cond[~accepted] * 2.3e-16)
    # where the LU was accepted every multiplier is <= 1: partial-pivoting quality
    assert np.all(rel_err(x, want)[accepted] <= 64 * cond[accepted] * 2.3e-16)


def test_multiplier_check_is_the_partial_pivoting_condition(tmp_path):
    # one entry below the diagonal just above / just below the pivot in magnitude
    base = np.eye(4) * 2.0
    cases = []
    for r, c in ((1, 0), (2, 0), (3, 0), (2, 1), (3, 1), (3, 2)):
        for v in (1.999, -1.999, 2.001, -2.001):
            m = base.copy()
            m[r, c] = v
            cases.append(m)
    a = np.array(cases)
    b = np.tile(np.eye(4), (a.shape[0], 1, 1))
    x, _, accepted, ok, _ = run_check(a, b, tmp_path)
    want_accept = np.array([abs(v) < 2.0 for _ in range(6) for v in (1.999, -1.999, 2.001, -2.001)])
    assert (accepted == want_accept).all() and ok.all()
    assert np.all(rel_err(x, np.linalg.inv(a)) <= 1e-14)


def test_singular_and_nan_matrices_are_flagged(tmp_path):
    a = np.zeros((4, 4, 4))
    a[0] = np.eye(4); a[0][3, 3] = 0.0                     # zero last pivot
    a[1] = np.ones((4, 4))                                 # rank 1
    a[2] = np.eye(4); a[2][1, 2] = np.nan                  # NaN entry
    a[3] = np.eye(4) * 3.0                                 # regular (control)
    b = np.tile(np.eye(4), (4, 1, 1))
    x, _, accepted, ok, _ = run_check(a, b, tmp_path)
    assert list(accepted) == [False, False, False, True]
    assert list(ok[:2]) == [False, False] and ok[3]        # det == 0 -> ok false (the kernels then return NaN + the flag)
    assert not np.isfinite(x[2]).all()                     # NaN propagates (NaN != 0, so `ok` stays as for the reference's inverse)
    assert np.allclose(x[3], np.eye(4) / 3.0, rtol=1e-15)
