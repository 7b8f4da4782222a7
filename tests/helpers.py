"""Test-side helpers (numpy).  count_windings / radial order follow the reference's
post-processing (src/postprocessing.rs:244-279,304-355) and are used only to pin the
eigenvector path against the reference's mode-id known answers."""
import numpy as np


def count_windings(y1, y2, x):
    cw = ccw = 0
    for i in range(len(y1) - 1):
        a1, a2 = y1[i], y1[i + 1]
        b1, b2 = y2[i], y2[i + 1]
        if a1 <= 0.0 and a2 > 0.0:
            yt = b1 - a1 * (b2 - b1) / (a2 - a1)
            if yt > 0.0:
                cw += 1
            else:
                ccw += 1
        elif a1 >= 0.0 and a2 < 0.0:
            yt = b1 - a1 * (b2 - b1) / (a2 - a1)
            if yt > 0.0:
                ccw += 1
            else:
                cw += 1
    return cw, ccw


def mode_id(eigenvector, ell, u):
    """(cw, ccw, radial_order) from a raw eigenvector [(n_pts)*4]; `u` = model.u on the grid."""
    e = np.asarray(eigenvector).reshape(-1, 4)
    y1, y2, y3, y4 = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
    x = np.arange(len(y1), dtype=float)
    if ell == 0:
        cw, ccw = count_windings(y1, y2, x)
        return cw, ccw, ccw
    if ell == 1:
        y1a = (1.0 - u / 3.0) * y1 + (y3 - y4) / 3.0
        y2a = y2 - y1
        cw, ccw = count_windings(y1a[2:], y2a[2:], x[2:])
        return (cw, ccw, ccw - cw) if cw > ccw else (cw, ccw, ccw - cw + 1)
    cw, ccw = count_windings(y1, y2 + y3, x)
    return cw, ccw, ccw - cw


def ndarray_linspace(a, b, n):
    """ndarray::linspace as used by the reference mode-id tests: start + step*i."""
    step = (b - a) / (n - 1)
    return np.array([a + step * i for i in range(n)], dtype=np.float64)
