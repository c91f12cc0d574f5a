"""Conic algebra of the reference's ``helperfunctions.my_ellipse`` needed on the path (host, float64):
param2mat (:25-33), transform (:124-129), mat2param with recover_theta / recover_C (:50-63,102-122)."""
import numpy as np

EPS = 1e-40


def _rot(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _trans(x, y):
    return np.array([[1.0, 0.0, x], [0.0, 1.0, y], [0.0, 0.0, 1.0]])


def param2mat(p):
    cx, cy, a, b, th = (float(v) for v in p)
    R, T = _rot(-th), _trans(-cx, -cy)
    Q = np.diag([1.0 / a ** 2, 1.0 / b ** 2, -1.0])
    return T.T @ R.T @ Q @ R @ T


def mat2param(m):
    a, b, c, d, e = m[0, 0], 2 * m[0, 1], m[1, 1], 2 * m[0, 2], 2 * m[1, 2]
    if abs(b) <= EPS:
        th = 0.0 if a <= c else np.pi / 2
    else:
        th = 0.5 * np.arctan2(b, a - c)
    den = b * b - 4 * a * c
    tx, ty = (2 * c * d - b * e) / den, (2 * a * e - b * d) / den
    R, T = _rot(th), _trans(tx, ty)
    n = R.T @ T.T @ m @ T @ R
    return np.array([tx, ty, np.sqrt(1.0 / n[0, 0]), np.sqrt(1.0 / n[1, 1]), th])


def transform(p, H):
    """Ellipse parameters after the homography H (my_ellipse(p).transform(H)[0][:-1])."""
    Hi = np.linalg.inv(H)
    return mat2param(np.linalg.inv(H.T) @ param2mat(p) @ Hi)
