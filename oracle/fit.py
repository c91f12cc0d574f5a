"""Oracle for the ellipse-fit stage of evaluate.py (test infrastructure; see oracle/__init__.py).

numpy restatement of utils.py:450-486 (search_proper_parameter_iou_for_our_data),
utils.py:176-204 (calc_ell_iou) and the conic algebra of helperfunctions.py:13-63,102-129
(my_ellipse.param2mat / transform / mat2param / recover_theta / recover_C).

Numerics that matter for bit-identical results ("Hard parts"):
  * the conic normalisation runs in float64 (numpy), its 5 outputs are then used as
    scalars against the float32 mesh, i.e. rounded to float32 first (torch scalar semantics);
  * every mesh operation is a separately rounded float32 op (no FMA contraction);
  * degrees <-> radians use 3.14159, not pi;
  * the score is float32(inter) / float32(nseg + nell - inter) (integer counts are exact in f32).
"""
import math

import numpy as np

EPS = 1e-40  # helperfunctions.py:10
PI_REF = 3.14159


def _rot(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1]])


def _trans(cx, cy):
    return np.array([[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1]])


def param2mat(param):
    """helperfunctions.py:25-33."""
    cx, cy, a, b, theta = tuple(param)
    Hr, Ht = _rot(-theta), _trans(-cx, -cy)
    Q = np.array([[1 / a ** 2, 0, 0], [0, 1 / b ** 2, 0], [0, 0, -1]])
    return Ht.T @ Hr.T @ Q @ Hr @ Ht


def mat2param(mat):
    """helperfunctions.py:50-63 with recover_theta (:102-116) and recover_C (:118-122)."""
    a, b, c, d, e = mat[0, 0], 2 * mat[0, 1], mat[1, 1], 2 * mat[0, 2], 2 * mat[1, 2]
    if abs(b) <= EPS and a <= c:
        theta = 0.0
    elif abs(b) <= EPS and a > c:
        theta = np.pi / 2
    else:
        theta = 0.5 * np.arctan2(b, (a - c))
    tx = (2 * c * d - b * e) / (b ** 2 - 4 * a * c)
    ty = (2 * a * e - b * d) / (b ** 2 - 4 * a * c)
    Hr, Ht = _rot(theta), _trans(tx, ty)
    mn = Hr.T @ Ht.T @ mat @ Ht @ Hr
    return np.array([tx, ty, np.sqrt(1 / mn[0, 0]), np.sqrt(1 / mn[1, 1]), theta])


def transform(param, H):
    """helperfunctions.py:124-129: ellipse params after the homography H (returns 5 params)."""
    Hi = np.linalg.inv(H)
    return mat2param(np.linalg.inv(H.T) @ param2mat(param) @ Hi)


def mesh_f32(H, W):
    """utils.py:27-60: float32 linspace(-1,1) grids."""
    import torch
    xs = torch.linspace(-1, 1, W).numpy()
    ys = torch.linspace(-1, 1, H).numpy()
    return np.broadcast_to(xs[None, :], (H, W)), np.broadcast_to(ys[:, None], (H, W))


def ell_iou(seg, el_px_deg, mesh):
    """utils.py:176-204 calc_ell_iou(seg, el, mesh, nor=False, angle_nor=True).
    seg: bool [H,W]; el_px_deg: (cx, cy, a, b, angle in degrees) in pixels."""
    Hh, Ww = seg.shape
    el = np.array(el_px_deg, dtype=np.float64)
    el[4] = el[4] / 180. * PI_REF
    Hm = np.array([[2 / Ww, 0, -1], [0, 2 / Hh, -1], [0, 0, 1]])
    el = transform(el, Hm)
    mx, my = mesh
    f = np.float32
    cx, cy, a, b = f(el[0]), f(el[1]), f(el[2]), f(el[3])
    ct, st = f(np.cos(el[4])), f(np.sin(el[4]))
    dx, dy = mx - cx, my - cy
    X = dx * ct + dy * st
    Y = (-dx) * st + dy * ct
    u, v = X / a, Y / b
    wt = u * u + v * v - f(1)
    ell = wt <= 0
    inter = int(np.count_nonzero(ell & seg))
    nseg = int(np.count_nonzero(seg))
    nell = int(np.count_nonzero(ell))
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(f(inter) / f(f(f(nseg) + f(nell)) - f(inter)))


def fit_ellipse(seg, ell_para, max_sweeps=40, count_evals=False):
    """utils.py:450-486: coordinate hill-climb on (a, b, angle_deg); centre fixed.
    ell_para = (cx, cy, a, b, theta_rad) in pixels.  Returns (cx, cy, a, b, theta_rad)."""
    Hh, Ww = seg.shape
    mesh = mesh_f32(Hh, Ww)
    center = [ell_para[0], ell_para[1]]
    now = [ell_para[2], ell_para[3], ell_para[4] * 180. / PI_REF]
    rt = ell_iou(seg, center + now, mesh)
    d = [1., 1., 1.]
    n_eval = 1
    for _ in range(max_sweeps):
        flag = False
        for j in range(3):
            now[j] -= d[j]
            n_eval += 1
            if ell_iou(seg, center + now, mesh) > rt:
                flag = True
                continue
            now[j] += 2. * d[j]
            n_eval += 1
            if ell_iou(seg, center + now, mesh) > rt:
                flag = True
                continue
            now[j] -= d[j]
            d[j] *= 0.8
        n_eval += 1
        score = ell_iou(seg, center + now, mesh)
        if score > rt:
            rt = score
        if not flag:
            break
    out = np.array(center + now, dtype=np.float64)
    out[4] = out[4] / 180.0 * PI_REF
    return (out, n_eval) if count_evals else out
