"""Host-side mirror of the reference's ``utils.py`` for the hot path.

Same names, argument meaning and error behaviour as the reference functions the entry scripts
call (file:line cited per function).  Compute runs on the HIP path; the metric helpers stay on
the host with numpy exactly as in the reference (they define "mIoU").
"""
import copy
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import require_cuda


# ---------------------------------------------------------------------------------------------
# parameter containers (same state_dict keys as the reference modules)
# ---------------------------------------------------------------------------------------------
class convBlock(nn.Module):
    """utils.py:1039-1050: conv3x3 -> leaky -> conv3x3 -> leaky -> BatchNorm2d (parameters only;
    executed by esf_engine)."""

    def __init__(self, in_c, inter_c, out_c, actfunc=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_c, inter_c, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(inter_c, out_c, kernel_size=3, padding=1)
        self.bn = nn.BatchNorm2d(num_features=out_c)


class regressionModule(nn.Module):
    """utils.py:983-1011 (parameters only)."""

    def __init__(self, feature_channels):
        super().__init__()
        inC = feature_channels if isinstance(feature_channels, int) else int(feature_channels["enc"]["op"][-1])
        self.c1 = nn.Conv2d(inC, 128, kernel_size=(2, 3), bias=True)
        self.c2 = nn.Conv2d(128, 128, kernel_size=3, bias=True)
        self.c3 = nn.Conv2d(128, 32, kernel_size=3, bias=False)
        self.l1 = nn.Linear(32 * 3 * 5, 256, bias=True)
        self.l2 = nn.Linear(256, 10, bias=True)


class linStack(nn.Module):
    """utils.py:953-981 (parameters only; used for the dataset-identity head)."""

    def __init__(self, num_layers, in_dim, hidden_dim, out_dim, bias, actBool, dp):
        super().__init__()
        self.layersLin = nn.ModuleList([
            nn.Linear(hidden_dim if i > 0 else in_dim, hidden_dim if i < num_layers - 1 else out_dim, bias=bias)
            for i in range(num_layers)])
        self.actBool = actBool
        self.dp_p = dp


class LinearBlock(nn.Module):
    """utils.py:1051-1090 with norm='none' (the only form the path uses)."""

    def __init__(self, input_dim, output_dim, norm="none", activation="relu"):
        super().__init__()
        if norm not in ("none", "sn"):
            raise AssertionError("Unsupported normalization: {}".format(norm))
        self.fc = nn.Linear(input_dim, output_dim, bias=True)
        self.activation_name = activation


class Conv2dBlock(nn.Module):
    """utils.py:1092-1149 with norm='none' (StyleEncoder's form)."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, norm="none", activation="relu",
                 pad_type="zero"):
        super().__init__()
        if norm not in ("none", "sn"):
            raise AssertionError("Unsupported normalization: {}".format(norm))
        if pad_type not in ("reflect", "zero"):
            raise AssertionError("Unsupported padding type: {}".format(pad_type))
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, bias=True)
        self.padding, self.pad_type, self.activation_name = padding, pad_type, activation


# ---------------------------------------------------------------------------------------------
# small host helpers
# ---------------------------------------------------------------------------------------------
def create_meshgrid(height, width, normalized_coordinates=True):
    """utils.py:27-60: [1,H,W,2] grid, x = linspace(-1,1,W), y = linspace(-1,1,H)."""
    if normalized_coordinates:
        xs, ys = torch.linspace(-1, 1, width), torch.linspace(-1, 1, height)
    else:
        xs, ys = torch.linspace(0, width - 1, width), torch.linspace(0, height - 1, height)
    return torch.stack([xs[None, :].expand(height, width), ys[:, None].expand(height, width)], dim=-1)[None]


def get_nparams(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def normPts(pts, sz):
    """utils.py:627-634."""
    o = copy.deepcopy(pts)
    shp = o.shape
    o = o.reshape(-1, 2)
    o[:, 0] = 2 * (o[:, 0] / sz[1]) - 1
    o[:, 1] = 2 * (o[:, 1] / sz[0]) - 1
    return o.reshape(shp)


def unnormPts(pts, sz):
    """utils.py:636-643."""
    o = copy.deepcopy(pts)
    shp = o.shape
    o = o.reshape(-1, 2)
    o[:, 0] = 0.5 * sz[1] * (o[:, 0] + 1)
    o[:, 1] = 0.5 * sz[0] * (o[:, 1] + 1)
    return o.reshape(shp)


def calc_edge(args, img, edge_model, device):
    """utils.py:645-656: grey -> 3 channels -> frozen BDCN -> fused map; optional >=0.1 -> 1
    (``args.edge_thres``), fused into the tail kernel here."""
    with torch.no_grad():
        x = torch.cat((img, img, img), dim=1).to(device).to(torch.float32)
        e = edge_model.forward_fuse(x, edge_thres=1 if getattr(args, "edge_thres", 0) == 1 else 0)
    return e.to(getattr(args, "prec", torch.float32))


def get_predictions(output):
    """utils.py:65-81: argmax over channels -> [B,H,W] int64 on the host (first max wins ties)."""
    bs, c, h, w = output.size()
    _, idx = output.cpu().max(1)
    return idx.view(bs, h, w)


def _jaccard(y_true, y_pred, labels):
    out = []
    for l in labels:
        t, p = y_true == l, y_pred == l
        union = np.count_nonzero(t | p)
        out.append(np.count_nonzero(t & p) / union if union else 0.0)
    return np.array(out)


def getSeg_metrics(y_true, y_pred, cond):
    """utils.py:120-150: per-sample Jaccard over the classes present in the ground truth, nan-mean."""
    assert y_pred.ndim == 3, "Incorrect number of dimensions"
    assert y_true.ndim == 3, "Incorrect number of dimensions"
    cond = cond.astype(bool)
    rows = []
    for i in range(y_true.shape[0]):
        present = np.unique(y_true[i])
        vals = np.full((3,), np.nan)
        if not cond[i]:
            sc = _jaccard(y_true[i].reshape(-1), y_pred[i].reshape(-1), present)
            for j, v in enumerate(present):
                vals[v] = sc[j]
        rows.append(vals)
    rows = np.stack(rows, axis=0)
    clean = rows[~cond, :]
    if len(clean) == 0:
        return np.nan, np.nan * np.ones(3), rows
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            per = np.nanmean(clean, axis=0)
            mean = np.nanmean(per)
    return mean, per, rows


def getPoint_metric(y_true, y_pred, cond, sz, do_unnorm):
    """utils.py:152-162: mean euclidean distance over valid samples."""
    if do_unnorm:
        y_pred = unnormPts(y_pred, sz)
    flag = (~cond.astype(bool)).astype(float)
    dist = flag * np.sqrt(((np.asarray(y_true) - np.asarray(y_pred)) ** 2).sum(axis=1))
    return (np.sum(dist) / np.sum(flag) if np.any(flag) else np.nan, dist)


def getAng_metric(y_true, y_pred, cond):
    """utils.py:164-170."""
    flag = (~cond.astype(bool)).astype(float)
    dist = np.rad2deg(flag * np.abs(y_true - y_pred))
    return (np.sum(dist) / np.sum(flag) if np.any(flag) else np.nan, dist)


# ---------------------------------------------------------------------------------------------
# ellipse fit (evaluate.py's "fit" stage) on the device
# ---------------------------------------------------------------------------------------------
_mesh_cache = {}


def _mesh_axes(H, W, dev):
    key = (H, W, str(dev))
    if key not in _mesh_cache:
        _mesh_cache[key] = (torch.linspace(-1, 1, W).to(dev), torch.linspace(-1, 1, H).to(dev))
    return _mesh_cache[key]


def fit_ellipses(mask, frame_of, cls, init, return_evals=False):
    """Batched device form of search_proper_parameter_iou_for_our_data (utils.py:450-486).

    mask [F,H,W] int64 class maps on the GPU; frame_of / cls [n] which frame / class each fit
    uses; init [n,5] (cx,cy,a,b,theta) pixels.  Returns [n,5] float64 (cx,cy,a,b,theta)."""
    require_cuda(mask, "mask")
    L = _lib.lib()
    dev = mask.device
    F, H, W = mask.shape
    n = len(frame_of)
    fo_h = np.asarray(frame_of, dtype=np.int32)
    if n and (fo_h.min() < 0 or fo_h.max() >= F):
        raise ValueError("fit_ellipses: frame index %d outside the %d class maps given" % (int(fo_h.max()), F))
    fo = torch.as_tensor(fo_h).to(dev)
    cl = torch.as_tensor(np.asarray(cls, dtype=np.int32)).to(dev)
    ini = torch.as_tensor(np.asarray(init, dtype=np.float64).reshape(n, 5)).to(dev)
    out = torch.empty((n, 5), dtype=torch.float64, device=dev)
    ev = torch.zeros((n,), dtype=torch.int32, device=dev)
    xs, ys = _mesh_axes(H, W, dev)
    m = mask.contiguous()
    _lib.check(L.egne_ellipse_fit(m.data_ptr(), F, fo.data_ptr(), cl.data_ptr(), n, H, W, xs.data_ptr(), ys.data_ptr(),
                                  ini.data_ptr(), out.data_ptr(), ev.data_ptr(), _lib.stream_ptr()), "ellipse_fit")
    return (out.cpu().numpy(), ev.cpu().numpy()) if return_evals else out.cpu().numpy()


def ellipse_seeds_from_pred(elPred, H, W):
    """evaluate.py:135-151 on the device: elPred [F,10] float32 (normalised ellipses of the regression head)
    -> (init [2F,5] float64 pixel ellipses, frame_of [2F] int32, cls [2F] int32), all on the GPU.
    Fit 2f is the iris (class 1) of frame f, fit 2f+1 its pupil (class 2)."""
    require_cuda(elPred, "elPred")
    L = _lib.lib()
    F = elPred.shape[0]
    ep = elPred.detach().to(torch.float32).contiguous()
    init = torch.empty((2 * F, 5), dtype=torch.float64, device=ep.device)
    fo = torch.empty((2 * F,), dtype=torch.int32, device=ep.device)
    cl = torch.empty((2 * F,), dtype=torch.int32, device=ep.device)
    _lib.check(L.egne_ellipse_init_from_pred(ep.data_ptr(), F, H, W, init.data_ptr(), fo.data_ptr(), cl.data_ptr(),
                                             _lib.stream_ptr()), "ellipse_init_from_pred")
    return init, fo, cl


def fit_ellipses_from_pred(mask, elPred, out=None):
    """The fit stage of evaluate.py:135-166 with no host hop: seeds from ``elPred`` on the device, both searches of
    every frame in one launch.  Returns a DEVICE tensor [F,2,5] float64 (iris, pupil) x (cx,cy,a,b,theta); nothing
    here synchronises -- the caller decides when to read it."""
    require_cuda(mask, "mask")
    L = _lib.lib()
    F, H, W = mask.shape
    if elPred.shape[0] != F:
        raise ValueError("fit_ellipses_from_pred: %d ellipse rows for %d class maps" % (elPred.shape[0], F))
    init, fo, cl = ellipse_seeds_from_pred(elPred, H, W)
    if out is None:
        out = torch.empty((F, 2, 5), dtype=torch.float64, device=mask.device)
    xs, ys = _mesh_axes(H, W, mask.device)
    m = mask.contiguous()
    _lib.check(L.egne_ellipse_fit(m.data_ptr(), F, fo.data_ptr(), cl.data_ptr(), 2 * F, H, W, xs.data_ptr(), ys.data_ptr(),
                                  init.data_ptr(), out.data_ptr(), None, _lib.stream_ptr()), "ellipse_fit")
    return out


def search_proper_parameter_iou_for_our_data(seg, ell_para):
    """utils.py:450-486, same signature: seg [H,W] bool tensor on the GPU, ell_para (5,) pixels."""
    require_cuda(seg, "seg")
    res = fit_ellipses(seg.to(torch.int64)[None], [0], [1], np.asarray(ell_para, dtype=np.float64)[None, :5])
    return res[0]
