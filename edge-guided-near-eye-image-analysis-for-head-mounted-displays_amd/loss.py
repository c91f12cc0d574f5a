"""Loss entry points of the reference's loss.py / RITnet_v2.get_allLoss on the device loss head.

The per-sample Python loops and ``.cpu()`` syncs of loss.py:48-137 are one fused kernel pair
(``egne_loss_fwd``); this module exposes it with the reference's call shape for callers that hold NCHW
logits (the model itself feeds the kernel from its NHWC buffer without this detour)."""
import ctypes as C

import torch

from . import _lib
from .engine import require_cuda


def get_allLoss(op, elOut, target, pupil_center, elNorm, spatWts, distMap, cond, ID, alpha):
    """models/RITnet_v2.py:372-432.  Returns (total_loss[0-d], pred_c_seg [B,2,2]); the individual terms
    (l_seg2pt, l_seg, l_pt, l_ellipse) are in ``get_allLoss.last_terms`` afterwards."""
    require_cuda(op, "op")
    L = _lib.lib()
    B, Cc, H, W = op.shape
    assert Cc == 3
    dev = op.device
    st = _lib.stream_ptr()
    nhwc = torch.zeros(B, H, W, 8, device=dev)
    _lib.check(L.egne_nchw_to_nhwc(op.contiguous().data_ptr(), B, 3, H, W, nhwc.data_ptr(), 8, 0, 8, st), "nchw_to_nhwc")
    f32 = lambda t: t.to(dev, torch.float32).contiguous()  # noqa: E731
    tg, sw, dm, cd = target.to(dev, torch.int64).contiguous(), f32(spatWts), f32(distMap), f32(cond)
    pc, en, eo = f32(pupil_center), f32(elNorm), f32(elOut)
    part = torch.zeros(int(L.egne_loss_workspace_floats(B, H, W)), device=dev)
    terms, pred_c, elp = torch.zeros(8, device=dev), torch.zeros(B, 2, 2, device=dev), torch.zeros(B, 10, device=dev)
    gx, gy = torch.linspace(-1, 1, W).to(dev), torch.linspace(-1, 1, H).to(dev)
    d = _lib.LossDesc()
    d.B, d.H, d.W = B, H, W
    d.logits, d.pix_stride, d.ch_off = nhwc.data_ptr(), 8, 0
    d.target, d.spatWts, d.distMap, d.cond = tg.data_ptr(), sw.data_ptr(), dm.data_ptr(), cd.data_ptr()
    d.pupil_center, d.elNorm, d.elOut, d.alpha = pc.data_ptr(), en.data_ptr(), eo.data_ptr(), float(alpha)
    d.grid_x, d.grid_y = gx.data_ptr(), gy.data_ptr()
    d.partials, d.out_terms, d.pred_c, d.elPred = part.data_ptr(), terms.data_ptr(), pred_c.data_ptr(), elp.data_ptr()
    _lib.check(L.egne_loss_fwd(C.byref(d), st), "loss_fwd")
    get_allLoss.last_terms = dict(l_seg2pt=terms[1], l_seg=terms[2], l_pt=terms[3], l_ellipse=terms[4])
    return terms[0], pred_c


def conf_Loss(x, gt, flag):
    """loss.py:139-157 on the device: L1(softmax, uniform) when ``flag`` else cross-entropy."""
    require_cuda(x, "x")
    L = _lib.lib()
    Bn, Cc = x.shape
    xs = x.to(torch.float32).contiguous()
    terms = torch.zeros(8, device=x.device)
    g = gt.to(x.device, torch.int64).contiguous() if gt is not None else None
    _lib.check(L.egne_conf_loss(xs.data_ptr(), Cc, g.data_ptr() if g is not None else None, Bn, Cc, 1 if flag else 0, 1.0,
                                terms.data_ptr(), _lib.stream_ptr()), "conf_loss")
    return terms[7]
