"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the comparator model models/deepvog_pytorch.py (registered as 'deepvog' in
modelSummary.py:26), evaluation and training mode.  Functional PyTorch on a state dict, pinned by tests/golden/deepvog_b2.npz (produced by
importing the reference itself, tests/golden/make_golden.py target "deepvog").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
import torch
import torch.nn.functional as F

from . import losses
from .esfnet import _bn, _conv


def _cbr(sd, p, k, x, training=False, update=None, **kw):
    return F.relu(_bn(sd, "%s.bn%d" % (p, k), _conv(sd, "%s.conv%d" % (p, k), x, **kw), training, update=update))


def deepvog_loss(op, target, pupil_center, cond):
    """models/deepvog_pytorch.py:148-167 get_allLoss.  Returns (loss, pred_c [B,2], terms)."""
    B, _, H, W = op.shape
    ok = 1 - cond[:, 1]
    t = (target == 2).long()
    l_pt, pred_c = losses.seg2pt(op[:, 1], losses.norm_pts(pupil_center, H, W), 4)
    l_seg = 10 * F.cross_entropy(torch.softmax(op, dim=1), t, reduction="none")          # the reference feeds probabilities
    l_seg = (l_seg.reshape(B, -1).mean(1) * ok).sum() / ok.sum() if ok.sum() else torch.zeros(())
    return l_seg + l_pt.mean(), pred_c, dict(l_seg=l_seg, l_pt=l_pt.mean())


def deepvog_forward(sd, x, target, pupil_center, cond, training=False, update=None):
    """models/deepvog_pytorch.py:115-146.  ``training``: BatchNorm with batch statistics (``update`` receives the new running
    statistics).  Returns (out [B,2,H,W], pred_c [B,2], loss [1], terms)."""
    h = torch.cat([x, x, x], 1)
    jumps = []
    for i in range(1, 5):
        p = "down_block%d" % i
        j = _cbr(sd, p, 1, h, training, update, padding=1)
        jumps.append(j)
        h = _cbr(sd, p, 2, j, training, update, stride=2)
    for i in range(1, 6):
        p = "up_block%d" % i
        if i > 1:
            h = torch.cat((h, jumps[5 - i]), 1)
        h = _cbr(sd, p, 1, h, training, update, padding=1)
        if i < 5:
            h = _cbr(sd, p, 2, F.interpolate(h, scale_factor=2, mode="nearest"), training, update, padding=1)
    out = _conv(sd, "conv1", h)
    loss, pred_c, terms = deepvog_loss(out, target, pupil_center, cond)
    return out, pred_c, loss.reshape(1), terms
