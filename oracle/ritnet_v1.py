"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the comparator model models/RITnet_v1.py (the EllSeg RITnet the reference keeps
next to ESF-Net; registered as 'ritnet_v1' in modelSummary.py:18-26).  Functional PyTorch on a state dict, pinned by
tests/golden/ritnet_v1_b2.npz (produced by importing the reference itself, tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
import torch
import torch.nn.functional as F

from . import losses
from .esfnet import _bn, _conv, regression


def down_block(sd, p, x, pool, training, update=None):
    """models/RITnet_v1.py:38-72 (dropout=False): AvgPool2d first, three LeakyReLU convolutions over growing concatenations,
    BatchNorm2d on the result."""
    if pool:
        x = F.avg_pool2d(x, 2)
    x1 = F.leaky_relu(_conv(sd, p + ".conv1", x, padding=1))
    x21 = torch.cat((x, x1), 1)
    x22 = F.leaky_relu(_conv(sd, p + ".conv22", _conv(sd, p + ".conv21", x21), padding=1))
    x31 = torch.cat((x21, x22), 1)
    out = F.leaky_relu(_conv(sd, p + ".conv32", _conv(sd, p + ".conv31", x31), padding=1))
    return _bn(sd, p + ".bn", out, training, update=update)


def up_block(sd, p, skip, x):
    """models/RITnet_v1.py:74-99: nearest x2, concatenate (x first), two 1x1 -> 3x3 pairs with LeakyReLU."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = torch.cat((x, skip), 1)
    x1 = F.leaky_relu(_conv(sd, p + ".conv12", _conv(sd, p + ".conv11", x), padding=1))
    x21 = torch.cat((x, x1), 1)
    return F.leaky_relu(_conv(sd, p + ".conv22", _conv(sd, p + ".conv21", x21), padding=1))


def ritnet_v1_forward(sd, x, x_edge, target, pupil_center, elNorm, spatWts, distMap, cond, ID, alpha, training=False,
                      disentangle=False, update=None):
    """models/RITnet_v1.py:244-309.  Returns (op, elPred, latent, loss[1], elOut, terms)."""
    xs = []
    h = x
    for i in range(1, 6):
        h = down_block(sd, "enc.down_block%d" % i, h, i > 1, training, update)
        xs.append(h)
    x1, x2, x3, x4, x5 = xs
    latent = x5.flatten(2).mean(-1)
    elOut = regression(sd, x5)
    h = x5
    for k, sk in zip((4, 3, 2, 1), (x4, x3, x2, x1)):
        h = up_block(sd, "dec.up_block%d" % k, sk, h)
    op = _conv(sd, "dec.final", h)
    total, pred_c, terms = losses.all_loss(op, elOut, target, pupil_center, elNorm, spatWts, distMap, cond, alpha)
    elPred = torch.cat([pred_c[:, 0, :], elOut[:, 2:5], pred_c[:, 1, :], elOut[:, 7:10]], dim=1)
    if disentangle:
        pd = F.linear(F.linear(latent, sd["dsIdentify_lin.layersLin.0.weight"], sd["dsIdentify_lin.layersLin.0.bias"]),
                      sd["dsIdentify_lin.layersLin.1.weight"], sd["dsIdentify_lin.layersLin.1.bias"])
        cl = losses.conf_loss(pd, ID.to(torch.long), True)
        terms["conf"] = cl
        total = total + 2 * cl
    if not torch.is_tensor(total):
        total = torch.tensor(float(total))
    return op, elPred, latent, total.reshape(1), elOut, terms
