"""Oracle for ESF-Net (``DenseNet2D``), both variants (test infrastructure; see oracle/__init__.py).

Functional form over a state_dict with the reference's key names.  ``variant='v2'`` follows
models/RITnet_v2.py, ``variant='concat'`` follows models/RITnet_concat.py.  Width is a
parameter (``chz``); only chz=32 has a reference to be checked against.
"""
import torch
import torch.nn.functional as F

from . import losses


def enc_sizes(chz, growth=1.2, blks=4):
    """models/RITnet_v2.py:15-29 getSizes."""
    inter = [chz * (i + 1) for i in range(blks)]
    op = [int(growth * chz * (i + 1)) for i in range(blks)]
    ip = [chz] + op[:-1]
    skip = [ip[::-1][i] + inter[::-1][i] for i in range(blks)]
    return dict(inter=inter, op=op, ip=ip, skip=skip)


def dec_sizes(chz, growth, add_edge, variant="v2"):
    """models/RITnet_v2.py:177-190 / RITnet_concat.py:164-169, generalised in chz
    (exact at chz=32)."""
    e = enc_sizes(chz, growth)
    fc = e["op"][-1]
    plain_ip = e["op"][::-1]
    plain_op = e["op"][::-1][1:] + [chz]
    if variant == "concat":
        return dict(ip=[2 * fc] + plain_op[:-1], op=plain_op, skip=[2 * s for s in e["skip"]])
    if add_edge:
        d = [int(round(chz * f)) for f in (5.625, 3.125, 1.9375)]
        return dict(ip=[2 * fc] + d, op=d + [chz], skip=e["skip"])
    return dict(ip=plain_ip, op=plain_op, skip=e["skip"])


def _conv(sd, p, x, **kw):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), **kw)


def _bn(sd, p, x, training, momentum=0.1, eps=1e-5, update=None):
    """BatchNorm2d; in training mode uses batch statistics (and, if ``update`` is a dict,
    records the new running stats as torch would: unbiased var, momentum 0.1)."""
    if not training:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], False, momentum, eps)
    rm = sd[p + ".running_mean"].clone()
    rv = sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, momentum, eps)
    if update is not None:
        update[p + ".running_mean"] = rm
        update[p + ".running_var"] = rv
    return y


def conv_block(sd, p, x, training, update=None):
    """utils.py:1039-1050 convBlock: conv3x3 -> leaky -> conv3x3 -> leaky -> BatchNorm2d."""
    x = F.leaky_relu(_conv(sd, p + ".conv1", x, padding=1))
    x = F.leaky_relu(_conv(sd, p + ".conv2", x, padding=1))
    return _bn(sd, p + ".bn", x, training, update=update)


def down_block(sd, p, x, pool):
    """models/RITnet_v2.py:46-66 (+ Transition_down :32-44).  Returns (skip, x_down)."""
    x1 = F.leaky_relu(_conv(sd, p + ".conv1", F.instance_norm(x), padding=1))
    x21 = torch.cat([x, x1], 1)
    x22 = F.leaky_relu(_conv(sd, p + ".conv22", _conv(sd, p + ".conv21", x21), padding=1))
    x31 = torch.cat([x21, x22], 1)
    out = F.leaky_relu(_conv(sd, p + ".conv32", _conv(sd, p + ".conv31", x31), padding=1))
    out = torch.cat([out, x], 1)
    t = _conv(sd, p + ".TD.conv", F.leaky_relu(F.instance_norm(out)))
    if pool:
        t = F.avg_pool2d(t, pool)
    return out, t


def encoder(sd, x, training, update=None, p="enc"):
    """models/RITnet_v2.py:167-174."""
    x = conv_block(sd, p + ".head", x, training, update)
    skips = []
    for i in (1, 2, 3, 4):
        s, x = down_block(sd, "%s.down_block%d" % (p, i), x, 2)
        skips.append(s)
    _, x = down_block(sd, p + ".bottleneck", x, 0)
    return skips[3], skips[2], skips[1], skips[0], x


def up_block(sd, p, skips, x):
    """models/RITnet_v2.py:79-88 (concat variant: RITnet_concat.py:79-88 takes two skips)."""
    x = F.interpolate(x, mode="bilinear", align_corners=False, scale_factor=2)
    x = torch.cat([x] + list(skips), 1)
    x1 = F.leaky_relu(_conv(sd, p + ".conv12", _conv(sd, p + ".conv11", x), padding=1))
    x21 = torch.cat([x, x1], 1)
    return F.leaky_relu(_conv(sd, p + ".conv22", _conv(sd, p + ".conv21", x21), padding=1))


def regression(sd, x, p="elReg"):
    """utils.py:1013-1037 regressionModule.forward."""
    B = x.shape[0]
    x = F.leaky_relu(_conv(sd, p + ".c1", x))
    x = F.avg_pool2d(x, 2)
    x = F.leaky_relu(_conv(sd, p + ".c2", x))
    x = F.leaky_relu(_conv(sd, p + ".c3", x))
    x = x.reshape(B, -1)
    x = F.linear(torch.selu(F.linear(x, sd[p + ".l1.weight"], sd[p + ".l1.bias"])),
                 sd[p + ".l2.weight"], sd[p + ".l2.bias"])
    return torch.cat([torch.tanh(x[:, 0:2]), torch.sigmoid(x[:, 2:4]), x[:, 4:5],
                      torch.tanh(x[:, 5:7]), torch.sigmoid(x[:, 7:9]), x[:, 9:10]], dim=1)


def style_encoder(sd, x, p="seg_encoder.model"):
    """models/RITnet_v2.py:91-107 StyleEncoder(4, 3, 64, style_dim, 'none', 'relu', 'reflect')
    built from utils.py:1082-1149 Conv2dBlock (reflect pad -> conv -> ReLU)."""
    x = F.relu(_conv(sd, p + ".0.conv", F.pad(x, (3, 3, 3, 3), mode="reflect")))
    for i in (1, 2, 3, 4):
        x = F.relu(_conv(sd, p + ".%d.conv" % i, F.pad(x, (1, 1, 1, 1), mode="reflect"), stride=2))
    x = x.mean(dim=(2, 3), keepdim=True)
    return _conv(sd, p + ".6", x)


def mlp(sd, x, p="mlp.model"):
    """models/RITnet_v2.py:110-121 MLP(style_dim, 2*fc, 256, 3): fc-relu, fc-relu, fc."""
    x = x.reshape(x.shape[0], -1)
    x = F.relu(F.linear(x, sd[p + ".0.fc.weight"], sd[p + ".0.fc.bias"]))
    x = F.relu(F.linear(x, sd[p + ".1.fc.weight"], sd[p + ".1.fc.bias"]))
    return F.linear(x, sd[p + ".2.fc.weight"], sd[p + ".2.fc.bias"])


def esf_forward(sd, setting, x, x_edge, target, pupil_center, elNorm, spatWts, distMap, cond, ID,
                alpha, variant="v2", training=False, disentangle=False, update=None):
    """models/RITnet_v2.py:261-354 (variant 'v2') / RITnet_concat.py:225-270 (variant 'concat').

    Returns (op, elPred, latent, loss[1], elOut, terms).
    """
    B = x.shape[0]
    if variant == "v2":
        assert setting["input_concat"] + setting["add_edge"] < 2
        if setting["only_edge"] == 1:
            x = x_edge
        if setting["input_concat"] == 1:
            x = torch.cat((x, x_edge), 1)
    s4, s3, s2, s1, xb = encoder(sd, x, training, update)
    latent = xb.flatten(2).mean(-1)
    e_sk = None
    if variant == "concat" or setting["add_edge"] == 1:
        e4, e3, e2, e1, xe = encoder(sd, x_edge, training, update)
        xb = torch.cat((xb, xe), 1)
        if variant == "concat":
            e_sk = (e4, e3, e2, e1)
    h = xb
    for k, sk in zip((4, 3, 2, 1), (s4, s3, s2, s1)):
        sks = [sk] if e_sk is None else [sk, e_sk[4 - k]]
        h = up_block(sd, "dec.up_block%d" % k, sks, h)
    op = conv_block(sd, "dec.final", h, training, update)
    if variant == "v2" and setting["add_seg"] == 1:
        sm = torch.softmax(op.detach() if setting["seg_detach"] else op, dim=1)
        ad = mlp(sd, style_encoder(sd, sm)).reshape(B, 2, -1)
        # RITnet_v2.py:251-259 calc_mean_std: unbiased variance + 1e-5
        flat = xb.flatten(2)
        std = (flat.var(dim=2) + 1e-5).sqrt()[:, :, None, None]
        mean = flat.mean(dim=2)[:, :, None, None]
        xb = (xb - mean) / std * ad[:, 0].reshape(B, -1, 1, 1) + ad[:, 1].reshape(B, -1, 1, 1)
    elOut = regression(sd, xb)
    total, pred_c, terms = losses.all_loss(op, elOut, target, pupil_center, elNorm, spatWts,
                                           distMap, cond, alpha)
    elPred = torch.cat([pred_c[:, 0, :], elOut[:, 2:5], pred_c[:, 1, :], elOut[:, 7:10]], dim=1)
    if disentangle and variant == "v2":
        # RITnet_v2.py:343-350 (toggle=True branch): + 2 * conf_Loss(linStack(latent))
        pd = F.linear(F.linear(latent, sd["dsIdentify_lin.layersLin.0.weight"],
                               sd["dsIdentify_lin.layersLin.0.bias"]),
                      sd["dsIdentify_lin.layersLin.1.weight"], sd["dsIdentify_lin.layersLin.1.bias"])
        cl = losses.conf_loss(pd, ID.to(torch.long), True)
        terms["conf"] = cl
        total = total + 2 * cl
    if not torch.is_tensor(total):
        total = torch.tensor(float(total))
    return op, elPred, latent, total.reshape(1), elOut, terms
