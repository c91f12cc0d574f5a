"""Oracle for the BDCN edge extractor (test infrastructure; see oracle/__init__.py).

Functional form over a state_dict ``sd`` with the reference's key names
(``features.conv1_1.weight`` ... ``fuse.bias``), so the same seeded dict drives the reference,
this oracle and the HIP path.
"""
import torch
import torch.nn.functional as F

# vgg16_c.py:11-39 -- (name, dilation); 'P2' = maxpool k2 s2 ceil, 'P1' = maxpool k2 s1 ceil
VGG_PLAN = [("conv1_1", 1), ("conv1_2", 1), "P2",
            ("conv2_1", 1), ("conv2_2", 1), "P2",
            ("conv3_1", 1), ("conv3_2", 1), ("conv3_3", 1), "P2",
            ("conv4_1", 1), ("conv4_2", 1), ("conv4_3", 1), "P1",
            ("conv5_1", 2), ("conv5_2", 2), ("conv5_3", 2)]

# bdcn_new.py:72-107 -- stage -> MSBlock names
STAGES = [("1", ["1_1", "1_2"]), ("2", ["2_1", "2_2"]), ("3", ["3_1", "3_2", "3_3"]),
          ("4", ["4_1", "4_2", "4_3"]), ("5", ["5_1", "5_2", "5_3"])]
# bdcn_new.py:108-111,127-164 -- stage -> (upsampler key, stride, crop offset)
UPS = {"2": ("upsample_2", 2, 1), "3": ("upsample_4", 4, 2), "4": ("upsample_8", 8, 4),
       "5": ("upsample_8_5", 8, 0)}


def vgg_features(sd, x, prefix="features."):
    """vgg16_c.py:65-88 -- 13 side features."""
    side = []
    for item in VGG_PLAN:
        if item == "P2":
            x = F.max_pool2d(x, 2, stride=2, ceil_mode=True)
        elif item == "P1":
            x = F.max_pool2d(x, 2, stride=1, ceil_mode=True)
        else:
            name, d = item
            x = F.relu(F.conv2d(x, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"],
                                padding=d, dilation=d))
            side.append(x)
    return side


def msblock(sd, p, x, rate=4):
    """bdcn_new.py:49-55."""
    o = F.relu(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1))
    acc = o
    for i in (1, 2, 3):
        d = rate * i
        acc = acc + F.relu(F.conv2d(o, sd[p + "conv%d.weight" % i], sd[p + "conv%d.bias" % i],
                                    padding=d, dilation=d))
    return acc


def bdcn_forward(sd, x, rate=4):
    """bdcn_new.py:116-191 -- returns the list of 11 sigmoid maps; fuse is [-1]."""
    H, W = x.shape[-2:]
    feats = vgg_features(sd, x)
    fi = 0
    s_a, s_b = [], []  # the two score maps per stage at input resolution
    for st, blocks in STAGES:
        tot = None
        for b in blocks:
            m = msblock(sd, "msblock%s." % b, feats[fi], rate)
            fi += 1
            dn = F.conv2d(m, sd["conv%s_down.weight" % b], sd["conv%s_down.bias" % b])
            tot = dn if tot is None else tot + dn
        a = F.conv2d(tot, sd["score_dsn%s.weight" % st], sd["score_dsn%s.bias" % st])
        b_ = F.conv2d(tot, sd["score_dsn%s_1.weight" % st], sd["score_dsn%s_1.bias" % st])
        if st in UPS:
            key, stride, off = UPS[st]
            w = sd[key + ".weight"]
            a = F.conv_transpose2d(a, w, stride=stride)[:, :, off:off + H, off:off + W]
            b_ = F.conv_transpose2d(b_, w, stride=stride)[:, :, off:off + H, off:off + W]
        s_a.append(a)
        s_b.append(b_)
    # bdcn_new.py:165-177 -- cascades (detach only matters for gradients)
    # same association as the reference: s_k + o_{k-1} + ... + o_1  /  s_k1 + o_{k+1,1} + ... + o_51
    p_a, p_b = [], []
    for k in range(5):
        t = s_a[k]
        for j in range(k - 1, -1, -1):
            t = t + s_a[j]
        p_a.append(t)
        t = s_b[k]
        for j in range(k + 1, 5):
            t = t + s_b[j]
        p_b.append(t)
    maps = p_a + p_b
    fuse = F.conv2d(torch.cat(maps, 1), sd["fuse.weight"], sd["fuse.bias"])
    return [torch.sigmoid(m) for m in maps] + [torch.sigmoid(fuse)]


def calc_edge(sd, img, edge_thres=0):
    """utils.py:645-656 -- grey -> 3 identical channels -> BDCN -> fused map (+ optional >=0.1 -> 1)."""
    with torch.no_grad():
        e = bdcn_forward(sd, torch.cat((img, img, img), dim=1))[-1]
    if edge_thres == 1:
        e = torch.where(e >= 0.1, torch.ones_like(e), e)
    return e
