"""RITnet (the EllSeg baseline the reference keeps as a comparator) on the HIP path -- drop-in for the reference's
``models/RITnet_v1.py``: same constructor arguments, ``state_dict`` keys, ``forward`` signature and return tuple
(models/RITnet_v1.py:193-309); registered as ``'ritnet_v1'`` (modelSummary.py:18-26).  One-channel input, no edge branch:
five dense blocks with BatchNorm (average pooling in front of blocks 2-5), four up blocks with nearest-neighbour up-sampling,
a 1x1 output convolution, the regression module on the bottleneck, the loss head of ESF-Net (get_allLoss is the same function).
Execution is a launch plan (``build_plan`` below) on the kernels of the ESF-Net path."""
import numpy as np
import torch
import torch.nn as nn

from .. import esf_engine as E
from ..engine import ACT_LEAKY, ACT_NONE, Piece, Plan, pad8
from ..utils import linStack, regressionModule
from . import RITnet_v2 as V2


def getSizes(chz, growth, blks=4):
    """models/RITnet_v1.py:22-36 (every width equals chz)."""
    sizes = {"enc": {"inter": np.array([chz] * blks), "op": np.array([chz] * blks), "ip": np.array([chz] * blks)}, "dec": {}}
    sizes["dec"]["skip"] = sizes["enc"]["ip"][::-1] + sizes["enc"]["inter"][::-1]
    sizes["dec"]["ip"] = sizes["enc"]["op"][::-1]
    sizes["dec"]["op"] = np.append(sizes["enc"]["op"][::-1][1:], chz)
    return sizes


class DenseNet2D_down_block(nn.Module):
    def __init__(self, input_channels, output_channels, down_size, dropout=False, prob=0):
        super().__init__()
        self.conv1 = nn.Conv2d(input_channels, output_channels, kernel_size=(3, 3), padding=(1, 1))
        self.conv21 = nn.Conv2d(input_channels + output_channels, output_channels, kernel_size=(1, 1), padding=(0, 0))
        self.conv22 = nn.Conv2d(output_channels, output_channels, kernel_size=(3, 3), padding=(1, 1))
        self.conv31 = nn.Conv2d(input_channels + 2 * output_channels, output_channels, kernel_size=(1, 1), padding=(0, 0))
        self.conv32 = nn.Conv2d(output_channels, output_channels, kernel_size=(3, 3), padding=(1, 1))
        self.bn = nn.BatchNorm2d(num_features=output_channels)
        self.down_size, self.dropout = down_size, dropout
        if dropout:
            raise NotImplementedError("the reference builds RITnet_v1 with dropout=False (models/RITnet_v1.py:214,222)")


class DenseNet2D_up_block(nn.Module):
    def __init__(self, skip_channels, input_channels, output_channels, up_stride=2, dropout=False, prob=0):
        super().__init__()
        self.conv11 = nn.Conv2d(skip_channels + input_channels, output_channels, kernel_size=(1, 1), padding=(0, 0))
        self.conv12 = nn.Conv2d(output_channels, output_channels, kernel_size=(3, 3), padding=(1, 1))
        self.conv21 = nn.Conv2d(skip_channels + input_channels + output_channels, output_channels, kernel_size=(1, 1), padding=(0, 0))
        self.conv22 = nn.Conv2d(output_channels, output_channels, kernel_size=(3, 3), padding=(1, 1))
        self.up_stride = up_stride


class DenseNet_encoder(nn.Module):
    def __init__(self, in_channels=1, channel_size=32):
        super().__init__()
        self.down_block1 = DenseNet2D_down_block(in_channels, channel_size, None)
        for i in range(2, 6):
            setattr(self, "down_block%d" % i, DenseNet2D_down_block(channel_size, channel_size, (2, 2)))


class DenseNet_decoder(nn.Module):
    def __init__(self, out_channels=3, channel_size=32):
        super().__init__()
        for i in range(1, 5):
            setattr(self, "up_block%d" % i, DenseNet2D_up_block(channel_size, channel_size, channel_size, (2, 2)))
        self.final = nn.Conv2d(channel_size, out_channels, kernel_size=1, padding=0)


class DenseNet2D(V2.DenseNet2D):
    variant = "v1"

    def __init__(self, chz=32, growth=1.2, actfunc=None, norm=None, selfCorr=False, disentangle=False, dropout=True, prob=0.2):
        nn.Module.__init__(self)
        # (the reference's `dropout` / `prob` arguments never reach its blocks: enc and dec are built with dropout=False)
        self.sizes = getSizes(chz, growth)
        self.chz, self.growth = chz, growth
        self.toggle = True
        self.selfCorr = selfCorr
        self.disentangle = disentangle
        self.disentangle_alpha = 2
        self.setting = {}
        self.enc = DenseNet_encoder(in_channels=1, channel_size=chz)
        self.dec = DenseNet_decoder(out_channels=3, channel_size=chz)
        self.elReg = regressionModule(self.sizes)
        self._initialize_weights()
        self._plans = {}
        self._events = None
        self.storage_dtype = torch.float32

    def setDatasetInfo(self, numSets=2):
        """models/RITnet_v1.py:232-242."""
        self.numSets = numSets
        self.dsIdentify_lin = linStack(num_layers=2, in_dim=int(self.sizes["enc"]["op"][-1]), hidden_dim=64, out_dim=numSets,
                                       bias=True, actBool=False, dp=0.0)

    def _build_plan(self, B, H, W, dev, training, dtype):
        return build_plan(self, B, H, W, dev, training, dtype)


def build_plan(model, B, H, W, dev, training, dtype=torch.float32):
    """Launch plan of models/RITnet_v1.py:244-309 for one (B, H, W, mode): every torch.cat is a list of slices read in place,
    eval-mode BatchNorm is folded behind its block's last convolution, training mode keeps batch statistics (E._train_bn)."""
    if H % 16 or W % 16:
        raise ValueError("RITnet_v1 pools four times: H and W must be multiples of 16 (got %dx%d)" % (H, W))
    pl = Plan(dev, train=training, dtype=dtype)
    L = pl.L
    pl.dbg = {}
    E._cl.eval_plan = not training
    chz = model.chz
    pl.in_img = pl.vec(B, 1, H, W)
    pl.in_edge = pl.vec(B, 1, H, W)          # (forward() copies the caller's edge map in; this model does not read it)
    xin = pl.buf(B, H, W, 8)
    pl.raw(L.egne_nchw_to_nhwc, (pl.in_img.data_ptr(), B, 1, H, W, xin.data_ptr(), 8, 0, 8), "in.img")
    x = Piece(xin, 0, 1, 8)
    x.nograd = True
    outs = []
    for i in range(1, 6):
        blk = getattr(model.enc, "down_block%d" % i)
        h, w = H >> (i - 1), W >> (i - 1)
        nm = "enc.b%d" % i
        if i > 1:       # AvgPool2d in FRONT of the block (models/RITnet_v1.py:57-58)
            (xp,) = E.concat_members(pl, B, h, w, [chz])
            pl.avgpool2(outs[-1], xp, B, 2 * h, 2 * w, name=nm + ".pool")
            x = xp
        x1, x22, pre = E.concat_members(pl, B, h, w, [chz, chz, chz])
        l = E._cl(blk.conv1, E._lay([x]), pad=(1, 1), act=ACT_LEAKY)
        pl.conv(l, [x], x1, B, h, w, name=nm + ".conv1")
        l1 = E._cl(blk.conv21, E._lay([x, x1]))
        l2 = E._cl(blk.conv22, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
        pl.conv_pair(l1, [x, x1], l2, x22, B, h, w, name=nm + ".conv2")
        l1 = E._cl(blk.conv31, E._lay([x, x1, x22]))
        l2 = E._cl(blk.conv32, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
        if not training:
            fold = E._BNFold(blk.bn, l2.CoutP, dev)
            pl.pre.append(fold.guard)
            l2.post = (fold.scale, fold.shift)
            pl.conv_pair(l1, [x, x1, x22], l2, pre, B, h, w, name=nm + ".conv3")
            out = pre
        else:
            pl.conv_pair(l1, [x, x1, x22], l2, pre, B, h, w, name=nm + ".conv3")
            (out,) = E.concat_members(pl, B, h, w, [chz])
            E._train_bn(pl, blk.bn, pre, out, 0, B, h * w, nm + ".bn")
        outs.append(out)
    x5 = outs[4]
    hb, wb = H >> 4, W >> 4
    fc = chz
    pl.latent_p = pl.buf(B, 1, 1, pad8(fc))
    pl.raw(L.egne_spatial_mean, (x5.ptr, x5.stride, x5.off, pad8(fc), B, hb * wb, pl.latent_p.data_ptr()), "latent")
    pl.latent = pl.latent_p.view(B, pad8(fc))[:, :fc]
    if training:
        def emit_latent(bw):
            gl, gb = pl.gbuf(pl.latent_p), pl.gp(x5)
            E.latent_upstream(pl, bw, gl, B, pad8(fc), fc)
            bw.raw(L.egne_spatial_mean_bwd, (gl.data_ptr(), pad8(fc), gb.ptr, gb.stride, gb.off, pad8(fc), B, hb * wb), "latent.bwd")
        pl.tape.append(emit_latent)
    E.regression_head(pl, model.elReg, [x5], B, hb, wb, training)
    cur, ch, cw = x5, hb, wb
    for k in (4, 3, 2, 1):
        ub = getattr(model.dec, "up_block%d" % k)
        h, w = 2 * ch, 2 * cw
        nm = "dec.up%d" % k
        skip = outs[k - 1]
        up, x1, y = E.concat_members(pl, B, h, w, [chz, chz, chz])
        pl.upsample2x_nearest(cur, up, B, ch, cw, name=nm + ".up")
        cat = [up, skip]
        l1 = E._cl(ub.conv11, E._lay(cat))
        l2 = E._cl(ub.conv12, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
        pl.conv_pair(l1, cat, l2, x1, B, h, w, name=nm + ".conv1")
        l1 = E._cl(ub.conv21, E._lay(cat + [x1]))
        l2 = E._cl(ub.conv22, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
        pl.conv_pair(l1, cat + [x1], l2, y, B, h, w, name=nm + ".conv2")
        cur, ch, cw = y, h, w
    opb = pl.buf(B, H, W, 8)
    l = E._cl(model.dec.final, E._lay([cur]), act=ACT_NONE)
    pl.conv(l, [cur], Piece(opb, 0, 3), B, H, W, name="dec.final")
    E.loss_head(pl, opb, B, H, W, dev, training)
    E.confusion_head(pl, model, fc, B, training, True)
    if training:
        pl.build_backward()
    return pl
