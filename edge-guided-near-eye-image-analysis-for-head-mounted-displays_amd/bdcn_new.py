"""BDCN edge extractor on the HIP path -- drop-in for the reference's ``bdcn_new.BDCN``.

Same constructor, same ``state_dict`` keys / shapes (``features.conv1_1.weight`` ...
``fuse.bias``; bdcn_new.py:66-114, vgg16_c.py:11-39) and the same ``forward`` contract
(bdcn_new.py:116-191): ``x [B,3,H,W] -> list of 11 sigmoid maps [B,1,H,W]``, fused map last.
The network is frozen in this pipeline (train.py:129, utils.py:646), so only forward exists.

Execution: 13 trunk convs and 13 MSBlocks run on the implicit-GEMM MFMA kernel (the three dilated
convs of a block + the 4-way sum fused into one launch), pools are NHWC kernels, and the side
path (1x1 down convs, score heads, transposed-conv upsampling, crops, cascades, fuse, sigmoids)
is two HBM-bound kernels.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from . import engine
from .engine import ACT_RELU, ConvLayer, Piece, PlanarPiece, Plan, VersionGuard, maxpool_out, pad8, require_cuda

import os
import types

PLANAR_IN = os.environ.get("EGNE_PLANAR_IN", "1") != "0"     # conv1_1 reads the NCHW frames in place
SCORES_FUSED = os.environ.get("EGNE_SCORES_FUSED", "1") != "0"     # stage score heads in the epilogue of the MSBlock kernel
MS_SIDE_STREAM = os.environ.get("EGNE_MS_SIDE", "1") != "0"        # MSBlocks on the plan's second stream next to the trunk
MS_SIDE_MIN_B = int(os.environ.get("EGNE_MS_SIDE_MIN_B", "8"))      # ... for batches of at least this many frames

# (name, cin, cout, dilation) / pool markers; vgg16_c.py:11-39
_VGG = [("conv1_1", 3, 64, 1), ("conv1_2", 64, 64, 1), ("P", 2),
        ("conv2_1", 64, 128, 1), ("conv2_2", 128, 128, 1), ("P", 2),
        ("conv3_1", 128, 256, 1), ("conv3_2", 256, 256, 1), ("conv3_3", 256, 256, 1), ("P", 2),
        ("conv4_1", 256, 512, 1), ("conv4_2", 512, 512, 1), ("conv4_3", 512, 512, 1), ("P", 1),
        ("conv5_1", 512, 512, 2), ("conv5_2", 512, 512, 2), ("conv5_3", 512, 512, 2)]
_STAGES = [("1", ["1_1", "1_2"], 64), ("2", ["2_1", "2_2"], 128), ("3", ["3_1", "3_2", "3_3"], 256),
           ("4", ["4_1", "4_2", "4_3"], 512), ("5", ["5_1", "5_2", "5_3"], 512)]
_UPS = {"2": ("upsample_2", 2, 1), "3": ("upsample_4", 4, 2), "4": ("upsample_8", 8, 4), "5": ("upsample_8_5", 8, 0)}


def get_upsampling_weight(in_channels, out_channels, kernel_size):
    """2-D bilinear kernel for ConvTranspose2d (bdcn_new.py:14-27)."""
    factor = (kernel_size + 1) // 2
    center = factor - 1 if kernel_size % 2 == 1 else factor - 0.5
    r = 1 - (torch.arange(kernel_size, dtype=torch.float64) - center).abs() / factor
    filt = r[:, None] * r[None, :]
    w = torch.zeros((in_channels, out_channels, kernel_size, kernel_size), dtype=torch.float64)
    for i in range(min(in_channels, out_channels)):
        w[i, i] = filt
    return w.float()


class VGG16_C(nn.Module):
    """Parameter container with the keys of vgg16_c.VGG16_C (vgg16_c.py:6-39)."""

    def __init__(self, pretrain=None, logger=None):
        super().__init__()
        for item in _VGG:
            if item[0] == "P":
                continue
            name, cin, cout, d = item
            conv = nn.Conv2d(cin, cout, 3, stride=1, padding=d, dilation=d)
            n = 9 * cout
            conv.weight.data.normal_(0, (2.0 / n) ** 0.5)   # vgg16_c.py:90-98
            conv.bias.data.zero_()
            setattr(self, name, conv)
        if pretrain:
            sd = torch.load(pretrain, map_location="cpu")
            own = self.state_dict()
            for k in own:
                if k in sd:
                    own[k].copy_(sd[k])


class MSBlock(nn.Module):
    """Parameter container of bdcn_new.MSBlock (bdcn_new.py:29-47)."""

    def __init__(self, c_in, rate=4):
        super().__init__()
        self.rate = rate
        self.conv = nn.Conv2d(c_in, 32, 3, stride=1, padding=1)
        for i in (1, 2, 3):
            d = rate * i if rate >= 1 else 1
            setattr(self, "conv%d" % i, nn.Conv2d(32, 32, 3, stride=1, dilation=d, padding=d))
        for m in (self.conv, self.conv1, self.conv2, self.conv3):
            m.weight.data.normal_(0, 0.01)
            m.bias.data.zero_()


class BDCN(nn.Module):
    def __init__(self, pretrain=None, logger=None, rate=4):
        super().__init__()
        self.pretrain = pretrain
        self.rate = rate
        self.features = VGG16_C(pretrain, logger)
        for st, blocks, cin in _STAGES:
            for b in blocks:
                setattr(self, "msblock" + b, MSBlock(cin, rate))
                setattr(self, "conv%s_down" % b, nn.Conv2d(32, 21, (1, 1), stride=1))
            setattr(self, "score_dsn" + st, nn.Conv2d(21, 1, (1, 1), stride=1))
            setattr(self, "score_dsn%s_1" % st, nn.Conv2d(21, 1, (1, 1), stride=1))
        self.upsample_2 = nn.ConvTranspose2d(1, 1, 4, stride=2, bias=False)
        self.upsample_4 = nn.ConvTranspose2d(1, 1, 8, stride=4, bias=False)
        self.upsample_8 = nn.ConvTranspose2d(1, 1, 16, stride=8, bias=False)
        self.upsample_8_5 = nn.ConvTranspose2d(1, 1, 16, stride=8, bias=False)
        self.fuse = nn.Conv2d(10, 1, 1, stride=1)
        self._initialize_weights()
        self._plans = {}
        self.f16_products = 0
        self._events = None  # bench.py: list collecting per-launch HIP events
        self.edge_thres = 0  # set by utils.calc_edge to fuse the >=0.1 -> 1 threshold into the tail

    def _initialize_weights(self, logger=None):
        """bdcn_new.py:193-217."""
        with torch.no_grad():
            for name, p in self.state_dict().items():
                if self.pretrain and "features" in name:
                    continue
                if "upsample" in name:
                    k = int(name.split(".")[0].split("_")[1])
                    p.copy_(get_upsampling_weight(1, 1, k * 2))
                elif "fuse" in name:
                    p.zero_() if "bias" in name else p.fill_(0.080)
                elif "bias" in name:
                    p.zero_()
                else:
                    p.normal_(0, 0.01)

    # ------------------------------------------------------------------------------------------
    def _build(self, B, H, W, dev, only_fuse, edge_thres, f16_storage=None):
        if f16_storage is None:
            # plain-f16 plans: conv1_1 / conv1_2 / pool1 as F16 tensors (egne_conv_desc.out_split = 2) -- their consumers round every operand
            # to exactly the stored value anyway, so nothing changes but the bytes; fp32 tensors where a chosen kernel does not know the storage
            # (level 2: also the outputs of conv3_1 .. conv5_3 and pool3 / pool4 -- deep trunk kernel, streamed-weights 3x3 and pooling read them)
            if engine.F16_STORAGE and getattr(self, "_plan_products", 0) == 1:
                for level in ((2, 1) if engine.F16_STORAGE_DEEP else (1,)):
                    try:
                        return self._build(B, H, W, dev, only_fuse, edge_thres, level)
                    except engine.NeedsFp32Storage:
                        pass
            return self._build(B, H, W, dev, only_fuse, edge_thres, 0)
        pl = Plan(dev)
        pl.f16_storage = int(f16_storage)
        # BDCN.f16_products = 1: plain f16 operands (one MFMA per product instead of the split's three) in the kernels that know
        # egne_conv_desc.f16_products -- the frozen edge network next to a training plan with bf16 activation storage, which rounds
        # the edge map to bf16 on entry (train.py / bench.py set it for --prec 16 only; inference and fp32 storage keep the split)
        pl.f16_products = getattr(self, "_plan_products", 0)
        L = pl.L
        f = self.features
        x_in = pl.vec(B, 3, H, W)
        if PLANAR_IN and engine.F16X3_ENABLED and engine.C4H_MODE != "off":
            cur = PlanarPiece(x_in)        # conv1_1 reads the NCHW frames in place (conv3x3_c4_f16.hip): no layout kernel
        else:
            xb = pl.buf(B, H, W, 8)
            pl.raw(L.egne_nchw_to_nhwc, (x_in.data_ptr(), B, 3, H, W, xb.data_ptr(), 8, 0, 8), "bdcn.in")
            cur = Piece(xb, 0, 3)
        ch, hh, ww = 3, H, W
        # MSBlocks + stage scores.  A block only needs its own trunk layer, so it is emitted right behind that layer and -- for batches
        # that fill the chip -- launched on the plan's SECOND stream (Plan.side_default): the narrow MSBlock launches of stages 3-5
        # (300 tiles on 512 workgroup slots) and the ragged frame tails of the wide trunk layers (88 of 256 slots) run next to each
        # other instead of one after the other.  The side stream is joined in front of the tail kernel, which reads every score map.
        tail = _lib.BdcnTailDesc()
        tail.B, tail.H, tail.W = B, H, W
        block_of = {}           # trunk layer name -> (stage index, block index)
        for si, (st, blocks, cin) in enumerate(_STAGES):
            for bi, b in enumerate(blocks):
                block_of["conv" + b] = (si, bi)
        stages = {}
        side = MS_SIDE_STREAM and B >= MS_SIDE_MIN_B

        def stage_state(si, h_s, w_s):
            if si in stages:
                return stages[si]
            st, blocks, cin = _STAGES[si]
            nb = len(blocks)
            S = types.SimpleNamespace(st=st, blocks=blocks, nb=nb, h=h_s, w=w_s, ms_bufs=[], fused=None)
            S.o_buf = pl.buf(B, h_s, w_s, 32)
            S.dn = [getattr(self, "conv%s_down" % b) for b in blocks]
            S.sa, S.sb = getattr(self, "score_dsn" + st), getattr(self, "score_dsn%s_1" % st)
            S.s, S.s1 = pl.vec(B, h_s, w_s), pl.vec(B, h_s, w_s)
            # Score heads fused into the dilated-branch kernel (bdcn_new.py:118-166 is linear after the MSBlock): per block
            # and head ONE 32-vector  head_w[21] @ down_w[21, 32]; the constant  head_w @ sum_k down_b[k] + head_b  rides with
            # the stage's first block.  The 32-channel block outputs are then never written.
            S.cw, S.cc = pl.vec(nb, 2, 32), pl.vec(2)

            def refresh_scores(cw=S.cw, cc=S.cc, dn=S.dn, sa=S.sa, sb=S.sb):
                with torch.no_grad():
                    heads = torch.stack([sa.weight.detach().reshape(21), sb.weight.detach().reshape(21)])      # [2, 21]
                    for k, m in enumerate(dn):
                        cw[k].copy_(heads @ m.weight.detach().reshape(21, 32))
                    bsum = torch.stack([m.bias.detach() for m in dn]).sum(0)
                    cc.copy_(heads @ bsum + torch.cat([sa.bias.detach(), sb.bias.detach()]))
            S.refresh_scores = refresh_scores
            stages[si] = S
            return S

        def emit_block(si, bi, src, c_in, hh, ww):
            S = stage_state(si, hh, ww)
            b = S.blocks[bi]
            pl.side_default = side
            mb = getattr(self, "msblock" + b)
            l0 = ConvLayer([mb.conv.weight], [mb.conv.bias], [(c_in, pad8(c_in))], pad=(1, 1), act=ACT_RELU)
            l0.split = True
            o = Piece(S.o_buf, 0, 32)
            r = self.rate
            dil = tuple(r * i if r >= 1 else 1 for i in (1, 2, 3))
            lg = ConvLayer([mb.conv1.weight, mb.conv2.weight, mb.conv3.weight],
                           [mb.conv1.bias, mb.conv2.bias, mb.conv3.bias], [(32, 32)], pad=(1, 1), dils=dil,
                           act=ACT_RELU)
            lg.split = True
            # `o` feeds the three dilated convolutions only (bdcn_new.py:50-54), which stage every element 13.5 times: where the
            # one-launch kernel runs them, the producer writes o as split hi / lo halves once (engine.SplitScale)
            if engine.PRESPLIT and pl.msdil_ok(lg, o, hh, ww):
                o.presplit = engine.SplitScale()
            pl.conv(l0, [src], o, B, hh, ww, name="ms%s.conv" % b)
            if not pl.last_presplit:
                o.presplit = None          # (the kernel chosen for this convolution writes plain fp32)
            if S.fused is None:
                S.fused = SCORES_FUSED and pl.msdil_ok(lg, o, hh, ww)
                if S.fused:
                    pl.pre.append(VersionGuard([p for m in S.dn + [S.sa, S.sb] for p in (m.weight, m.bias)], S.refresh_scores))
            if S.fused:
                pl.conv(lg, [o], Piece(S.o_buf, 0, 32), B, hh, ww, residual=o, name="ms%s.dil" % b, scores=(S.cw[bi], S.cc, S.s, S.s1, bi > 0))
            else:
                msb = pl.buf(B, hh, ww, 32)
                pl.conv(lg, [o], Piece(msb, 0, 32), B, hh, ww, residual=o, name="ms%s.dil" % b)
                S.ms_bufs.append(msb)
            if bi == S.nb - 1:
                finish_stage(si, S)
            pl.side_default = False

        def finish_stage(si, S):
            st, nb, dn, sa, sb, s, s1 = S.st, S.nb, S.dn, S.sa, S.sb, S.s, S.s1
            if not S.fused:
                wd, bd = pl.vec(nb, 21, 32), pl.vec(nb, 21)
                heads = pl.vec(2, 21)
                hb = pl.vec(2)

                def refresh(wd=wd, bd=bd, heads=heads, hb=hb, dn=dn, sa=sa, sb=sb):
                    with torch.no_grad():
                        for k, m in enumerate(dn):
                            wd[k].copy_(m.weight.detach().reshape(21, 32))
                            bd[k].copy_(m.bias.detach())
                        heads[0].copy_(sa.weight.detach().reshape(21))
                        heads[1].copy_(sb.weight.detach().reshape(21))
                        hb[0:1].copy_(sa.bias.detach())
                        hb[1:2].copy_(sb.bias.detach())
                pl.pre.append(VersionGuard([p for m in dn + [sa, sb] for p in (m.weight, m.bias)], refresh))
                arr = (C.c_void_p * nb)(*[t.data_ptr() for t in S.ms_bufs])
                pl.keep.append(arr)
                pl.raw(L.egne_bdcn_stage_scores,
                       (arr, nb, 32, B * S.h * S.w, wd.data_ptr(), bd.data_ptr(), heads.data_ptr(), hb.data_ptr(),
                        heads.data_ptr() + 4 * 21, hb.data_ptr() + 4, s.data_ptr(), s1.data_ptr()), "bdcn.scores" + st)
            tail.s[si], tail.s1[si] = s.data_ptr(), s1.data_ptr()
            tail.h[si], tail.w[si] = S.h, S.w
            if st in _UPS:
                key, stride, crop = _UPS[st]
                upw = pl.vec(2 * stride, 2 * stride)
                mod = getattr(self, key)
                pl.pre.append(VersionGuard([mod.weight], lambda upw=upw, mod=mod: upw.copy_(
                    mod.weight.detach().reshape(upw.shape))))
                tail.up[si], tail.stride[si], tail.crop[si] = upw.data_ptr(), stride, crop
            else:
                tail.stride[si], tail.crop[si] = 1, 0

        pooled = None               # a stride-2 pooling already written by the convolution in front of it
        for idx, item in enumerate(_VGG):
            if item[0] == "P":
                s = item[1]
                ho, wo = maxpool_out(hh, s), maxpool_out(ww, s)
                if pooled is not None:
                    dst = pooled
                else:
                    c16 = getattr(cur, "f16s", None)
                    ob = (pl.buf16 if c16 is not None else pl.buf)(B, ho, wo, cur.Cp)
                    dst = Piece(ob, 0, cur.C)
                    dst.f16s = c16
                    pl.maxpool2(cur, dst, B, hh, ww, s, "vgg.pool")
                cur, hh, ww, pooled = dst, ho, wo, None
                continue
            name, cin, cout, d = item
            conv = getattr(f, name)
            layer = ConvLayer([conv.weight], [conv.bias], [(cin, pad8(cin))], pad=(1, 1), dils=(d,), act=ACT_RELU)
            layer.split = cin % 32 == 0      # frozen trunk: split-f16 MFMA (conv_f16x3.hip), edge map tolerance 1e-3
            layer.split_c4 = cin <= 4        # conv1_1: streaming split-f16 first-layer kernel (conv3x3_c4_f16.hip)
            h16 = (f16_storage >= 1 and name in ("conv1_1", "conv1_2")) or (f16_storage >= 2 and name[4] in "345")
            ob = (pl.buf16 if h16 else pl.buf)(B, hh, ww, cout)
            dst = Piece(ob, 0, cout)
            if h16:
                dst.f16s = engine.SplitScale()
            nxt = _VGG[idx + 1] if idx + 1 < len(_VGG) else None
            pq = None
            if nxt is not None and nxt[0] == "P" and nxt[1] == 2:         # vgg16_c.py:70: pooling right behind this convolution
                pq = Piece((pl.buf16 if h16 else pl.buf)(B, maxpool_out(hh, 2), maxpool_out(ww, 2), pad8(cout)), 0, cout)
                pq.f16s = dst.f16s
            pl.conv(layer, [cur], dst, B, hh, ww, name="vgg." + name, pool=pq)
            pooled = pq if pl.last_pooled else None
            cur = dst
            si, bi = block_of[name]
            emit_block(si, bi, dst, cout, hh, ww)
        pl.serial_timing = True
        pl.join_before.add(len(pl.calls))       # the tail kernel reads every stage's score maps: the side stream is joined first
        fw, fb = pl.vec(10), pl.vec(1)
        pl.pre.append(VersionGuard([self.fuse.weight, self.fuse.bias], lambda: (
            fw.copy_(self.fuse.weight.detach().reshape(10)), fb.copy_(self.fuse.bias.detach()))))
        tail.fuse_w, tail.fuse_b = fw.data_ptr(), fb.data_ptr()
        outs = pl.vec(11, B, 1, H, W)
        for k in range(11):
            tail.out[k] = outs[k].data_ptr() if (k == 10 or not only_fuse) else None
        tail.edge_thres = int(edge_thres)
        pl.keep.append(tail)
        pl.raw(L.egne_bdcn_tail, (C.byref(tail),), "bdcn.tail")
        pl.x_in, pl.outs = x_in, outs
        return pl

    def _plan(self, x, only_fuse, edge_thres=0):
        require_cuda(x, "BDCN input")
        for p in self.parameters():
            require_cuda(p, "BDCN parameters (call .cuda())")
            break
        B, Cc, H, W = x.shape
        if Cc != 3:
            raise ValueError("BDCN expects a 3-channel input, got %d" % Cc)
        prod = 1 if int(getattr(self, "f16_products", 0)) == 1 else 0
        key = (B, H, W, x.device, bool(only_fuse), int(edge_thres), prod)
        if key not in self._plans:
            self._plan_products = prod
            self._plans[key] = self._build(B, H, W, x.device, only_fuse, edge_thres)
        return self._plans[key]

    def forward(self, x):
        """bdcn_new.py:116-191: returns [p1_1..p5_1, p1_2..p5_2, fuse], each [B,1,H,W]."""
        pl = self._plan(x, only_fuse=False)
        self._last_plan = pl
        pl.x_in.copy_(x.to(torch.float32))
        pl.run()
        return [pl.outs[k].clone() for k in range(11)]

    def forward_fuse(self, x, edge_thres=0):
        """Only the fused edge map (what utils.calc_edge consumes, utils.py:648)."""
        pl = self._plan(x, only_fuse=True, edge_thres=edge_thres)
        self._last_plan = pl
        pl.x_in.copy_(x.to(torch.float32))
        pl.run(self._events)
        return pl.outs[10].clone()

    def overflowed(self):
        """True if the LAST call produced non-finite values inside a split-f16 convolution: a frame whose activations exceed 32x those
        of the batch the pre-scales were calibrated on left the f16 range (engine.Plan.overflowed) -- the edge maps of that call
        are invalid.  Synchronises; the plan is re-calibrated by the next call, so the answer to True is to call again."""
        pl = getattr(self, "_last_plan", None)
        return pl is not None and pl.overflowed()
