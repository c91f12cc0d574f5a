"""DeepVOG (the second comparator the reference keeps) on the HIP path -- drop-in for the reference's
``models/deepvog_pytorch.py``: same constructor arguments, ``state_dict`` keys, ``forward`` signature and return
tuple (models/deepvog_pytorch.py:83-146); registered as ``'deepvog'`` (modelSummary.py:26) and selectable in evaluate.py
(``--model deepvog``, evaluate.py:364-365).

A four-level U-Net: 3x3 conv + BatchNorm + ReLU and a 2x2 / stride-2 conv + BatchNorm + ReLU per encoder level, decoder levels
with nearest-neighbour up-sampling, a 1x1 convolution to TWO channels (background / pupil), its own loss (get_allLoss, :148-167).
Execution is a launch plan on the kernels of the ESF-Net path; eval-mode BatchNorm sits between convolution and ReLU here, so it is
folded into the convolution's weights and bias (derived tensors refreshed when a parameter or running statistic changes).

Training (``model.train()``; the reference's loop can train any registered model, train.py:262-287): fp32 storage only.  Every
convolution is followed by BatchNorm with batch statistics (a statistics pass, an affine + ReLU pass; running statistics updated as
torch does) and the backward plan runs ReLU backward, the BatchNorm backward of the ESF-Net path, the generic weight gradient and -- for
the 2x2 / stride-2 convolutions -- a phase-packed 1x1 data gradient that one kernel un-shuffles; the loss has its own backward kernel.
Simple passes, not tuned: the comparator is trained for comparison, not for throughput.
"""
import types

import torch
import torch.nn as nn

from .. import esf_engine as E
from ..engine import ACT_NONE, ACT_RELU, Piece, Plan, VersionGuard, require_cuda
from . import RITnet_v2 as V2


class encoding_block(nn.Module):
    def __init__(self, input_channels, filter_size, filters_num, layer_num, block_type, stage, s=1, X_skip=0):
        super().__init__()
        self.conv1 = nn.Conv2d(input_channels, filters_num, kernel_size=filter_size, stride=(s, s), padding=(1, 1))
        self.conv2 = nn.Conv2d(filters_num, filters_num * 2, kernel_size=(2, 2), stride=(2, 2), padding=(0, 0))
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm2d(num_features=filters_num)
        self.bn2 = nn.BatchNorm2d(num_features=filters_num * 2)
        _xavier(self)


class decoding_block(nn.Module):
    def __init__(self, skip_channels, input_channels, filter_size, filters_num, layer_num, block_type, stage, s=1, up_stride=(2, 2),
                 X_jump=0, up_sampling=True):
        super().__init__()
        self.conv1 = nn.Conv2d(input_channels + skip_channels, filters_num, kernel_size=filter_size, stride=(s, s), padding=(1, 1))
        self.conv2 = nn.Conv2d(filters_num, filters_num, kernel_size=filter_size, stride=(1, 1), padding=(1, 1))
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm2d(num_features=filters_num)
        self.bn2 = nn.BatchNorm2d(num_features=filters_num)
        self.X_jump, self.up_sampling, self.up_stride = X_jump, up_sampling, up_stride
        _xavier(self)


def _xavier(mod):
    """models/deepvog_pytorch.py:29-33 (every block and the model itself re-initialise their convolutions this way)."""
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight, gain=nn.init.calculate_gain("relu"))


class _FoldedConvBN:
    """conv -> eval BatchNorm as ONE convolution: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta.
    ``sum_in``: the model feeds three copies of its one-channel frame (:129), i.e. a one-channel convolution with the input
    channels of the weight summed."""

    def __init__(self, conv, bn, dev, sum_in=False):
        self.conv, self.bn, self.sum_in = conv, bn, sum_in
        shape = list(conv.weight.shape)
        if sum_in:
            shape[1] = 1
        self.weight = torch.zeros(shape, device=dev)
        self.bias = torch.zeros(shape[0], device=dev)
        self.guard = VersionGuard([conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], self._fold)

    def _fold(self):
        bn = self.bn
        s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        w = self.conv.weight.detach()
        if self.sum_in:
            w = w.sum(1, keepdim=True)
        self.weight.copy_(w * s.view(-1, 1, 1, 1))
        self.bias.copy_((self.conv.bias.detach() - bn.running_mean) * s + bn.bias.detach())


class DeepVOG_pytorch(nn.Module):
    def __init__(self, in_channels=3, out_channels=2, filter_size=(3, 3)):
        super().__init__()
        if in_channels != 3 or out_channels != 2 or tuple(filter_size) != (3, 3):
            raise NotImplementedError("DeepVOG_pytorch: the reference's configuration only (3 input copies, 2 classes, 3x3 filters)")
        c = self.output_channels = 16
        kw = dict(filter_size=filter_size, layer_num=1, stage=1, s=1)
        self.down_block1 = encoding_block(input_channels=in_channels, filters_num=c, block_type="down", **kw)
        self.down_block2 = encoding_block(input_channels=c * 2, filters_num=c * 2, block_type="down", **kw)
        self.down_block3 = encoding_block(input_channels=c * 4, filters_num=c * 4, block_type="down", **kw)
        self.down_block4 = encoding_block(input_channels=c * 8, filters_num=c * 8, block_type="down", **kw)
        self.up_block1 = decoding_block(skip_channels=0, input_channels=c * 16, filters_num=c * 16, block_type="up", **kw)
        self.up_block2 = decoding_block(skip_channels=c * 8, input_channels=c * 16, filters_num=c * 16, block_type="up", **kw)
        self.up_block3 = decoding_block(skip_channels=c * 4, input_channels=c * 16, filters_num=c * 8, block_type="up", **kw)
        self.up_block4 = decoding_block(skip_channels=c * 2, input_channels=c * 8, filters_num=c * 4, block_type="up", **kw)
        self.up_block5 = decoding_block(skip_channels=c, input_channels=c * 4, filters_num=c * 2, block_type="up", up_sampling=False, **kw)
        self.conv1 = nn.Conv2d(c * 2, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0))
        _xavier(self)
        self._plans = {}
        self._events = None

    _ensure_grad_arena = V2.DenseNet2D._ensure_grad_arena       # one flat gradient buffer, p.grad are views (the plans hold the pointers)

    def to(self, *args, **kwargs):
        if any(a in (torch.float16, torch.bfloat16) for a in list(args) + [kwargs.get("dtype")]):
            raise NotImplementedError("DeepVOG_pytorch: fp32 only (bf16 activation storage is built for ESF-Net and RITnet_v1)")
        return super().to(*args, **kwargs)

    def _plan(self, B, H, W, dev):
        key = (B, H, W, dev, bool(self.training))          # (index 4 = training: _ensure_grad_arena drops those plans when the arena moves)
        if key not in self._plans:
            self._plans[key] = build_plan(self, B, H, W, dev, bool(self.training))
        return self._plans[key]

    def forward(self, x, x_gt, target, pupil_center, elNorm, spatWts, distMap, cond, ID, alpha):
        """models/deepvog_pytorch.py:115-146.  Returns (out [B,2,H,W], elPred [B,10], embedding [B,5], loss [1], embedding); as in the
        reference, elPred carries the predicted pupil centre at [0:2] and [5:7] and uniform random numbers elsewhere, and the embedding
        is a tensor of ones."""
        require_cuda(x, "x")
        want_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if want_grad:
            self._ensure_grad_arena()
        B, _, H, W = x.shape
        pl = self._plan(B, H, W, x.device)
        self._last_plan = pl
        pl.in_img.copy_(x.expand(-1, pl.in_img.shape[1], -1, -1))
        pl.t_target.copy_(target)
        pl.t_pc.copy_(pupil_center)
        pl.t_cond.copy_(cond)
        if want_grad:
            if getattr(self, "_dummy", None) is None or self._dummy.device != x.device:
                self._dummy = torch.zeros(1, device=x.device, requires_grad=True)
            out, pc, loss = _DeepVOGFunction.apply(self, pl, self._dummy)
        else:
            pl.run(self._events)
            out, pc, loss = pl.op.clone(), pl.pred_c.clone(), pl.terms[0:1].clone()
        r = torch.rand(B, 6, device=x.device)
        elPred = torch.cat([pc, r[:, :3], pc, r[:, 3:]], dim=1)
        emb = torch.ones(B, 5, device=x.device)
        return out, elPred, emb, loss, emb

    def overflowed(self):
        """As DenseNet2D.overflowed (engine.Plan.overflowed): the last inference call left the f16 range of its calibrated pre-scales."""
        last = getattr(self, "_last_plan", None)
        return last is not None and last.overflowed()

    def predictions(self):
        """Argmax mask [B,H,W] int64 of the last forward (utils.get_predictions on the two-channel output): 1 = pupil."""
        last = getattr(self, "_last_plan", None)
        if last is None:
            raise RuntimeError("predictions(): no forward pass has run yet")
        return last.mask.clone()


class _DeepVOGFunction(torch.autograd.Function):
    """Forward = the launch plan; backward = its backward plan scaled by the incoming gradient of the loss (train.py:285-286)."""

    @staticmethod
    def forward(ctx, model, pl, dummy):
        ctx.set_materialize_grads(False)
        ctx.model, ctx.pl = model, pl
        pl.run(model._events)
        return pl.op.clone(), pl.pred_c.clone(), pl.terms[0:1].clone()

    @staticmethod
    def backward(ctx, g_op, g_pc, g_loss):
        if g_op is not None or g_pc is not None:
            raise NotImplementedError("the HIP path back-propagates the returned loss only (train.py:285-286 calls loss.backward())")
        if g_loss is None:
            return None, None, None
        model, pl = ctx.model, ctx.pl
        model._ensure_grad_arena()
        pl.zero_grads()
        pl.gscale.copy_(g_loss.reshape(1))
        pl.bw.run(model._events)
        return None, None, None


def _train_cbr(pl, conv, bn, srcs, B, h, w, name, stride=1):
    """Training mode: conv -> BatchNorm2d with batch statistics (running statistics updated as torch does: momentum, unbiased variance)
    -> ReLU, as explicit passes; returns the activated piece.  Backward: ReLU, BatchNorm (egne_norm_bwd, per_sample = 0), then the
    convolution's own backward from the tape."""
    L = pl.L
    k = conv.kernel_size[0]
    l = E._cl(conv, E._lay(srcs), pad=(k // 2, k // 2) if stride == 1 else (0, 0), act=ACT_NONE, stride=stride)
    ho, wo = h // stride, w // stride
    C_ = conv.out_channels
    (pre,) = E.concat_members(pl, B, ho, wo, [C_])
    (dst,) = E.concat_members(pl, B, ho, wo, [C_])
    pl.conv(l, srcs, pre, B, h, w, name=name)
    n = B * ho * wo
    rstd, nshift, mean, var = pl.norm_stats(pre, B, ho * wo, per_sample=False, eps=bn.eps, want_moments=True, name=name + ".bn.stats")
    sc, sh, gpad = pl.vec(pre.Cp), pl.vec(pre.Cp), pl.vec(pre.Cp)

    def finish():
        with torch.no_grad():
            g = bn.weight.detach()
            gpad[:C_].copy_(g)
            sc[:C_].copy_(rstd[0, :C_] * g)
            sh[:C_].copy_(nshift[0, :C_] * g + bn.bias.detach())
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(mean[0, :C_], alpha=m)
            bn.running_var.mul_(1 - m).add_(var[0, :C_] * (n / max(n - 1, 1)), alpha=m)
            bn.num_batches_tracked.add_(1)
    pl.raw(E._PyCall(finish), (), name + ".bn.coef")
    pl.raw(L.egne_affine_act, (pre.ptr, pre.stride, pre.off, dst.ptr, dst.stride, dst.off, pre.Cp, n, sc.data_ptr(), sh.data_ptr(), ACT_RELU),
           name + ".bn.apply")

    def emit(bw):
        gq, gpre = pl.gp(dst), pl.gp(pre)
        ws = bw.vec((int(L.egne_act_bwd_bias_workspace_bytes(n, pre.Cp)) + 7) // 8, dtype=torch.float64)
        bw.raw(L.egne_act_bwd_bias, (gq.ptr, gq.stride, gq.off, dst.ptr, dst.stride, dst.off, ACT_RELU, pre.Cp, n, None, C_, 1, ws.data_ptr()),
               name + ".relu.bwd")
        sums = bw.vec(pre.Cp * 2)
        wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, ho * wo, pre.Cp, 0)) + 7) // 8, dtype=torch.float64)
        bw.raw(L.egne_norm_bwd, (pre.ptr, pre.stride, pre.off, rstd.data_ptr(), nshift.data_ptr(), gpad.data_ptr(), gq.ptr, gq.stride, gq.off,
                                 0, pre.Cp, B, ho * wo, 0, gpre.ptr, gpre.stride, gpre.off, sums.data_ptr(), bn.weight.grad.data_ptr(),
                                 bn.bias.grad.data_ptr(), C_, wsn.data_ptr()), name + ".bn.bwd")
    pl.tape.append(emit)
    return dst


def build_plan(model, B, H, W, dev, training=False):
    if H % 16 or W % 16:
        raise ValueError("DeepVOG halves the frame four times: H and W must be multiples of 16 (got %dx%d)" % (H, W))
    if training:
        model._ensure_grad_arena()          # the backward plan holds the gradient pointers
    pl = Plan(dev, train=training)
    L = pl.L
    pl.dbg = {}
    E._cl.eval_plan = not training
    nin = 3 if training else 1          # training keeps the three input copies (the first layer's weight gradient has three channels)
    pl.in_img = pl.vec(B, nin, H, W)
    xin = pl.buf(B, H, W, 8)
    pl.raw(L.egne_nchw_to_nhwc, (pl.in_img.data_ptr(), B, nin, H, W, xin.data_ptr(), 8, 0, 8), "in.img")

    def cbr(conv, bn, srcs, h, w, name, stride=1, sum_in=False):
        """conv + BatchNorm + ReLU at INPUT size h x w -> a new piece."""
        if training:
            return _train_cbr(pl, conv, bn, srcs, B, h, w, name, stride)
        f = _FoldedConvBN(conv, bn, dev, sum_in)
        pl.pre.append(f.guard)
        k = conv.kernel_size[0]
        l = E._cl(types.SimpleNamespace(weight=f.weight, bias=f.bias), E._lay(srcs), pad=(k // 2, k // 2) if stride == 1 else (0, 0),
                  act=ACT_RELU, stride=stride)
        pl.keep.append(f)
        (dst,) = E.concat_members(pl, B, h // stride, w // stride, [conv.out_channels])
        pl.conv(l, srcs, dst, B, h, w, name=name)
        return dst

    cur, jumps = Piece(xin, 0, nin, 8), []
    cur.nograd = True
    for i in range(1, 5):
        blk = getattr(model, "down_block%d" % i)
        h, w = H >> (i - 1), W >> (i - 1)
        j = cbr(blk.conv1, blk.bn1, [cur], h, w, "down%d.conv1" % i, sum_in=(i == 1 and not training))
        jumps.append(j)
        cur = cbr(blk.conv2, blk.bn2, [j], h, w, "down%d.conv2" % i, stride=2)
    h, w = H >> 4, W >> 4
    for i in range(1, 6):
        blk = getattr(model, "up_block%d" % i)
        srcs = [cur] if i == 1 else [cur, jumps[5 - i]]           # torch.cat((x, prev_feature_map)) (:66)
        t = cbr(blk.conv1, blk.bn1, srcs, h, w, "up%d.conv1" % i)
        if blk.up_sampling:
            (up,) = E.concat_members(pl, B, 2 * h, 2 * w, [t.C])
            pl.upsample2x_nearest(t, up, B, h, w, name="up%d.up" % i)
            h, w = 2 * h, 2 * w
            t = cbr(blk.conv2, blk.bn2, [up], h, w, "up%d.conv2" % i)
        cur = t
    opb = pl.buf(B, H, W, 8)
    l = E._cl(model.conv1, E._lay([cur]), act=ACT_NONE)
    pl.conv(l, [cur], Piece(opb, 0, 2), B, H, W, name="final")
    pl.t_target = pl.vec(B, H, W, dtype=torch.int64)
    pl.t_pc, pl.t_cond = pl.vec(B, 2), pl.vec(B, 4)
    pl.terms, pl.pred_c = pl.vec(8), pl.vec(B, 2)
    pl.op = pl.vec(B, 2, H, W)
    pl.mask = pl.vec(B, H, W, dtype=torch.int64)
    part = pl.vec(int(L.egne_deepvog_loss_workspace_floats(B, H, W)))
    pl.raw(L.egne_deepvog_loss_fwd, (opb.data_ptr(), 8, 0, pl.t_target.data_ptr(), pl.t_pc.data_ptr(), pl.t_cond.data_ptr(), B, H, W,
                                     part.data_ptr(), pl.terms.data_ptr(), pl.pred_c.data_ptr(), pl.op.data_ptr(), pl.mask.data_ptr()),
           "loss")
    if training:
        pl.gscale = pl.vec(1)

        def emit_loss(bw):
            go = pl.gbuf(opb)
            bw.raw(L.egne_deepvog_loss_bwd, (opb.data_ptr(), 8, 0, pl.t_target.data_ptr(), pl.t_pc.data_ptr(), pl.t_cond.data_ptr(), B, H, W,
                                             part.data_ptr(), pl.pred_c.data_ptr(), pl.gscale.data_ptr(), go.data_ptr(), go.shape[-1], 0), "loss.bwd")
        pl.tape.append(emit_loss)
        pl.build_backward()
    return pl
