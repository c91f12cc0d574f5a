"""ESF-Net (``DenseNet2D``) on the HIP path -- drop-in for the reference's ``models/RITnet_v2.py``.

Same constructor arguments, attributes (``selfCorr``, ``disentangle``, ``toggle``,
``setDatasetInfo``), ``state_dict`` keys and ``forward`` signature / return tuple
(models/RITnet_v2.py:204-354).  The modules below only hold parameters under the reference's
names; execution is a launch plan built by ``esf_engine`` (no torch ops on the compute path).
"""
import numpy as np
import torch
import torch.nn as nn

from ..esf_engine import build_forward_plan, dec_sizes, enc_sizes
from ..engine import require_cuda
from ..utils import Conv2dBlock, LinearBlock, convBlock, linStack, regressionModule


def getSizes(chz, growth, blks=4):
    """models/RITnet_v2.py:15-29 (same dict layout)."""
    e = enc_sizes(chz, growth, blks)
    sizes = {"enc": {"inter": np.array(e["inter"]), "ip": np.array(e["ip"]), "op": np.array(e["op"])},
             "dec": {}}
    sizes["dec"]["skip"] = sizes["enc"]["ip"][::-1] + sizes["enc"]["inter"][::-1]
    sizes["dec"]["ip"] = sizes["enc"]["op"][::-1]
    sizes["dec"]["op"] = np.append(sizes["enc"]["op"][::-1][1:], chz)
    return sizes


class Transition_down(nn.Module):
    def __init__(self, in_c, out_c, down_size, norm=None, actfunc=None):
        super().__init__()
        self.conv = nn.Conv2d(in_c, out_c, kernel_size=1, padding=0)
        self.down_size = down_size


class DenseNet2D_down_block(nn.Module):
    def __init__(self, in_c, inter_c, op_c, down_size, norm=None, actfunc=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_c, inter_c, kernel_size=3, padding=1)
        self.conv21 = nn.Conv2d(in_c + inter_c, inter_c, kernel_size=1, padding=0)
        self.conv22 = nn.Conv2d(inter_c, inter_c, kernel_size=3, padding=1)
        self.conv31 = nn.Conv2d(in_c + 2 * inter_c, inter_c, kernel_size=1, padding=0)
        self.conv32 = nn.Conv2d(inter_c, inter_c, kernel_size=3, padding=1)
        self.TD = Transition_down(inter_c + in_c, op_c, down_size, norm, actfunc)


class DenseNet2D_up_block(nn.Module):
    def __init__(self, skip_c, in_c, out_c, up_stride, actfunc=None):
        super().__init__()
        self.conv11 = nn.Conv2d(skip_c + in_c, out_c, kernel_size=1, padding=0)
        self.conv12 = nn.Conv2d(out_c, out_c, kernel_size=3, padding=1)
        self.conv21 = nn.Conv2d(skip_c + in_c + out_c, out_c, kernel_size=1, padding=0)
        self.conv22 = nn.Conv2d(out_c, out_c, kernel_size=3, padding=1)
        self.up_stride = up_stride


class StyleEncoder(nn.Module):
    """models/RITnet_v2.py:91-107."""

    def __init__(self, n_downsample, input_dim, dim, style_dim, norm, activ, pad_type):
        super().__init__()
        m = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type=pad_type)]
        for _ in range(2):
            m += [Conv2dBlock(dim, 2 * dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type)]
            dim *= 2
        for _ in range(n_downsample - 2):
            m += [Conv2dBlock(dim, dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type)]
        m += [nn.AdaptiveAvgPool2d(1)]
        m += [nn.Conv2d(dim, style_dim, 1, 1, 0)]
        self.model = nn.Sequential(*m)
        self.output_dim = dim


class MLP(nn.Module):
    """models/RITnet_v2.py:110-121."""

    def __init__(self, input_dim, output_dim, dim, n_blk, norm="none", activ="relu"):
        super().__init__()
        m = [LinearBlock(input_dim, dim, norm=norm, activation=activ)]
        for _ in range(n_blk - 2):
            m += [LinearBlock(dim, dim, norm=norm, activation=activ)]
        m += [LinearBlock(dim, output_dim, norm="none", activation="none")]
        self.model = nn.Sequential(*m)


class DenseNet_encoder(nn.Module):
    def __init__(self, in_c=1, chz=32, actfunc=None, growth=1.5, norm=None):
        super().__init__()
        s = enc_sizes(chz, growth)
        self.head = convBlock(in_c=in_c, inter_c=chz, out_c=chz, actfunc=actfunc)
        for i in range(4):
            setattr(self, "down_block%d" % (i + 1),
                    DenseNet2D_down_block(s["ip"][i], s["inter"][i], s["op"][i], 2, norm, actfunc))
        self.bottleneck = DenseNet2D_down_block(s["op"][3], s["inter"][3], s["op"][3], 0, norm, actfunc)


class DenseNet_decoder(nn.Module):
    def __init__(self, setting, chz, out_c, growth, actfunc=None, norm=None, variant="v2"):
        super().__init__()
        d = dec_sizes(chz, growth, setting["add_edge"] == 1, variant)
        for k in range(4):
            setattr(self, "up_block%d" % (4 - k), DenseNet2D_up_block(d["skip"][k], d["ip"][k], d["op"][k], 2, actfunc))
        self.final = convBlock(chz, chz, out_c, actfunc)


class _ESFFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward replays the launch plan, backward replays the
    reversed tape.  Parameter gradients are written straight into ``p.grad`` (views of one flat
    arena, ready for a single RCCL all-reduce), so nothing is returned for them."""

    @staticmethod
    def forward(ctx, model, pl, dummy):
        ctx.set_materialize_grads(False)
        ctx.model, ctx.pl = model, pl
        pl.run(model._events)
        return pl.op.clone(), pl.elPred.clone(), pl.latent.to(torch.float32, copy=True), pl.terms[0:1].clone(), pl.elOut.clone()

    @staticmethod
    def backward(ctx, g_op, g_elPred, g_latent, g_loss, g_elOut):
        """``loss.backward()`` alone (train.py:285-286) is the fast path.  A caller that adds its own terms on ``op`` / ``elPred`` /
        ``latent`` / ``elOut`` (all four carry grad in the reference, models/RITnet_v2.py:334-354) hands their gradients in here: the
        loss head's backward kernel adds them to its own (egne_loss_desc.g_op_nchw / g_pred_c / g_elOut_up), the latent's joins the
        bottleneck gradient in front of its spatial-mean backward.  No synchronisation either way."""
        model, pl = ctx.model, ctx.pl
        if g_loss is None and all(g is None for g in (g_op, g_elPred, g_latent, g_elOut)):
            return None, None, None
        model._ensure_grad_arena()
        pl.zero_grads_join()
        if g_loss is None:
            pl.gscale.zero_()
        else:
            pl.gscale.copy_(g_loss.reshape(1))
        ld, keep = pl.loss_desc, []
        B = pl.elOut.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()  # noqa: E731
        if g_op is not None:
            keep.append(f32(g_op))
            ld.g_op_nchw = keep[-1].data_ptr()
        if g_elPred is not None or g_elOut is not None:
            up = torch.zeros(B, 10, device=pl.elOut.device)
            if g_elOut is not None:
                up += f32(g_elOut)
            if g_elPred is not None:
                ge = f32(g_elPred)
                up[:, 2:5] += ge[:, 2:5]
                up[:, 7:10] += ge[:, 7:10]
                # no mask in the batch: the iris centre of elPred is a copy of elOut[:, 5:7] (RITnet_v2.py:404), else the soft-argmax
                # of the logits; out_terms[5] = number of samples with a mask, read on the device
                has_mask = (pl.terms[5] > 0).to(torch.float32)
                up[:, 5:7] += (1.0 - has_mask) * ge[:, 0:2]
                keep.append(torch.cat((ge[:, 0:2], ge[:, 5:7]), 1).contiguous())
                ld.g_pred_c = keep[-1].data_ptr()
            keep.append(up)
            ld.g_elOut_up = up.data_ptr()
        pl._g_latent_up = f32(g_latent) if g_latent is not None else None
        gc = getattr(model, "grad_comm", None)        # parallel.overlap_grads: the early bucket of the gradient all-reduce
        pl.bw.tail_hook = gc.tail_ready if gc is not None else None
        try:
            pl.bw.run(model._events)
        finally:
            ld.g_op_nchw = ld.g_pred_c = ld.g_elOut_up = None
            pl._g_latent_up = None
            pl.bw.tail_hook = None
        pl._upstream_keep = keep          # alive until the next backward (the launches are asynchronous)
        pl.zero_grads_ahead()
        return None, None, None


class DenseNet2D(nn.Module):
    variant = "v2"

    def __init__(self, setting, chz=32, growth=1.2, actfunc=None, norm=None, selfCorr=False, disentangle=False):
        super().__init__()
        self.sizes = getSizes(chz, growth)
        self.chz, self.growth = chz, growth
        self.toggle = True
        self.selfCorr = selfCorr
        self.disentangle = disentangle
        self.disentangle_alpha = 2
        self.setting = setting
        input_channels = 2 if (self.variant == "v2" and setting["input_concat"] == 1) else 1
        self.enc = DenseNet_encoder(in_c=input_channels, chz=chz, actfunc=actfunc, growth=growth, norm=norm)
        self.dec = DenseNet_decoder(setting, chz=chz, out_c=3, actfunc=actfunc, growth=growth, norm=norm,
                                    variant=self.variant)
        fc_enc = int(self.sizes["enc"]["op"][-1])
        feature_channels = setting["feature_channels"] if chz == 32 else fc_enc
        if self.variant == "concat" or setting["add_edge"] == 1:
            feature_channels *= 2
            # reference: assert feature_channels == 306 (RITnet_v2.py:227-230), i.e. 2 * enc.op[-1]
            assert feature_channels == 2 * fc_enc, "feature_channels must equal the encoder width"
        if self.variant == "v2" and setting["add_seg"] == 1:
            style_dim = setting["style_dim"]
            self.seg_encoder = StyleEncoder(4, 3, 64, style_dim, norm="none", activ="relu", pad_type="reflect")
            self.mlp = MLP(style_dim, feature_channels * 2, 256, 3, norm="none", activ="relu")
        self.elReg = regressionModule(feature_channels)
        self._initialize_weights()
        self._plans = {}
        self._events = None  # bench.py: list collecting per-launch HIP events
        self.storage_dtype = torch.float32   # activation storage of TRAINING plans: fp32, or bf16 after .to(torch.bfloat16 / float16)

    def setDatasetInfo(self, numSets=2):
        """models/RITnet_v2.py:240-249."""
        self.numSets = numSets
        self.dsIdentify_lin = linStack(num_layers=2, in_dim=int(self.sizes["enc"]["op"][-1]), hidden_dim=64,
                                       out_dim=numSets, bias=True, actBool=False, dp=0.0)

    def _initialize_weights(self):
        """models/RITnet_v2.py:356-369."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, np.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    # ------------------------------------------------------------------------------------------
    def _ensure_grad_arena(self):
        """All parameter gradients live in ONE flat fp32 buffer (``self._grad_flat``); ``p.grad`` are views.
        Re-attaches views dropped by ``optimizer.zero_grad(set_to_none=True)`` (zeroing them first)."""
        params = [p for p in self.parameters()]
        dev = params[0].device
        sig = tuple((id(p), p.numel()) for p in params) + (dev,)
        if getattr(self, "_grad_sig", None) != sig:
            self._grad_flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
            self._grad_views, o = [], 0
            for p in params:
                self._grad_views.append(self._grad_flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
            self._grad_sig = sig
            self._plans = {k: v for k, v in self._plans.items() if not k[4]}  # training plans hold grad pointers
        stale = [i for i, (p, v) in enumerate(zip(params, self._grad_views)) if p.grad is None or p.grad.data_ptr() != v.data_ptr()]
        if len(stale) == len(params):
            self._grad_flat.zero_()           # optimizer.zero_grad() dropped every view: ONE fill instead of one per parameter
            for p, v in zip(params, self._grad_views):
                p.grad = v
        else:
            for i in stale:
                self._grad_views[i].zero_()
                params[i].grad = self._grad_views[i]
        return self._grad_flat

    def to(self, *args, **kwargs):
        """``model.to(prec)`` of the entry scripts (train.py:206, test.py:297; ``--prec`` args.py:17-28).  The reference casts the
        whole module; here a half-precision ``prec`` (torch.float16 of ``--prec 16``, or torch.bfloat16) selects **bf16 activation
        storage for the training plans** (BASELINE.json configs[2..4]): activations and activation gradients live in HBM as bf16,
        every product accumulates in fp32, and the parameters stay fp32 master weights (so do the gradient arena, the optimiser
        state and the checkpoints).  gfx950 has bf16 MFMAs with fp32's exponent range; an fp16 module would need loss scaling the
        reference's loop does not have.  Inference plans keep fp32 tensors either way.  float32 switches back."""
        dtype = kwargs.get("dtype")
        rest = []
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                rest.append(a)
        kwargs.pop("dtype", None)
        if dtype is not None:
            if dtype in (torch.float16, torch.bfloat16):
                self.storage_dtype = torch.bfloat16
            elif dtype in (torch.float32, torch.float64):
                self.storage_dtype = torch.float32
            else:
                raise RuntimeError("DenseNet2D.to(%s): floating point precisions only" % dtype)
        return super().to(*rest, **kwargs) if (rest or kwargs) else self

    def _plan(self, B, H, W, dev):
        st = self.storage_dtype if self.training else torch.float32
        key = (B, H, W, dev, bool(self.training), bool(self.disentangle), bool(self.toggle), st)
        if key not in self._plans:
            self._plans[key] = self._build_plan(B, H, W, dev, bool(self.training), st)
        return self._plans[key]

    def _build_plan(self, B, H, W, dev, training, dtype):
        return build_forward_plan(self, B, H, W, dev, training, dtype=dtype)

    def forward(self, x, x_edge, target, pupil_center, elNorm, spatWts, distMap, cond, ID, alpha):
        """models/RITnet_v2.py:261-354.  Returns (op, elPred, latent, loss[1], elOut)."""
        if self.variant == "v2":
            assert (self.setting["input_concat"] + self.setting["add_edge"] < 2), "edge can use only 1 time!"
        elif self.variant == "concat":
            assert self.setting["add_edge"] == 1
        require_cuda(x, "x")
        want_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if want_grad:
            self._ensure_grad_arena()
        if self.selfCorr:
            raise NotImplementedError("selfCorr is disabled in the reference pipeline (--selfCorr 0, args.py:41)")
        B, _, H, W = x.shape
        pl = self._plan(B, H, W, x.device)
        self._last_plan = pl
        pl.in_img.copy_(x)
        pl.in_edge.copy_(x_edge)
        # the loss head reads the caller's ground-truth tensors in place when they already have the layout it wants
        # (contiguous, on this device, int64 / float32): its descriptor is re-read at every launch, so only pointers change
        ld = pl.loss_desc
        keep = []
        for field, buf, src, dt in (("target", pl.t_target, target, torch.int64), ("spatWts", pl.t_spat, spatWts, torch.float32),
                                    ("distMap", pl.t_dist, distMap, torch.float32), ("cond", pl.t_cond, cond, torch.float32),
                                    ("pupil_center", pl.t_pc, pupil_center, torch.float32), ("elNorm", pl.t_eln, elNorm, torch.float32)):
            if (torch.is_tensor(src) and src.dtype == dt and src.device == buf.device and src.is_contiguous()
                    and tuple(src.shape) == tuple(buf.shape) and not src.requires_grad):
                setattr(ld, field, src.data_ptr())
                keep.append(src)
            else:
                buf.copy_(src)
                setattr(ld, field, buf.data_ptr())
        pl._inputs = keep          # alive until the next forward replaces them (the launches are asynchronous)
        pl.loss_desc.alpha = float(alpha)
        if self.disentangle and torch.is_tensor(ID):
            pl.t_id.copy_(ID.to(torch.long))
        if want_grad:
            if getattr(self, "_dummy", None) is None or self._dummy.device != x.device:
                self._dummy = torch.zeros(1, device=x.device, requires_grad=True)
            return _ESFFunction.apply(self, pl, self._dummy)
        pl.run(self._events)
        loss = pl.terms[0:1].clone()
        return pl.op.clone(), pl.elPred.clone(), pl.latent.to(torch.float32, copy=True), loss, pl.elOut.clone()

    def overflowed(self):
        """True if the LAST inference call produced non-finite values inside a split-f16 convolution (engine.Plan.overflowed: a batch
        whose activations exceed 32x those of the batch the f16 pre-scales were calibrated on): its outputs are invalid.
        Synchronises; the plan re-calibrates on the next call, so the answer to True is to run the batch again.  Training plans take
        their scales on the device (or have none: bf16 storage) and never report."""
        last = getattr(self, "_last_plan", None)
        return last is not None and last.overflowed()

    def loss_flags(self):
        """Device scalar: the number of samples of the last forward whose ground-truth mask lacks TWO classes.  The reference's
        wCE dies there (``rmIdx.item()`` on a two-element tensor, loss.py:132); a kernel cannot raise, so the loss head counts
        such samples (out_terms[6]) and the entry scripts call ``raise_on_loss_flags`` where they synchronise anyway."""
        last = getattr(self, "_last_plan", None)
        if last is None:
            raise RuntimeError("loss_flags(): no forward pass has run yet")
        return last.terms[6:7].clone()

    @staticmethod
    def raise_on_loss_flags(flags):
        """``flags``: what loss_flags() returned for a batch.  Synchronises (``.item()``)."""
        n = int(flags.item())
        if n:
            raise RuntimeError("%d sample(s) of the batch have a segmentation mask with two absent classes: the weighted "
                               "cross-entropy of loss.py:127-135 supports at most one (the reference raises here too)" % n)

    def predictions(self):
        """Argmax mask [B,H,W] int64 of the last forward (device-side get_predictions, utils.py:65-81).
        A copy: the plan buffer it comes from is overwritten by the next forward of the same shape."""
        last = getattr(self, "_last_plan", None)
        if last is None:
            raise RuntimeError("predictions(): no forward pass has run yet")
        return last.mask.clone()
