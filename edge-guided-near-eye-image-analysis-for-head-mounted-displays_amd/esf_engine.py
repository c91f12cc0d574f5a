"""Launch-plan builder for ESF-Net (``DenseNet2D``), shared by models/RITnet_v2.py and
models/RITnet_concat.py.

Data flow follows models/RITnet_v2.py:261-354 but is laid out for the GPU:

* the shared encoder runs ONCE on a 2B batch (image frames then edge maps; the weights are the
  same and InstanceNorm is per sample; in training mode the two BatchNorm passes still use
  their own batch statistics, RITnet_v2.py:281,285);
* every torch.cat of the reference is a channel slice of a shared NHWC buffer
  (``[out | x | x1 | x22]`` per down block, ``[up(x) | x1]`` per up block); the 1x1 convs read
  up to 6 slices directly, the 3x3 convs write into their slice;
* InstanceNorm is never materialised: a statistics kernel produces per-(n,c) scale/shift that
  the consuming conv applies while loading (Transition_down's LeakyReLU included);
* eval-mode BatchNorm is folded into the producing conv's epilogue.
"""
import ctypes as C

import torch

from . import _lib
from . import engine as _eng
from .engine import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvLayer, Piece, PlanarPiece, Plan, VersionGuard, pad8, pad32


import os
import types

PLANAR_IN = os.environ.get("EGNE_PLANAR_IN", "1") != "0"   # one-channel inputs read in place by the fused head (no NHWC staging)
FOLD_UP = os.environ.get("EGNE_FOLD_UP", "1") != "0"     # up blocks: 1x1 of the up-sampled operand at half resolution
# inference plans: the ellipse regression head (eight latency-bound launches on the 15x20 bottleneck) on the plan's second stream
# next to the decoder, joined in front of the loss head
ELREG_SIDE = os.environ.get("EGNE_ELREG_SIDE", "1") != "0"
FOLD_UP_TRAIN = os.environ.get("EGNE_FOLD_UP_TRAIN", "1") != "0"       # ... and in bf16-storage TRAINING plans (backward: up2x^T of the 1x1's output gradient)
FOLD_UP_STREAM = os.environ.get("EGNE_FOLD_UP_STREAM", "1") != "0"     # ... also where the pair is not fused (streaming 1x1 with the addend in its epilogue)


def enc_sizes(chz, growth=1.2, blks=4):
    """models/RITnet_v2.py:15-29 getSizes (encoder part)."""
    inter = [chz * (i + 1) for i in range(blks)]
    op = [int(growth * chz * (i + 1)) for i in range(blks)]
    ip = [chz] + op[:-1]
    return dict(inter=inter, op=op, ip=ip)


def dec_sizes(chz, growth, add_edge, variant):
    """Decoder widths: RITnet_v2.py:177-190 / RITnet_concat.py:160-169, generalised in chz
    (exact at chz=32; other widths have no reference)."""
    e = enc_sizes(chz, growth)
    skip = [e["ip"][::-1][i] + e["inter"][::-1][i] for i in range(4)]
    fc = e["op"][-1]
    plain_ip = e["op"][::-1]
    plain_op = e["op"][::-1][1:] + [chz]
    if variant == "concat":
        return dict(ip=[2 * fc] + plain_op[:-1], op=plain_op, skip=[2 * s for s in skip])
    if add_edge:
        d = [int(round(chz * f)) for f in (5.625, 3.125, 1.9375)]
        return dict(ip=[2 * fc] + d, op=d + [chz], skip=skip)
    return dict(ip=plain_ip, op=plain_op, skip=skip)


TD_POOL_FIRST_TRAIN = os.environ.get("EGNE_TD_POOL_FIRST_TRAIN", "1") != "0"


def _cl(conv, layout, pad=(0, 0), act=ACT_NONE, **kw):
    l = ConvLayer([conv.weight], [conv.bias] if conv.bias is not None else None, layout, pad=pad, act=act, **kw)
    from . import engine
    ok = engine.ESF_SPLIT and (_cl.eval_plan or engine.TRAIN_SPLIT)
    l.split = ok and (l.kh > 1 or l.kw > 1)
    l.split1 = ok and l.kh == 1 and l.kw == 1
    return l


_cl.eval_plan = False


def concat_members(pl, nb, h, w, chans):
    """Pieces for the members of a would-be torch.cat.  fp32 plans: channel slices of ONE buffer (a 32-channel slice is 128
    contiguous bytes per pixel: a whole cache line).  bf16 plans: a buffer per member -- the memory system fetches 128-byte
    lines, so a 64-byte slice of a wider pixel row streams at half rate (measured: 2.55 vs 5.04 TB/s, scratch/bf16_stride.py),
    while a tensor of its own is contiguous from pixel to pixel."""
    if pl.bf16:
        return [Piece(pl.buf(nb, h, w, pad8(c)), 0, c) for c in chans]
    b, out, off = pl.buf(nb, h, w, sum(pad8(c) for c in chans)), [], 0
    for c in chans:
        out.append(Piece(b, off, c))
        off += pad8(c)
    return out


def _lay(pieces):
    return [(p.C, p.Cp) for p in pieces]


class _BNFold:
    """Eval-mode BatchNorm2d folded to per-channel (scale, shift) after the activation."""

    def __init__(self, bn, CoutP, dev):
        self.bn = bn
        self.scale = torch.zeros(CoutP, device=dev)
        self.shift = torch.zeros(CoutP, device=dev)
        self.guard = VersionGuard([bn.weight, bn.bias, bn.running_mean, bn.running_var], self._fold)

    def _fold(self):
        bn, c = self.bn, self.bn.num_features
        s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        self.scale[:c].copy_(s)
        self.shift[:c].copy_(bn.bias.detach() - bn.running_mean * s)


def _train_bn(pl, bn, pre, dst, n0, B, HW, name, producer=None, hw=None, partials=None):
    """Training-mode BatchNorm2d over samples [n0, n0+B) (utils.py:1049): batch statistics of ``pre``
    (the activated conv output, kept for backward) -> dst = (pre-mean)*rstd*gamma + beta; running
    statistics updated as torch does (momentum 0.1, unbiased variance).  ``producer``: the convolution whose activated output
    ``pre`` is -- bf16 plans then mask its output gradient and take its bias sums inside the BatchNorm's backward
    (egne_bn_act_bwd: no pass of its own for either)."""
    p, q = pre.samples(n0), dst.samples(n0)
    if partials is not None and partials[2] == pre.Cp:
        # batch statistics from the partial sums the producing convolution left (engine.Plan._conv_bf16, `_want_partials`): the chunks of
        # samples [n0, n0 + B) as ONE sample of B * nchunk chunks -- no pass over `pre`
        ws_, nch, Cs_ = partials
        rstd, nshift, mean, var = pl.vec(1, pre.Cp), pl.vec(1, pre.Cp), pl.vec(1, pre.Cp), pl.vec(1, pre.Cp)
        pl._add(pl.L.egne_norm_stats_finish_moments, (ws_.data_ptr() + 16 * n0 * nch * Cs_, Cs_, 1, B * nch, B * HW, bn.eps, rstd.data_ptr(), nshift.data_ptr(),
                                                      mean.data_ptr(), var.data_ptr()), name + ".stats", kind="norm_stats")
    else:
        rstd, nshift, mean, var = pl.norm_stats(p, B, HW, per_sample=False, eps=bn.eps, want_moments=True, name=name + ".stats")
    sc, sh, gpad = pl.vec(pre.Cp), pl.vec(pre.Cp), pl.vec(pre.Cp)
    c = bn.num_features
    n = B * HW

    def finish(rstd=rstd, nshift=nshift, mean=mean, var=var, sc=sc, sh=sh):
        # tiny [C]-sized bookkeeping on the stream (no sync): affine coefficients + running stats
        with torch.no_grad():
            g = bn.weight.detach()
            gpad[:c].copy_(g)
            sc[:c].copy_(rstd[0, :c] * g)
            sh[:c].copy_(nshift[0, :c] * g + bn.bias.detach())
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(mean[0, :c], alpha=m)
            bn.running_var.mul_(1 - m).add_(var[0, :c] * (n / max(n - 1, 1)), alpha=m)
            bn.num_batches_tracked.add_(1)
    pl.raw(_PyCall(finish), (), name + ".coef")
    pl.raw(pl.L.egne_affine, (p.ptr, p.stride, p.off, q.ptr, q.stride, q.off, pre.Cp, n, sc.data_ptr(), sh.data_ptr()),
           name + ".apply")
    if pl.train:
        def emit(bw):
            L = pl.L
            pl.flush_deferred_norm(bw, Piece(dst.buf, dst.off, dst.C, dst.Cp, 0), name)      # (block 0's input: its readers' InstanceNorm backward first)
            gq, gpre = pl.gp(q, B), pl.gp(p, B)
            bias = producer.biases[0] if (producer is not None and producer.biases is not None) else None
            if (_eng.BN_ACT_FUSE and pl.bf16 and producer is not None and hw is not None and producer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
                    and id(producer) not in pl._pair_links and pre.Cp % 8 == 0 and pre.off % 8 == 0 and q.Cp == pre.Cp):
                # gz of the producer = act'(pre) * BatchNorm-backward(gy): the only source of pre's gradient, so it is stored
                sums = bw.vec(pre.Cp * 2)
                wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, HW, pre.Cp, 1)) + 7) // 8, dtype=torch.float64)
                wsb = bw.vec((int(L.egne_act_bwd_bias_workspace_bytes(B * HW, pre.Cp)) + 7) // 8, dtype=torch.float64)
                bw.raw(L.egne_bn_act_bwd, (p.ptr, p.stride, p.off, producer.act, rstd.data_ptr(), nshift.data_ptr(), gpad.data_ptr(),
                                           gq.ptr, gq.stride, gq.off, pre.Cp, B, hw[0], hw[1], gpre.ptr, gpre.stride, gpre.off,
                                           sums.data_ptr(), wsn.data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), c,
                                           bias.grad.data_ptr() if bias is not None else None, producer.Cout, wsb.data_ptr()), name + ".bn_act_bwd")
                if bias is not None:
                    seen = bw.__dict__.setdefault("_bias_writers", {})
                    assert seen.setdefault(id(bias), "main") == "main", name
                pl.mark_stored(p, B)
                pl._premasked.setdefault((id(pre.buf), pre.off), set()).update(range(n0, n0 + B))
                return
            sums = bw.vec(pre.Cp * 2)
            ws = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, HW, pre.Cp, 0)) + 7) // 8, dtype=torch.float64)
            bw.raw(L.egne_norm_bwd, (p.ptr, p.stride, p.off, rstd.data_ptr(), nshift.data_ptr(), gpad.data_ptr(),
                                     gq.ptr, gq.stride, gq.off, 0, pre.Cp, B, HW, 0, gpre.ptr, gpre.stride, gpre.off,
                                     sums.data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), c, ws.data_ptr()),
                   name + ".bwd")
        pl.tape.append(emit)


def latent_upstream(pl, bw, gl, B, fcp, fc):
    """In front of the latent's spatial-mean backward: a caller's gradient w.r.t. the RETURNED latent (models/RITnet_v2.py:282,354;
    ``_ESFFunction.backward`` leaves it in ``pl._g_latent_up``) joins what the dataset-confusion head left in the twin ``gl``.
    Nothing happens on the loss-only path."""
    pl._g_latent_up = None

    def add_upstream():
        if pl._g_latent_up is not None:
            gl.view(B, fcp)[:, :fc].add_(pl._g_latent_up.to(gl.dtype))
    bw.raw(_PyCall(add_upstream), (), "latent.upstream")


class _PyCall:
    """Adapter so that a python callable can sit in a plan's launch list (main stream only: it queues torch work on the current one)."""
    python = True

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        self.fn()
        return 0


def build_forward_plan(model, B, H, W, dev, training, dtype=torch.float32):
    """``dtype``: storage of the plan's activation (and activation-gradient) buffers -- torch.bfloat16 for training plans of a
    model switched to bf16 storage (``model.to(torch.bfloat16)`` / ``--prec 16``), fp32 otherwise (engine.Plan)."""
    st = model.setting
    variant = model.variant
    chz, growth = model.chz, model.growth
    add_edge = variant == "concat" or st["add_edge"] == 1
    in_c = 2 if (variant == "v2" and st["input_concat"] == 1) else 1
    only_edge = variant == "v2" and st["only_edge"] == 1
    NB = 2 * B if add_edge else B
    es = enc_sizes(chz, growth)
    ds = dec_sizes(chz, growth, add_edge, variant)
    fc = es["op"][-1]
    pl = Plan(dev, train=training, dtype=dtype)
    L = pl.L
    esz = pl.esz
    pl.dbg = {}
    _cl.eval_plan = not training

    # ---- inputs (persistent; forward() copies the caller's tensors in) -----------------------------
    # Inference with one input channel: the image and edge batches sit back to back in ONE [NB][H][W] tensor that the fused
    # convBlock head reads in place (NCHW with C = 1 is NHWC with a pixel pitch of one float) -- no layout kernels, no 8-channel
    # padded staging copy (2 x 315 MB written and read back per step at B = 64).
    planar_in = (PLANAR_IN and not training and in_c == 1 and _eng.ESF_SPLIT and _eng.FUSE_1X1 and _eng.FUSE_C4 and _eng.F16X3_ENABLED
                 and W >= _eng.FUSE_1X1_MIN_W and chz == 32)
    if planar_in:
        pin = pl.vec(NB, 1, H, W)
        first = pin[:B]
        pl.in_img, pl.in_edge = (pl.vec(B, 1, H, W), first) if only_edge else (first, pin[B:] if add_edge else pl.vec(B, 1, H, W))
        xin = None
    else:
        pl.in_img = pl.vec(B, 1, H, W)
        pl.in_edge = pl.vec(B, 1, H, W)
        xin = pl.buf(NB, H, W, 8)
        first = pl.in_edge if only_edge else pl.in_img
        pl.raw(L.egne_nchw_to_nhwc, (first.data_ptr(), B, 1, H, W, xin.data_ptr(), 8, 0, 8), "in.img")
        if in_c == 2:
            pl.raw(L.egne_nchw_to_nhwc, (pl.in_edge.data_ptr(), B, 1, H, W, xin.data_ptr(), 8, 1, 1), "in.edge_ch")
        if add_edge:
            pl.raw(L.egne_nchw_to_nhwc, (pl.in_edge.data_ptr(), B, 1, H, W, xin.data_ptr() + esz * B * H * W * 8, 8, 0, 8),
                   "in.edge")

    # ---- encoder on NB samples -------------------------------------------------------------------
    enc = model.enc
    blocks = [enc.down_block1, enc.down_block2, enc.down_block3, enc.down_block4, enc.bottleneck]
    ins = es["ip"] + [es["op"][3]]
    inters = es["inter"] + [es["inter"][3]]
    outs = es["op"] + [es["op"][3]]
    pools = [2, 2, 2, 2, 0]
    res = [(H >> i, W >> i) for i in range(5)]

    def slices(nb, h, w, chans):
        return concat_members(pl, nb, h, w, chans)

    def dbuf(i):
        h, w = res[i]
        o, x, x1, x22 = slices(NB, h, w, [inters[i], ins[i], inters[i], inters[i]])
        return dict(out=o, x=x, x1=x1, x22=x22)
    D = [dbuf(i) for i in range(5)]
    if training:
        # blocks 0-3: `out` is produced by conv32 and read normalised by the pooled Transition_down only; the input x of blocks 1-3 is
        # produced by the previous Transition_down's 1x1 and read normalised by conv1 and the pooled Transition_down: their InstanceNorm
        # backward rides on the producer's masking pass (engine.Plan.defer_norm_bwd).  Block 0's x comes out of the head's BatchNorm: its
        # deferred backward is one fused pass in front of the BatchNorm's; the bottleneck's Transition_down does not pool: separate passes.
        for i in range(4):
            D[i]["out"].norm_fuse = TD_POOL_FIRST_TRAIN and res[i][0] % 2 == 0 and res[i][1] % 2 == 0
            D[i]["x"].norm_fuse = D[i]["out"].norm_fuse       # (block 0: flushed by the head BatchNorm's backward, Plan.flush_deferred_norm)

    t0 = pl.buf(NB, H, W, pad8(chz))
    l1 = _cl(enc.head.conv1, [(in_c, 8)], pad=(1, 1), act=ACT_LEAKY)
    if planar_in:
        xin_p = PlanarPiece(pin)
    else:
        xin_p = Piece(xin, 0, in_c, 8)
        xin_p.nograd = True
    l = _cl(enc.head.conv2, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
    x_stats = None
    if not training:
        fold = _BNFold(enc.head.bn, l.CoutP, dev)
        pl.pre.append(fold.guard)
        l.post = (fold.scale, fold.shift)
        # conv2(leaky(conv1(x))) as one launch where the kernel allows it; InstanceNorm statistics of block 0's input from its epilogue
        pl.conv_pair(l1, [xin_p], l, D[0]["x"], NB, H, W, tmp=Piece(t0, 0, chz), name="enc.head", stats=True)
        x_stats = pl.last_stats
    else:
        pl.conv(l1, [xin_p], Piece(t0, 0, chz), NB, H, W, name="enc.head.conv1")
        pre = pl.buf(NB, H, W, pad8(chz))
        pl._want_partials = True
        pl.conv(l, [Piece(t0, 0, chz)], Piece(pre, 0, chz), NB, H, W, name="enc.head.conv2")
        head_partials, pl._want_partials = pl.last_partials, False
        pl.dbg["head_pre"] = pre
        _train_bn(pl, enc.head.bn, Piece(pre, 0, chz), D[0]["x"], 0, B, H * W, "enc.head.bn", producer=l, hw=(H, W), partials=head_partials)
        if add_edge:
            _train_bn(pl, enc.head.bn, Piece(pre, 0, chz), D[0]["x"], B, B, H * W, "enc.head.bn.edge", producer=l, hw=(H, W), partials=head_partials)

    bott = pl.buf(NB, res[4][0], res[4][1], pad8(fc))
    pl.dbg.update(D=D, bott=bott, t0=t0)
    for i, blk in enumerate(blocks):
        h, w = res[i]
        d = D[i]
        nm = "enc.b%d" % i
        if x_stats is not None:
            sc, sh = x_stats
            x_stats = None
        else:
            sc, sh, _, _ = pl.norm_stats(d["x"], NB, h * w, name=nm + ".in_x")
        l = _cl(blk.conv1, _lay([d["x"]]), pad=(1, 1), act=ACT_LEAKY)
        pl.conv(l, [d["x"].with_norm(sc, sh)], d["x1"], NB, h, w, name=nm + ".conv1")
        # conv22(conv21(cat(x, x1))) and conv32(conv31(cat(x, x1, x22))): each 1x1 feeds exactly one 3x3 (RITnet_v2.py:59-62)
        l1 = _cl(blk.conv21, _lay([d["x"], d["x1"]]))
        l2 = _cl(blk.conv22, [(inters[i], pad8(inters[i]))], pad=(1, 1), act=ACT_LEAKY)
        pl.conv_pair(l1, [d["x"], d["x1"]], l2, d["x22"], NB, h, w, name=nm + ".conv2")
        l1 = _cl(blk.conv31, _lay([d["x"], d["x1"], d["x22"]]))
        l2 = _cl(blk.conv32, [(inters[i], pad8(inters[i]))], pad=(1, 1), act=ACT_LEAKY)
        pl.conv_pair(l1, [d["x"], d["x1"], d["x22"]], l2, d["out"], NB, h, w, name=nm + ".conv3", stats=True)
        sc2, sh2 = pl.last_stats
        tdl = _cl(blk.TD.conv, _lay([d["out"], d["x"]]))
        tin = [d["out"].with_norm(sc2, sh2, ACT_LEAKY), d["x"].with_norm(sc, sh, ACT_LEAKY)]
        if pools[i] and not training and pl.td_pool_fusable(tdl, tin, D[i + 1]["x"]) and h % 2 == 0 and w % 2 == 0:
            # eval plans: the 2x2 average folded into the 1x1's operand load (linear ops commute): one launch, no pooled tensor
            pl.conv1x1_pooled(tdl, tin, D[i + 1]["x"], NB, h, w, name=nm + ".TD")
        elif pools[i] and (not training or (TD_POOL_FIRST_TRAIN and h % 2 == 0 and w % 2 == 0)):
            # pool first, then the 1x1 conv at quarter resolution (linear ops commute; avg_pool2d drops an odd last row / column
            # on both routes).  Training plans too: the 1x1, its weight and data gradients all run on a quarter of the pixels, and
            # the InstanceNorm backward takes a quarter of the pooled cell's gradient (egne_norm_pool2_bwd)
            q_out, q_x = slices(NB, h // 2, w // 2, [d["out"].C, d["x"].C])
            for src, dstp, (a_, b_) in ((d["out"], q_out, (sc2, sh2)), (d["x"], q_x, (sc, sh))):
                pl.raw(L.egne_norm_act_pool2, (src.ptr, src.stride, src.off, a_.data_ptr(), b_.data_ptr(), ACT_LEAKY,
                                               dstp.ptr, dstp.stride, dstp.off, NB, h, w, src.Cp), nm + ".TDpool")
                if training:
                    def emit_pool(bw, src=src, dstp=dstp, a_=a_, b_=b_, h=h, w=w, nm=nm):
                        if pl.norm_fusable(src):          # the producer of `src` takes it (egne_act_norm_bwd in its masking pass)
                            pl.defer_norm_bwd(src, a_, b_, NB, h, w, gq=pl.gp(dstp), act_q=ACT_LEAKY)
                            return
                        gq = pl.gp(dstp)
                        first = pl.first_touch(src.buf, src.off, src.Cp)       # the first writer of a gradient slice stores
                        gs_ = pl.gp(src)
                        if first:
                            pl.mark_stored(src, NB)
                        sums = bw.vec(NB * src.Cp * 2)
                        wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(NB, h * w, src.Cp, 1)) + 7) // 8, dtype=torch.float64)
                        bw.raw(L.egne_norm_pool2_bwd, (src.ptr, src.stride, src.off, a_.data_ptr(), b_.data_ptr(), gq.ptr, gq.stride,
                                                       gq.off, ACT_LEAKY, src.Cp, NB, h, w, gs_.ptr, gs_.stride, gs_.off,
                                                       0 if first else 1, sums.data_ptr(), wsn.data_ptr()), nm + ".TDpool.bwd")
                    pl.tape.append(emit_pool)
            pl.conv(tdl, [q_out, q_x], D[i + 1]["x"], NB, h // 2, w // 2, name=nm + ".TD")
        elif pools[i]:
            td = pl.buf(NB, h, w, pad8(outs[i]))
            pl.conv(tdl, tin, Piece(td, 0, outs[i]), NB, h, w, name=nm + ".TD")
            pl.avgpool2(Piece(td, 0, outs[i]), D[i + 1]["x"], NB, h, w, name=nm + ".pool")
        else:
            pl.conv(tdl, tin, Piece(bott, 0, fc), NB, h, w, name=nm + ".TD")

    # latent = mean over H*W of the image pass' bottleneck (RITnet_v2.py:282)
    hb, wb = res[4]
    fcp = pad8(fc)
    pl.latent_p = pl.buf(B, 1, 1, fcp)   # padded row stride so that it can feed a 1x1 conv directly
    pl.raw(L.egne_spatial_mean, (bott.data_ptr(), bott.shape[-1], 0, fcp, B, hb * wb, pl.latent_p.data_ptr()), "latent")
    pl.latent = pl.latent_p.view(B, fcp)[:, :fc]
    if training:
        def emit_latent(bw):
            gl, gb = pl.gbuf(pl.latent_p), pl.gbuf(bott)
            latent_upstream(pl, bw, gl, B, fcp, fc)
            bw.raw(L.egne_spatial_mean_bwd, (gl.data_ptr(), fcp, gb.data_ptr(), gb.shape[-1], 0, fcp, B, hb * wb), "latent.bwd")
        pl.tape.append(emit_latent)

    # ---- decoder on B samples ---------------------------------------------------------------------
    xb = [Piece(bott, 0, fc)] + ([Piece(bott, 0, fc, n0=B)] if add_edge else [])
    # (not at one or two frames: there the fork and join cost what the head's launches take -- 3.29 -> 3.38 ms per two-frame call)
    early_head = ELREG_SIDE and B >= 8 and not training and not (variant == "v2" and st["add_seg"] == 1)
    if early_head:       # (with AdaIN the head reads the bottleneck modulated by the decoder's own output: it stays behind it)
        pl.serial_timing = True          # run(events): per-launch timing on one stream (overlapping kernels stretch each other)
        pl.side_default = True
        regression_head(pl, model.elReg, xb, B, hb, wb, training)
        pl.side_default = False
    prev, ph, pw = xb, hb, wb
    dec = model.dec
    ups = [dec.up_block4, dec.up_block3, dec.up_block2, dec.up_block1]
    for k, ub in enumerate(ups):
        lvl = 3 - k                      # encoder level whose skip is used
        h, w = res[lvl]
        oc = ds["op"][k]
        nm = "dec.up%d" % (4 - k)
        skip = [D[lvl]["out"], D[lvl]["x"]]
        if variant == "concat":
            skip = skip + [D[lvl]["out"].samples(B), D[lvl]["x"].samples(B)]
        Cl = sum(p.C for p in prev)                  # channels of the up-sampled operand (first in the reference's torch.cat)
        if FOLD_UP and not training and variant != "concat" and D[lvl]["x1"].C == oc:
            # inference: the encoder is finished, its x1 slice of this level is dead -- the up block's x1 takes its place, so that
            # conv21 reads skip and x1 from ONE buffer (the fused kernel's uniform-buffer path)
            x1 = Piece(D[lvl]["x1"].buf, D[lvl]["x1"].off, oc)
        else:
            x1b = pl.buf(B, h, w, pad8(oc))
            x1 = Piece(x1b, 0, oc)
        y = pl.buf(B, h, w, pad8(oc))
        l12 = _cl(ub.conv12, [(oc, pad8(oc))], pad=(1, 1), act=ACT_LEAKY)
        l22 = _cl(ub.conv22, [(oc, pad8(oc))], pad=(1, 1), act=ACT_LEAKY)
        # Inference plans, narrow blocks: conv11(cat(up(x), skip)) = up(W_up x) + W_skip skip -- the 1x1 and the bilinear
        # interpolation commute, so W_up x is evaluated at HALF resolution (for conv11 and conv21 at once) and the fused
        # 1x1 -> 3x3 kernel adds its up-sampling on the fly: the up-sampled tensor never exists (RITnet_v2.py:80-88)
        l11s = ConvLayer([ub.conv11.weight[:, Cl:]], [ub.conv11.bias], _lay(skip))
        l21s = ConvLayer([ub.conv21.weight[:, Cl:]], [ub.conv21.bias], _lay(skip + [x1]))
        for l in (l11s, l21s):
            l.split1 = l12.split1 or l12.split
        fold_up = (FOLD_UP and not training and oc == 32 and sum((p.Cp + 15) // 16 for p in skip + [x1]) <= 8
                   and pl.pair_fusable(l11s, skip, l12, x1, h, w) and pl.pair_fusable(l21s, skip + [x1], l22, Piece(y, 0, oc), h, w))
        # wider blocks whose pair is not fused (62 channels at 120x160: more channel groups than the fused kernel stages): the same
        # identity on the STREAMING 1x1 kernel, which adds the up-sampled W_up x in its epilogue -- no up-sampled tensor, and both 1x1s
        # read the skip slices only (K = 104 / 168 instead of 204 / 268 channels)
        ocp = pad8(oc)
        # (only where the 1x1 over cat(up(x), skip) would run on the streaming kernel as well: against the LDS-staged GEMM of the wider
        #  blocks the per-pixel gathers of the addend cost more than the halved K saves -- up block 3: 947 -> 1010 us)
        full = [types.SimpleNamespace(C=p.C, Cp=pad8(p.C), scale=None) for p in prev] + skip
        l11f, l21f = _cl(ub.conv11, _lay(full)), _cl(ub.conv21, _lay(full + [x1]))
        fold_up_s = (FOLD_UP and FOLD_UP_STREAM and not training and not fold_up and variant != "concat"
                     and pl.stream1x1_ok(l11s, skip, B, h, w, up_add=True) and pl.stream1x1_ok(l21s, skip + [x1], B, h, w, up_add=True)
                     and pl.stream1x1_ok(l11f, full, B, h, w) and pl.stream1x1_ok(l21f, full + [x1], B, h, w))
        # bf16-storage training plans (round 5): the same identity -- P = [W11_up; W21_up] x at half resolution, the two 1x1s read the skip
        # slices (and x1) only and add up2x(P) in their epilogue (conv1x1_bf16.hip).  Backward: gP = up2x^T(gz) for either 1x1
        # (egne_upsample2x_bwd), then P's own 1x1 gives W_up's weight gradient and x's data gradient at a quarter of the pixels.  The
        # up-sampled operand (nine passes over a full-resolution tensor per block and step: written, read by two 1x1s and their two
        # weight gradients, its gradient written, accumulated and read back) never exists.  The weight slices are derived tensors with
        # gradient buffers of their own, added into conv11 / conv21's .grad behind the block's last weight gradient (on its stream).
        x1p_ = Piece(x1.buf, x1.off, oc)
        fold_train = (FOLD_UP and FOLD_UP_TRAIN and training and pl.bf16 and variant != "concat" and h == 2 * ph and w == 2 * pw
                      and ub.conv11.bias is not None and ub.conv21.bias is not None
                      and pl.bf16_stream1x1_ok(l11s, skip, Piece(y, 0, oc), B, h, w) and pl.bf16_stream1x1_ok(l21s, skip + [x1p_], Piece(y, 0, oc), B, h, w)
                      and B * ph * pw * 2 * ocp * 2 < 2 ** 31)
        if fold_train:
            Ks = sum(p.C for p in skip)
            wp = torch.zeros(2 * ocp, Cl, 1, 1, device=dev)
            w11 = torch.zeros(oc, Ks, 1, 1, device=dev)
            w21 = torch.zeros(oc, Ks + oc, 1, 1, device=dev)
            for t in (wp, w11, w21):
                t.grad = torch.zeros_like(t)

            def refresh_wt(wp=wp, w11=w11, w21=w21, ub=ub, Cl=Cl, oc=oc, ocp=ocp):
                with torch.no_grad():
                    wp[:oc].copy_(ub.conv11.weight.detach()[:, :Cl])
                    wp[ocp:ocp + oc].copy_(ub.conv21.weight.detach()[:, :Cl])
                    w11.copy_(ub.conv11.weight.detach()[:, Cl:])
                    w21.copy_(ub.conv21.weight.detach()[:, Cl:])
            pl.pre.append(VersionGuard([ub.conv11.weight, ub.conv21.weight], refresh_wt))
            refresh_wt()
            lpw = ConvLayer([wp], None, _lay(prev))
            l11t = ConvLayer([w11], [ub.conv11.bias], _lay(skip))
            l21t = ConvLayer([w21], [ub.conv21.bias], _lay(skip + [x1]))

            def emit_scatter(bw, wp=wp, w11=w11, w21=w21, ub=ub, Cl=Cl, oc=oc, ocp=ocp, nm=nm):
                def scatter():
                    with torch.no_grad():
                        g11, g21 = ub.conv11.weight.grad, ub.conv21.weight.grad
                        g11[:, :Cl].add_(wp.grad[:oc])
                        g21[:, :Cl].add_(wp.grad[ocp:ocp + oc])
                        g11[:, Cl:].add_(w11.grad)
                        g21[:, Cl:].add_(w21.grad)
                        for t in (wp, w11, w21):
                            t.grad.zero_()
                bw._add(_PyCall(scatter), (), nm + ".wgrad_scatter", kind="host", side=_eng.WGRAD_SIDE_STREAM)
            pl.tape.append(emit_scatter)          # (replayed in reverse: behind the backward of everything below)
            Pb = pl.buf(B, ph, pw, 2 * ocp)
            pl.conv(lpw, prev, Piece(Pb, 0, 2 * ocp), B, ph, pw, name=nm + ".up_w")
            P1, P2 = Piece(Pb, 0, oc, ocp), Piece(Pb, ocp, oc, ocp)
            pl.conv_pair(l11t, skip, l12, x1, B, h, w, name=nm + ".conv1", up_add=(P1, ph, pw))
            pl.conv_pair(l21t, skip + [x1], l22, Piece(y, 0, oc), B, h, w, name=nm + ".conv2", up_add=(P2, ph, pw))
        elif fold_up_s:
            wp = torch.zeros(2 * ocp, Cl, 1, 1, device=dev)

            def refresh_wps(wp=wp, ub=ub, Cl=Cl, oc=oc, ocp=ocp):
                wp.zero_()
                wp[:oc].copy_(ub.conv11.weight.detach()[:, :Cl])
                wp[ocp:ocp + oc].copy_(ub.conv21.weight.detach()[:, :Cl])
            pl.pre.append(VersionGuard([ub.conv11.weight, ub.conv21.weight], refresh_wps))
            refresh_wps()
            lpw = ConvLayer([wp], None, _lay(prev))
            lpw.split1 = l11s.split1
            Pb = pl.buf(B, ph, pw, 2 * ocp)
            pl.conv(lpw, prev, Piece(Pb, 0, 2 * ocp), B, ph, pw, name=nm + ".up_w")
            P1, P2 = Piece(Pb, 0, oc, ocp), Piece(Pb, ocp, oc, ocp)
            t1, t2 = Piece(pl.buf(B, h, w, ocp), 0, oc), Piece(pl.buf(B, h, w, ocp), 0, oc)
            pl.conv(l11s, skip, t1, B, h, w, name=nm + ".conv1.a", up_add=(P1, ph, pw))
            pl.conv(l12, [t1], x1, B, h, w, name=nm + ".conv1.b")
            pl.conv(l21s, skip + [x1], t2, B, h, w, name=nm + ".conv2.a", up_add=(P2, ph, pw))
            pl.conv(l22, [t2], Piece(y, 0, oc), B, h, w, name=nm + ".conv2.b")
        elif fold_up:
            wp = torch.zeros(2 * oc, Cl, 1, 1, device=dev)

            def refresh_wp(wp=wp, ub=ub, Cl=Cl, oc=oc):
                wp[:oc].copy_(ub.conv11.weight.detach()[:, :Cl])
                wp[oc:].copy_(ub.conv21.weight.detach()[:, :Cl])
            pl.pre.append(VersionGuard([ub.conv11.weight, ub.conv21.weight], refresh_wp))
            refresh_wp()
            lpw = ConvLayer([wp], None, _lay(prev))
            lpw.split1 = l11s.split1
            Pb = pl.buf(B, ph, pw, 2 * pad32(oc))
            pl.conv(lpw, prev, Piece(Pb, 0, 2 * oc), B, ph, pw, name=nm + ".up_w")
            P1, P2 = Piece(Pb, 0, oc, 32), Piece(Pb, 32, oc, 32)
            pl.conv_pair(l11s, skip, l12, x1, B, h, w, name=nm + ".conv1", up_add=(P1, ph, pw))
            pl.conv_pair(l21s, skip + [x1], l22, Piece(y, 0, oc), B, h, w, name=nm + ".conv2", up_add=(P2, ph, pw))
        else:
            up_pieces = slices(B, h, w, [p.C for p in prev])
            for p, q in zip(prev, up_pieces):
                pl.upsample2x(p, q, B, ph, pw, name=nm + ".up")
            cat = up_pieces + skip
            l1 = _cl(ub.conv11, _lay(cat))
            pl.conv_pair(l1, cat, l12, x1, B, h, w, name=nm + ".conv1")
            l1 = _cl(ub.conv21, _lay(cat + [x1]))
            pl.conv_pair(l1, cat + [x1], l22, Piece(y, 0, oc), B, h, w, name=nm + ".conv2")
        prev, ph, pw = [Piece(y, 0, oc)], h, w
        pl.dbg[nm] = prev[0]

    tf = pl.buf(B, H, W, pad8(chz))
    l = _cl(dec.final.conv1, _lay(prev), pad=(1, 1), act=ACT_LEAKY)
    pl.conv(l, prev, Piece(tf, 0, chz), B, H, W, name="dec.final.conv1")
    opb = pl.buf(B, H, W, 8)
    l = _cl(dec.final.conv2, [(chz, pad8(chz))], pad=(1, 1), act=ACT_LEAKY)
    if not training:
        fold = _BNFold(dec.final.bn, l.CoutP, dev)
        pl.pre.append(fold.guard)
        l.post = (fold.scale, fold.shift)
    if not training:
        pl.conv(l, [Piece(tf, 0, chz)], Piece(opb, 0, 3), B, H, W, name="dec.final.conv2")
    else:
        pre = pl.buf(B, H, W, 8)
        pl.conv(l, [Piece(tf, 0, chz)], Piece(pre, 0, 3), B, H, W, name="dec.final.conv2")
        _train_bn(pl, dec.final.bn, Piece(pre, 0, 3), Piece(opb, 0, 3), 0, B, H * W, "dec.final.bn", producer=l, hw=(H, W))

    if variant == "v2" and st["add_seg"] == 1:
        # ---- AdaIN fusion (RITnet_v2.py:289-308): softmax(op) -> StyleEncoder -> MLP -> modulate bottleneck ----
        from .engine import ACT_RELU
        sm = pl.buf(B, H, W, 8)
        pl.raw(L.egne_softmax3, (opb.data_ptr(), 8, 0, sm.data_ptr(), 8, 0, 8, B * H * W), "adain.softmax")
        se = model.seg_encoder.model
        cur, cc, ch, cw = Piece(sm, 0, 3), 3, H, W
        if st.get("seg_detach", 0):
            cur.nograd = True            # softmx(op.detach()), RITnet_v2.py:291-292
        elif training:
            def emit_softmax(bw):
                gs, go = pl.gbuf(sm), pl.gbuf(opb)
                bw.raw(L.egne_softmax3_bwd, (sm.data_ptr(), 8, 0, gs.data_ptr(), 8, 0, go.data_ptr(), 8, 0, B * H * W),
                       "adain.softmax.bwd")
            pl.tape.append(emit_softmax)
        for i in range(5):
            blk = se[i]
            k = blk.conv.kernel_size[0]
            l = ConvLayer([blk.conv.weight], [blk.conv.bias], [(cc, pad8(cc))], stride=blk.conv.stride[0],
                          pad=(blk.padding, blk.padding), act=ACT_RELU, pad_mode=1)
            oh, ow = l.out_hw(ch, cw)
            ob = pl.buf(B, oh, ow, pad8(l.Cout))
            pl.conv(l, [cur], Piece(ob, 0, l.Cout), B, ch, cw, name="adain.enc%d" % i)
            cur, cc, ch, cw = Piece(ob, 0, l.Cout), l.Cout, oh, ow
        gap = pl.buf(B, 1, 1, pad8(cc))
        pl.raw(L.egne_spatial_mean, (cur.ptr, cur.stride, cur.off, cur.Cp, B, ch * cw, gap.data_ptr()), "adain.gap")
        if training:
            def emit_gap(bw, cur=cur, n=ch * cw):
                gg, gc = pl.gbuf(gap), pl.gp(cur)
                bw.raw(L.egne_spatial_mean_bwd, (gg.data_ptr(), gg.shape[-1], gc.ptr, gc.stride, gc.off, cur.Cp, B, n),
                       "adain.gap.bwd")
            pl.tape.append(emit_gap)
        l = ConvLayer([se[6].weight], [se[6].bias], [(cc, pad8(cc))])
        sty = pl.buf(B, 1, 1, pad8(l.Cout))
        pl.conv(l, [Piece(gap, 0, cc)], Piece(sty, 0, l.Cout), B, 1, 1, name="adain.style")
        cur, cc = Piece(sty, 0, l.Cout), l.Cout
        mlp = model.mlp.model
        for i in range(len(mlp)):
            fcm = mlp[i].fc
            l = ConvLayer([fcm.weight], [fcm.bias], [(cc, pad8(cc))], kernel_hw=(1, 1),
                          act=ACT_RELU if mlp[i].activation_name == "relu" else ACT_NONE)
            ob = pl.buf(B, 1, 1, pad8(l.Cout))
            pl.conv(l, [cur], Piece(ob, 0, l.Cout), B, 1, 1, name="adain.mlp%d" % i)
            cur, cc = Piece(ob, 0, l.Cout), l.Cout
        nfc = fc * len(xb)
        assert cc == 2 * nfc, (cc, nfc)
        xa = pl.buf(B, hb, wb, pad8(fc) * len(xb))
        mod = []
        for j, pc in enumerate(xb):
            q = Piece(xa, j * pad8(fc), fc)
            # gamma = adain_params[:,0] (first nfc values of the MLP row), beta = adain_params[:,1] (next nfc)
            pl.raw(L.egne_adain, (pc.ptr, pc.stride, pc.off, fc, cur.ptr + esz * (j * fc), cur.ptr + esz * (nfc + j * fc),
                                  cur.stride, 0, q.ptr, q.stride, q.off, B, hb * wb, 1e-5), "adain.apply")
            if training:
                def emit_adain(bw, pc=pc, q=q, j=j, cur=cur):
                    gq, gx, gm = pl.gp(q), pl.gp(pc), pl.gp(cur)
                    bw.raw(L.egne_adain_bwd, (pc.ptr, pc.stride, pc.off, fc, cur.ptr + esz * (j * fc), cur.stride, 0,
                                              gq.ptr, gq.stride, gq.off, gx.ptr, gx.stride, gx.off,
                                              gm.ptr + esz * (j * fc), gm.ptr + esz * (nfc + j * fc), gm.stride, 0,
                                              B, hb * wb, 1e-5), "adain.apply.bwd")
                pl.tape.append(emit_adain)
            mod.append(q)
        xb = mod

    if early_head:
        pl.join_before.add(len(pl.calls))
        pl.raw(pl._elout_copy, (), "elOut.copy")
    else:
        regression_head(pl, model.elReg, xb, B, hb, wb, training)
    loss_head(pl, opb, B, H, W, dev, training)
    confusion_head(pl, model, fc, B, training, variant == "v2")
    if training:
        bw = pl.build_backward()
        # where the backward pass reaches the encoder: every gradient of the decoder, the AdaIN modules, the regression and the
        # dataset-identity heads is final (parallel.GradOverlap issues their share of the all-reduce from here)
        enc_calls = [i for i, (_, _, n) in enumerate(bw.calls) if n.startswith("enc.")]
        bw.tail_at = enc_calls[0] if enc_calls else None
    return pl


def regression_head(pl, rg, xb, B, hb, wb, training):
    """regressionModule (utils.py:983-1037) on the bottleneck pieces ``xb`` [B, hb, wb]: sets pl.elOut (+ pl.g_elOut)."""
    L = pl.L
    l = _cl(rg.c1, _lay(xb), act=ACT_LEAKY)
    h1, w1 = l.out_hw(hb, wb)
    r1 = pl.buf(B, h1, w1, 128)
    pl.conv(l, xb, Piece(r1, 0, 128), B, hb, wb, name="elReg.c1")
    r2 = pl.buf(B, h1 // 2, w1 // 2, 128)
    pl.avgpool2(Piece(r1, 0, 128), Piece(r2, 0, 128), B, h1, w1, name="elReg.pool")
    l = _cl(rg.c2, [(128, 128)], act=ACT_LEAKY)
    h3, w3 = l.out_hw(h1 // 2, w1 // 2)
    r3 = pl.buf(B, h3, w3, 128)
    pl.conv(l, [Piece(r2, 0, 128)], Piece(r3, 0, 128), B, h1 // 2, w1 // 2, name="elReg.c2")
    l = _cl(rg.c3, [(128, 128)], act=ACT_LEAKY)
    h4, w4 = l.out_hw(h3, w3)
    r4 = pl.buf(B, h4, w4, 32)
    pl.conv(l, [Piece(r3, 0, 128)], Piece(r4, 0, 32), B, h3, w3, name="elReg.c3")
    if 32 * h4 * w4 != rg.l1.weight.shape[1]:
        raise ValueError("regressionModule expects a %d-feature map, got 32x%dx%d (input must be 240x320)"
                         % (rg.l1.weight.shape[1], h4, w4))
    l = ConvLayer([rg.l1.weight], [rg.l1.bias], [(32, 32)], kernel_hw=(h4, w4))
    r5 = pl.buf(B, 1, 1, 256)
    pl.conv(l, [Piece(r4, 0, 32)], Piece(r5, 0, 256), B, h4, w4, name="elReg.l1")
    pl.raw(L.egne_selu_inplace, (r5.data_ptr(), B * 256), "elReg.selu")
    if training:
        pl.tape.append(lambda bw: bw.raw(L.egne_selu_bwd, (pl.gbuf(r5).data_ptr(), r5.data_ptr(), B * 256), "elReg.selu.bwd"))
    l = ConvLayer([rg.l2.weight], [rg.l2.bias], [(256, 256)], kernel_hw=(1, 1))
    r6 = pl.buf(B, 1, 1, 16)
    pl.conv(l, [Piece(r5, 0, 256)], Piece(r6, 0, 10, 16), B, 1, 1, name="elReg.l2")
    pl.raw(L.egne_ellipse_head_act, (r6.data_ptr(), B, 16), "elReg.act")
    pl.elOut = pl.vec(B, 10)
    copy_out = _PyCall(lambda: pl.elOut.copy_(r6.view(B, 16)[:, :10]))
    if pl.side_default:
        # on the second stream (inference plans): the torch copy is queued by the caller on the main stream, behind the join -- a
        # python call issued under another torch stream inside a hipGraph capture gave wrong replays (GraphedFrames)
        pl._elout_copy = copy_out
    else:
        pl.raw(copy_out, (), "elOut.copy")
    if training:
        pl.g_elOut = pl.vec(B, 10)
        pl.tape.append(lambda bw: bw.raw(L.egne_ellipse_head_act_bwd, (pl.gbuf(r6).data_ptr(), r6.data_ptr(), B, 16),
                                         "elReg.act.bwd"))
        pl.tape.append(lambda bw: bw.raw(_PyCall(lambda: pl.gbuf(r6).view(B, 16)[:, :10].copy_(pl.g_elOut)), (),
                                         "elOut.copy.bwd"))



def loss_head(pl, opb, B, H, W, dev, training):
    """get_allLoss (models/RITnet_v2.py:372-432 = models/RITnet_v1.py:322-372) on the logits buffer ``opb`` [B,H,W,8]: the loss
    kernel's descriptor, ground-truth staging tensors, argmax mask and NCHW logits."""
    L = pl.L
    ld = _lib.LossDesc()
    pl.t_target = pl.vec(B, H, W, dtype=torch.int64)
    pl.t_spat = pl.vec(B, H, W)
    pl.t_dist = pl.vec(B, 3, H, W)
    pl.t_cond = pl.vec(B, 4)
    pl.t_pc = pl.vec(B, 2)
    pl.t_eln = pl.vec(B, 2, 5)
    pl.terms = pl.vec(8)
    pl.pred_c = pl.vec(B, 2, 2)
    pl.elPred = pl.vec(B, 10)
    pl.op = pl.vec(B, 3, H, W)
    pl.mask = pl.vec(B, H, W, dtype=torch.int64)
    part = pl.vec(int(L.egne_loss_workspace_floats(B, H, W)))
    gx = torch.linspace(-1, 1, W).to(dev)   # create_meshgrid (utils.py:27-60) builds the axes this way
    gy = torch.linspace(-1, 1, H).to(dev)
    pl.keep += [gx, gy]
    ld.B, ld.H, ld.W = B, H, W
    ld.logits, ld.pix_stride, ld.ch_off = opb.data_ptr(), 8, 0
    ld.dtype = 1 if pl.bf16 else 0
    ld.target, ld.spatWts, ld.distMap = pl.t_target.data_ptr(), pl.t_spat.data_ptr(), pl.t_dist.data_ptr()
    ld.cond, ld.pupil_center, ld.elNorm = pl.t_cond.data_ptr(), pl.t_pc.data_ptr(), pl.t_eln.data_ptr()
    ld.elOut = pl.elOut.data_ptr()
    ld.alpha = 0.0
    ld.grid_x, ld.grid_y = gx.data_ptr(), gy.data_ptr()
    ld.partials, ld.out_terms, ld.pred_c, ld.elPred = part.data_ptr(), pl.terms.data_ptr(), pl.pred_c.data_ptr(), pl.elPred.data_ptr()
    ld.mask, ld.op_nchw = pl.mask.data_ptr(), pl.op.data_ptr()
    pl.loss_desc = ld
    if training:
        pl.coef = pl.vec(B, 32)
        ld.coef = pl.coef.data_ptr()
        pl.gscale = pl.vec(1)
    pl.raw(L.egne_loss_fwd, (C.byref(ld),), "loss")
    if training:
        def emit_loss(bw):
            go = pl.gbuf(opb)
            bw.raw(L.egne_loss_bwd, (C.byref(ld), pl.gscale.data_ptr(), go.data_ptr(), go.shape[-1], 0,
                                     pl.g_elOut.data_ptr()), "loss.bwd")
        pl.tape.append(emit_loss)



def confusion_head(pl, model, fc, B, training, enabled):
    """dataset-confusion head (models/RITnet_v2.py:343-350 = RITnet_v1.py:297-307; loss.py:139-157) on pl.latent_p."""
    L = pl.L
    pl.t_id = pl.vec(B, dtype=torch.int64)
    if model.disentangle and enabled:
        lins = model.dsIdentify_lin.layersLin
        cur, cc = Piece(pl.latent_p, 0, fc), fc
        for i, lin in enumerate(lins):   # actBool=False, dropout 0: plain linear stack (utils.py:953-981)
            l = ConvLayer([lin.weight], [lin.bias] if lin.bias is not None else None, [(cc, pad8(cc))], kernel_hw=(1, 1))
            ob = pl.buf(B, 1, 1, pad8(l.Cout))
            pl.conv(l, [cur], Piece(ob, 0, l.Cout), B, 1, 1, name="dsIdentify.lin%d" % i)
            cur, cc = Piece(ob, 0, l.Cout), l.Cout
        pl.raw(L.egne_conf_loss, (cur.ptr, cur.stride, pl.t_id.data_ptr(), B, cc, 1 if model.toggle else 0,
                                  float(model.disentangle_alpha), pl.terms.data_ptr()), "conf_loss")
        if training:
            if not model.toggle:
                raise NotImplementedError("toggle=False (secondary dataset loss) has no backward here: the reference "
                                          "never steps that optimizer (train.py:182-186)")
            pred = cur

            def emit_conf(bw):
                gpd = pl.gp(pred)
                tmp = bw.vec(1)
                # conf enters the total as alpha*conf; its gradient scale is gscale*alpha
                bw.raw(_PyCall(lambda: tmp.copy_(pl.gscale * float(model.disentangle_alpha))), (), "conf.scale")
                bw.raw(L.egne_conf_loss_bwd, (pred.ptr, pred.stride, pl.t_id.data_ptr(), B, cc, 1, tmp.data_ptr(),
                                                gpd.ptr, gpd.stride), "conf_loss.bwd")
            # tape order: the conf-loss emitter must run BEFORE the linear layers' backward, i.e. be appended last
            pl.tape.append(emit_conf)
