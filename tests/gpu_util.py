"""Helpers for the -m gpu tests: run single C-ABI ops on NCHW torch tensors."""
import torch

import egne_amd  # noqa: F401
from egne_amd import _lib
from egne_amd.engine import ConvLayer, Piece, Plan, pad8

DEV = "cuda:0"


def to_nhwc_buf(pl, xs, B, H, W):
    """xs: list of NCHW CPU tensors -> one NHWC buffer holding them as padded slices; returns pieces."""
    tot = sum(pad8(x.shape[1]) for x in xs)
    buf = pl.buf(B, H, W, tot)
    pieces, off = [], 0
    for x in xs:
        C = x.shape[1]
        buf[..., off:off + C] = x.permute(0, 2, 3, 1).to(DEV)
        pieces.append(Piece(buf, off, C))
        off += pad8(C)
    return pieces


def conv_hip(xs, weights, biases, stride=1, pad=(0, 0), dils=(1,), act=0, pad_mode=0, residual=None,
             norm=None, post=None, kernel_hw=None):
    """xs: list of NCHW tensors (concat order).  weights: list (groups) of OIHW tensors.
    norm: optional dict {seg_index: (scale[B,C], shift[B,C], act_in)}.  Returns NCHW CPU tensor."""
    B, _, H, W = xs[0].shape
    pl = Plan(torch.device(DEV))
    pieces = to_nhwc_buf(pl, xs, B, H, W)
    wd = [torch.nn.Parameter(w.to(DEV)) for w in weights]
    bd = [torch.nn.Parameter(b.to(DEV)) for b in biases] if biases is not None else None
    layer = ConvLayer(wd, bd, [(p.C, p.Cp) for p in pieces], stride=stride, pad=pad, dils=dils, act=act,
                      pad_mode=pad_mode, kernel_hw=kernel_hw)
    if norm:
        for i, (sc, sh, ai) in norm.items():
            scp = torch.zeros(B, pieces[i].Cp, device=DEV)
            shp = torch.zeros(B, pieces[i].Cp, device=DEV)
            scp[:, :sc.shape[1]] = sc.to(DEV)
            shp[:, :sh.shape[1]] = sh.to(DEV)
            pl.keep += [scp, shp]
            pieces[i] = pieces[i].with_norm(scp, shp, ai)
    if post is not None:
        ps = torch.zeros(layer.CoutP, device=DEV)
        pt = torch.zeros(layer.CoutP, device=DEV)
        ps[:layer.Cout] = post[0].to(DEV)
        pt[:layer.Cout] = post[1].to(DEV)
        layer.post = (ps, pt)
    Ho, Wo = layer.out_hw(H, W)
    out = pl.buf(B, Ho, Wo, pad8(layer.Cout) + 8)
    out.fill_(777.0)  # poison: stores must not touch anything outside the slice
    dst = Piece(out, 8, layer.Cout)
    res = None
    if residual is not None:
        res = to_nhwc_buf(pl, [residual], B, Ho, Wo)[0]
    pl.conv(layer, pieces, dst, B, H, W, residual=res)
    pl.run()
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[..., :8] == 777.0).all(), "conv wrote outside its output slice"
    if layer.Cout < pad8(layer.Cout):
        assert (o[..., 8 + layer.Cout:8 + pad8(layer.Cout)] == 0).all(), "padding channels must be written as zeros"
    return o[..., 8:8 + layer.Cout].permute(0, 3, 1, 2).contiguous()
