"""Outputs of the bf16 3x3 kernel on fixed inputs (forward, with affine, with residual), saved for a comparison between two builds of the library
(EGNE_LIB).  usage: EGNE_LIB=... python scratch/b3_diff.py out.pt ; python scratch/b3_diff.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        print("%-40s max diff %.3e (scale %.3e)  differing elements %d of %d" % (k, d.max().item(), a[k].float().abs().max().item(), int((d > 0).sum()), d.numel()))
    sys.exit(0)
import egne_amd  # noqa
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
DEV = torch.device('cuda:0'); BF = torch.bfloat16
G = torch.Generator().manual_seed(5)
out = {}
for name, B, Cin, Cout, H, W, norm, res, act in [("32->32", 4, 32, 32, 48, 64, False, False, 2), ("32->32 norm", 4, 32, 32, 48, 64, True, False, 2), ("64->64", 4, 64, 64, 40, 64, False, False, 2),
                                                  ("64->64 res", 4, 64, 64, 40, 64, False, True, 0), ("38->64 norm", 4, 38, 64, 30, 40, True, False, 2), ("128->128", 2, 128, 128, 30, 40, False, False, 2),
                                                  ("32->3", 4, 32, 3, 48, 64, False, False, 2), ("96->96 norm", 2, 96, 96, 30, 40, True, False, 2), ("180->180", 2, 180, 180, 30, 40, False, False, 2)]:
    pl = Plan(DEV, dtype=BF)
    xb = pl.buf(B, H, W, pad8(Cin)); xb.zero_(); xb[..., :Cin] = torch.randn(B, H, W, Cin, generator=G).to(DEV).to(BF)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3, generator=G) / (3 * Cin ** 0.5)).to(DEV)); b = torch.nn.Parameter((torch.randn(Cout, generator=G) * 0.1).to(DEV))
    layer = ConvLayer([w], [b], [(Cin, pad8(Cin))], pad=(1, 1), act=act)
    piece = Piece(xb, 0, Cin)
    if norm:
        sc = (0.5 + torch.rand(B, pad8(Cin), generator=G)).to(DEV).contiguous(); sh = (torch.randn(B, pad8(Cin), generator=G) * 0.3).to(DEV).contiguous()
        pl.keep += [sc, sh]
        piece = piece.with_norm(sc, sh, 2)
    ob = pl.buf(B, H, W, pad8(Cout)); ob.zero_()
    rb = None
    if res:
        rbuf = pl.buf(B, H, W, pad8(Cout)); rbuf.zero_(); rbuf[..., :Cout] = torch.randn(B, H, W, Cout, generator=G).to(DEV).to(BF)
        rb = Piece(rbuf, 0, Cout)
    pl.conv(layer, [piece], Piece(ob, 0, Cout), B, H, W, residual=rb)
    pl.run(); torch.cuda.synchronize()
    out[name] = ob.float().cpu()[..., :Cout].clone()
torch.save(out, sys.argv[1])
