"""Error pattern of the fused 1x1 -> 3x3 launch against float64 (debugging aid)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
import egne_amd
from egne_amd import engine
from egne_amd.engine import ConvLayer, Piece, Plan, pad8
from gpu_util import DEV, to_nhwc_buf
G = torch.Generator().manual_seed(1)
chans, C1, C2, B, H, W = (32, 32), 32, 32, 2, 61, 83
xs = [torch.randn(B, c, H, W, generator=G) * 2 for c in chans]
Cin = sum(chans)
w1, b1 = torch.randn(C1, Cin, 1, 1, generator=G) / Cin ** 0.5, torch.randn(C1, generator=G)
w2, b2 = torch.randn(C2, C1, 3, 3, generator=G) / (3 * C1 ** 0.5), torch.randn(C2, generator=G)
t = F.conv2d(torch.cat(xs, 1).double(), w1.double(), b1.double())
truth = F.leaky_relu(F.conv2d(t, w2.double(), b2.double(), padding=1), 0.01)
pl = Plan(torch.device(DEV))
pieces = to_nhwc_buf(pl, xs[:-1], B, H, W) + to_nhwc_buf(pl, xs[-1:], B, H, W)
l1 = ConvLayer([torch.nn.Parameter(w1.to(DEV))], [torch.nn.Parameter(b1.to(DEV))], [(p.C, p.Cp) for p in pieces])
l2 = ConvLayer([torch.nn.Parameter(w2.to(DEV))], [torch.nn.Parameter(b2.to(DEV))], [(C1, pad8(C1))], pad=(1, 1), act=2)
l1.split1 = l2.split = True
out = pl.buf(B, H, W, pad8(C2) + 16); out.fill_(777.0)
engine.FUSE_1X1_MIN_W = 0
pl.conv_pair(l1, pieces, l2, Piece(out, 8, C2), B, H, W)
pl.run(); torch.cuda.synchronize()
o = out.cpu()
got = o[..., 8:8 + C2].permute(0, 3, 1, 2).double()
err = (got - truth).abs()
print("max err", err.max().item(), "truth max", truth.abs().max().item(), "untouched 777:", (got == 777).sum().item())
bad = err > 1e-3
print("bad fraction", bad.float().mean().item())
print("bad per row (frame 0, ch 0):", bad[0, 0].sum(1).tolist())
print("bad per col (frame 0, ch 0):", bad[0, 0].sum(0).tolist())
print("bad per channel:", bad.sum((0, 2, 3)).tolist())
y, x = 3, 5
print("got", got[0, :4, y, x].tolist(), "truth", truth[0, :4, y, x].tolist())
for (yy, xx) in ((3, 5), (3, 6), (3, 1), (10, 33), (9, 2)):
    g = got[0, 3, yy, xx]
    d = (truth[0, 3] - g).abs()
    idx = (d < 1e-4).nonzero().tolist()
    d2 = (truth[0] - g).abs()
    print((yy, xx), "value", g.item(), "matches truth at (same channel)", idx[:4], " any channel:", (d2 < 1e-5).nonzero().tolist()[:4])
